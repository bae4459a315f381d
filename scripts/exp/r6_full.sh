#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=15 > gpurun_out/r6_full.log 2>&1; echo "rc=$?" >> gpurun_out/r6_full.log
tail -30 gpurun_out/r6_full.log | cut -c1-250
