#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
RLHIP_POISON=1 timeout 2400 python -m pytest tests/ -q -m gpu -p no:cacheprovider > gpurun_out/r6_poison.log 2>&1; echo "rc=$?" >> gpurun_out/r6_poison.log
grep -n "^FAILED\|passed\|failed" gpurun_out/r6_poison.log | tail -40 | cut -c1-250
