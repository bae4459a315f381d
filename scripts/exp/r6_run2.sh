#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_sharded_f32.py -q -k "deficient" > gpurun_out/r6_t3.log 2>&1; echo "rc=$?" >> gpurun_out/r6_t3.log
tail -15 gpurun_out/r6_t3.log
timeout 900 python -m pytest tests/test_gpu_repeat.py -q --durations=10 > gpurun_out/r6_t4.log 2>&1; echo "rc=$?" >> gpurun_out/r6_t4.log
tail -25 gpurun_out/r6_t4.log
timeout 1200 python scripts/c4_pivots_determined.py > gpurun_out/r6_c4_pivots2.json 2> gpurun_out/r6_c4_pivots2.err; echo "piv rc=$?"
tail -5 gpurun_out/r6_c4_pivots2.err
