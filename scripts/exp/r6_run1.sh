#!/bin/bash
# round 6, GPU call 1: the new sharded fp32 / deficient / look-ahead tests, the full-size pivot experiment, the PMC pass of the persistent Jacobi
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_sharded_f32.py tests/test_gpu_lookahead.py -x -q > gpurun_out/r6_t1.log 2>&1; echo "rc=$?" >> gpurun_out/r6_t1.log
tail -5 gpurun_out/r6_t1.log
timeout 600 python -m pytest tests/test_gpu_sharded.py -x -q > gpurun_out/r6_t2.log 2>&1; echo "rc=$?" >> gpurun_out/r6_t2.log
tail -3 gpurun_out/r6_t2.log
timeout 900 python scripts/c4_pivots_determined.py > gpurun_out/r6_c4_pivots.json 2> gpurun_out/r6_c4_pivots.err; echo "piv rc=$?"
tail -40 gpurun_out/r6_c4_pivots.json
timeout 600 python scripts/pmc_all.py round6 jacobi_persist > gpurun_out/r6_pmc_jp.log 2>&1; echo "pmc rc=$?"; tail -5 gpurun_out/r6_pmc_jp.log
