#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "geqrf_q or hqrq or orhr or geqrf" > gpurun_out/r6_t8.log 2>&1; echo "rc=$?" >> gpurun_out/r6_t8.log; tail -15 gpurun_out/r6_t8.log | cut -c1-250
timeout 900 python -m pytest tests/test_gpu_drivers.py tests/test_gpu_linops.py tests/test_gpu_fullsize.py -q -x -k "abrik or hqrq or rsvd" > gpurun_out/r6_t9.log 2>&1; echo "rc=$?" >> gpurun_out/r6_t9.log; tail -8 gpurun_out/r6_t9.log | cut -c1-250
timeout 300 python scripts/bench_other.py abrik --steps 3 > gpurun_out/round6_c5_abrik_line.json 2> gpurun_out/r6_c5.err; cut -c1-400 gpurun_out/round6_c5_abrik_line.json
