#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "geqrf or hqrq or orhr or cholqr" > gpurun_out/r6_t10.log 2>&1; echo "rc=$?" >> gpurun_out/r6_t10.log; tail -12 gpurun_out/r6_t10.log | cut -c1-250
timeout 900 python -m pytest tests/test_gpu_drivers.py tests/test_gpu_linops.py tests/test_gpu_fullsize.py tests/test_gpu_benchmarks.py -q -x -k "abrik or hqrq or rsvd or hqrrp" > gpurun_out/r6_t11.log 2>&1; echo "rc=$?" >> gpurun_out/r6_t11.log; tail -8 gpurun_out/r6_t11.log | cut -c1-250
timeout 300 python scripts/bench_other.py abrik --steps 3 > gpurun_out/round6_c5_abrik_line.json 2> gpurun_out/r6_c5.err; cut -c1-300 gpurun_out/round6_c5_abrik_line.json
bash scripts/exp/abrik_timeline.sh; cat gpurun_out/c5tl/out.txt; cp gpurun_out/c5tl/c5_timeline.txt gpurun_out/round6_c5_abrik_timeline.txt
