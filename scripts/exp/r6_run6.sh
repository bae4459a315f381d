#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
( cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/sv3 -- python $R/bench.py --m 25000 --steps 3 --warmup 2 --no-cpu-baseline --opt jacobi_persist=0 > /dev/null 2> $R/gpurun_out/r6_sv3.err; echo "no cooperative launch (jacobi_persist=0): rocprofv3 rc=$?" )
( cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/sv4 -- python $R/bench.py --m 25000 --steps 3 --warmup 2 --no-cpu-baseline --opt jacobi_persist=3 > /dev/null 2> $R/gpurun_out/r6_sv4.err; echo "persistent kernel by ordinary launch (jacobi_persist=3): rocprofv3 rc=$?" )
rm -rf gpurun_out/sv3 gpurun_out/sv4
timeout 300 python scripts/bench_other.py cqrrpt --steps 4 > gpurun_out/round6_c3_cqrrpt_line.json 2> gpurun_out/r6_c3.err; python -c "import json; o=json.load(open('gpurun_out/round6_c3_cqrrpt_line.json')); print('C3', o['ms_per_step'], o['roofline']['also'])"
timeout 1500 python -m pytest tests/ -x -q -m gpu > gpurun_out/r6_full2.log 2>&1; echo "rc=$?" >> gpurun_out/r6_full2.log; tail -4 gpurun_out/r6_full2.log | cut -c1-200
