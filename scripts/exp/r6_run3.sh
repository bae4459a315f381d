#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_sharded_f32.py tests/test_gpu_linops.py -q -x -k "deficient or coo or view or block" > gpurun_out/r6_t5.log 2>&1; echo "rc=$?" >> gpurun_out/r6_t5.log
tail -12 gpurun_out/r6_t5.log | cut -c1-300
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -x -k "same_xcd or persistent" > gpurun_out/r6_t6.log 2>&1; echo "rc=$?" >> gpurun_out/r6_t6.log
tail -8 gpurun_out/r6_t6.log | cut -c1-300
timeout 600 python scripts/ranks_on_one_device.py bqrrp --check --steps 1 --decades 4 > gpurun_out/round6_ranks8_on_one_device_bqrrp.json 2> gpurun_out/r6_ranks8.err; echo "ranks8 rc=$?"; cat gpurun_out/round6_ranks8_on_one_device_bqrrp.json
timeout 600 python scripts/bench_other.py bqrrp_full --triple default --steps 1 > gpurun_out/round6_c4_bqrrp_f32_65536_default_triple_line.json 2> gpurun_out/r6_c4def.err; echo "c4 default rc=$?"; cut -c1-700 gpurun_out/round6_c4_bqrrp_f32_65536_default_triple_line.json; tail -3 gpurun_out/r6_c4def.err
R=$GRAFT_REPO_ROOT
( cd /tmp && RLHIP_ROCTX=1 timeout 600 rocprofv3 --marker-trace --kernel-trace --output-format csv -d $R/gpurun_out/mk -- python $R/scripts/bq_prof.py 32768 2048 f32 > $R/gpurun_out/r6_mk.log 2>&1 ); echo "marker rc=$?"
python scripts/marker_summary.py gpurun_out/mk > gpurun_out/round6_bqrrp_32768_phase_ranges.txt 2>&1; cat gpurun_out/round6_bqrrp_32768_phase_ranges.txt; ls gpurun_out/mk/* | head; rm -rf gpurun_out/mk
