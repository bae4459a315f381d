#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python scripts/exp/tri_gram_ab.py > gpurun_out/r6_tri_gram_ab.txt 2>&1; cat gpurun_out/r6_tri_gram_ab.txt
# does rocprofv3's exit-time segfault depend on the library's roctx binding?
( cd /tmp; RLHIP_ROCTX=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/sv0 -- python $R/bench.py --m 25000 --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/r6_sv0.err; echo "RLHIP_ROCTX=0 rocprofv3 rc=$?" )
( cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/sv1 -- python $R/bench.py --m 25000 --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/r6_sv1.err; echo "default rocprofv3 rc=$?" )
( cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/sv2 -- python -c "import torch; x=torch.randn(4096,4096,device='cuda'); y=x@x; torch.cuda.synchronize(); print(float(y.sum()))" > /dev/null 2> $R/gpurun_out/r6_sv2.err; echo "torch-only rocprofv3 rc=$?" )
rm -rf gpurun_out/sv0 gpurun_out/sv1 gpurun_out/sv2
timeout 900 python scripts/pmc_all.py round6 trsm_fused_oop gemm_sk_tri trsm_fused gemm_sk_nn > gpurun_out/r6_pmc2.log 2>&1; cp gpurun_out/pmc/round6_pmc_*.json profiles/
timeout 300 python scripts/bench_other.py cqrrpt --steps 4 > gpurun_out/round6_c3_cqrrpt_line.json 2> gpurun_out/r6_c3.err; cut -c1-300 gpurun_out/round6_c3_cqrrpt_line.json
python -c "import json; o=json.load(open('gpurun_out/round6_c3_cqrrpt_line.json')); print(o['roofline']['traffic'], o['roofline']['traffic_source'])"
timeout 300 python bench.py > gpurun_out/round6_bench_line.json 2> gpurun_out/r6_bench.err; python -c "import json; o=json.load(open('gpurun_out/round6_bench_line.json')); print(o['ms_per_step'], o['roofline'])"
