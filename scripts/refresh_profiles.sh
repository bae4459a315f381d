#!/bin/bash
# re-measure every committed line under profiles/ with the current build (run on the GPU box; outputs under gpurun_out/refresh/)
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
O=$R/gpurun_out/refresh; mkdir -p $O
cd $R
timeout 300 python bench.py < /dev/null > $O/round1_bench_line.json 2> $O/bench.err
timeout 300 python scripts/bench_other.py cqrrpt --steps 3 < /dev/null > $O/round1_c3_cqrrpt_line.json 2> $O/c3.err
timeout 300 python scripts/bench_other.py bqrrp --steps 3 < /dev/null > $O/round1_c4cut_bqrrp_f32_line.json 2> $O/c4cut.err
timeout 300 python scripts/bench_other.py bqrrp64 --steps 3 < /dev/null > $O/round1_bqrrp_f64_16k_line.json 2> $O/b64.err
timeout 300 python scripts/bench_other.py bqrrp_full --steps 2 < /dev/null > $O/round1_c4_bqrrp_f32_65536_line.json 2> $O/c4.err
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $R/bench.py --no-cpu-baseline < /dev/null > $O/round1_bench_line_profiled.json 2> $O/prof.err
f=$(find $O/prof -name '*kernel_stats.csv' 2>/dev/null | head -1)
if [ -n "$f" ]; then cp "$f" $O/round1_bench_kernel_stats.csv; fi
rm -rf $O/prof
for j in $O/*.json; do echo "$(basename $j): $(cut -c1-260 $j)"; done
