#!/bin/bash
# re-measure the C3 lines + kernel statistics only (same commands as scripts/refresh_profiles.sh), outputs under gpurun_out/refresh_c3/
R=$GRAFT_REPO_ROOT
TAG=${1:-round5}
export PYTHONPATH=$R
O=$R/gpurun_out/refresh_c3; mkdir -p $O
cd $R
timeout 300 python scripts/bench_other.py cqrrpt --steps 5 < /dev/null > $O/${TAG}_c3_cqrrpt_line.json 2> $O/c3.err
timeout 300 python scripts/bench_other.py cqrrpt --steps 3 --opt saso_mode=0 < /dev/null > $O/${TAG}_c3_cqrrpt_affine_saso_line.json 2>> $O/c3.err
timeout 300 python scripts/saso_time.py > $O/saso_time.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $R/scripts/bench_other.py cqrrpt --steps 3 < /dev/null > $O/${TAG}_c3_cqrrpt_line_profiled.json 2> $O/prof.err
f=$(find $O/prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $O/${TAG}_c3_cqrrpt_kernel_stats.csv; rm -rf $O/prof
for j in $O/*line*.json; do echo "$(basename $j): $(cut -c1-200 $j)"; done; tail -5 $O/saso_time.txt
