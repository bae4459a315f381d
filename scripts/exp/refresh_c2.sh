#!/bin/bash
# re-measure the C2 lines (whole matrix and 1/8 row shard), their kernel statistics and the shard timeline -- the same commands as scripts/refresh_profiles.sh
R=$GRAFT_REPO_ROOT
TAG=${1:-round5}
export PYTHONPATH=$R
O=$R/gpurun_out/refresh_c2; mkdir -p $O
cd $R
timeout 300 python bench.py < /dev/null > $O/${TAG}_bench_line.json 2> $O/bench.err
timeout 300 python bench.py --m 25000 --steps 8 --warmup 3 --no-cpu-baseline < /dev/null > $O/${TAG}_rank_of_8_line.json 2> $O/r8.err
cd /tmp && export TMPDIR=/tmp
prof() {
    local name=$1; shift
    timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -- "$@" < /dev/null > $O/${TAG}_${name}_line_profiled.json 2> $O/prof_$name.err
    local f=$(find $O/prof_$name -name '*kernel_stats.csv' 2>/dev/null | head -1)
    if [ -n "$f" ]; then cp "$f" $O/${TAG}_${name}_kernel_stats.csv; fi
    rm -rf $O/prof_$name
}
prof bench python $R/bench.py --no-cpu-baseline
prof rank_of_8 python $R/bench.py --m 25000 --steps 5 --warmup 3 --no-cpu-baseline
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/tl -- python $R/bench.py --m 25000 --steps 6 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
python $R/scripts/timeline.py $O/tl 5 > $O/${TAG}_rank_of_8_timeline.txt 2>&1; rm -rf $O/tl
for j in $O/${TAG}_*line.json; do echo "$(basename $j): $(cut -c1-200 $j)"; done
grep -n "jacobi\|step span" $O/${TAG}_rank_of_8_timeline.txt
