#!/bin/bash
# per-rank view of the 8-way row-sharded RSVD (m/8 rows on one device, no exchange): where the replicated tail goes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python $R/bench.py --m 25000 --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/tail_line.json 2> $R/gpurun_out/tail_err.log
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/tail_prof -- python $R/bench.py --m 25000 --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/tail_line_prof.json 2>> $R/gpurun_out/tail_err.log
f=$(find $R/gpurun_out/tail_prof -name '*kernel_stats.csv' | head -1)
head -30 "$f" | cut -c1-200
cat $R/gpurun_out/tail_line.json | cut -c1-400
