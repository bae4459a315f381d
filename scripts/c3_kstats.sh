#!/bin/bash
# kernel-level breakdown of CQRRPT at BASELINE configs[2] (1048576 x 1024 fp64)
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp PYTHONPATH=$R
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/c3prof -- python $R/scripts/bench_other.py cqrrpt --steps 3 < /dev/null > $R/gpurun_out/c3_line_profiled.json 2> $R/gpurun_out/c3_prof.err
f=$(find $R/gpurun_out/c3prof -name '*kernel_stats.csv' 2>/dev/null | head -1)
if [ -n "$f" ]; then cp "$f" $R/gpurun_out/c3_kernel_stats.csv; head -14 "$f" | cut -c1-200; rm -rf $R/gpurun_out/c3prof; else echo "no stats"; tail -5 $R/gpurun_out/c3_prof.err; fi
