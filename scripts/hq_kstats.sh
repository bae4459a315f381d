#!/bin/bash
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp PYTHONPATH=$R
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/hqprof -- python $R/scripts/hq_prof.py 16384 256 < /dev/null 2>&1 | grep "hqrrp ms"
f=$(find $R/gpurun_out/hqprof -name '*kernel_stats.csv' 2>/dev/null | head -1)
if [ -n "$f" ]; then head -14 "$f" | cut -c1-210; rm -rf $R/gpurun_out/hqprof; fi
