#!/bin/bash
# kernel-level breakdown of BQRRP at BASELINE configs[3] (65536^2 fp32, b = 2048) on one device
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/c4prof -- python $R/scripts/bq_prof.py 65536 2048 f32 < /dev/null 2>&1 | tail -3
f=$(find $R/gpurun_out/c4prof -name '*kernel_stats.csv' 2>/dev/null | head -1)
if [ -n "$f" ]; then cp "$f" $R/gpurun_out/c4_kernel_stats.csv; head -32 "$f" | cut -c1-230; rm -rf $R/gpurun_out/c4prof; else echo "no stats file"; fi
