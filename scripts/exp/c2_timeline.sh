#!/bin/bash
# kernel timeline of one full-size C2 step (start, duration, gap before each kernel), same tool as the shard timeline
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
O=$R/gpurun_out/c2tl; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/tl -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python $R/scripts/timeline.py $O/tl 3 > $O/c2_timeline.txt 2>&1; rm -rf $O/tl
