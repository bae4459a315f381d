#!/bin/bash
# every kernel of the last ABRIK call at C5 (scripts/exp/abrik_one.py marks it with a torch fill), gaps included
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
O=$R/gpurun_out/c5tl; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof -- python $R/scripts/exp/abrik_one.py < /dev/null > $O/out.txt 2> $O/prof.err
python $R/scripts/exp/last_call_timeline.py $O/prof > $O/c5_timeline.txt 2>&1
rm -rf $O/prof
