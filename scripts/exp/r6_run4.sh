#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_sharded_f32.py tests/test_gpu_sharded.py -q > gpurun_out/r6_t7.log 2>&1; echo "rc=$?" >> gpurun_out/r6_t7.log
tail -12 gpurun_out/r6_t7.log | cut -c1-300
O=gpurun_out/round6_clock_holders_ab.txt
echo "shard step (bench.py --m 25000 --steps 8 --warmup 3): clock holders of the persistent Jacobi on / off, device idle / a second stream busy (fp64 FMA chains on N workgroups)" > $O
for busy in 0 64 256; do for h in 1 0; do
  for rep in 1 2; do
  timeout 200 python bench.py --m 25000 --steps 8 --warmup 3 --no-cpu-baseline --opt jacobi_clock_holders=$h --busy-side $busy 2>/dev/null | python -c "import sys,json; o=json.loads(sys.stdin.read()); print('holders=$h busy_side_workgroups=$busy rep=$rep ms_per_step', o['ms_per_step'], 'launch_ms', o['roofline']['launch_ms'])" >> $O
  done
done; done
for h in 1 0; do for rep in 1 2; do
  timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --opt jacobi_clock_holders=$h 2>/dev/null | python -c "import sys,json; o=json.loads(sys.stdin.read()); print('FULL C2 holders=$h rep=$rep ms_per_step', o['ms_per_step'], 'launch_ms', o['roofline']['launch_ms'])" >> $O
done; done
cat $O
