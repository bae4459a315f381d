#!/bin/bash
# C3 (CQRRPT 1048576 x 1024 fp64) with and without the split QRCP: bench lines, kernel statistics, and the last call's kernel
# timeline around the pivoted QR (which kernels ran beside each other).  Run on the GPU box; outputs under gpurun_out/c3split/.
R=$GRAFT_REPO_ROOT
TAG=${1:-round5}
export PYTHONPATH=$R
O=$R/gpurun_out/c3split; mkdir -p $O
cd $R
timeout 300 python scripts/bench_other.py cqrrpt --steps 4 < /dev/null > $O/${TAG}_c3_cqrrpt_line.json 2> $O/c3.err
timeout 300 python scripts/bench_other.py cqrrpt --steps 4 --opt cqrrpt_split_qrcp=0 < /dev/null > $O/${TAG}_c3_cqrrpt_one_piece_qrcp_line.json 2>> $O/c3.err
timeout 300 python scripts/bench_other.py cqrrpt --steps 4 --opt saso_mode=0 < /dev/null > $O/${TAG}_c3_cqrrpt_affine_saso_line.json 2>> $O/c3.err
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $R/scripts/bench_other.py cqrrpt --steps 3 < /dev/null > $O/${TAG}_c3_cqrrpt_line_profiled.json 2> $O/prof.err
f=$(find $O/prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $O/${TAG}_c3_cqrrpt_kernel_stats.csv
python - <<PY > $O/${TAG}_c3_split_timeline.txt 2>&1
import csv, glob
f = glob.glob("$O/prof/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the LAST untimed CQRRPT call: find the last-but-one saso_apply launch (the last call of the bench is the timed one, which runs unsplit)
sa = [i for i, r in enumerate(rows) if "saso_apply_dma" in r["Kernel_Name"]]
i0 = sa[-2]; i1 = sa[-1]
t0 = int(rows[i0]["Start_Timestamp"])
qk = "Queue_Id" if "Queue_Id" in rows[0] else "Stream_Id"
print("one untimed CQRRPT call (1048576 x 1024 fp64, split QRCP), kernels of >= 20 us: start (ms after the sketch apply starts), duration (ms), queue, kernel")
for r in rows[i0:i1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if e - s < 20000: continue
    nm = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:70]
    print(f"{(s - t0) / 1e6:8.3f} {(e - s) / 1e6:8.3f}  q{r.get(qk, '?'):>3s}  {nm}")
PY
rm -rf $O/prof
for j in $O/*line.json; do echo "$(basename $j): $(cut -c1-200 $j)"; done
cat $O/${TAG}_c3_split_timeline.txt
