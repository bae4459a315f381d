#!/bin/bash
# every kernel of one untimed CQRRPT call at C3 (no duration threshold), with the gap to the previous kernel's end on any queue
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
O=$R/gpurun_out/c3tl; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --output-format csv -d $O/prof -- python $R/scripts/bench_other.py cqrrpt --steps 3 < /dev/null > $O/line.json 2> $O/prof.err
python - <<PY > $O/c3_full_timeline.txt 2>&1
import csv, glob
f = glob.glob("$O/prof/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sa = [i for i, r in enumerate(rows) if "saso_apply_dma" in r["Kernel_Name"]]
i0 = sa[-2]; i1 = sa[-1]
t0 = int(rows[i0]["Start_Timestamp"])
qk = "Queue_Id" if "Queue_Id" in rows[0] else "Stream_Id"
last = t0
for r in rows[i0:i1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    nm = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:64]
    print(f"{(s - t0) / 1e6:8.3f} {(e - s) / 1e3:9.1f} us  gap {(s - last) / 1e3:8.1f}  q{r.get(qk, '?'):>2s}  {nm}")
    last = max(last, e)
PY
rm -rf $O/prof
