"""Aggregate a rocprofv3 kernel trace by kernel family: launches, summed duration, share.  usage: trace_agg.py <trace dir> [top]"""
import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    nm = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("rlhip::", "").split("(")[0][:64]
    a = agg[nm]; a[0] += 1; a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = sum(a[1] for a in agg.values())
print(f"{len(rows)} kernels, {tot / 1e6:.2f} ms busy")
for nm, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{n:6d} x {t / 1e6:9.3f} ms ({100 * t / tot:5.1f} %)  avg {t / n / 1e3:9.1f} us  {nm}")
