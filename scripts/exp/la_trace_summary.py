"""Per queue and kernel family of a rocprofv3 --kernel-trace csv: launches, summed duration, summed gap to the previous kernel of the same
queue (gaps > 5 ms are host-side and dropped).  usage: la_trace_summary.py <trace dir> [skip_first_fraction]"""
import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t_lo = int(rows[0]["Start_Timestamp"]); t_hi = max(int(r["End_Timestamp"]) for r in rows)
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
cut = t_lo + skip * (t_hi - t_lo)
last_end = {}
dur = collections.defaultdict(float); gap = collections.defaultdict(float); n = collections.Counter()
span = {}
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    q = r["Queue_Id"]
    nm = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("rlhip_lu::", "").split("(")[0][:48]
    if s >= cut:
        k = (q, nm)
        n[k] += 1; dur[k] += e - s
        if q in last_end and 0 < s - last_end[q] < 5e6: gap[k] += s - last_end[q]
        lo, hi = span.get(q, (s, e)); span[q] = (min(lo, s), max(hi, e))
    last_end[q] = max(e, last_end.get(q, 0))
for q in sorted(span):
    ks = [k for k in n if k[0] == q]
    td = sum(dur[k] for k in ks); tg = sum(gap[k] for k in ks)
    print(f"queue {q}: span {(span[q][1] - span[q][0]) / 1e6:.1f} ms, {sum(n[k] for k in ks)} kernels, busy {td / 1e6:.1f} ms, gaps {tg / 1e6:.1f} ms")
    for k in sorted(ks, key=lambda k: -(dur[k] + gap[k]))[:16]:
        print(f"    {n[k]:6d} x  dur {dur[k] / 1e6:8.2f} ms ({dur[k] / n[k] / 1e3:8.1f} us)  gap-before {gap[k] / 1e6:8.2f} ms ({gap[k] / n[k] / 1e3:7.1f} us)  {k[1]}")
