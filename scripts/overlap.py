"""From a rocprofv3 --kernel-trace csv: per queue, busy time; time with >= 2 kernels in flight; the side queue's kernels with what ran beside them.
usage: overlap.py <trace dir> [min_start_fraction]"""
import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
key = "Queue_Id" if "Queue_Id" in rows[0] else ("Stream_Id" if "Stream_Id" in rows[0] else None)
ev = []
busy = collections.Counter(); cnt = collections.Counter()
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    qid = r.get(key, "?")
    busy[qid] += e - s; cnt[qid] += 1
    ev.append((s, 1, qid)); ev.append((e, -1, qid))
ev.sort()
depth = 0; last = ev[0][0]; t_by_depth = collections.Counter()
for t, dlt, qid in ev:
    t_by_depth[depth] += t - last; last = t; depth += dlt
print("columns:", list(rows[0].keys()))
for qid in busy: print(f"queue {qid}: {cnt[qid]} kernels, busy {busy[qid]/1e6:.2f} ms")
for dpt in sorted(t_by_depth): print(f"depth {dpt}: {t_by_depth[dpt]/1e6:.2f} ms")
# top kernels by total time
tot = collections.Counter(); n = collections.Counter()
for r in rows:
    nm = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:60]
    tot[nm] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); n[nm] += 1
for nm, t in tot.most_common(14): print(f"{t/1e6:9.2f} ms  {n[nm]:6d}  {nm}")
