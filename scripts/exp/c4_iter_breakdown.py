"""One block iteration of the SERIAL BQRRP loop at C4 from a rocprofv3 kernel trace: kernels between two consecutive sketch-QR launches, aggregated by
name in order of first appearance: launches, summed duration, summed gap before (idle time of the device in front of the kernel).
usage: c4_iter_breakdown.py <trace dir> <iteration index>"""
import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
qr = [i for i, r in enumerate(rows) if "qr_blk_kernel" in r["Kernel_Name"]]
it = int(sys.argv[2])
i0, i1 = qr[it], qr[it + 1]
agg = collections.OrderedDict()
last = int(rows[i0]["Start_Timestamp"])
t0 = last
for r in rows[i0:i1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    nm = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("rlhip_lu::", "").replace("rlhip::", "").split("(")[0][:60]
    a = agg.setdefault(nm, [0, 0.0, 0.0])
    a[0] += 1; a[1] += e - s; a[2] += max(0, s - last)
    last = max(last, e)
tot_d = sum(a[1] for a in agg.values()); tot_g = sum(a[2] for a in agg.values())
print(f"iteration {it}: span {(last - t0) / 1e6:.2f} ms, kernels {sum(a[0] for a in agg.values())}, busy {tot_d / 1e6:.2f} ms, gaps {tot_g / 1e6:.2f} ms")
for nm, (n, dsum, g) in agg.items():
    print(f"{n:5d} x  dur {dsum / 1e3:9.1f} us  gaps {g / 1e3:8.1f} us  {nm}")
