"""Print the kernel timeline of the LAST step of `bench.py --m M` from a rocprofv3 --kernel-trace csv (start, duration, gap before, name)."""
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a step starts with the first stream-K NN launch (gemm_sk_kernel<double, false,) after a fill
starts = [i for i, r in enumerate(rows) if "gemm_sk_kernel<double, false," in r["Kernel_Name"]]
# bench runs the roofline probe launches at the end: take the step before those = the (steps)th occurrence from warmup; choose by argv
which = int(sys.argv[2]) if len(sys.argv) > 2 else 3
i0 = starts[which]
i1 = starts[which + 1]
t0 = int(rows[i0]["Start_Timestamp"])
prev_end = t0
tot_gap = 0
for r in rows[i0:i1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = s - prev_end
    tot_gap += max(gap, 0)
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:70]
    print(f"{(s - t0)/1e3:9.1f} us  dur {(e - s)/1e3:8.1f}  gap {gap/1e3:7.1f}  {name}")
    prev_end = max(prev_end, e)
print(f"step span {(prev_end - t0)/1e3:.1f} us, gaps {tot_gap/1e3:.1f} us, kernels {i1 - i0}")
