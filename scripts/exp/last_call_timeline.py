"""All kernels after the LAST torch fill marker of a rocprofv3 kernel trace: start (ms), duration (us), gap before (us), name."""
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f))); rows.sort(key=lambda r: int(r["Start_Timestamp"]))
mk = [i for i, r in enumerate(rows) if "FillFunctor" in r["Kernel_Name"]]
i0 = mk[-1] + 1
t0 = int(rows[i0]["Start_Timestamp"]); last = t0
for r in rows[i0:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    nm = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("rlhip::", "")[:70]
    print(f"{(s - t0) / 1e6:8.3f} {(e - s) / 1e3:8.1f} us  gap {(s - last) / 1e3:7.1f}  {nm}")
    last = max(last, e)
