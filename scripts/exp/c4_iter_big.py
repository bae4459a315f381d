"""kernels of >= 100 us of one block iteration of the serial BQRRP loop, in launch order, with grid sizes.  usage: c4_iter_big.py <trace dir> <iteration>"""
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f))); rows.sort(key=lambda r: int(r["Start_Timestamp"]))
qr = [i for i, r in enumerate(rows) if "qr_blk_kernel" in r["Kernel_Name"]]
it = int(sys.argv[2]); i0, i1 = qr[it], qr[it + 1]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if e - s < 100000: continue
    nm = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("rlhip_lu::", "").replace("rlhip::", "").split("(")[0][:58]
    print(f"{(s - t0) / 1e6:8.2f} ms  {(e - s) / 1e3:9.1f} us  grid {r['Grid_Size_X']:>8s} x {r['Grid_Size_Y']:>5s}  wg {r['Workgroup_Size_X']:>4s}  {nm}")
