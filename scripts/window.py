import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
qr = [i for i, r in enumerate(rows) if "qr_blk_kernel" in r["Kernel_Name"]]
i0, i1 = qr[int(sys.argv[2])], qr[int(sys.argv[2]) + 1]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if e - s < 300000: continue
    print(f"q{r['Queue_Id']} {(s-t0)/1e6:8.2f} .. {(e-t0)/1e6:8.2f} ms  dur {(e-s)/1e6:7.2f}  grid {r['Grid_Size_X']}  {r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','')[:70]}")
