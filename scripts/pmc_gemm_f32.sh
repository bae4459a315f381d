#!/bin/bash
# PMC pass (counters only, own run) over scripts/pmc_gemm_f32.py -> gpurun_out/pmc_f32/summary.txt
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp PYTHONPATH=$R
O=$R/gpurun_out/pmc_f32; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/run -- python $R/scripts/pmc_gemm_f32.py < /dev/null > $O/log.txt 2>&1
python - <<PY
import csv, glob, collections
O = "$O"
rows = []
for f in glob.glob(O + "/run/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
kt = {}
for f in glob.glob(O + "/run/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        kt[r.get("Dispatch_Id")] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
disp = collections.OrderedDict()
for r in rows:
    if "gemm_kernel" not in r["Kernel_Name"]: continue
    d = disp.setdefault(r["Dispatch_Id"], {"name": r["Kernel_Name"][:110]})
    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
with open(O + "/summary.txt", "w") as out:
    out.write("rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -- python scripts/pmc_gemm_f32.py\n")
    out.write("generic MFMA GEMM, fp32, 128x128 tiles, 2 workgroups/CU: TN = W (2048 x 30720) = V^T C over 32768 rows; NN = C (32768 x 30720) -= V W (K = 2048); GRBM_GUI_ACTIVE is summed over 8 XCDs\n")
    for k, d in disp.items():
        cyc = d.get("GRBM_GUI_ACTIVE", 0) / 8.0
        ms = kt.get(k)
        line = f"{d['name']}  ms={ms}  clock={(cyc / (ms * 1e6)) if ms else 0:.2f}GHz  mfma_busy/simd_cycles={d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (cyc * 1024) if cyc else 0:.3f}  cu_busy={d.get('SQ_BUSY_CU_CYCLES', 0) / (cyc * 256) if cyc else 0:.3f}  wait_inst/wave_cycles={d.get('SQ_WAIT_INST_ANY', 0) / max(d.get('SQ_WAVE_CYCLES', 1), 1):.3f}  lds_bank_conflict={d.get('SQ_LDS_BANK_CONFLICT', 0):.0f}"
        out.write(line + "\n"); print(line)
PY
rm -rf $O/run
