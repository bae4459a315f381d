#!/bin/bash
# HBM-side traffic of the dominant kernel (gemm_sk_kernel<double, NN>, Y = A*Omega at the C2 shape) from the PMC counters:
# two SEPARATE passes (FETCH_SIZE, WRITE_SIZE), kernel trace only -- as MI355X_MICROARCH.md prescribes.
# usage: bash scripts/pmc_traffic.sh <round tag>   ->  gpurun_out/pmc/<tag>_pmc_traffic.json (+ the raw per-dispatch csv rows)
R=$GRAFT_REPO_ROOT; TAG=${1:-round2}
cd /tmp; export TMPDIR=/tmp PYTHONPATH=$R
O=$R/gpurun_out/pmc; mkdir -p $O
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/run_$C -- python $R/scripts/pmc_gemm.py < /dev/null > $O/log_$C.txt 2>&1
done
python - <<PY
import csv, glob, json
O, TAG = "$O", "$TAG"
def total(counter):
    vals = {}
    for f in glob.glob(f"{O}/run_{counter}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "gemm_sk_kernel<double, false>" in r["Kernel_Name"] and r["Counter_Name"] == counter:
                vals[r["Dispatch_Id"]] = vals.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
    v = sorted(vals.values())
    return v[len(v) // 2] if v else None, len(v)
f, nf = total("FETCH_SIZE"); w, nw = total("WRITE_SIZE")
m, n, k = 200000, 20000, 256
alg = 8 * (m * n + n * k + m * k)
out = {"kernel": "gemm_sk_kernel<double, false> (Y = A*Omega, 200000 x 20000 x 256 fp64)",
       "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, in a separate pass, --pmc WRITE_SIZE) -- python scripts/pmc_gemm.py  (scripts/pmc_traffic.sh)",
       "dispatches": [nf, nw], "fetch_size_kib_raw": f, "write_size_kib_raw": w,
       "correction": "gfx950: FETCH_SIZE reports 1/2 of the bytes of 16 B/lane streaming reads (MI355X_MICROARCH.md, HBM section) -> x2; WRITE_SIZE taken as is",
       "traffic_bytes": (2 * f + w) * 1024 if f is not None and w is not None else None, "algorithmic_bytes": alg}
json.dump(out, open(f"{O}/{TAG}_pmc_traffic.json", "w"), indent=1)
print(json.dumps(out))
PY
rm -rf $O/run_FETCH_SIZE $O/run_WRITE_SIZE
