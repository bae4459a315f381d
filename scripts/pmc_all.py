#!/usr/bin/env python3
"""Counter evidence for every roofline claim: for each workload of scripts/pmc_workloads.py run THREE separate rocprofv3 passes
(kernel trace + one counter group each, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE do not fit one pass, and no
pass mixes --pmc with any other trace domain):

    rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python scripts/pmc_workloads.py <name>
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -- ...
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -- ...

and write gpurun_out/pmc/<tag>_pmc_<name>.json (copy the ones to be judged into profiles/).  Per dispatch of the kernel under test
(median over the launches): duration, FETCH / WRITE bytes (KiB counters x 1024; FETCH x 2 = the gfx950 correction for 16 B/lane
streaming reads, stated in the file), the ratio to the ALGORITHMIC bytes, MFMA busy / SIMD cycles, CU busy, LDS bank-conflict cycles,
wait fractions.

    python scripts/pmc_all.py <tag> [name ...]          (on the GPU box: run from the repo root)"""
import csv
import glob
import json
import os
import shutil
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
SQ = "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT".split()
GROUPS = {"fetch": ["FETCH_SIZE"], "write": ["WRITE_SIZE"], "sq": SQ}


def run_pass(name, group, outdir):
    shutil.rmtree(outdir, ignore_errors=True)
    env = dict(os.environ, TMPDIR="/tmp", PYTHONPATH=ROOT)
    cmd = ["rocprofv3", "--kernel-trace", "--pmc", *GROUPS[group], "--output-format", "csv", "-d", outdir, "--", sys.executable,
           os.path.join(ROOT, "scripts", "pmc_workloads.py"), name]
    p = subprocess.run(cmd, cwd="/tmp", env=env, stdin=subprocess.DEVNULL, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    return p.returncode, p.stdout[-2000:]


def collect(outdir, filt):
    """{dispatch id: {counter: value summed over the counter's instances, 'ms': duration}} for dispatches whose kernel name contains filt"""
    disp = {}
    for f in glob.glob(outdir + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if filt not in r["Kernel_Name"]:
                continue
            e = disp.setdefault(r["Dispatch_Id"], {"name": r["Kernel_Name"]})
            e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    for f in glob.glob(outdir + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Dispatch_Id") in disp:
                disp[r["Dispatch_Id"]]["ms"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    return disp


def median_of(disp, key):
    v = [e[key] for e in disp.values() if key in e]
    return statistics.median(v) if v else None


def main():
    from pmc_workloads import WORKLOADS

    tag = sys.argv[1] if len(sys.argv) > 1 else "round4"
    names = sys.argv[2:] or list(WORKLOADS)
    out_root = os.path.join(ROOT, "gpurun_out", "pmc")
    os.makedirs(out_root, exist_ok=True)
    for name in names:
        fn, filt, what, alg_bytes, flops, bound = WORKLOADS[name]
        from randlapack_amd import _lib as _rl
        rec = {"workload": name, "kernel_filter": filt, "what": what, "bound": bound, "algorithmic_bytes": alg_bytes, "flops_per_launch": flops,
               "librlhip_sha256": _rl.lib_sha256(),
               "command": "rocprofv3 --kernel-trace --pmc <group> --output-format csv -- python scripts/pmc_workloads.py " + name +
                          "   (three separate passes: FETCH_SIZE | WRITE_SIZE | " + " ".join(SQ) + "; scripts/pmc_all.py)",
               "correction": "FETCH_SIZE / WRITE_SIZE are in KiB; gfx950 reports 1/2 of the bytes of 16 B/lane streaming reads (MI355X_MICROARCH.md, HBM) -> "
                             "fetch_bytes_corrected = 2 x raw; WRITE_SIZE taken as is (uncalibrated).  The counters sit on the L2's fabric side: Infinity-Cache hits are counted."}
        groups = ("fetch", "sq") if bound == "latency" else ("fetch", "write", "sq")     # latency-bound kernels: no WRITE_SIZE pass
        if os.environ.get("PMC_GROUPS"):
            groups = tuple(x for x in os.environ["PMC_GROUPS"].split(",") if x in GROUPS)
        for group in groups:
            od = f"/tmp/pmc_{name}_{group}"
            try:
                rc, tail = run_pass(name, group, od)
            except subprocess.TimeoutExpired:
                rec[group + "_error"] = "timeout"
                continue
            disp = collect(od, filt)
            shutil.rmtree(od, ignore_errors=True)
            if not disp:
                rec[group + "_error"] = f"rc={rc}, no dispatches; " + tail[-400:]
                continue
            if rc != 0:       # rocprofv3 7.2 aborts at EXIT of a process that made a cooperative launch; the counter rows are complete by then
                rec.setdefault("profiler_exit_codes", {})[group] = rc
            rec.setdefault("dispatches", {})[group] = len(disp)
            rec.setdefault("kernel_name", next(iter(disp.values()))["name"][:160])
            ms = median_of(disp, "ms")
            if group == "fetch":
                raw = median_of(disp, "FETCH_SIZE")
                rec.update(fetch_size_kib_raw=raw, fetch_bytes_corrected=2 * raw * 1024, launch_ms_fetch_pass=ms)
                if len(disp) > 8:       # many small launches (latency-bound kernels): report the sum over one call as well
                    rec["fetch_bytes_corrected_sum"] = 2 * 1024 * sum(e.get("FETCH_SIZE", 0) for e in disp.values())
            elif group == "write":
                raw = median_of(disp, "WRITE_SIZE")
                rec.update(write_size_kib_raw=raw, write_bytes=raw * 1024, launch_ms_write_pass=ms)
            else:
                g = {k: median_of(disp, k) for k in SQ}
                cyc = g["GRBM_GUI_ACTIVE"] / 8.0                      # summed over the 8 XCDs
                rec.update(launch_ms_sq_pass=ms, clock_ghz=round(cyc / (ms * 1e6), 3) if ms else None,
                           mfma_busy_per_simd_cycle=round(g["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024), 4) if cyc else None,
                           cu_busy=round(g["SQ_BUSY_CU_CYCLES"] / (cyc * 256), 4) if cyc else None,
                           lds_bank_conflict_cycles=g["SQ_LDS_BANK_CONFLICT"],
                           lds_bank_conflict_per_lds_inst_cycle=round(g["SQ_LDS_BANK_CONFLICT"] / max(g["SQ_ACTIVE_INST_LDS"], 1), 4),
                           wait_any_per_wave_cycle=round(g["SQ_WAIT_ANY"] / max(g["SQ_WAVE_CYCLES"], 1), 4),
                           wait_inst_any_per_wave_cycle=round(g["SQ_WAIT_INST_ANY"] / max(g["SQ_WAVE_CYCLES"], 1), 4), raw_sq=g)
        if rec.get("fetch_bytes_corrected") is not None and rec.get("write_bytes") is not None:
            rec["traffic_bytes"] = rec["fetch_bytes_corrected"] + rec["write_bytes"]
            if alg_bytes:
                rec["traffic_over_algorithmic"] = round(rec["traffic_bytes"] / alg_bytes, 3)
            ms = rec.get("launch_ms_sq_pass") or rec.get("launch_ms_fetch_pass")
            if ms:
                rec["fabric_side_TBps"] = round(rec["traffic_bytes"] / (ms * 1e-3) / 1e12, 3)
                if alg_bytes:
                    rec["algorithmic_TBps"] = round(alg_bytes / (ms * 1e-3) / 1e12, 3)
                if flops:
                    rec["TFLOPs_under_counters"] = round(flops / (ms * 1e-3) / 1e12, 2)
        path = os.path.join(out_root, f"{tag}_pmc_{name}.json")
        json.dump(rec, open(path, "w"), indent=1)
        print(json.dumps({k: v for k, v in rec.items() if k not in ("raw_sq", "command", "correction")}), flush=True)


if __name__ == "__main__":
    main()
