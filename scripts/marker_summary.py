#!/usr/bin/env python3
"""Phase attribution from the library's roctx ranges (rlhip_range_push / _pop; names = the reference's NVTX ranges, drivers/rl_bqrrp_gpu.hh:335-403):

    cd /tmp && rocprofv3 --marker-trace --kernel-trace --output-format csv -d OUT -- python scripts/bq_prof.py 65536 2048 f32
    python scripts/marker_summary.py OUT

For every range name: number of ranges, total host time inside them, and the device time of the kernels whose START lies inside a range of that
name (innermost range wins) -- so a timeline is attributed to phases without a per-script table of kernel names.  With the drivers' subroutine
timers on (every lap drains the stream, as in bq_prof.py) the two columns agree; without them the host column is enqueue time."""
import csv, glob, sys
from collections import defaultdict


def main(out):
    marks = []
    for f in glob.glob(out + "/**/*marker_api_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            marks.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Function") or r.get("Name") or "?"))
    kernels = []
    for f in glob.glob(out + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            kernels.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    if not marks:
        print("no marker ranges found under", out)
        return 1
    host = defaultdict(lambda: [0, 0])
    for s, e, n in marks:
        host[n][0] += 1; host[n][1] += e - s
    marks.sort(key=lambda t: (t[0], -t[1]))
    dev = defaultdict(lambda: [0, 0])
    for ks, ke in kernels:
        inner = None
        for s, e, n in marks:                      # (small inputs: a linear scan is fine)
            if s > ks: break
            if e >= ks and (inner is None or s >= inner[0]): inner = (s, e, n)
        name = inner[2] if inner else "(outside every range)"
        dev[name][0] += 1; dev[name][1] += ke - ks
    print(f"{'range':28s} {'count':>7s} {'host ms':>12s} {'kernels':>9s} {'device ms':>12s}")
    for n in sorted(set(host) | set(dev), key=lambda k: -dev[k][1]):
        print(f"{n:28s} {host[n][0]:7d} {host[n][1] / 1e6:12.2f} {dev[n][0]:9d} {dev[n][1] / 1e6:12.2f}")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
