#!/usr/bin/env python3
"""Static audit of the gfx950 assembly hipcc produced for the library's kernels (hipcc --cuda-device-only -S file.hip): per kernel the VGPR
count, the spill count, and -- what costs time -- every LOOP block that holds a scratch (spill) access or a compiler-inserted
`s_waitcnt vmcnt(0)` next to LDS-DMA requests.  Round 5: the stream-K GEMM reloaded a spilled LDS offset between the DMA requests of every
K-tile (a full HBM round trip per tile, 4.4 ms of the 66.4 ms headline); this script is how the other kernels were checked.
usage: asm_audit.py file.s [...]"""
import re, sys

def audit(path):
    lines = open(path).read().split("\n")
    meta = {}
    name = None
    for l in lines:
        m = re.match(r"\s+\.name:\s+(\S+)", l)
        if m: name = m.group(1); meta[name] = {}
        for k in ("vgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size"):
            m = re.match(r"\s+\." + k + r":\s+(\d+)", l)
            if m and name: meta[name][k] = int(m.group(1))
    fn = None; cur = None; rows = {}
    for l in lines:
        m = re.match(r"^(_Z\w+):", l)
        if m: fn = m.group(1); cur = None; continue
        m = re.match(r"^(\.LBB\d+_\d+):(.*)", l)
        if m:
            depth = re.search(r"Depth=(\d+)", m.group(2))
            cur = (m.group(1), int(depth.group(1)) if depth else 0, [])
            rows.setdefault(fn, []).append(cur); continue
        if cur is not None: cur[2].append(l.strip().split(";")[0].strip())
    out = []
    for fn, blocks in rows.items():
        if fn not in meta: continue
        hot = []
        for (lab, depth, body) in blocks:
            if depth < 1: continue
            sc = sum(1 for x in body if x.startswith("scratch_"))
            if sc: hot.append(f"{lab}(d{depth}): {sc} scratch")
        mm = meta[fn]
        if mm.get("vgpr_spill_count", 0) or hot:
            out.append((fn, mm.get("vgpr_count"), mm.get("vgpr_spill_count"), mm.get("private_segment_fixed_size"), hot))
    return out

if __name__ == "__main__":
    import subprocess
    for p in sys.argv[1:]:
        for fn, v, sp, sz, hot in audit(p):
            try: dn = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", fn], capture_output=True, text=True).stdout.strip()[:110]
            except Exception: dn = fn[:110]
            print(f"{p.split('/')[-1]:12s} vgpr {v:3d} spill {sp:3d} scratch {sz:4d}B  {dn}")
            for h in hot[:6]: print(f"{'':14s}in a loop: {h}")
