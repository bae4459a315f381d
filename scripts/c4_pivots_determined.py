#!/usr/bin/env python3
"""BASELINE configs[3] at FULL size (65536 x 65536 fp32, b = 2048) on an input with graded columns (scales over `--decades` decades in a random
order, the construction of the small-size pivot tests): do the pivots survive 32 blocks of 2048?

Four factorizations of the SAME matrix with the SAME sketching operator (Philox key), fast triple {luqr, cholqr, gemqrt}:
  serial     one device, look-ahead off (the reference's order of operations, drivers/rl_bqrrp.hh:318-661)
  lookahead  one device, default (side queue: down-date + next QRCP beside the tail of the apply)
  ranks8     8 row-sharded ranks (block-cyclic) on one device, the real call_sharded path (tests/_world.py), sketch summed over the ranks
  f64        the same fp32 matrix and the serial run's fp32 sketch promoted to fp64, factored in fp64 with tol = eps32: what the decisions
             are when rounding is 9 digits further away
and for every pair: J equal?, fraction of equal positions, first differing position, block-wise overlap of the selected column SETS,
max relative difference of |diag R|.  Precedent: test/drivers/test_bqrrp_gpu.cu:231-249 (J / tau / R agreement across back ends).

usage: c4_pivots_determined.py [--n 65536] [--b 2048] [--decades 16] [--world 8] [--no-f64]
"""
import argparse, json, os, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from randlapack_amd import device as d
from _world import World, block_cyclic_rows

NEVER = 1 << 62


def compare(Ja, Jb, b):
    same = (Ja == Jb)
    n = Ja.numel()
    first = int((~same).nonzero()[0]) if not bool(same.all()) else -1
    ov = []
    for k in range(0, n, b):
        sa, sb = set(Ja[k:k + b].tolist()), set(Jb[k:k + b].tolist())
        ov.append(len(sa & sb) / max(len(sa), 1))
    return dict(J_equal=bool(same.all()), positions_equal=float(same.double().mean()), first_difference=first,
                first_difference_block=(first // b if first >= 0 else -1),
                blocks_with_identical_order=int(sum(bool(same[k:k + b].all()) for k in range(0, n, b))),
                blocks_with_identical_set=int(sum(o == 1.0 for o in ov)), min_block_set_overlap=float(min(ov)), mean_block_set_overlap=float(np.mean(ov)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=65536); ap.add_argument("--b", type=int, default=2048)
    ap.add_argument("--decades", type=float, default=16.0); ap.add_argument("--top", type=float, default=10.0)
    ap.add_argument("--world", type=int, default=8); ap.add_argument("--no-f64", action="store_true")
    a = ap.parse_args()
    n = m = a.n
    b, N = a.b, a.world
    ctx = d.Context(0)
    A = d.cm_empty(m, n, dtype=torch.float32)                         # (n, m) tensor: column j of the matrix is A[j]
    ctx.fill_dense(A, m, n, key=(7, 0))
    g = torch.Generator().manual_seed(1)
    scales = torch.logspace(a.top, a.top - a.decades, n, dtype=torch.float64)[torch.randperm(n, generator=g)]
    A.mul_(scales.to(torch.float32).cuda().unsqueeze(1))
    ctx.sync()
    out = {"workload": f"BQRRP {m}x{n} fp32 b={b}, columns graded over {a.decades} decades (top 1e{a.top:g}) in a random order",
           "triple": "{luqr, cholqr, gemqrt}", "runs": {}, "pairs": {}}
    res = {}

    def run_single(name, thresh):
        W = A.clone()
        with ctx.options(bqrrp_lookahead_min_elems=thresh):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            r = d.drv_bqrrp(ctx, W, m, n, b, 1.0, want_sketch=(name == "serial"), key=(4, 0), qrcp_wide=0, qr_tall=1, apply_trans_q=1)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        res[name] = dict(J=r["J"].clone(), diag=W.diagonal().abs().double().clone(), rank=r["rank"])
        out["runs"][name] = dict(seconds=round(dt, 3), rank=r["rank"])
        return r

    run_single("warm", NEVER); res.pop("warm"); out["runs"].pop("warm")
    rs = run_single("serial", NEVER)
    sk32 = rs["sketch"]
    run_single("lookahead", -1)
    # ---- 8 ranks, block-cyclic rows, on this device
    Wd = World(N)
    rows = [block_cyclic_rows(r, N, m, b) for r in range(N)]
    shards = [A[:, torch.from_numpy(rows[r]).cuda()].contiguous() for r in range(N)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    rr = Wd.run(lambda r, c: d.drv_bqrrp(c, shards[r], len(rows[r]), n, b, 1.0, key=(4, 0), qrcp_wide=0, qr_tall=1, apply_trans_q=1, m_global=m, block_cyclic=True))
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    agree = all(torch.equal(rr[0]["J"], rr[r]["J"]) for r in range(N))
    dg = torch.zeros(n, dtype=torch.float64, device="cuda")
    for r in range(N):                                                 # diagonal entry i lives with the owner of global row i
        gi = torch.from_numpy(rows[r]).cuda()
        dg[gi] = shards[r][gi, torch.arange(len(rows[r]), device="cuda")].abs().double()
    res["ranks8"] = dict(J=rr[0]["J"].clone(), diag=dg, rank=rr[0]["rank"])
    out["runs"]["ranks8"] = dict(seconds_all_ranks=round(dt, 3), seconds_per_rank=round(dt / N, 3), rank=rr[0]["rank"], ranks_agree=bool(agree),
                                 collectives=Wd.collectives, bytes_all_reduced=Wd.bytes_reduced, lookaheads_taken=int(Wd.ctx[0].path_count(12)))
    del shards
    Wd.close()
    # ---- fp64 legs on the same float matrix: rounding is 9 digits further away from every decision
    if not a.no_f64:
        eps32 = float(np.finfo(np.float32).eps)
        A64 = A.double()
        sk64 = sk32.double()

        def run64(name, thresh, sketch):
            W64 = A64.clone()
            with ctx.options(bqrrp_lookahead_min_elems=thresh):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                r = d.drv_bqrrp(ctx, W64, m, n, b, 1.0, sketch_in=sketch, key=(4, 0), qrcp_wide=0, qr_tall=1, apply_trans_q=1, tol=eps32)
                torch.cuda.synchronize(); dt = time.perf_counter() - t0
            res[name] = dict(J=r["J"].clone(), diag=W64.diagonal().abs().clone(), rank=r["rank"])
            out["runs"][name] = dict(seconds=round(dt, 3), rank=r["rank"])
        # (1) the float sketch promoted: the decisions of exact arithmetic on the float problem
        run64("f64", NEVER, sk64)
        # (2) the SCHEDULES in fp64: look-ahead and 8 row-sharded ranks against the serial order, each forming its own fp64 sketch
        run64("f64_serial_own_sketch", NEVER, None)
        run64("f64_lookahead_own_sketch", -1, None)
        Wd = World(N)
        shards = [A64[:, torch.from_numpy(rows[r]).cuda()].contiguous() for r in range(N)]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rr = Wd.run(lambda r, c: d.drv_bqrrp(c, shards[r], len(rows[r]), n, b, 1.0, key=(4, 0), qrcp_wide=0, qr_tall=1, apply_trans_q=1, tol=eps32,
                                            m_global=m, block_cyclic=True))
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        dg = torch.zeros(n, dtype=torch.float64, device="cuda")
        for r in range(N):
            gi = torch.from_numpy(rows[r]).cuda()
            dg[gi] = shards[r][gi, torch.arange(len(rows[r]), device="cuda")].abs()
        res["f64_ranks8_own_sketch"] = dict(J=rr[0]["J"].clone(), diag=dg, rank=rr[0]["rank"])
        out["runs"]["f64_ranks8_own_sketch"] = dict(seconds_all_ranks=round(dt, 3), rank=rr[0]["rank"],
                                                    ranks_agree=bool(all(torch.equal(rr[0]["J"], rr[r]["J"]) for r in range(N))),
                                                    lookaheads_taken=int(Wd.ctx[0].path_count(12)))
        del shards
        Wd.close()
        # (3) margin probe: the SAME fp64 factorization with every sketch entry moved by a relative 1e-7 / 1e-10 (one float rounding / a
        #     thousandth of it): where do the decisions first move?  A decision that a 1e-7 perturbation of the sketch moves is not separated
        #     beyond float rounding, whatever kernel does the rounding
        g2 = torch.Generator(device="cuda").manual_seed(3)
        for tag, rel in (("f64_sketch_perturbed_1e-7", 1e-7), ("f64_sketch_perturbed_1e-10", 1e-10)):
            run64(tag, NEVER, sk64 * (1.0 + rel * torch.randn(sk64.shape, generator=g2, device="cuda", dtype=torch.float64)))
        del A64
    names = list(res)
    for i in range(len(names)):
        for j in range(i + 1, len(names)):
            x, y = res[names[i]], res[names[j]]
            c = compare(x["J"].cpu(), y["J"].cpu(), b)
            c["diagR_max_rel_diff"] = float(((x["diag"] - y["diag"]).abs() / y["diag"].clamp_min(1e-300)).max())
            c["diagR_max_rel_diff_sorted_within_blocks"] = float(max(
                ((x["diag"][k:k + b].sort().values - y["diag"][k:k + b].sort().values).abs() / y["diag"][k:k + b].sort().values.clamp_min(1e-300)).max().item()
                for k in range(0, n, b)))
            out["pairs"][f"{names[i]} vs {names[j]}"] = c
    print(json.dumps(out, indent=1), flush=True)


if __name__ == "__main__":
    main()
