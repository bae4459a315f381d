"""-m gpu: the row-sharded C++ drivers with world_size 2 and 3 on ONE GPU (ranks share cuda:0; the library's
all-reduce hook exchanges over gloo).  This runs the real sharded code path -- every Queue::allreduce_sum in
CholQRQ / RS / QB / RSVD -- which the round-end 8-GPU bench uses with RCCL as the transport."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,m,n,k,p", [(2, 3001, 256, 32, 0), (3, 2000, 300, 40, 2), (2, 4096, 512, 64, 1),
                                            (3, 210, 160, 16, 0)])  # last: row blocks shorter than n -> BQRRP top blocks straddle ranks, ranks run dry
def test_rowsharded_drivers_match_single_device_and_oracle(world, m, n, k, p):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "_sharded_worker.py"), str(m), str(n), str(k), str(p)]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    line = [ln for ln in res.stdout.splitlines() if ln.startswith("SHARDED_RESULT ")]
    assert res.returncode == 0 and line, res.stdout[-2000:] + res.stderr[-4000:]
    out = json.loads(line[-1][len("SHARDED_RESULT "):])
    assert out["k"] == k and out["k2"] == k
    assert out["S_vs_single"] <= 1e-12          # same arithmetic up to the order of the cross-rank sums
    assert out["S_vs_oracle"] <= 1e-10
    assert out["recon"] <= out["recon_ref"] * (1 + 1e-6) + 1e-12
    assert out["orthU"] <= 1e-10 and out["orthV"] <= 1e-10
    assert out["recon2"] <= out["recon_ref"] * 1.5 + 1e-12 and out["orthU2"] <= 1e-9
    # row-sharded CQRRPT == single-device CQRRPT (same SASO, same pivots; R to rounding)
    assert out["cq_rank"] == out["cq_rank1"] and out["cq_J_equal"]
    assert out["cq_R"] <= 1e-10 and out["cq_resid"] <= 1e-12 and out["cq_orth"] <= 1e-11
    # row-sharded BQRRP == single-device BQRRP: same pivots, GEQP3-format output (V, R, tau) equal to rounding, valid factorization --
    # with the reference's DEFAULT subroutines {luqr, geqrf, ormqr} (sharded: TSQR panels) and with the fast triple {luqr, cholqr, gemqrt}
    assert out["bq_rank"] == out["bq_rank1"] and out["bq_J_equal"]
    assert out["bq_A"] <= 1e-10 and out["bq_tau"] <= 1e-10
    assert out["bqf_rank"] == out["bq_rank1"] and out["bqf_J_equal"]
    assert out["bqf_A"] <= 1e-10 and out["bqf_tau"] <= 1e-10
    # ... and with the sharded look-ahead forced (side queue + this process group's collectives): every rank took it in every iteration but the last
    assert out["bql_rank"] == out["bq_rank1"] and out["bql_J_equal"]
    assert out["bql_A"] <= 1e-10 and out["bql_tau"] <= 1e-10
    assert all(c >= 1 for c in out["bql_lookaheads"]), out["bql_lookaheads"]
    # row-sharded standalone hqrrp (pivoted / Householder / Cholesky-QR panels) == the single-device factorization
    for tag, h in out["hqrrp"].items():
        assert h["rc"] == [0, 0] and h["J_equal"], (tag, h)
        assert h["A"] <= 1e-10 and h["tau"] <= 1e-10 and h["resid"] <= 1e-12 and h["orth"] <= 1e-11, (tag, h)
    # linop QR drivers and ABRIK on a row-sharded CSR operator: same R / Ritz values as on one device
    assert all(v <= 1e-10 for v in out["lin_R"].values()), out["lin_R"]
    assert out["sp_abrik_trip"][0] == out["sp_abrik_trip"][1] and out["sp_abrik_S"] <= 1e-9
    # block-cyclic row layout: same pivots, reflectors and R as the single-device run
    assert out["bqc_rank"] == out["bq_rank1"] and out["bqc_J_equal"]
    assert out["bqc_A"] <= 1e-10 and out["bqc_tau"] <= 1e-10
    assert out["bq_resid"] <= 1e-12 and out["bq_orth"] <= 1e-11
    # row-sharded ABRIK with the reference's DEFAULT panels (geqrf_ungqr): same trajectory, Ritz values and RNG state as on one device
    assert out["abh_iters"][0] == out["abh_iters"][1] and out["abh_trip"][0] == out["abh_trip"][1] and out["abh_next_ctr"][0] == out["abh_next_ctr"][1]
    assert out["abh_S_vs_single"] <= 1e-9 and out["abh_orthU"] <= 1e-9 and out["abh_res"] <= 1e-9
    # row-sharded ABRIK (CQRRT panels): same iteration count, same leading Ritz values as on one device
    assert (out["ab_iters"], out["ab_trip"]) == (out["ab_iters1"], out["ab_trip1"])
    assert out["ab_S_vs_single"] <= 1e-9 and out["ab_orthU"] <= 1e-9 and out["ab_res"] <= 1e-9


@pytest.mark.parametrize("world,m,n,k,p", [(2, 3000, 400, 32, 2), (3, 2400, 300, 40, 2), (3, 5000, 256, 64, 3)])
def test_rowsharded_stabilisers_power_scheme_and_replicated_qrcp(world, m, n, k, p):
    """Row-sharded PLUL (tournament-pivoted LU, one exchange) and HQRQ (TSQR, one exchange) -- the stabilisers the reference's own RSVD / QB
    object graphs put inside the power scheme (test/drivers/test_rsvd.cc:70, test/comps/test_qb.cc:59) -- at world 2 and 3 with p >= 2:
    valid bases of the same column space, and the RSVD they feed delivers the single-device singular values to 1e-10.  Also: CQRRPT with
    the hqrrp / bqrrp QRCP (replicated on every rank: identical pivots) and with the orthonormal completion, on a sharded queue."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "_sharded_stab_worker.py"), str(m), str(n), str(k), str(p)]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    line = [ln for ln in res.stdout.splitlines() if ln.startswith("STAB_RESULT ")]
    assert res.returncode == 0 and line, res.stdout[-2000:] + res.stderr[-4000:]
    out = json.loads(line[-1][len("STAB_RESULT "):])
    assert out["hqrq_rc"] == 0 and out["plul_rc"] == 0
    assert out["hqrq_orth"] <= 1e-12 and out["hqrq_span"] <= 1e-11              # orthonormal basis of span(Y)
    assert out["plul_span"] <= 1e-10 and out["plul_unit_rows"] == k              # same span; the k pivot rows form a unit lower triangle
    assert out["plul_max"] <= 50.0 and out["plul_cond"] <= 1e4                   # tournament pivoting: bounded multipliers, a well-conditioned basis
    for name in ("cholqrq", "hqrq", "plul"):
        r = out[f"rsvd_{name}"]
        assert r["k"] == k, (name, r)
        assert r["S_vs_single"] <= 1e-10 and r["S_vs_exact"] <= 1e-9, (name, r)
        assert r["orthU"] <= 1e-9 and r["recon"] <= 2e-7, (name, r)          # (the 1e-9 noise floor of the test matrix)
    for name in ("hqrrp", "bqrrp"):
        c = out[f"cq_{name}"]
        assert c["rank"][0] == c["rank"][1] and c["J_equal"], (name, c)
        assert c["R"] <= 1e-10 and c["resid"] <= 1e-12 and c["orth"] <= 1e-11, (name, c)
    co = out["cq_orth"]
    assert co["rank"] < co["ncols"] and co["orth_all"] <= 1e-10 and co["resid"] <= 1e-11, co


def test_bench_multi_rank_code_path_on_one_gpu():
    """bench.py --gpus 2 end to end (row sharding, barrier + max-over-ranks timing, JSON line) with two ranks sharing the GPU and
    gloo as the process group; the driver's real runs differ only in the transport (RCCL)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"]
    env = dict(os.environ)
    env["RLHIP_BENCH_BACKEND"] = "gloo"
    env["RLHIP_BENCH_SHAPE"] = "16384,2048,64"
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert res.returncode == 0 and len(lines) == 1, res.stdout[-1500:] + res.stderr[-3000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["scaling"] == "strong" and out["value"] > 0
    assert out["config"]["k_out"] == 64 and out["config"]["collectives"] == "torch.distributed"
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(out["roofline"])
