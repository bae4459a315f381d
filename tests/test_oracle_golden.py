"""Pins the CPU oracle against every exact vector that exists for this path:
Random123's Philox4x32-10 KATs and the reference's col_swap KATs (test/misc/test_util.cc:195-312,510-551)."""
import json
from pathlib import Path

import numpy as np
import pytest

G = Path(__file__).resolve().parent / "golden"


def test_philox_kat(orc):
    for v in json.loads((G / "philox_kat.json").read_text()):
        out = orc.philox(v["ctr"], v["key"])
        assert [int(x) for x in out] == v["out"]


def test_fill_dense_golden_and_state_rule(orc):
    g = json.loads((G / "fill_dense_golden.json").read_text())
    for name, dist in (("gaussian", 0), ("uniform", 1)):
        c = g[name]
        buf, nxt = orc.fill_dense(c["rows"], c["cols"], ctr=c["ctr"], key=c["key"], dist=dist)
        np.testing.assert_allclose(buf.T.ravel(), np.array(c["values"]), rtol=0, atol=1e-15)
        assert list(nxt) == c["next_ctr"]
    # next state = ctr + ceil(rows*cols/4); the stream is position-addressable: two half fills == one fill
    full, nxt = orc.fill_dense(8, 4)
    a, mid = orc.fill_dense(8, 2)
    b, end = orc.fill_dense(8, 2, ctr=mid)
    assert nxt == (8, 0, 0, 0) and mid == (4, 0, 0, 0) and end == nxt
    np.testing.assert_array_equal(np.hstack([a, b]), full)
    # counter carry into the second word
    _, nxt = orc.fill_dense(4, 4, ctr=(0xFFFFFFFE, 0, 0, 0))
    assert nxt == (2, 1, 0, 0)


def test_fill_dense_moments(orc):
    g, _ = orc.fill_dense(2000, 100, key=(3, 0))
    assert abs(g.mean()) < 0.01 and abs(g.std() - 1) < 0.01
    assert abs(np.mean(g**3)) < 0.03 and abs(np.mean(g**4) - 3) < 0.1
    u, _ = orc.fill_dense(2000, 100, key=(3, 0), dist=1)
    assert u.min() > -1 and u.max() < 1 and abs(u.mean()) < 0.01 and abs(u.var() - 1 / 3) < 0.01


def test_col_swap_structured(orc):
    cs = json.loads((G / "col_swap_kats.json").read_text())
    for c in cs["structured"]:
        A = np.array(c["A"]).reshape(c["n"], c["m"]).T
        rc, B, idx = orc.col_swap(A, c["J"])
        assert rc == 0 and list(idx) == c["J"]  # pivot vector restored
        np.testing.assert_array_equal(B.T.ravel(), np.array(c["expect"]))


def test_col_swap_respects_lda(orc):
    c = json.loads((G / "col_swap_kats.json").read_text())["lda"]
    rc, buf, idx = orc.col_swap_lda(c["A"], c["m"], c["lda"], c["n"], c["J"])
    assert rc == 0 and list(idx) == c["J"]
    np.testing.assert_array_equal(buf, np.array(c["expect"]))


def test_col_swap_int_vector(orc):
    for c in json.loads((G / "col_swap_kats.json").read_text())["int_vector"]:
        rc, v, idx = orc.col_swap_int(c["A"], c["J"])
        assert rc == 0 and list(idx) == c["J"]
        assert list(v) == c["expect"]


@pytest.mark.parametrize("m,n,k,seed", [(10, 7, 7, 0), (10, 7, 4, 1), (1000, 200, 200, 2), (8, 12, 5, 3), (5, 1, 1, 5),
                                        (6, 9, 1, 6)])
def test_col_swap_gather_contract_and_lapmt(orc, m, n, k, seed):
    # test_util.cc:195-213 with params :510-526; permutations from numpy (the contract is the property)
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((m, n))
    J = rng.permutation(n) + 1
    rc, B, idx = orc.col_swap(A, J, k)
    assert rc == 0
    np.testing.assert_array_equal(idx, J)
    np.testing.assert_array_equal(B[:, :k], A[:, J[:k] - 1])
    # the restatement agrees with LAPACK's own lapmt (what the reference literally calls, rl_util.hh:163)
    B2, idx2 = orc.lapmt(A, J)
    np.testing.assert_array_equal(B, B2)
    np.testing.assert_array_equal(idx2, J)


def test_col_swap_rejects_k_gt_n(orc):
    rc, _, _ = orc.col_swap(np.zeros((3, 4)), [1, 2, 3, 4], k=5)
    assert rc != 0  # reference throws std::runtime_error (rl_util.hh:159-160)


def test_col_swap_large_single_cycle_is_linear(orc):
    # efficiency canary of test_util.cc:295-312 (n = 1e6, one n-cycle, must stay O(n))
    import time

    n = 1_000_000
    A = np.arange(n, dtype=np.float64).reshape(1, n)
    J = (np.arange(n) + 1) % n + 1
    t0 = time.time()
    rc, B, _ = orc.col_swap(A, J)
    assert time.time() - t0 < 10.0
    np.testing.assert_array_equal(B[0], (np.arange(n) + 1) % n)


def test_numpy_philox_restatement_matches_kats_and_cpp(orc):
    """the vectorised numpy Philox4x32-10 used by the SASO restatement: Random123 KATs + the C++ restatement on carries"""
    for v in json.loads((G / "philox_kat.json").read_text()):
        assert [int(x) for x in orc.philox_np([v["ctr"]], v["key"])[0]] == v["out"]
    for base in [(0xFFFFFFFD, 7, 0, 0), (0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 3)]:
        ctrs = orc._ctr_array(base, range(6))
        out = orc.philox_np(ctrs, (11, 13))
        for i in range(6):
            assert list(out[i]) == list(orc.philox(tuple(int(x) for x in ctrs[i]), (11, 13)))


@pytest.mark.parametrize("mode", [0, 1])
def test_saso_restatement_structure(orc, mode):
    """d x m operator with nnz nonzeros per column: distinct rows, +-1 values, documented state advance; the
    block-affine structure additionally gives every sketch row exactly nnz sources per full block of d input rows"""
    for (d, m, nnz) in [(40, 1000, 4), (25, 333, 2), (64, 64, 8), (7, 50, 7), (1, 5, 1)]:
        S, nxt = orc.saso_dense(d, m, nnz, (7, 0, 0, 0), (5, 0), mode)
        assert S.shape == (d, m) and set(np.unique(S)) <= {-1.0, 0.0, 1.0}
        assert set((S != 0).sum(0)) == {nnz}
        T = (m + d - 1) // d
        assert nxt == ((7 + T + m) if mode == 0 else (7 + m * ((nnz + 1) // 2)), 0, 0, 0)
        if mode == 0:
            for t in range(m // d):
                assert set((S[:, t * d:(t + 1) * d] != 0).sum(1)) == {nnz}
        if m * nnz >= 2000:
            assert abs(S.sum()) < 6 * np.sqrt(m * nnz)                              # balanced signs
    # different keys / counters give different operators; same inputs the same operator
    a, _ = orc.saso_dense(16, 64, 2, (0, 0, 0, 0), (1, 0), mode)
    b, _ = orc.saso_dense(16, 64, 2, (0, 0, 0, 0), (2, 0), mode)
    c, _ = orc.saso_dense(16, 64, 2, (0, 0, 0, 0), (1, 0), mode)
    assert not np.array_equal(a, b) and np.array_equal(a, c)
