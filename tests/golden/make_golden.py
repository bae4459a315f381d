"""Regenerates the fixtures under tests/golden/.  Run from the repo root: python tests/golden/make_golden.py

philox_kat.json    : Random123 Philox4x32-10 known-answer vectors (public kat_vectors values as quoted in
                     SURVEY.md section 8c) -- these are DATA that pin both the oracle and the HIP generator.
col_swap_kats.json : the reference's exact col_swap cases (test/misc/test_util.cc:215-293,510-547): inputs
                     and the expected outputs implied by its ASSERT_EQ contract.
fill_dense_golden.json : first entries of this library's own sketch stream as produced by the oracle
                     (regression pin only; the stream is 'parity unpinned' w.r.t. RandBLAS).
"""
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
OUT = Path(__file__).resolve().parent


def main():
    kat = [
        dict(ctr=[0, 0, 0, 0], key=[0, 0], out=[0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]),
        dict(ctr=[0xFFFFFFFF] * 4, key=[0xFFFFFFFF] * 2, out=[0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]),
        dict(ctr=[0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], key=[0xA4093822, 0x299F31D0],
             out=[0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]),
    ]
    (OUT / "philox_kat.json").write_text(json.dumps(kat, indent=1))

    cs = {}
    # structured permutations on iota 3x8 (test_util.cc:215-243, 528-537)
    m, n = 3, 8
    A = np.arange(m * n, dtype=np.float64).reshape(n, m).T  # column-major iota
    cases = []
    for name, J in (("identity", list(range(1, n + 1))), ("reversal", [n - i for i in range(n)]),
                    ("rotation", [(i + 1) % n + 1 for i in range(n)])):
        exp = A[:, [j - 1 for j in J]]
        cases.append(dict(name=name, m=m, n=n, J=J, A=A.T.ravel().tolist(), expect=exp.T.ravel().tolist()))
    cs["structured"] = cases
    # lda case (test_util.cc:245-266, 539-541): m=4, lda=7, n=6, J={3,1,6,2,5,4}, A = iota over lda*n
    m, lda, n = 4, 7, 6
    J = [3, 1, 6, 2, 5, 4]
    buf = np.arange(lda * n, dtype=np.float64)
    exp = buf.copy()
    for j in range(n):
        exp[j * lda:j * lda + m] = buf[(J[j] - 1) * lda:(J[j] - 1) * lda + m]
    cs["lda"] = dict(m=m, lda=lda, n=n, J=J, A=buf.tolist(), expect=exp.tolist())
    # integer vector prefix contract (test_util.cc:268-293, 543-547): A = iota from 100, J = perm of 1..k
    iv = []
    rng = np.random.default_rng(20260928)
    for (nn, k) in ((7, 7), (200, 200), (2800, 100)):
        Jk = (rng.permutation(k) + 1).tolist()
        vec = list(range(100, 100 + nn))
        exp = [vec[Jk[i] - 1] for i in range(k)] + vec[k:]
        iv.append(dict(n=nn, k=k, J=Jk, A=vec, expect=exp))
    cs["int_vector"] = iv
    (OUT / "col_swap_kats.json").write_text(json.dumps(cs))

    import oracle
    g, nxt = oracle.fill_dense(5, 3, ctr=(0, 0, 0, 0), key=(0, 0), dist=0)
    u, nxt_u = oracle.fill_dense(5, 3, ctr=(0, 0, 0, 0), key=(42, 0), dist=1)
    (OUT / "fill_dense_golden.json").write_text(json.dumps(dict(
        gaussian=dict(rows=5, cols=3, ctr=[0, 0, 0, 0], key=[0, 0], values=g.T.ravel().tolist(), next_ctr=list(nxt)),
        uniform=dict(rows=5, cols=3, ctr=[0, 0, 0, 0], key=[42, 0], values=u.T.ravel().tolist(), next_ctr=list(nxt_u)),
    ), indent=1))
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    main()
