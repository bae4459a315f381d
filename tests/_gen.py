"""Test-matrix generators, restating the constructions of the reference's RandLAPACK/testing/rl_gen.hh
(gen_singvec :62-101, gen_poly_singvals :105-132) in numpy.  Random factors come from numpy, not from the
(absent) RandBLAS stream -- the reference's tests only need *some* orthonormal factors."""
import numpy as np


def poly_singvals(k, frac_spectrum_one=0.1, cond=1e6, p=2.0):
    # rl_gen.hh:112-131: s_i = 1 / (a (i + b)^p) past the first floor(k*frac) ones
    s = np.ones(k)
    offset = int(np.floor(k * frac_spectrum_one))
    first, last = 1.0, 1.0 / cond
    neg_invp = -1.0 / p
    a = ((last**neg_invp - first**neg_invp) / (k - offset)) ** p
    b = (a * first) ** neg_invp - offset
    idx = np.arange(offset, k, dtype=np.float64)
    s[offset:] = 1.0 / (a * (idx + b) ** p)
    return s


def with_singvals(m, n, s, rng):
    # rl_gen.hh:62-101: A = U diag(s) V^T with U, V orthonormalised Gaussians
    k = len(s)
    U = np.linalg.qr(rng.standard_normal((m, k)))[0]
    V = np.linalg.qr(rng.standard_normal((n, k)))[0]
    return (U * s) @ V.T


def poly_mat(m, n, k, rng, cond=1e6, p=2.0):
    return with_singvals(m, n, poly_singvals(k, 0.1, cond, p), rng)
