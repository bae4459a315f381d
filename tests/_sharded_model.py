"""numpy statement of the row-sharded RSVD's exchange pattern: tests run it under gloo with world_size 2 on CPU to pin
down WHICH quantities are all-reduced.  Test scaffolding only -- the product's sharded path is the C++ drivers over librlhip.so."""
import numpy as np


def rowsharded_rsvd_model(A_local, k, Omega, allreduce):
    """numpy model of the exchange pattern for p = 0, one QB block:
         Y_g = A_g Omega | G = allreduce(Y_g^T Y_g) | R = chol(G) | Q_g = Y_g R^-1 |
         B^T = allreduce(A_g^T Q_g) | SVD(B^T) replicated | U_g = Q_g Uhat
    `allreduce(x)` must return the element-wise sum over ranks."""
    Y = A_local @ Omega
    G = allreduce(Y.T @ Y)
    R = np.linalg.cholesky(G).T
    Q = np.linalg.solve(R.T, Y.T).T
    BT = allreduce(A_local.T @ Q)
    V, S, UT = np.linalg.svd(BT, full_matrices=False)
    U = Q @ UT.T
    return U, S, V
