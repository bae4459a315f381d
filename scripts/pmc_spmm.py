import numpy as np, torch
from randlapack_amd import device as d
m, n, r, b = 1_000_000, 1000, 10, 1000
ctx = d.Context(0)
rng = np.random.default_rng(0)
cols = np.sort(rng.integers(0, n, size=(m, r)), axis=1).astype(np.int64).ravel()
rp = torch.as_tensor(np.arange(m + 1, dtype=np.int64) * r, device="cuda:0")
ci = torch.as_tensor(cols, device="cuda:0")
v = torch.as_tensor(rng.standard_normal(m * r), device="cuda:0")
nnz = m * r
rpt = torch.zeros(n + 1, dtype=torch.int64, device="cuda:0"); cit = torch.zeros(nnz, dtype=torch.int64, device="cuda:0"); vt = torch.zeros(nnz, dtype=torch.float64, device="cuda:0")
ctx.lib.rlhip_csr_transpose_f64(ctx.h, m, n, rp.data_ptr(), ci.data_ptr(), v.data_ptr(), rpt.data_ptr(), cit.data_ptr(), vt.data_ptr())
B = torch.randn(n * b, dtype=torch.float64, device="cuda:0"); C = torch.zeros(m * b, dtype=torch.float64, device="cuda:0")
Ct = torch.zeros(n * b, dtype=torch.float64, device="cuda:0")
for _ in range(3):
    ctx.lib.rlhip_csr_spmm_f64(ctx.h, b"R", m, b, n, 1.0, rp.data_ptr(), ci.data_ptr(), v.data_ptr(), B.data_ptr(), b, 0.0, C.data_ptr(), b)      # forward  A * M
    ctx.lib.rlhip_csr_spmm_f64(ctx.h, b"R", n, b, m, 1.0, rpt.data_ptr(), cit.data_ptr(), vt.data_ptr(), C.data_ptr(), b, 0.0, Ct.data_ptr(), b)   # adjoint  A^T * X
torch.cuda.synchronize()
