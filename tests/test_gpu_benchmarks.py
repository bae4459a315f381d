"""-m gpu: the benchmark mains (benchmarks/*.py) run at toy sizes and write the reference's file formats
(benchmark/bench_BQRRP/*.cc, bench_CQRRPT/*.cc): `num_info_lines` header lines, then comma-separated rows."""
import re

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rows(path):
    lines = open(path).read().rstrip("\n").split("\n")
    assert lines[0].startswith("Description:")
    # the header is the reference's, verbatim in structure: it ends with the "Additional parameters" line (6 lines; the
    # "num_info_lines_7" in two of the file names counts one more than the reference itself writes)
    last = max(i for i, ln in enumerate(lines) if ln.startswith("Additional parameters"))
    assert last == 5
    body = lines[last + 1:]
    return [[x for x in re.split(r",\s*", ln.strip()) if x] for ln in body]


def test_bqrrp_mains(tmp_path):
    from benchmarks import bqrrp

    p = bqrrp.speed_mat_size([str(tmp_path), "2", "1", "8", "512", "768"])
    rows = _rows(p)
    assert rows[-1][0].startswith("Total benchmark execution time:")
    data = rows[:-1]
    assert len(data) == 4 and all(len(r) == 7 and all(int(x) > 0 for x in r) for r in data)      # 2 sizes x 2 runs, 7 algorithms
    p = bqrrp.runtime_breakdown([str(tmp_path), "cholqr", "2", "1024", "512", "64", "128"])
    data = _rows(p)[:-1]
    assert len(data) == 4 and all(len(r) == 10 for r in data)
    for r in data:
        t = [int(x) for x in r]
        assert t[9] > 0 and abs(sum(t[:9]) - t[9]) <= 2                      # the columns add up to the total
    p1, p2 = bqrrp.pivot_quality([str(tmp_path), "600", "256", "32", "polynomial"])
    r1 = _rows(p1)
    assert len(r1) == 1 and len(r1[0]) == 256
    ratios = np.array([float(x) for x in r1[0]])
    assert abs(ratios[0] - 1.0) < 1e-12 and np.all((ratios[:200] > 0.2) & (ratios[:200] < 5))   # BQRRP's R tracks QP3's
    r2 = _rows(p2)
    assert len(r2) == 2 and all(len(r) == 256 for r in r2)
    q = np.array([[float(x) for x in r] for r in r2])
    assert np.all((q[:, :200] > 0.05) & (q[:, :200] < 20))                   # |R_ii| within a modest factor of sigma_i


def test_bqrrp_error_analysis_main(tmp_path):
    from benchmarks import bqrrp

    p = bqrrp.error_analysis([str(tmp_path), "bqrrp", "2", "512", "512", "64"])
    lines = open(p).read().rstrip("\n").split("\n")
    assert lines[0].startswith("Description:") and len(lines) == 5 + 4       # 5 info lines, 4 matrix types
    for ln in lines[5:]:
        v = [float(x) for x in ln.rstrip(", ").split(",")]
        assert len(v) == 4 and v[0] < 1e-12 and v[2] < 1e-11 and v[1] >= 0 and v[3] >= 0      # backward stable on every matrix type
    p2 = bqrrp.error_analysis([str(tmp_path / "g"), "geqp3", "1", "400", "300", "32"]) if (tmp_path / "g").mkdir() is None else None
    rows = open(p2).read().rstrip("\n").split("\n")[5:]
    assert len(rows) == 3 and all(float(r.split(",")[0]) < 1e-13 for r in rows)               # no Kahan row for a non-square input


def test_cqrrpt_mains(tmp_path):
    from benchmarks import cqrrpt

    p = cqrrpt.speed([str(tmp_path), "1", "4096", "64", "128"])
    data = _rows(p)[:-1]
    assert len(data) == 2 and all(len(r) == 8 and all(int(x) > 0 for x in r) for r in data)
    p = cqrrpt.runtime_breakdown([str(tmp_path), "2", "4096", "128"])
    data = _rows(p)[:-1]
    assert len(data) == 2 and all(len(r) == 8 for r in data)
    p1, p2 = cqrrpt.pivot_quality([str(tmp_path), "2000", "128", "polynomial"])
    r1, r2 = _rows(p1), _rows(p2)
    assert len(r1) == 1 and len(r1[0]) == 128 and len(r2) == 2
    ratios = np.array([float(x) for x in r1[0]])
    assert abs(ratios[0] - 1.0) < 1e-10


def test_abrik_and_cqrrt_linops_mains(tmp_path):
    from benchmarks import abrik, cqrrt_linops

    p = abrik.speed([str(tmp_path), "polynomial", "1", "1500", "400", "10", "2", "2", "8", "16", "4", "8"])
    rows = _rows(p)
    data = rows[:-1]
    assert len(data) == 4 and all(len(r) == 15 for r in data)
    for r in data:
        v = [float(x) for x in r]
        assert v[5] > 0 and v[8] > 0 and v[14] > 0                       # times
        assert v[12] < 1e-8                                               # the full SVD has no residual
    best = min(float(r[4]) for r in data)
    assert best < 1e-3                                                    # with 8 matmuls of block 16 ABRIK nails the leading 10 triplets
    p = cqrrt_linops.basic([str(tmp_path), "2", "1", "20000", "40000", "100", "8", "2.0", "2", "0"])
    lines = [ln for ln in open(p).read().splitlines() if not ln.startswith("#")]
    assert lines[0].startswith("m,n,run,aspect_ratio") and len(lines) == 3
    for ln in lines[1:]:
        f = ln.split(",")
        assert len(f) == len(lines[0].split(","))
        assert float(f[6]) < 1e-10 and int(f[8]) == 1                    # CQRRT_linops: orthonormal Q
        assert float(f[14]) < 1e-10 and float(f[18]) < 1e-10              # sCholQR3 and dense CQRRT too
        assert int(f[7]) == int(f[1])                                     # every column of the prefix test passes


def test_bqrrp_block_size_main(tmp_path):
    from benchmarks import bqrrp

    p = bqrrp.speed_block_size([str(tmp_path), "1", "768", "512", "64", "128", "256"])
    rows = _rows(p)
    assert rows[-1][0].startswith("Total benchmark execution time:")
    data = rows[:-1]
    assert len(data) == 3 and all(len(r) == 7 and all(int(x) > 0 for x in r) for r in data)      # 3 block sizes x 1 run, 7 algorithms


def test_cqrrpt_error_analysis_main(tmp_path):
    from benchmarks import cqrrpt

    p = cqrrpt.error_analysis([str(tmp_path), "cqrrpt", "2", "4096", "64", "128"])
    lines = open(p).read().rstrip("\n").split("\n")
    assert lines[0].startswith("Description:") and lines[3].startswith("Input size:4096 by 64, 128, ")
    body = lines[4:]
    assert len(body) == 2 * 3                                         # two column sizes x (polynomial, staircase, spiked); no Kahan row (m != n)
    for ln in body:
        v = [float(x) for x in ln.rstrip(", ").split(",")]
        # CQRRPT's guarantees: reconstruction error at the working precision, Q orthonormal to eps * cond(preconditioned A)
        assert len(v) == 4 and v[0] < 1e-12 and v[2] < 1e-10 and v[1] >= 0 and v[3] >= 0
    (tmp_path / "g").mkdir()
    p2 = cqrrpt.error_analysis([str(tmp_path / "g"), "geqp3", "1", "300", "300"])
    rows = open(p2).read().rstrip("\n").split("\n")[4:]
    assert len(rows) == 4 and all(float(r.split(",")[0]) < 1e-13 for r in rows)                   # square input: the Kahan row is there


@pytest.mark.parametrize("m_type", ["gaussian", "sparse:0.01"])
def test_abrik_runtime_breakdown_main(tmp_path, m_type):
    from benchmarks import abrik

    p = abrik.runtime_breakdown([str(tmp_path), m_type, "2", "1500", "1000", "20", "2", "2", "8", "16", "4", "8"])
    lines = open(p).read().rstrip("\n").split("\n")
    # header as the reference writes it: 5 lines (the "File format" and "rows correspond" strings are concatenated without a line break)
    assert lines[0].startswith("Description:") and lines[4].startswith("Additional parameters")
    data = [[x for x in re.split(r",\s*", ln.strip()) if x] for ln in lines[5:]]
    assert len(data) == 2 * 2 * 2 and all(len(r) == 15 for r in data)          # block size, matmuls, 13 timers
    for r in data:
        t = [int(x) for x in r]
        assert t[0] in (8, 16) and t[1] in (4, 8)
        alloc, factors, ungqr, reorth, qr, gemm_a, main_loop, sketch, r_cpy, s_cpy, norm, rest, total = t[2:]
        assert total > 0 and gemm_a > 0 and qr > 0 and factors > 0
        assert alloc + factors + ungqr + reorth + qr + gemm_a + sketch + r_cpy + s_cpy + norm + rest == total   # the reference's identity (:732)
        assert 0 < main_loop <= total and rest >= 0


def test_bqrrp_subroutines_speed_main(tmp_path):
    """BQRRP_subroutines_speed.cc: three blocks of rows (wide QRCP, tall QR, apply Q^T) in the reference's order and widths"""
    from benchmarks import bqrrp

    p = bqrrp.subroutines_speed([str(tmp_path), "2", "2048", "64", "128"])
    lines = open(p).read().rstrip("\n").split("\n")
    assert lines[0].startswith("Description:") and any(ln.startswith("Additional parameters num runs per size 2 nb_start 64") for ln in lines[:10])
    body = [[x for x in re.split(r",\s*", ln.strip()) if x] for ln in lines if ln and ln[0].isdigit()]
    wide, rest = body[:4], body[4:]                                   # 2 sizes x 2 runs, (GEQP3, LUQR)
    assert all(len(r) == 2 and all(int(x) > 0 for x in r) for r in wide)
    tall, apply_q = rest[:4], rest[4:8]
    # n = 64: nb = 64 only -> 6 + 1 columns; n = 128: nb = 64, 128 -> 6 + 2 columns
    assert [len(r) for r in tall] == [7, 7, 8, 8] and all(int(x) > 0 for r in tall for x in r)
    assert [len(r) for r in apply_q] == [2, 2, 3, 3] and all(int(x) > 0 for r in apply_q for x in r)
    assert lines[-1].startswith("Total benchmark execution time:")


def test_hqrrp_runtime_breakdown_main(tmp_path):
    """HQRRP_runtime_breakdown.cc: 27 numbers per run; the top-level entries add up to the total (rl_hqrrp.hh:1150-1158)"""
    from benchmarks import bqrrp

    p = bqrrp.hqrrp_runtime_breakdown([str(tmp_path), "2", "1024", "768", "64", "128"])
    lines = open(p).read().rstrip("\n").split("\n")
    # 7 header lines, as the file name says (this main's header has a line break before "rows correspond", HQRRP_runtime_breakdown.cc:147-149)
    assert lines[0].startswith("Description:") and lines[6].startswith("Additional parameters: HQRRP block sizes: 64, 128, ")
    rows = [[x for x in re.split(r",\s*", ln.strip()) if x] for ln in lines[7:]]
    assert rows[-1][0].startswith("Total benchmark execution time:")
    data = rows[:-1]
    assert len(data) == 4 and all(len(r) == 27 for r in data)
    for r in data:
        t = [float(x) for x in r]
        assert t[8] > 0 and t[1] > 0 and t[3] > 0 and t[4] > 0 and t[5] > 0
        assert abs(sum(t[:8]) - t[8]) <= 1e-6 * t[8] + 1                  # preallocation + ... + other == total
        assert t[17] == t[3] and t[26] == t[4]                            # the kernels' own totals


def test_abrik_speed_sparse_main(tmp_path):
    """ABRIK_speed_comparisons_sparse.cc: 7 columns; ABRIK on the CSR operator reaches the host SVDS's residual level"""
    from benchmarks import abrik

    p = abrik.speed_sparse([str(tmp_path), "sparse:0.02:1500:1000", "1", "5", "1", "2", "8", "4", "8"])
    lines = open(p).read().rstrip("\n").split("\n")
    assert lines[0].startswith("Description:") and lines[3].startswith("Input type:sparse:")
    data = [[x for x in re.split(r",\s*", ln.strip()) if x] for ln in lines[6:]]
    assert len(data) == 2 and all(len(r) == 7 for r in data)
    for r in data:
        assert int(r[0]) == 8 and int(r[1]) in (4, 8) and int(r[2]) == 5 and int(r[4]) > 0 and int(r[6]) > 0
    assert float(data[1][3]) < float(data[0][3]) * 1.01                  # more matmuls: residual does not grow
    # a Matrix Market input takes the same path
    import scipy.io
    import scipy.sparse as sp

    M = sp.random(300, 200, density=0.05, random_state=1, format="coo")
    scipy.io.mmwrite(str(tmp_path / "a.mtx"), M)
    (tmp_path / "o").mkdir()
    p2 = abrik.speed_sparse([str(tmp_path / "o"), str(tmp_path / "a.mtx"), "1", "4", "1", "1", "8", "6"])
    assert "Input size:300 by 200" in open(p2).read()


def test_cqrrt_linops_composite_applications_main(tmp_path):
    """CQRRT_linop_composite_applications.cc: the generalized-LS / generalized-SVD study on L^{-1} V"""
    from benchmarks import cqrrt_linops

    res, brk = cqrrt_linops.composite_applications(["double", str(tmp_path), "1", "gen:3000", "gen:60:6", "2.0", "4", "0", "0", "0"])
    lines = [ln for ln in open(res).read().splitlines() if not ln.startswith("#")]
    hdr = lines[0].split(",")
    assert hdr[:6] == ["m", "n", "run", "algorithm", "chol_time_us", "qr_time_us"] and len(hdr) == 18
    rows = [ln.split(",") for ln in lines[1:]]
    assert [r[3] for r in rows] == ["CQRRT_linop", "CholQR", "sCholQR3", "sCholQR3_basic"]
    for r in rows:
        assert len(r) == 18 and int(r[0]) == 3000 and int(r[1]) == 60 and int(r[5]) > 0
        assert float(r[6]) < 1e-10 and int(r[7]) == 60                   # Q = (L^-1 V) R^-1 is orthonormal: R is the factor of the composite
        assert float(r[9]) < 1e-9                                         # generalized least squares recovers x_true
        assert float(r[12]) < 1e-10                                       # right singular vectors of R orthonormal
    blines = [ln for ln in open(brk).read().splitlines() if not ln.startswith("#")]
    assert blines[0].startswith("m,n,run,algorithm,t0") and len(blines) == 5 and all(len(ln.split(",")) == 22 for ln in blines)


def test_bqrrp_gpu_benchmark_main(tmp_path):
    """benchmarks/bqrrp_gpu.py = BQRRP_GPU_benchmark.cu on the BQRRP_GPU class: the reference's four files (names, 6 header lines,
    `qrf  cholqr  geqrf` speed rows, 15-entry breakdown rows that add up to their total), both modes."""
    from benchmarks import bqrrp_gpu

    f3, f1, f2, rows = bqrrp_gpu.run_block_size_sweep(str(tmp_path), 1024, 1024, [64, 128, 256], True, True)
    for path, name in ((f3, "_BQRRP_GPU_speed_comparisons_block_size_num_info_lines_6.txt"), (f1, "_BQRRP_GPU_runtime_breakdown_qrf_num_info_lines_6.txt"),
                       (f2, "_BQRRP_GPU_runtime_breakdown_cholqr_num_info_lines_6.txt")):
        assert path.endswith(name)
    # the reference writes FIVE header lines into these files (Description / File format / Input type / Input size / Additional
    # parameters -- there is no OMP-threads line in the GPU main), whatever the "num_info_lines_6" of the names says: kept as it is
    lines = open(f3).read().rstrip("\n").split("\n")
    assert lines[0].startswith("Description:") and lines[4].startswith("Additional parameters: BQRRP block sizes: 64,128,256")
    body = lines[5:]
    assert body[-1].startswith("Total benchmark execution time:") and len(body) == 4
    for ln, r in zip(body[:3], rows):
        t = [int(x) for x in ln.split()]
        assert len(t) == 3 and all(x > 0 for x in t) and tuple(t) == tuple(r)
    for path, chol in ((f1, False), (f2, True)):
        lines = open(path).read().rstrip("\n").split("\n")
        assert lines[1].startswith("File format: 15 data columns") and lines[4].startswith("Additional parameters: Tall QR subroutine " + ("cholqr" if chol else "geqrf"))
        data = [[x for x in re.split(r",\s*", ln.strip()) if x] for ln in lines[5:]]
        assert len(data) == 3
        for r in data:
            t = [int(x) for x in r]
            assert len(t) == 15 and sum(t[:14]) == t[14] and t[2] == t[4] == t[6] == 0
            assert (t[8] > 0) == chol and (t[10] > 0) == chol
    path, rows = bqrrp_gpu.run_mat_size_sweep(str(tmp_path), [512, 1024], False, False)
    assert path.endswith("BQRRP_GPU_speed_comparisons_mat_size_num_info_lines_6.txt") and not path.split("/")[-1].startswith("_")
    lines = open(path).read().rstrip("\n").split("\n")
    assert lines[3].startswith("Input size: dim start: 512,1024") and len(lines) == 7
    assert all(len(ln.split()) == 3 and int(ln.split()[2]) == 0 for ln in lines[5:])
    assert bqrrp_gpu.main([]) == 1


def test_flop_count_mains(capsys):
    """benchmarks/flop_count.py = bench_general/GEMM_flop_count.cc + LAPACK_flop_count.cc: LAWN-41 flop counts, best of N, the
    reference's output lines"""
    from benchmarks import flop_count

    g = flop_count.gemm_flops(2048, 3)
    assert g > 5e3                                                    # an MFMA GEMM, not a fallback (GFLOP/s)
    q = flop_count.geqrf_flops(4096, 1024, 2)
    t = flop_count.getrf_flops(4096, 1024, 2)
    p = flop_count.potrf_flops(1024, 2)
    assert min(q, t, p) > 10
    out = capsys.readouterr().out
    assert "THE SYSTEM IS CAPABLE OF" in out and "RUNNING GEQRF." in out and "RUNNING GETRF." in out and "RUNNING POTRF." in out
    assert flop_count.main(["potrf", "300", "200", "1"]) == 0
    with pytest.raises(RuntimeError):
        flop_count.main(["gesvd", "10", "10", "1"])


def test_general_and_side_mains(tmp_path):
    """benchmarks/general.py = benchmark/bench_general/{Chol_check, Gemm_vs_ormqr, basic_blas_speed, convert_time}.cc and
    bench_BQRRP/{HQRRP_sanity_check, find_test_mat_spectrum}.cc at toy sizes: the checks they print hold and the files have the reference's layout."""
    from benchmarks import general

    out = general.chol_check(m=300, k=120, seeds=2)
    for rc, nrm in out:
        assert rc > 120                                    # the indefinite trailing part stops potrf past the Gram block ...
        assert nrm < 1e-10                                 # ... whose factor is already final: R'R = A[:k, :k]
    rows = general.gemm_vs_ormqr(sizes=((1024, 32), (2048, 64)), runs=3)
    assert len(rows) == 2 and all(r[2] > 0 and r[3] > 0 for r in rows)
    p = general.basic_blas_speed(256, 512, 2, str(tmp_path))
    lines = open(p).read().strip().split("\n")
    assert len(lines) == 4 and lines[0].startswith("256,") and lines[-1].startswith("512,") and all(len(ln.split(",")) == 5 for ln in lines)
    raw = tmp_path / "embedding_combined.dat"
    raw.write_text("1000000 2500000 10\n3 4 5\n")
    general.convert_time(str(raw))
    assert raw.read_text() == "\n1.000000  2.500000  0.000010  \n0.000003  0.000004  0.000005  "
    p = general.hqrrp_sanity_check([str(tmp_path), "2", "256", "512"])
    lines = open(p).read().split("\n")
    assert lines[0].startswith("Description: Results from the sanity check") and lines[5].startswith("Input row sizes:256, 512, ")
    data = [ln for ln in lines[7:] if ln.strip()]
    assert len(data) == 4 and all(len(ln.split(",")) == 3 and int(ln.split(",")[0]) > 0 for ln in data)
    paths = general.find_test_mat_spectrum([str(tmp_path), "300", "200"])
    assert [pp.split("/")[-1] for pp in paths] == [f"_{t}_spectrum_num_info_lines_4.txt" for t in ("poly", "stair", "spike", "kahan")]
    poly = open(paths[0]).read().split("\n")
    assert poly[2].startswith("Input type:0") and poly[3] == "Input size:300 by 200"
    sv = np.array([float(x) for x in poly[4].split(",") if x.strip()])
    assert sv.size == 200 and abs(sv[0] - 1.0) < 1e-6 and abs(sv[-1] / 1e-10 - 1.0) < 0.05 and np.all(np.diff(sv) <= 1e-15)      # (six printed digits)
    stair = np.array([float(x) for x in open(paths[1]).read().split("\n")[4].split(",") if x.strip()])
    assert stair.size == 200 and abs(stair[0] / stair[-1] - 1e10) / 1e10 < 1e-3
