"""The C++ object layer used the way a caller of the reference uses it: tests/cxx/test_qb_caller.cpp (QB / RSVD object graph) and
tests/cxx/test_gpu_classes_caller.cpp (the two device classes) are user programs written from scratch against RandLAPACK_amd.hh,
built with plain g++ against librlhip.so by `make -C tests/cxx` (__graft_entry__.build()) and run on the device here."""
import subprocess
from pathlib import Path

import pytest

CXX = Path(__file__).resolve().parent / "cxx"


def test_cxx_caller_program_builds_and_links():
    """no GPU needed: the program compiles against RandLAPACK_amd.hh with the host compiler and links against the C ABI only"""
    subprocess.run(["make", "-C", str(CXX), "-s"], check=True)
    exe = CXX / "test_qb_caller"
    assert exe.exists()
    out = subprocess.check_output(["nm", "-D", "--undefined-only", str(exe)], text=True)
    assert "rlhip_drv" not in out                      # the object layer is header-only C++ over the kernel-level C ABI ...
    assert "rlhip_gemm_f64" in out and "rlhip_malloc" in out
    assert "hip" not in out.replace("rlhip", "")       # ... and needs no HIP runtime symbol of its own


@pytest.mark.gpu
def test_qb_rsvd_object_graph_from_a_cxx_caller():
    """tests/cxx/test_qb_caller.cpp: planted-rank QB at several block sizes and stabilisers, the tolerance exit, RSVD through the
    abstract bases in double and float, argument errors -- every check is made by the program itself"""
    exe = CXX / "test_qb_caller"
    if not exe.exists():
        subprocess.run(["make", "-C", str(CXX), "-s"], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("PASSED"), r.stdout + r.stderr
    assert r.stdout.count("\nQB ") + r.stdout.startswith("QB ") == 5 and r.stdout.count("RSVD ") == 3 and "3 of 3 invalid calls raised" in r.stdout


def test_gpu_class_caller_builds_and_links():
    subprocess.run(["make", "-C", str(CXX), "-s"], check=True)
    exe = CXX / "test_gpu_classes_caller"
    assert exe.exists()
    out = subprocess.check_output(["nm", "-D", "--undefined-only", str(exe)], text=True)
    assert "rlhip_drv" not in out and "hip" not in out.replace("rlhip", "")


@pytest.mark.gpu
def test_bqrrp_gpu_and_cqrrpt_gpu_classes_from_a_cxx_caller():
    """tests/cxx/test_gpu_classes_caller.cpp: a freshly written caller of BQRRP_GPU / CQRRPT_GPU (the reference's device classes,
    rl_bqrrp_gpu.hh:27-149, rl_cqrrpt_gpu.hh:23-146) that sets `.qr_tall`, goes through the `_alg` base classes, reads `.times`
    (15 / 8 entries) and `.rank`, and verifies the factorizations itself."""
    exe = CXX / "test_gpu_classes_caller"
    if not exe.exists():
        subprocess.run(["make", "-C", str(CXX), "-s"], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("PASSED"), r.stdout + r.stderr
    assert r.stdout.count("BQRRP_GPU<") == 3 and "CQRRPT_GPU 900 x 60" in r.stdout
