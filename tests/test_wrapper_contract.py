"""Runs the C++ contract test of the drop-in classes (tests/cpp/wrapper_contract.cpp, built by rails_amd/csrc/Makefile into
rails_amd/lib/wrapper_contract): HostDenseMatrix cases on the CPU, HipMultiVectorWrapper / HipOperatorWrapper cases on the GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "rails_amd", "lib", "wrapper_contract")


def _run(args, exe=EXE, marker="ALL PASSED"):
    if not os.path.exists(exe):
        import rails_amd.build

        rails_amd.build.build()
    p = subprocess.run([exe] + args, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert p.returncode == 0 and marker in p.stdout, p.stdout[-4000:]
    return p.stdout


def test_dense_matrix_contract_on_host():
    out = _run(["--host"])
    assert "host cases" in out


@pytest.mark.gpu
def test_wrapper_contract_on_gpu():
    out = _run([])
    assert "all cases" in out


@pytest.mark.gpu
def test_cpp_example_program_solves_on_both_back_ends():
    """examples/solve_laplace.cpp: a pure C++ host program on the header-only classes over the C ABI (no Python in the loop)"""
    out = _run([], exe=os.path.join(ROOT, "rails_amd", "lib", "solve_laplace"), marker="OK")
    assert "direct back end: return 0" in out and "coordinate-space back end: return 0" in out
