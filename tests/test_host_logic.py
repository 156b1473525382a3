"""Host logic of the product without a GPU: the solver template rails::Solver (rails_amd/include/rails/
LyapunovSolver.hpp) instantiated on a plain CPU backend (tests/cpu_backend, test scaffolding) must walk the same
trajectory as the oracle on identical inputs and counter-RNG streams; plus the row-partition helpers."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    import rails_amd

    rails_amd.load()
    out = tmp_path_factory.mktemp("cpu_backend") / "solver_cpu_driver"
    cmd = ["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "rails_amd", "include"),
           "-I" + os.path.join(ROOT, "tests", "cpu_backend"), os.path.join(ROOT, "tests", "cpu_backend", "solver_cpu_driver.cpp"),
           "-o", str(out), "-L" + os.path.join(ROOT, "rails_amd", "lib"), "-lrails_hip", "-L/opt/rocm/lib",
           "-Wl,-rpath," + os.path.join(ROOT, "rails_amd", "lib"), "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    return str(out)


def run_driver(driver, tmp_path, A, B, params, seed=1, V0=None):
    n, p = B.shape
    np.asfortranarray(A).T.copy().tofile(tmp_path / "A.bin")  # column-major on disk
    np.asfortranarray(B).T.copy().tofile(tmp_path / "B.bin")
    args = [driver, str(tmp_path / "A.bin"), str(tmp_path / "B.bin"), str(n), str(p), str(seed), str(tmp_path / "out")]
    args += ["%s=%r" % (k, float(v)) for k, v in params.items()]
    if V0 is not None:
        np.asfortranarray(V0).T.copy().tofile(tmp_path / "V0.bin")
        args += ["V0=%s" % (tmp_path / "V0.bin"), "V0cols=%d" % V0.shape[1]]
    subprocess.check_call(args, stdout=subprocess.DEVNULL)
    lines = open(str(tmp_path / "out.txt")).read().split()
    rc, trips, k = int(lines[0]), int(lines[1]), int(lines[2])
    hist = np.array([float(x) for x in lines[3:]])
    V = T = None
    if k:
        V = np.fromfile(str(tmp_path / "out.V")).reshape(k, n).T
        T = np.fromfile(str(tmp_path / "out.T")).reshape(k, k).T
    return rc, trips, hist, V, T


def _tridiagonal_problem(n, seed):
    g = np.random.default_rng(seed)
    A = g.uniform(-1, 1, (n, n))
    for i in range(n):
        for j in range(n):
            if abs(i - j) > 1:
                A[i, j] = 0.0
            elif i == j:
                A[i, j] *= 3.0
    B = np.zeros((n, 1))
    B[n - 1, 0] = g.uniform(-1, 1)
    return A, B


def test_solver_template_matches_oracle_config1(driver, tmp_path, oracle):
    from rails_amd import problems as P

    A = P.dense_stable(256, seed=1)
    B = P.rhs(256, 4, seed=2)
    params = {"Restart size": 32, "Reduced size": 16, "Expand size": 3, "Lanczos iterations": 10, "Tolerance": 1e-3}
    rc, trips, hist, V, T = run_driver(driver, tmp_path, A, B, params, seed=1)
    out = oracle.solve(A, B, oracle.params({**params, "rng_mode": 1, "seed": 1}))
    assert rc == out["ret"] == 0
    assert abs(trips - out["trips"]) <= 1
    n = min(8, len(hist), len(out["res_hist"]))
    np.testing.assert_allclose(hist[:n], out["res_hist"][:n], rtol=1e-7)
    Xp, Xo = V @ T @ V.T, out["V"] @ out["T"] @ out["V"].T
    assert np.linalg.norm(Xp - Xo) / np.linalg.norm(Xo) < 1e-2
    R = A @ Xp + Xp @ A.T + B @ B.T
    assert np.linalg.norm(R) / np.linalg.norm(B @ B.T) < 2e-3


def test_solver_template_reference_cases(driver, tmp_path):
    # the shapes of test/LyapunovSolver_test.cpp:118-352 through the product's solver template
    def resid(A, B, V, T):
        X = V @ T @ V.T
        return np.abs(A @ X + X @ A.T + B @ B.T).max()

    A, B = _tridiagonal_problem(20, 2)
    rc, trips, hist, V, T = run_driver(driver, tmp_path, A, B, {"Restart Size": 19, "Reduced Size": 15, "Expand Size": 1, "Minimize solution space": 0})
    assert rc == 0 and V.shape[1] < 20 and resid(A, B, V, T) < 1e-3
    A, B = _tridiagonal_problem(20, 3)
    rc, trips, hist, V, T = run_driver(driver, tmp_path, A, B, {"Minimize solution space": 0, "Tolerance": 1e-8})
    assert rc == 0 and V.shape[1] == 20 and resid(A, B, V, T) < 1e-3
    rc, trips, hist, V, T = run_driver(driver, tmp_path, A, B, {"Minimize solution space": 1, "Tolerance": 1e-8})
    assert rc == 0 and V.shape[1] < 20 and resid(A, B, V, T) < 1e-3
    A, B = _tridiagonal_problem(20, 4)
    rc, trips, hist, V, T = run_driver(driver, tmp_path, A, B, {"Restart iterations": 10, "Minimize solution space": 0, "Expand size": 1})
    assert rc == 0 and V.shape[1] < 20 and resid(A, B, V, T) < 1e-3
    # warm start (:312-352)
    A, B = _tridiagonal_problem(20, 5)
    rc, trips, hist, V, T = run_driver(driver, tmp_path, A, B, {"Minimize solution space": 1, "Tolerance": 1e-8})
    assert rc == 0 and V.shape[1] < 20
    A[19, 19] = 4.0
    rc, trips2, hist, V2, T2 = run_driver(driver, tmp_path, A, B, {"Minimize solution space": 1, "Tolerance": 1e-8, "Restart from solution": 1}, V0=V)
    assert rc == 0 and V2.shape[1] < 20 and resid(A, B, V2, T2) < 1e-3
    # parameter validation (src/LyapunovSolver.hpp:89-95): set_parameters returns 1
    rc, *_ = run_driver(driver, tmp_path, A, B, {"Lanczos iterations": 3, "Expand size": 3})
    assert rc == 101


def test_row_ranges_and_halo_plan():
    from rails_amd import partition, problems as P

    starts = partition.row_ranges(1003, 4)
    assert starts[0] == 0 and starts[-1] == 1003 and np.all(np.diff(starts) >= 250)
    A = P.banded_random(1003, 9, 40, seed=3)
    plans = []
    # emulate the all-gather of the request lists on one process
    reqs = []
    for r in range(4):
        rp, col, val = P.csr_rows(A, starts[r], starts[r + 1])
        colg = col.astype(np.int64)
        own = (colg >= starts[r]) & (colg < starts[r + 1])
        ghosts = np.unique(colg[~own])
        owner = np.searchsorted(starts, ghosts, side="right") - 1
        reqs.append([ghosts[owner == d] for d in range(4)])
    for r in range(4):
        rp, col, val = P.csr_rows(A, starts[r], starts[r + 1])
        plans.append(partition.HaloPlan(starts, r, col.astype(np.int64), lambda obj: reqs))
    for r, pl in enumerate(plans):
        assert pl.col_local.min() >= 0 and pl.col_local.max() < pl.m_local + pl.n_ghost
        assert pl.recv_counts[r] == 0 and pl.send_counts[r] == 0
        for d in range(4):
            assert pl.send_counts[d] == plans[d].recv_counts[r]
        # every remapped column points at the right global row
        rp, col, val = P.csr_rows(A, starts[r], starts[r + 1])
        glob = np.where(pl.col_local < pl.m_local, pl.col_local + starts[r], pl.ghost_globals[np.maximum(pl.col_local - pl.m_local, 0)])
        assert np.array_equal(glob, col.astype(np.int64))


def test_compiled_reference_stays_in_the_build_container():
    """The binary built from the reference's own sources (oracle/_ref, oracle/Makefile) is test infrastructure of the build container: it is
    listed in .gpurunignore (it does not travel to the GPU box) and in .gitignore (it stays out of history), and where /root/reference does
    not exist nothing may sit under oracle/_ref."""
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ignore = open(os.path.join(root, ".gpurunignore")).read().split()
    assert "oracle/_ref/" in ignore
    assert "oracle/_ref/" in open(os.path.join(root, ".gitignore")).read().split()
    ref_dir = os.path.join(root, "oracle", "_ref")
    if not os.path.isdir("/root/reference"):
        assert not os.path.isdir(ref_dir) or not os.listdir(ref_dir), "a binary built from the reference has travelled: %s" % os.listdir(ref_dir)
