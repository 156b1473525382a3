"""The host-side schedule of the sweep SpMM kernel (rails_amd/csrc/sweep_plan.cpp), checked on the CPU.

The kernel (rails_amd/csrc/spmm_sweep.hip, the product path for `A_ * W`, src/LyapunovSolver.hpp:146, on banded
patterns) interprets a schedule; the oracle's interpreter (oracle/rails_oracle.cpp: orc_sweep_interpret) executes the
same schedule in the same order on the CPU and checks what the kernel relies on: every ring row read holds the X row it
is meant to (arrived, not being refilled), every row of Y is written exactly once.  Its Y is compared with the oracle's
CSR product: both use the same order of additions per row, so the match is exact.
"""
import numpy as np
import pytest

from rails_amd import problems as P
from rails_amd._lib import RailsError
from rails_amd.sweep import SweepPlan


def _check(oracle, A, params, n_chunks, ncols=None, seed=0):
    pl = SweepPlan(*A, ncols=ncols, params=params)
    m = A[0].size - 1
    X = np.random.default_rng(seed).uniform(-1, 1, (pl.ncols, 16 * n_chunks))
    rc, Y = oracle.sweep_interpret(pl, X, n_chunks)
    assert rc == 0, rc
    ref = oracle.csr_spmm(A[0], A[1], A[2], X) if pl.ncols == m else None
    if ref is None:
        ref = np.zeros((m, X.shape[1]))
        for i in range(m):
            for q in range(A[0][i], A[0][i + 1]):
                ref[i] += A[2][q] * X[A[1][q]]
    assert np.array_equal(Y, ref), np.abs(Y - ref).max()
    if pl.trips:
        assert abs(pl.efficiency - pl.nnz / (pl.slots * float(pl.trips))) < 1e-12
    return pl


@pytest.mark.parametrize("entry_trips", [4, 2])
@pytest.mark.parametrize("waves,groups,seg,phases", [(1, 4, 8, 4), (2, 10, 16, 2), (3, 5, 16, 3), (2, 6, 32, 4)])
def test_banded_small_geometries(oracle, waves, groups, seg, phases, entry_trips):
    bw = 7 * seg // 2
    R = waves * groups * 16
    assert (phases - 1) * R >= 2 * bw + 1 + seg  # the feasibility rule of sweep_plan.h
    A = P.banded_random(5 * phases * R + 37, 9, bw, seed=waves + groups)
    pl = _check(oracle, A, (waves, groups, seg, 5, 3, phases, 1, entry_trips), 2)
    assert pl.entry_trips == entry_trips and 0.2 < pl.efficiency <= 1.0


@pytest.mark.parametrize("entry_trips", [4, 2])
def test_kernel_geometry_on_the_bench_pattern(oracle, entry_trips):
    # the kernel's own geometry (8 waves x 22 groups of 16 rows, 256-row steps, 5 segments, 8 parts x 4 phases) at 1/8 of the bench size;
    # entries of half a unit (two trips) waste fewer slots than whole units
    A = P.banded_random(131072, 27, 4096, seed=1)
    pl = _check(oracle, A, (8, 22, 256, 5, 8, 4, 1, entry_trips), 8)
    assert (pl.waves, pl.groups, pl.seg_rows, pl.nseg, pl.parts, pl.phases, pl.slots, pl.entry_trips) == (8, 22, 256, 5, 8, 4, 16, entry_trips)
    assert pl.efficiency > (0.5 if entry_trips == 4 else 0.6)


def test_ragged_rows_empty_rows_duplicates_and_rectangular(oracle):
    g = np.random.default_rng(5)
    m, ncols = 3000, 3300
    rows = []
    for i in range(m):
        n = int(g.integers(0, 14)) if i % 97 else 0  # some rows have no entries at all: their Y rows must still be zeroed
        c = np.sort(np.clip(i + g.integers(-40, 330, n), 0, ncols - 1))  # duplicates allowed, window shifted to the right
        rows.append(c)
    rowptr = np.zeros(m + 1, dtype=np.int64)
    rowptr[1:] = np.cumsum([r.size for r in rows])
    col = np.concatenate(rows).astype(np.int32)
    val = g.uniform(-1, 1, col.size)
    _check(oracle, (rowptr, col, val), (2, 10, 16, 5, 4, 4), 1, ncols=ncols)
    _check(oracle, (rowptr, col, val), (2, 10, 16, 5, 4, 4, 1, 2), 1, ncols=ncols)


def test_long_rows_and_single_part(oracle):
    # 300 entries of one row inside one ring: several units per step for that group
    g = np.random.default_rng(6)
    m = 900
    rows = [np.sort(np.clip(i + g.integers(-60, 60, 300 if i == 450 else 5), 0, m - 1)) for i in range(m)]
    rowptr = np.zeros(m + 1, dtype=np.int64)
    rowptr[1:] = np.cumsum([r.size for r in rows])
    col = np.concatenate(rows).astype(np.int32)
    val = g.uniform(-1, 1, col.size)
    _check(oracle, (rowptr, col, val), (1, 8, 16, 5, 1, 4), 2)
    _check(oracle, (rowptr, col, val), (1, 8, 16, 5, 1, 4, 1, 2), 2)


def test_tiny_and_empty_matrices(oracle):
    A = P.banded_random(5, 3, 2, seed=0)
    _check(oracle, A, (1, 2, 8, 5, 8, 2), 1)  # more parts than rows
    rowptr = np.zeros(41, dtype=np.int64)
    _check(oracle, (rowptr, np.zeros(0, np.int32), np.zeros(0)), (1, 2, 8, 5, 2, 2), 1)  # no nonzeros: Y = 0


def test_patterns_that_do_not_fit_are_refused():
    A = P.banded_random(20000, 9, 3000, seed=2)  # window 6001 rows > (phases - 1) * 160 rows
    with pytest.raises(RailsError, match="does not fit"):
        SweepPlan(*A, params=(2, 10, 16, 5, 2, 2))
    U = P.uniform_random(4000, 5, seed=1)
    with pytest.raises(RailsError, match="does not fit"):
        SweepPlan(*U, params=(2, 10, 16, 5, 2, 4))
    # unsorted columns inside a row
    rowptr = np.array([0, 2], dtype=np.int64)
    with pytest.raises(RailsError, match="not sorted"):
        SweepPlan(rowptr, np.array([0, 0], np.int32)[::-1] + np.array([1, 0], np.int32), np.ones(2), ncols=4, params=(1, 2, 8, 5, 1, 2))


def test_compiler_stays_out_of_the_assembly_registers():
    """spmm_sweep.hip: the unit loop's inline assembly owns v24-v255; `amdgpu_num_vgpr(24)` keeps the compiler below them only while its
    own values fit, so the generated code is checked (compile only: hipcc cross-compiles gfx950 without a GPU)."""
    import importlib.util
    import os

    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "check_sweep_regs.py")
    spec = importlib.util.spec_from_file_location("check_sweep_regs", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    kernels, bad = mod.check()
    assert kernels >= 2 and not bad, bad[:5]
    # the object that ships, not only a separate compile: metadata notes of the code objects inside the built librails_hip.so
    nb, badb = mod.check_built()
    assert nb >= 2 and not badb, badb[:5]
    meta = mod.kernel_metadata()
    assert all(v.get("private_segment_fixed_size", 0) == 0 for k, v in meta.items() if "k_spmm_planes" in k or "k_panel_gemm_wide" in k)  # no spills in the new kernels
