"""The sweep SpMM kernel (rails_amd/csrc/spmm_sweep.hip; `A_ * W`, src/LyapunovSolver.hpp:146, on banded patterns) through the C ABI.

Every row's nonzeros are consumed in ascending column order with one fused multiply-add each, from zero: the same chain as
the row-gather kernel, so the two HIP kernels must agree BIT FOR BIT; against the CPU oracle (separate multiply and add)
the bound is the SpMM tolerance of tests/test_gpu_kernels.py, 1e-14 * sqrt(nnz/row) relative.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import rails_amd

    c = rails_amd.Context(device=0, seed=77)
    yield c
    c.close()


def _panels(ctx, m, nc, xoff=0, yoff=0, seed=0):
    import rails_amd

    Xh = np.random.default_rng(seed).uniform(-1, 1, (m, nc))
    big = rails_amd.HipMultiVectorWrapper(ctx, m=m, n=nc + xoff, capacity=nc + xoff)
    X = big.view(xoff, xoff + nc - 1)
    X.from_host(Xh)
    outp = rails_amd.HipMultiVectorWrapper(ctx, m=m, n=nc + yoff, capacity=nc + yoff + 2)
    Y = outp.view(yoff, yoff + nc - 1)
    return Xh, X, Y, outp


@pytest.mark.parametrize("m,nc,xoff,yoff", [(131072, 128, 0, 0), (150001, 128, 2, 4), (131072, 64, 0, 0), (100000, 32, 0, 2), (70000, 16, 6, 0)])
def test_sweep_is_bitwise_the_rowgather_result_and_matches_the_oracle(ctx, oracle, m, nc, xoff, yoff):
    import rails_amd
    from rails_amd import problems as P

    A = P.banded_random(m, 27, 4096 if nc >= 64 else 1500, seed=m % 7)
    op = rails_amd.HipOperatorWrapper(ctx, *A)
    Xh, X, Y, outp = _panels(ctx, m, nc, xoff, yoff, seed=nc)
    outp.assign(0.0)
    op.set_variant(7)
    op.apply(X, Y)
    assert op.last_kernel().startswith("k_spmm_sweep")
    st = op.sweep_stats(nc)
    assert st["built"] and st["efficiency"] > 0.5
    Ys = Y.to_host()
    op.set_variant(3)
    op.apply(X, Y)
    assert op.last_kernel() == "k_spmm_rowgather"
    Yr = Y.to_host()
    assert np.array_equal(Ys, Yr), np.abs(Ys - Yr).max()
    rows = np.random.default_rng(1).choice(m, 4000, replace=False)
    rows = np.concatenate([rows, np.arange(64), np.arange(m - 64, m)])
    rp, col, val = A
    ref = np.zeros((rows.size, nc))
    for k, i in enumerate(rows):
        ref[k] = val[rp[i]:rp[i + 1]] @ Xh[col[rp[i]:rp[i + 1]]]
    assert np.abs(Ys[rows] - ref).max() <= 4e-14 * np.sqrt(27) * np.abs(ref).max()
    if yoff:
        assert np.array_equal(outp.to_host()[:, :yoff], np.zeros((m, yoff)))  # columns outside the window untouched


@pytest.mark.parametrize("kind,nc", [("laplace7", 128), ("stencil27", 128), ("stencil27_random", 64), ("banded_narrow", 256)])
def test_sweep_on_stencils_and_at_256_columns(ctx, kind, nc):
    """Structured patterns (every row of a plane reads the same few offsets: many slots ask for neighbouring ring rows at once) and the widest
    panel the kernel takes (16 column chunks x 2 phases: the window has to fit ONE block of 2816 rows): forced sweep against the row-gather
    kernel, bit for bit."""
    import rails_amd
    from rails_amd import problems as P

    if kind == "laplace7":
        A = P.laplace7(50, 50, 60)  # window 2 * 2500 + 1 rows
    elif kind == "stencil27":
        A = P.stencil27(30, 30, 150)  # window 2 * 931 + 1
    elif kind == "stencil27_random":
        A = P.stencil27(40, 40, 90, random_values=True, seed=3)
    else:
        A = P.banded_random(140000, 27, 1000, seed=5)
    m = A[0].size - 1
    op = rails_amd.HipOperatorWrapper(ctx, *A)
    Xh, X, Y, outp = _panels(ctx, m, nc, seed=nc + 1)
    op.set_variant(7)
    op.apply(X, Y)
    assert op.last_kernel().startswith("k_spmm_sweep")
    Ys = Y.to_host()
    op.set_variant(3)
    op.apply(X, Y)
    assert op.last_kernel() == "k_spmm_rowgather"
    assert np.array_equal(Ys, Y.to_host())
    rp, col, val = A
    rows = np.concatenate([np.arange(40), np.random.default_rng(2).choice(m, 2000, replace=False), np.arange(m - 40, m)])
    ref = np.stack([val[rp[i]:rp[i + 1]] @ Xh[col[rp[i]:rp[i + 1]]] for i in rows])
    assert np.abs(Ys[rows] - ref).max() <= 4e-14 * np.sqrt(27) * max(1.0, np.abs(ref).max())


def test_sweep_handles_ragged_rows_and_rows_without_entries(ctx, oracle):
    import rails_amd

    g = np.random.default_rng(9)
    m = 120000
    cnt = g.integers(0, 40, m)
    cnt[::113] = 0
    rowptr = np.zeros(m + 1, dtype=np.int64)
    rowptr[1:] = np.cumsum(cnt)
    col = np.empty(rowptr[-1], dtype=np.int32)
    for i in range(m):
        c = i + g.integers(-3000, 3000, cnt[i])
        c = np.where(c < 0, -c, c)
        c = np.where(c >= m, 2 * (m - 1) - c, c)  # reflected at the ends (clipping would pile dozens of entries of a row on one column)
        col[rowptr[i]:rowptr[i + 1]] = np.sort(c)
    val = g.uniform(-1, 1, col.size)
    op = rails_amd.HipOperatorWrapper(ctx, rowptr, col, val)
    Xh, X, Y, outp = _panels(ctx, m, 128, seed=3)
    Y.assign(7.0)
    op.set_variant(7)
    op.apply(X, Y)
    Ys = Y.to_host()
    ref = oracle.csr_spmm(rowptr, col, val, Xh)
    assert np.abs(Ys - ref).max() <= 4e-14 * np.sqrt(40) * np.abs(ref).max()
    assert np.array_equal(Ys[::113], np.zeros_like(Ys[::113]))
    op.set_variant(3)
    op.apply(X, Y)
    assert np.array_equal(Ys, Y.to_host())


def test_sweep_declines_what_it_cannot_do(ctx):
    import rails_amd
    from rails_amd import problems as P

    U = P.uniform_random(200000, 9, seed=4)
    op = rails_amd.HipOperatorWrapper(ctx, *U)
    Xh, X, Y, _ = _panels(ctx, 200000, 128)
    op.set_variant(7)
    with pytest.raises(rails_amd.RailsError, match="does not fit"):
        op.apply(X, Y)
    op.set_variant(0)  # auto falls back to the row-gather kernel
    op.apply(X, Y)
    assert op.last_kernel().startswith("k_spmm_rowgather")
    A = P.banded_random(140000, 27, 4096, seed=1)
    op2 = rails_amd.HipOperatorWrapper(ctx, *A)
    Xh, X, Y, _ = _panels(ctx, 140000, 24)
    op2.set_variant(7)
    with pytest.raises(rails_amd.RailsError, match="multiple of 16"):
        op2.apply(X, Y)


def test_auto_choice_on_the_bench_matrix_and_its_transpose(ctx):
    """BASELINE configs[2] at full size: after prepare(128) auto picks the sweep kernel at 128 columns; linearity and the adjoint identity hold."""
    import rails_amd
    from rails_amd import problems as P

    m = 1000000
    A = P.banded_random(m, 27, 4096, seed=1)
    op = rails_amd.HipOperatorWrapper(ctx, *A)
    X = rails_amd.HipMultiVectorWrapper(ctx, m=m, n=128, capacity=128)
    Z = rails_amd.HipMultiVectorWrapper(ctx, m=m, n=128, capacity=128)
    X.random()
    Z.random()
    # without set-up the first products of a width stay with the row kernels: the schedule costs a thousand products' worth of what it saves
    # (RAILS_SWEEP_AFTER = 16 products of the width, or rails_csr_prepare -- used for the transpose below and in test_gpu_fullsize.py)
    Y0 = op.apply(X)
    for _ in range(14):
        op.apply(X, Y0)
    assert not op.last_kernel().startswith("k_spmm_sweep") and not op.sweep_stats(128)["built"]
    Y = op.apply(X)
    assert op.last_kernel().startswith("k_spmm_sweep") and op.sweep_stats(128)["built"]
    d0 = Y.copy()
    d0 -= Y0
    assert d0.norm() <= 1e-13 * Y.norm()
    op.set_variant(3)
    Yr = op.apply(X)
    d = Y.copy()
    d -= Yr
    assert d.norm() == 0.0  # bitwise equal to the row-gather kernel on all 128M entries
    op.set_variant(0)
    # A * ones = row sums (the generator makes them -1 exactly up to rounding of the sum)
    ones = rails_amd.HipMultiVectorWrapper(ctx, m=m, n=128, capacity=128)
    ones.assign(1.0)
    rs = op.apply(ones).to_host()[:, [0, 127]]
    rp, col, val = A
    assert np.abs(rs - np.add.reduceat(val, rp[:-1])[:, None]).max() < 1e-12
    # adjoint identity <Z, A X> = <A^T Z, X> through the transposed operator (its own schedule)
    lhs = Z.dot(Y)
    At = op.transpose()
    assert At.prepare(128)
    AtZ = At.apply(Z)
    assert op.last_kernel().startswith("k_spmm_sweep")
    rhs = AtZ.dot(X)
    assert np.abs(lhs - rhs).max() <= 1e-9 * np.abs(lhs).max()


def test_auto_leaves_structured_stencils_to_the_box_kernel(ctx):
    """A 7-point Laplacian whose window fits the sweep's ring goes to the stencil kernels in auto mode, not to the sweep (few nonzeros per row
    leave the sweep at its floor of one LDS-DMA latency per step: 0.82 ms at 50 x 50 x 400, 128 columns, against 0.65 for the box kernel and
    0.50 for the plane-sweep kernel, which takes complete stencils since round 3)"""
    import rails_amd
    from rails_amd import problems as P

    A = P.laplace7(50, 50, 400)
    m = A[0].size - 1
    op = rails_amd.HipOperatorWrapper(ctx, *A)
    X = rails_amd.HipMultiVectorWrapper(ctx, m=m, n=128, capacity=128)
    X.random()
    Y = op.apply(X)
    assert op.last_kernel() == "k_spmm_planes"
    op.set_variant(7)
    Ys = op.apply(X)
    assert op.last_kernel().startswith("k_spmm_sweep")
    d = Y.copy()
    d -= Ys
    assert d.norm() <= 1e-13 * Y.norm()  # (the box kernel adds a row's terms in its own order)


@pytest.mark.gpu
@pytest.mark.parametrize("entry_trips", [2, 4])
@pytest.mark.parametrize("kind,nc", [("banded", 128), ("banded", 64), ("ragged", 128), ("laplace7", 128), ("banded_narrow", 256)])
def test_both_entry_sizes_are_bitwise_the_rowgather_result(ctx, monkeypatch, kind, nc, entry_trips):
    """Both entry sizes of the schedule -- half units (the default: k_spmm_sweep_h2, every half of a unit of the code fetches its own
    entry) and whole units (RAILS_SWEEP_ENTRY_TRIPS=4: k_spmm_sweep) -- against the row-gather kernel, bit for bit (`A_ * W`,
    src/LyapunovSolver.hpp:146)."""
    import rails_amd
    from rails_amd import problems as P

    monkeypatch.setenv("RAILS_SWEEP_ENTRY_TRIPS", str(entry_trips))
    g = np.random.default_rng(nc)
    if kind == "banded":
        A = P.banded_random(131072 + 77, 27, 4096, seed=3)
    elif kind == "banded_narrow":
        A = P.banded_random(140000, 27, 1000, seed=5)
    elif kind == "laplace7":
        A = P.laplace7(50, 50, 60)
    else:
        m = 120000
        lens = g.integers(0, 22, m)
        lens[::53] = 0
        rowptr = np.zeros(m + 1, dtype=np.int64)
        rowptr[1:] = np.cumsum(lens)
        col = np.concatenate([np.sort(g.integers(max(0, i - 3000), min(m, i + 3000), n)) for i, n in enumerate(lens)]).astype(np.int32)
        A = (rowptr, col, g.uniform(-1, 1, col.size))
    m = A[0].size - 1
    op = rails_amd.HipOperatorWrapper(ctx, *A)
    Xh, X, Y, outp = _panels(ctx, m, nc, seed=nc + 3)
    op.set_variant(7)
    op.apply(X, Y)
    assert op.last_kernel() == ("k_spmm_sweep_h2" if entry_trips == 2 else "k_spmm_sweep")
    Ys = Y.to_host()
    op.set_variant(3)
    op.apply(X, Y)
    assert op.last_kernel() == "k_spmm_rowgather"
    assert np.array_equal(Ys, Y.to_host())


def test_leftover_experiment_switch_is_refused_not_obeyed():
    """RAILS_SWEEP_ABLATE selects experiment builds whose results are wrong by construction; the shipped library has none of them and a
    product with the variable set fails loudly instead of returning a wrong panel (a leftover export of a profiling script)."""
    import os
    import subprocess
    import sys

    code = (
        "import numpy as np, rails_amd\n"
        "from rails_amd import problems as P\n"
        "ctx = rails_amd.Context(device=0, seed=1)\n"
        "A = P.banded_random(131072, 27, 4096, seed=1)\n"
        "op = rails_amd.HipOperatorWrapper(ctx, *A)\n"
        "X = rails_amd.HipMultiVectorWrapper(ctx, m=131072, n=128, capacity=128)\n"
        "X.random()\n"
        "op.set_variant(7)\n"
        "try:\n"
        "    op.apply(X)\n"
        "    print('COMPUTED', op.last_kernel())\n"
        "except rails_amd.RailsError as e:\n"
        "    print('REFUSED', e)\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RAILS_SWEEP_ABLATE="16", PYTHONPATH=root)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300, cwd=root)
    assert "REFUSED" in out.stdout and "experiment builds" in out.stdout, out.stdout + out.stderr
