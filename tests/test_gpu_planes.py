"""The plane-sweep SpMM kernel for structured-grid stencils (rails_amd/csrc/spmm_planes.hip; `A_ * W`, src/LyapunovSolver.hpp:146, on
the operators of BASELINE configs[1] and configs[3]) through the C ABI.

Every row's nonzeros meet the same chain of fused multiply-adds in column order as in the row-gather kernel (plus +0 terms for halo
rows outside the grid), so the two HIP kernels agree BIT FOR BIT up to the sign of an exact zero (`np.array_equal` does not see it);
against the CPU oracle (separate multiply and add) the bound is the SpMM tolerance of tests/test_gpu_kernels.py.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import rails_amd

    c = rails_amd.Context(device=0, seed=78)
    yield c
    c.close()


def _panels(ctx, m, nc, xoff=0, yoff=0, seed=0, pad=0):
    import rails_amd

    Xh = np.random.default_rng(seed).uniform(-1, 1, (m, nc))
    big = rails_amd.HipMultiVectorWrapper(ctx, m=m, n=nc + xoff, capacity=nc + xoff + pad)
    X = big.view(xoff, xoff + nc - 1)
    X.from_host(Xh)
    outp = rails_amd.HipMultiVectorWrapper(ctx, m=m, n=nc + yoff, capacity=nc + yoff + 2 + pad)
    Y = outp.view(yoff, yoff + nc - 1)
    return Xh, X, Y, outp


def _check(ctx, oracle, A, nc, xoff=0, yoff=0, pad=0):
    import rails_amd

    m = A[0].size - 1
    op = rails_amd.HipOperatorWrapper(ctx, *A)
    Xh, X, Y, outp = _panels(ctx, m, nc, xoff, yoff, seed=nc + m % 5, pad=pad)
    outp.assign(0.0)
    op.set_variant(9)
    op.apply(X, Y)
    assert op.last_kernel() == "k_spmm_planes"
    Yp = Y.to_host()
    op.set_variant(3)
    op.apply(X, Y)
    assert op.last_kernel() == "k_spmm_rowgather"
    Yr = Y.to_host()
    assert np.array_equal(Yp, Yr), np.abs(Yp - Yr).max()
    Yo = oracle.csr_spmm(*A, Xh)
    assert np.abs(Yp - Yo).max() <= 4e-14 * np.sqrt(27) * max(1.0, np.abs(Yo).max())
    if yoff:
        assert np.array_equal(outp.to_host()[:, :yoff], np.zeros((m, yoff)))  # columns outside the window untouched
    return op


@pytest.mark.parametrize("shape,nc,xoff,yoff,pad", [((20, 12, 9), 128, 0, 0, 0), ((17, 13, 7), 128, 2, 4, 3), ((33, 9, 5), 64, 0, 0, 0),
                                                   ((8, 4, 3), 32, 0, 2, 0), ((9, 8, 1), 16, 6, 0, 0), ((5, 3, 40), 2, 0, 0, 0),
                                                   ((24, 21, 6), 256, 0, 0, 0), ((11, 10, 4), 200, 0, 0, 9), ((19, 21, 7), 16, 0, 0, 1),
                                                   ((23, 37, 5), 32, 2, 0, 0), ((9, 19, 6), 48, 0, 0, 0), ((30, 30, 30), 10, 0, 4, 0),
                                                   ((21, 20, 11), 16, 0, 3, 0), ((12, 12, 12), 128, 2, 1, 0), ((15, 14, 9), 32, 0, 5, 1)])
def test_planes_27_point_is_bitwise_the_rowgather_result_and_matches_the_oracle(ctx, oracle, shape, nc, xoff, yoff, pad):
    """27-point stencil with random coefficients: grids that are not multiples of the patch (partial patches, a single plane, fewer
    planes than a segment), panel windows on even offsets, widths below, at and above one 128-column chunk, padded row strides."""
    from rails_amd import problems as P

    _check(ctx, oracle, P.stencil27(*shape, random_values=True, seed=sum(shape)), nc, xoff, yoff, pad)


@pytest.mark.parametrize("shape,nc", [((50, 50, 8), 128), ((13, 7, 11), 64), ((6, 5, 4), 16)])
def test_planes_7_point(ctx, oracle, shape, nc):
    from rails_amd import problems as P

    _check(ctx, oracle, P.laplace7(*shape), nc)


def test_planes_declines_what_it_cannot_take(ctx):
    """Incomplete stencils (a neighbour inside the grid without an entry), odd panel windows and odd widths go to the other kernels:
    asked for by name the product fails loudly, in automatic mode another kernel computes it."""
    import rails_amd
    from rails_amd import problems as P

    rp, col, val = P.stencil27(12, 10, 6, random_values=True, seed=2)
    # drop one off-diagonal entry of one interior row
    i = 12 * 10 * 3 + 12 * 4 + 5
    k = rp[i] + 3
    rp2 = rp.copy()
    rp2[i + 1:] -= 1
    A2 = (rp2, np.delete(col, k), np.delete(val, k))
    m = rp.size - 1
    op = rails_amd.HipOperatorWrapper(ctx, *A2)
    Xh, X, Y, _ = _panels(ctx, m, 64)
    op.set_variant(9)
    with pytest.raises(rails_amd.RailsError):
        op.apply(X, Y)
    op.set_variant(0)
    op.apply(X, Y)
    assert op.last_kernel() != "k_spmm_planes"
    ref = np.zeros((m, 64))
    for r in range(m):
        ref[r] = A2[2][rp2[r]:rp2[r + 1]] @ Xh[A2[1][rp2[r]:rp2[r + 1]]]
    assert np.abs(Y.to_host() - ref).max() <= 1e-12
    # an odd X window offset: the complete stencil, but the 16-byte LDS-DMA pieces of the kernel need even columns (an odd Y offset is
    # fine: A * W written behind an odd number of basis columns, tested above)
    op = rails_amd.HipOperatorWrapper(ctx, rp, col, val)
    Xh, X, Y, _ = _panels(ctx, m, 64, xoff=1)
    op.set_variant(9)
    with pytest.raises(rails_amd.RailsError):
        op.apply(X, Y)


def test_planes_does_not_let_unreferenced_rows_leak(ctx):
    """A non-finite value in an X row that no entry of a matrix row references must not reach that row of the product (the kernel only
    takes complete stencils, and rows outside the grid are staged as zeros)."""
    import rails_amd
    from rails_amd import problems as P

    shape = (10, 9, 8)
    A = P.laplace7(*shape)
    m = A[0].size - 1
    op = rails_amd.HipOperatorWrapper(ctx, *A)
    Xh, X, Y, _ = _panels(ctx, m, 32)
    bad = 10 * 9 * 4 + 10 * 4 + 5
    Xh[bad, :] = np.inf
    X.from_host(Xh)
    op.set_variant(9)
    op.apply(X, Y)
    Yp = Y.to_host()
    rp, col, val = A
    touched = np.zeros(m, bool)
    for r in range(m):
        if bad in col[rp[r]:rp[r + 1]]:
            touched[r] = True
    assert np.isfinite(Yp[~touched]).all()
    assert not np.isfinite(Yp[touched]).all()


def test_planes_is_the_automatic_choice_for_wide_products_on_grid_stencils(ctx):
    import rails_amd
    from rails_amd import problems as P

    A = P.stencil27(16, 16, 16, random_values=True, seed=5)
    m = A[0].size - 1
    op = rails_amd.HipOperatorWrapper(ctx, *A)
    _, X, Y, _ = _panels(ctx, m, 128)
    op.apply(X, Y)
    assert op.last_kernel() == "k_spmm_planes"
