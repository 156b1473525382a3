"""Seeded synthetic inputs of the BASELINE.json configurations.

All generators are deterministic functions of their arguments, return host arrays
(rowptr int64, col int32, val float64) and are shared by the tests, smoke() and bench.py so
that the HIP path and the CPU oracle always see identical (A, M, B).
"""
import numpy as np


def _csr_from_offsets(m, cols, vals):
    """cols/vals: (m, w) arrays with col == -1 marking an absent entry."""
    mask = cols >= 0
    counts = mask.sum(1)
    rowptr = np.zeros(m + 1, dtype=np.int64)
    np.cumsum(counts, out=rowptr[1:])
    return rowptr, cols[mask].astype(np.int32), vals[mask].astype(np.float64)


def laplace7(nx, ny, nz):
    """3D 7-point Laplacian, diag -6, off-diagonals +1, Dirichlet (config C2: 50 x 50 x 40)."""
    m = nx * ny * nz
    idx = np.arange(m, dtype=np.int64)
    x = idx % nx
    y = (idx // nx) % ny
    z = idx // (nx * ny)
    cols = np.full((m, 7), -1, dtype=np.int64)
    vals = np.zeros((m, 7))
    # sorted by column: z-1, y-1, x-1, centre, x+1, y+1, z+1
    specs = [(z > 0, -nx * ny), (y > 0, -nx), (x > 0, -1), (None, 0), (x < nx - 1, 1), (y < ny - 1, nx), (z < nz - 1, nx * ny)]
    for s, (cond, off) in enumerate(specs):
        if cond is None:
            cols[:, s] = idx
            vals[:, s] = -6.0
        else:
            cols[cond, s] = idx[cond] + off
            vals[cond, s] = 1.0
    return _csr_from_offsets(m, cols, vals)


def stencil27(nx, ny, nz, random_values=False, seed=0):
    """27-point stencil.  Fixed values: centre -26, neighbours +1 (config C4).  random_values: off-diagonals
    U(0,1), diagonal = -(sum |off| + 1): strictly diagonally dominant, hence stable."""
    m = nx * ny * nz
    idx = np.arange(m, dtype=np.int64)
    x = idx % nx
    y = (idx // nx) % ny
    z = idx // (nx * ny)
    cols = np.full((m, 27), -1, dtype=np.int64)
    vals = np.zeros((m, 27))
    g = np.random.default_rng(seed)
    s = 0
    centre = None
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                ok = (x + dx >= 0) & (x + dx < nx) & (y + dy >= 0) & (y + dy < ny) & (z + dz >= 0) & (z + dz < nz)
                off = dx + nx * dy + nx * ny * dz
                cols[ok, s] = idx[ok] + off
                if dx == 0 and dy == 0 and dz == 0:
                    centre = s
                else:
                    vals[ok, s] = g.uniform(0.0, 1.0, int(ok.sum())) if random_values else 1.0
                s += 1
    vals[:, centre] = -(np.abs(vals).sum(1) + 1.0) if random_values else -26.0
    return _csr_from_offsets(m, cols, vals)


def banded_random(m, nnz_row=27, bandwidth=4096, seed=0):
    """nnz_row entries per row incl. the diagonal; off-diagonal columns uniform in |j - i| <= bandwidth
    (reflected at the boundary), values U(0,1); diagonal = -(row sum + 1).  Config C3 (primary)."""
    g = np.random.default_rng(seed)
    idx = np.arange(m, dtype=np.int64)[:, None]
    w = nnz_row - 1
    off = g.integers(1, bandwidth + 1, size=(m, w)) * g.choice(np.array([-1, 1]), size=(m, w))
    cols = idx + off
    cols = np.where(cols < 0, idx - off, cols)
    cols = np.where(cols >= m, idx - off, cols)
    cols = np.clip(cols, 0, m - 1)
    vals = g.uniform(0.0, 1.0, size=(m, w))
    diag = -(vals.sum(1) + 1.0)
    cols = np.concatenate([cols, idx], axis=1)
    vals = np.concatenate([vals, diag[:, None]], axis=1)
    order = np.argsort(cols, axis=1, kind="stable")
    cols = np.take_along_axis(cols, order, 1)
    vals = np.take_along_axis(vals, order, 1)
    rowptr = np.arange(m + 1, dtype=np.int64) * nnz_row
    return rowptr, cols.ravel().astype(np.int32), vals.ravel()


def banded_random_block(m_global, r0, r1, nnz_row=27, bandwidth=4096, seed=0):
    """Rows [r0, r1) of a banded-random matrix with m_global rows; column indices are GLOBAL.  Each row block is
    generated independently (seeded by (seed, r0)) so that every rank of a row-partitioned run builds only its block."""
    g = np.random.default_rng([seed, int(r0)])
    ml = r1 - r0
    idx = np.arange(r0, r1, dtype=np.int64)[:, None]
    w = nnz_row - 1
    off = g.integers(1, bandwidth + 1, size=(ml, w)) * g.choice(np.array([-1, 1]), size=(ml, w))
    cols = idx + off
    cols = np.where(cols < 0, idx - off, cols)
    cols = np.where(cols >= m_global, idx - off, cols)
    cols = np.clip(cols, 0, m_global - 1)
    vals = g.uniform(0.0, 1.0, size=(ml, w))
    diag = -(vals.sum(1) + 1.0)
    cols = np.concatenate([cols, idx], axis=1)
    vals = np.concatenate([vals, diag[:, None]], axis=1)
    order = np.argsort(cols, axis=1, kind="stable")
    cols = np.take_along_axis(cols, order, 1)
    vals = np.take_along_axis(vals, order, 1)
    rowptr = np.arange(ml + 1, dtype=np.int64) * nnz_row
    return rowptr, cols.ravel(), vals.ravel()


def stencil27_block(nx, ny, nz, z0, z1, random_values=False, seed=0):
    """Rows of the z-planes [z0, z1) of the 27-point stencil on an nx x ny x nz grid (z-slab partition of
    config C4); column indices are GLOBAL."""
    rowptr, col, val = stencil27(nx, ny, z1 - z0 + (1 if z0 > 0 else 0) + (1 if z1 < nz else 0), random_values, seed + z0)
    # build on the slab extended by one ghost plane on each interior side, then cut the owned rows
    lo = 1 if z0 > 0 else 0
    plane = nx * ny
    r0, r1 = lo * plane, (lo + (z1 - z0)) * plane
    p0, p1 = rowptr[r0], rowptr[r1]
    shift = (z0 - lo) * plane
    return (rowptr[r0:r1 + 1] - p0).astype(np.int64), col[p0:p1].astype(np.int64) + shift, val[p0:p1]


def uniform_random(m, nnz_row=27, seed=0):
    """Like banded_random but off-diagonal columns uniform over all rows (report-only variant of C3)."""
    g = np.random.default_rng(seed)
    idx = np.arange(m, dtype=np.int64)[:, None]
    w = nnz_row - 1
    cols = g.integers(0, m, size=(m, w))
    vals = g.uniform(0.0, 1.0, size=(m, w))
    diag = -(vals.sum(1) + 1.0)
    cols = np.concatenate([cols, idx], axis=1)
    vals = np.concatenate([vals, diag[:, None]], axis=1)
    order = np.argsort(cols, axis=1, kind="stable")
    cols = np.take_along_axis(cols, order, 1)
    vals = np.take_along_axis(vals, order, 1)
    rowptr = np.arange(m + 1, dtype=np.int64) * nnz_row
    return rowptr, cols.ravel().astype(np.int32), vals.ravel()


def dense_stable(m, seed=1, shift=12.0):
    """Config C1: U(-1,1)^{m x m} - shift*I (raw U(-1,1) is not stable)."""
    g = np.random.default_rng(seed)
    return g.uniform(-1.0, 1.0, (m, m)) - shift * np.eye(m)


def dense_to_csr(A):
    m = A.shape[0]
    rowptr = np.arange(m + 1, dtype=np.int64) * A.shape[1]
    col = np.tile(np.arange(A.shape[1], dtype=np.int32), m)
    return rowptr, col, np.ascontiguousarray(A).ravel().astype(np.float64)


def rhs(m, p, seed=7):
    """B = U(-1,1)^{m x p}, column-major."""
    g = np.random.default_rng(seed)
    return np.asfortranarray(g.uniform(-1.0, 1.0, (m, p)))


def mass_diag(m, seed=11):
    """C5: M = diag(U(0.5, 1.5)) as CSR."""
    g = np.random.default_rng(seed)
    d = g.uniform(0.5, 1.5, m)
    return np.arange(m + 1, dtype=np.int64), np.arange(m, dtype=np.int32), d


def mass_tridiag(m):
    """C5 variant: SPD mass matrix tridiag(1/6, 2/3, 1/6)."""
    idx = np.arange(m, dtype=np.int64)
    cols = np.stack([idx - 1, idx, idx + 1], 1)
    cols[0, 0] = -1
    cols[m - 1, 2] = -1
    vals = np.tile(np.array([1.0 / 6.0, 2.0 / 3.0, 1.0 / 6.0]), (m, 1))
    return _csr_from_offsets(m, cols, vals)


def csr_rows(A, r0, r1):
    """Row block [r0, r1) of a CSR triple (global column indices kept)."""
    rowptr, col, val = A
    p0, p1 = rowptr[r0], rowptr[r1]
    return (rowptr[r0:r1 + 1] - p0).astype(np.int64), col[p0:p1], val[p0:p1]
