"""The reference's application problem as its own test builds it (matlab/test/test_MOC.m:94-163) from the data set it ships
(matlab/DataErik/, packed into tests/golden/moc_erik.npz by tests/golden/make_moc_fixture.py): the MOC ocean model, n = 8*8*4*6 = 1536
unknowns (6 per grid cell: u, v, w, p, T, S), mass matrix zero except on temperature and salinity, forcing on salinity only, and the two
border rows / columns that pin the checkerboard pressure modes."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load():
    import scipy.sparse as sp

    d = np.load(os.path.join(ROOT, "tests", "golden", "moc_erik.npz"))
    n = int(d["n"])
    A = sp.csr_matrix((d["co"], d["jco"] - 1, d["beg"] - 1), shape=(n, n))  # beg / jco are 1-based (test_MOC.m:108-121)
    pos = np.arange(n) % 6
    mdiag = d["mdiag"].copy()
    mdiag[pos < 4] = 0.0          # "set everything but temperature and salinity to zero" (:123-126)
    F = d["frc"].copy()
    F[pos < 5] = 0.0              # "set everything but salinity to zero" (:128-131)
    B = 0.1 * F[:, None]          # "spatially correlated noise" (:133-134)
    return A, mdiag, B


def add_border(A, mdiag, B):
    """test_MOC.m:137-163: the nullspace of the pressure added as a border"""
    import scipy.sparse as sp

    n = A.shape[0]
    j = np.arange(n)
    sel = j[j % 6 == 3]
    cell = sel // 6
    which = ((cell % 4) + ((cell // 4) % 16)) % 2  # 0 -> row n, 1 -> row n+1 (0-based)
    Cb = sp.csr_matrix((np.ones(sel.size), (sel, which)), shape=(n, 2))
    A2 = sp.bmat([[A, Cb], [Cb.T, None]], format="csr")
    A2.sort_indices()
    return A2, np.concatenate([mdiag, [0.0, 0.0]]), np.vstack([B, np.zeros((2, B.shape[1]))])


def schur_dense(A2, m2diag, B2, tol=1e-12):
    """matlab/RAILSschur.m:23-45 with dense algebra (host reference for the tests)"""
    d = np.asarray(m2diag)
    i1, i2 = np.flatnonzero(np.abs(d) < tol), np.flatnonzero(np.abs(d) >= tol)
    Ad = A2.toarray()
    S = Ad[np.ix_(i2, i2)] - Ad[np.ix_(i2, i1)] @ np.linalg.solve(Ad[np.ix_(i1, i1)], Ad[np.ix_(i1, i2)])
    return S, d[i2], B2[i2], i1, i2
