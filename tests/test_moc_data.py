"""The reference's MOC data set (tests/golden/moc_erik.npz) on the CPU: the fixture is what test_MOC.m expects, and the oracle's
generalized solve on its Schur complement satisfies the reference's acceptance (matlab/test/test_MOC.m:27-36) and agrees with an
independent dense solve (scipy's Bartels-Stewart)."""
import numpy as np

from moc_problem import add_border, load, schur_dense

PARAMS = {"Maximum iterations": 1000, "Tolerance": 1e-3, "Expand size": 3, "Lanczos iterations": 10}


def test_fixture_is_the_problem_of_test_MOC():
    A, mdiag, B = load()
    n = A.shape[0]
    assert n == 8 * 8 * 4 * 6 and A.nnz == 17364       # test_MOC.m:17, DataErik/Ap1.info
    assert np.count_nonzero(mdiag) == 2 * n // 6 and np.all(mdiag[np.arange(n) % 6 < 4] == 0)
    assert B.shape == (n, 1) and np.count_nonzero(B) > 0 and np.all(B[np.arange(n) % 6 != 5] == 0)
    A2, m2, B2 = add_border(A, mdiag, B)
    assert A2.shape == (n + 2, n + 2) and A2.nnz == A.nnz + 2 * (n // 6)
    assert abs(A2 - A2.T)[n:, :].nnz == 0               # the border is symmetric


def test_oracle_on_the_schur_complement_of_the_moc_problem(oracle):
    import scipy.linalg as sl

    from rails_amd import problems as P

    A, mdiag, B = load()
    A2, m2, B2 = add_border(A, mdiag, B)
    S, ms, BS, i1, i2 = schur_dense(A2, m2, B2)
    assert S.shape == (512, 512) and np.abs(B2[i1]).max() == 0.0
    m = S.shape[0]
    Mcsr = (np.arange(m + 1, dtype=np.int64), np.arange(m, dtype=np.int32), ms.copy())
    oracle.srand(1)
    out = oracle.solve(P.dense_to_csr(S), BS, oracle.params({**PARAMS, "rng_mode": 1, "seed": 1}), M=Mcsr)
    assert out["ret"] == 0
    V, T = out["V"], out["T"]
    # the reference's acceptance: || S V T V' MS' + MS V T V' S' + BS BS' ||_F < 1e-3 (test_MOC.m:30-31)
    X = V @ T @ V.T
    R = S @ X * ms[None, :] + (ms[:, None] * X) @ S.T + BS @ BS.T
    assert np.linalg.norm(R) < 1e-3
    assert np.linalg.norm(R, 2) < 1e-3 * np.linalg.norm(BS.T @ BS, 2) * 1.5
    # independent dense solve of the same generalized equation: (MS^-1 S) X + X (MS^-1 S)' + (MS^-1 BS)(MS^-1 BS)' = 0
    Sm, Bm = S / ms[:, None], BS / ms[:, None]
    Xref = sl.solve_continuous_lyapunov(Sm, -Bm @ Bm.T)
    assert np.linalg.norm(X - Xref) / np.linalg.norm(Xref) < 2e-2  # tolerance 1e-3 on the residual, not on X


def test_begjco_text_format_round_trip(tmp_path):
    """the plain-text CSR format of matlab/DataErik/ (test_MOC.m:94-121) through rails_amd.mmio"""
    import pytest

    from rails_amd import mmio

    A, mdiag, B = load()
    stem = str(tmp_path / "Ap1")
    mmio.write_begjco(stem, A.indptr, A.indices, A.data)
    np.savetxt(str(tmp_path / "Bp1.co"), mdiag, fmt="%26.16E")
    text = open(stem + ".co").read().splitlines()
    assert len(text) == A.nnz and "E" in text[0]
    m, n, rp, col, val = mmio.read_csr(stem)
    As = A.copy()
    As.sort_indices()
    assert (m, n) == A.shape and np.array_equal(rp, As.indptr) and np.array_equal(col, As.indices) and np.array_equal(val, As.data)
    m, n, rp, col, val = mmio.read_csr(str(tmp_path / "Bp1.co"))
    assert m == n == mdiag.size and np.array_equal(val, mdiag) and np.array_equal(col, np.arange(n))
    assert np.array_equal(mmio.read_dense(str(tmp_path / "Bp1.co"))[:, 0], mdiag)
    open(stem + ".beg", "w").write("1\n5\n3\n")
    with pytest.raises(mmio.MatrixMarketError):
        mmio.read_csr(stem)
