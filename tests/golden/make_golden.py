#!/usr/bin/env python3
"""Generate tests/golden/ref_stl.npz from the reference's own Stl sources.

Run in the build container only (needs /root/reference and `make -C oracle ref`):

    python tests/golden/make_golden.py

Every array in the fixture is DATA: inputs drawn here with numpy, and the outputs the
compiled, unmodified reference (oracle/_ref/librails_ref.so: StlWrapper, StlVector,
LapackWrapper, Timer and the header-only Solver::resid_lanczos / compute_restart_vectors)
produced for them.  Solver::solve / dense_solve are not covered: they need SLICOT's
sb03md_, which is absent from this image (see oracle/README.md); those are pinned by the
reference's own known-answer tests instead (tests/test_oracle_kats.py).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.oracle import Reference, build  # noqa: E402


def main():
    build(ref=True)
    r = Reference()
    g = np.random.default_rng(20261004)
    out = {}

    # StlWrapper::random after srand(1)  (src/StlWrapper.cpp:414-423)
    r.srand(1)
    out["rng_srand1_6x3"] = r.random(6, 3)
    r.srand(12345)
    out["rng_srand12345_4x1"] = r.random(4, 1)

    # dot / operator* / transpose-multiply / norms
    m = 97
    X = g.uniform(-1, 1, (m, 5))
    Y = g.uniform(-1, 1, (m, 3))
    Cs = g.uniform(-1, 1, (5, 4))
    out["ops_X"], out["ops_Y"], out["ops_C"] = X, Y, Cs
    out["ops_dot"] = r.dot(X, Y)
    out["ops_mult"] = r.mult(X, Cs)
    out["ops_mult_t"] = r.mult_t(X, Y)
    out["ops_norm_X"] = np.array(r.norm(X))
    out["ops_norm_col0"] = np.array(r.norm(X[:, :1]))
    out["ops_norm_inf_C"] = np.array(r.norm_inf(Cs))

    # orthogonalize with the watermark: 3 columns, then push_back 2 more
    m = 60
    V1 = g.uniform(-1, 1, (m, 3))
    V2 = g.uniform(-1, 1, (m, 2))
    out["orth_V1"], out["orth_V2"] = V1, V2
    out["orth_out1"] = r.orthogonalize(V1)
    out["orth_out12"] = r.orthogonalize(V1, V2)
    # nearly dependent new column (exercises the second CGS pass)
    V3 = V1[:, :1] + 1e-7 * g.uniform(-1, 1, (m, 1))
    out["orth_V3"] = V3
    out["orth_out13"] = r.orthogonalize(V1, V3)

    # symmetric eigs + eigenvalue selection
    S = g.uniform(-1, 1, (7, 7))
    S = S + S.T
    d, Vv, info = r.eigs(S)
    out["eigs_S"], out["eigs_d"], out["eigs_V"] = S, d, Vv
    vals = np.array([0.3, -2.0, 1.5, -0.1, 4.0, -4.5, 0.0, 2.5])
    out["largest_vals"] = vals
    out["largest_idx5"] = r.find_largest(vals, 5)

    # resid_lanczos: well-separated from Krylov exhaustion (rank of R >> Lanczos steps)
    m, k, p, L = 240, 12, 4, 10
    A = g.uniform(-1, 1, (m, m)) - 14.0 * np.eye(m)
    V = r.orthogonalize(g.uniform(-1, 1, (m, k)))
    AV = A @ V
    B = g.uniform(-1, 1, (m, p))
    Tm = g.uniform(-1, 1, (k, k))
    Tm = 0.05 * (Tm + Tm.T)
    r.srand(5)
    q0 = r.random(m, 1)  # the start vector resid_lanczos will draw after srand(5)
    r.srand(5)
    lz = r.resid_lanczos(AV, V, Tm, B, L)
    out["lz_AV"], out["lz_V"], out["lz_T"], out["lz_B"] = AV, V, Tm, B
    out["lz_q0"] = q0
    out["lz_steps"] = np.array(lz["steps"])
    out["lz_H"] = lz["H"]
    out["lz_eigenvalues"] = lz["eigenvalues"]
    out["lz_eigenvectors"] = lz["eigenvectors"]

    # compute_restart_vectors
    Tr = g.uniform(-1, 1, (12, 12))
    Tr = Tr + Tr.T
    Tr[:, 3] *= 1e-9
    Tr[3, :] *= 1e-9
    out["rv_T"] = Tr
    out["rv_X_num8"] = r.compute_restart_vectors(Tr, 8, 1e-6)
    out["rv_X_all"] = r.compute_restart_vectors(Tr, -1, 1e-6)
    out["rv_X_tol"] = r.compute_restart_vectors(Tr, 12, 2.0)

    path = os.path.join(ROOT, "tests", "golden", "ref_stl.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
