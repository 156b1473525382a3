#!/usr/bin/env python3
"""Diagnostic (lives under tests/ because it uses the CPU oracle, which only test code may touch): residual-estimate history of the
direct back end, its coefficient-space Lanczos and the coordinate-space back end against the oracle, trip by trip.
    python tests/diag_backends.py        (on the GPU box)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rails_amd
from rails_amd import problems as P
from oracle.oracle import Oracle

def run(ctx, A, B, params, seed, proj):
    ctx.set_seed(seed, 0)
    csr = A if isinstance(A, tuple) else P.dense_to_csr(A)
    op = rails_amd.HipOperatorWrapper(ctx, *csr)
    s = rails_amd.Solver(ctx, op, B)
    assert s.set_parameters(params) == 0
    s.set_option("verbose", 0)
    s.set_option("projected_lanczos", 1 if proj == 1 else 0)
    s.set_option("subspace", 1 if proj == 2 else 0)
    b = ctx.stats()
    code, V, T = s.solve()
    a = ctx.stats()
    if proj == 2:
        print("   backend:", s.backend_stats(), "orth %.1e" % np.abs(V.T @ V - np.eye(V.shape[1])).max(), "rel %.2e" % s.relative_residual())
    return code, s.history(), s.trips(), a["lanczos_start"] - b["lanczos_start"], V @ T @ V.T

def main():
    ctx = rails_amd.Context(device=0, seed=1)
    orc = Oracle()
    cases = []
    A = P.dense_stable(256, seed=1)
    cases.append(("dense256 p8 L8", A, P.rhs(256, 8, seed=2), {"Restart size": 64, "Reduced size": 32, "Expand size": 4, "Lanczos iterations": 8, "Tolerance": 1e-3}, 1))
    A = P.laplace7(20, 20, 15)
    cases.append(("laplace 20x20x15 p8 L10", A, P.rhs(6000, 8, seed=5), {"Restart size": 64, "Reduced size": 32, "Expand size": 8, "Lanczos iterations": 10, "Tolerance": 1e-3}, 3))
    cases.append(("laplace 20x20x15 p8 L10 tol1e-6", A, P.rhs(6000, 8, seed=5), {"Restart size": 96, "Reduced size": 48, "Expand size": 8, "Lanczos iterations": 10, "Tolerance": 1e-6}, 3))
    A = P.laplace7(40, 40, 30)
    cases.append(("laplace 40x40x30 p8 L10", A, P.rhs(48000, 8, seed=5), {"Restart size": 96, "Reduced size": 48, "Expand size": 8, "Lanczos iterations": 10, "Tolerance": 1e-4}, 3))
    for name, A, B, prm, seed in cases:
        out = orc.solve(A, B, orc.params({**prm, "rng_mode": 1, "seed": seed}))
        ho = np.array(out["res_hist"])
        Xo = out["V"] @ out["T"] @ out["V"].T
        for proj in (0, 1, 2):
            code, h, trips, nstart, X = run(ctx, A, B, prm, seed, proj)
            h = np.array(h)
            n = min(len(h), len(ho))
            rel = np.abs(h[:n] - ho[:n]) / np.abs(ho[:n])
            print("%s proj=%d code=%d trips=%d (oracle %d) projected_trips=%d |dX|/|X|=%.2e" % (name, proj, code, trips, out["trips"], nstart, np.linalg.norm(X - Xo) / np.linalg.norm(Xo)))
            print("   res/oracle-1 per trip:", " ".join("%.0e" % r for r in rel))
            print("   res:", " ".join("%.2e" % r for r in h[:n]), flush=True)
    ctx.close()

main()
