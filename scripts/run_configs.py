#!/usr/bin/env python3
"""Run the BASELINE.json configurations that fit one GPU to convergence (SURVEY.md section 8(d) definitions) and
print one JSON line per configuration: trips, wall time, iterations/s, V.N(), relative residual, kernel counters.

    [RAILS_RUN_SUBSPACE=1] [RAILS_RUN_TOL=1e-8] python scripts/run_configs.py [c1 c2 c3 c3s c3u c4slab c5]

c3  = banded-random (SURVEY primary), c3s = the same size with the 27-point stencil pattern, c4slab = ONE rank's share of
config 4 (27-point stencil, 1M rows, B m x 32, Restart 256 / Reduced 128 / Expand 32 / Lanczos 40) on one GPU,
c5  = generalized SPD mass matrix + warm start after a 1 % perturbation of A's diagonal."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(ctx, name, A, B, params, M=None, V0=None, max_trips=400, seed=1):
    import rails_amd

    ctx.set_seed(seed, 0)
    if os.environ.get("RAILS_RUN_TOL"):  # the same configurations at another tolerance (robustness sweeps)
        params = dict(params, Tolerance=float(os.environ["RAILS_RUN_TOL"]))
    op = rails_amd.HipOperatorWrapper(ctx, *A)
    mop = rails_amd.HipOperatorWrapper(ctx, *M) if M is not None else None
    s = rails_amd.Solver(ctx, op, B, M=mop)
    assert s.set_parameters(params) == 0
    s.set_option("verbose", 0)
    s.set_option("max_trips", max_trips)
    if M is None and os.environ.get("RAILS_RUN_PROJECTED", "0") == "1":
        s.set_option("projected_lanczos", 1)
    s.set_option("subspace", 1 if os.environ.get("RAILS_RUN_SUBSPACE", "0") == "1" else 0)
    if M is not None:
        s.set_option("mass", 1)
    ctx.sync()
    t0 = time.perf_counter()
    code, V, T = s.solve(V0=V0, fetch=False)
    ctx.sync()
    dt = time.perf_counter() - t0
    rel = s.relative_residual()
    out = {"config": name, "m": int(A[0].size - 1), "p": int(B.shape[1]), "params": params, "code": code, "trips": s.trips(), "seconds": dt,
           "iterations_per_s": s.trips() / dt, "k_final": s.k, "relative_residual": rel, "host_sections": s.profile(), "backend": s.backend_stats(), "spmm_kernel": op.last_kernel(), "counters_cumulative": ctx.stats()}
    print(json.dumps(out), flush=True)
    return s, op


def main():
    import rails_amd
    from rails_amd import problems as P

    which = sys.argv[1:] or ["c1", "c2", "c3", "c3s", "c4slab", "c5"]
    ctx = rails_amd.Context(device=0, seed=1)
    if "c1" in which:
        A = P.dense_to_csr(P.dense_stable(256, seed=1))
        run(ctx, "C1 dense m=256 (CSR of the dense matrix)", A, P.rhs(256, 4, seed=2),
            {"Restart size": 32, "Reduced size": 16, "Expand size": 3, "Lanczos iterations": 10, "Tolerance": 1e-3})
    if "c2" in which:
        A = P.laplace7(50, 50, 40)
        run(ctx, "C2 7-pt Laplacian 50x50x40", A, P.rhs(100000, 8, seed=3),
            {"Restart size": 64, "Reduced size": 32, "Expand size": 8, "Lanczos iterations": 20, "Tolerance": 1e-3})
    prm3 = {"Restart size": 200, "Reduced size": 128, "Expand size": 16, "Lanczos iterations": 20, "Tolerance": 1e-3}
    if "c3" in which:
        A = P.banded_random(1000000, 27, 4096, seed=1)
        run(ctx, "C3 banded-random m=1M", A, P.rhs(1000000, 16, seed=8), prm3)
    if "c3s" in which:
        A = P.stencil27(100, 100, 100, random_values=True, seed=1)
        run(ctx, "C3 stencil-27 pattern m=1M", A, P.rhs(1000000, 16, seed=8), prm3)
    if "c3u" in which:  # SURVEY 8(d)'s secondary pattern: uniformly random columns (report-only)
        A = P.uniform_random(1000000, 27, seed=1)
        run(ctx, "C3 uniform-random m=1M", A, P.rhs(1000000, 16, seed=8), prm3)
    if "c4slab" in which:
        A = P.stencil27(200, 200, 25)
        run(ctx, "C4 one rank's slab 200x200x25 on one GPU", A, P.rhs(1000000, 32, seed=9),
            {"Restart size": 256, "Reduced size": 128, "Expand size": 32, "Lanczos iterations": 40, "Tolerance": 1e-3})
    if "c5" in which:
        A = P.banded_random(1000000, 27, 4096, seed=1)
        M = P.mass_diag(1000000, seed=11)
        B = P.rhs(1000000, 16, seed=8)
        s, op = run(ctx, "C5 generalized M=diag(U(0.5,1.5)), cold", A, B, prm3, M=M)
        V = s.V()
        rowptr, col, val = A
        val2 = val.copy()
        diag = col == np.repeat(np.arange(1000000), np.diff(rowptr))
        val2[diag] *= 1.01
        s.close()
        del op
        run(ctx, "C5 warm start after 1% diagonal perturbation", (rowptr, col, val2), B, {**prm3, "Restart from solution": 1}, M=M, V0=V, seed=2)
    print(json.dumps({"counters": ctx.stats()}), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
