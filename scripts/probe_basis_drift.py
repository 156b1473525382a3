"""Orthonormality of the coordinate-space back end's device basis over a long run (RAILS_SUBSPACE_VERIFY: P'P - I measured on the device
after every synchronous block; an overlapped block's columns are seen by the next synchronous one).
    PYTHONPATH=. RAILS_SUBSPACE_VERIFY=1 python scripts/probe_basis_drift.py [m] [trips]"""
import os, sys, time
import numpy as np
os.environ.setdefault("RAILS_SUBSPACE_VERIFY", "1")
import rails_amd
from rails_amd import problems as P

m = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
trips = int(sys.argv[2]) if len(sys.argv) > 2 else 300
if len(sys.argv) > 3 and sys.argv[3] == "laplace":  # a run that stagnates: many restarts, many nearly dependent blocks
    n = int(round(m ** (1 / 3)))
    A = P.laplace7(n, n, n)
    m = n * n * n
else:
    A = P.banded_random(m, 27, seed=1)  # BASELINE configs[2] pattern
B = P.rhs(m, 16, seed=2)
for overlap in ("1", "0"):
    os.environ["RAILS_SUBSPACE_OVERLAP"] = overlap
    ctx = rails_amd.Context(device=0, seed=1)
    op = rails_amd.HipOperatorWrapper(ctx, *A)
    s = rails_amd.Solver(ctx, op, B)
    s.set_parameters({"Restart size": 200, "Reduced size": 128, "Expand size": 16, "Lanczos iterations": 20, "Tolerance": 1e-12, "Maximum iterations": trips})
    s.set_option("verbose", 0)
    s.set_option("subspace", 1)
    t0 = time.time()
    code, V, T = s.solve()
    st = s.backend_stats()
    print("overlap", overlap, "code", code, "trips", s.trips(), "%.1f s" % (time.time() - t0), "V'V - I %.1e" % np.abs(V.T @ V - np.eye(V.shape[1])).max(),
          {k: st[k] for k in ("dim", "absorb", "one_by_one", "dropped", "compress", "overlapped_blocks", "verify_representation", "verify_orthonormality")}, flush=True)
    s.close(); ctx.close()
