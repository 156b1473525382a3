"""Round 3 probe: where the coordinate-space back end's deviation from the oracle in the trip-by-trip test of configs[1] comes from
(tests/test_gpu_solver.py::test_reference_parameter_sets_trip_by_trip): per trip, estimate and solution deviations with the overlapped
block orthogonalisation on and off."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rails_amd
from rails_amd import problems as P
from oracle.oracle import Oracle
orc = Oracle()
A = P.laplace7(12, 12, 10)
B = P.rhs(1440, 8, seed=3)
params = {"Restart size": 64, "Reduced size": 32, "Expand size": 8, "Lanczos iterations": 20, "Tolerance": 1e-3}
seed = 4
for overlap in ("1", "0"):
    os.environ["RAILS_SUBSPACE_OVERLAP"] = overlap
    for sub in (1, 0):
        ctx = rails_amd.Context(device=0, seed=1)
        line = []
        for j in range(1, 10):
            head = orc.solve(A, B, orc.params({**params, "Maximum iterations": j, "rng_mode": 1, "seed": seed}))
            if head["ret"] == 0:
                break
            V0 = np.ascontiguousarray(head["V"])
            one = {**params, "Restart from solution": 1, "Maximum iterations": 1}
            out = orc.solve(A, B, orc.params({**one, "rng_mode": 1, "seed": seed + j}), V0=V0)
            ctx.set_seed(seed + j, 0)
            op = rails_amd.HipOperatorWrapper(ctx, *A)
            s = rails_amd.Solver(ctx, op, B)
            s.set_parameters(one); s.set_option("verbose", 0); s.set_option("subspace", sub)
            code, V, T = s.solve(V0=V0)
            h, ho = s.history(), out["res_hist"]
            Xg, Xo = V @ T @ V.T, out["V"] @ out["T"] @ out["V"].T
            k0 = V0.shape[1]
            As = __import__("scipy.sparse", fromlist=["x"]).csr_matrix((A[2], A[1], A[0]), shape=(1440, 1440))
            line.append("%d: est %.1e X %.1e |V-V0| %.1e |V'V-I| %.1e |T-To| %.1e" % (j, abs(h[0] - ho[0]) / abs(ho[0]), np.linalg.norm(Xg - Xo) / np.linalg.norm(Xo),
                        np.abs(V[:, :k0] - V0).max() if V.shape[1] >= k0 else -1, np.abs(V.T @ V - np.eye(V.shape[1])).max(), np.abs(T - out["T"]).max() / np.abs(out["T"]).max()) + (" " + str({kk: vv for kk, vv in s.backend_stats().items() if kk in ("dim", "one_by_one", "dropped", "delicate_blocks", "reprojected_blocks", "replaced_columns")}) if sub else ""))
            s.close()
        print("overlap", overlap, "subspace", sub, " | ".join(line), flush=True)
        ctx.close()
