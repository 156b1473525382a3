"""Orthogonality of the final V of the full-size configuration on both back ends, before and after the sweep schedule exists
(diagnostic for tests/test_gpu_fullsize.py::test_solver_full_size_both_back_ends)."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rails_amd
from rails_amd import problems as P
from rails_amd.wrappers import HipMultiVectorWrapper as MV
M_ROWS = 1_000_000
PARAMS = {"Restart size": 200, "Reduced size": 128, "Expand size": 16, "Lanczos iterations": 20, "Tolerance": 1e-6}
ctx = rails_amd.Context(device=0, seed=1)
A = P.banded_random(M_ROWS, 27, 4096, seed=0)
op = rails_amd.HipOperatorWrapper(ctx, *A)
B = P.rhs(M_ROWS, 16, seed=7)
for prep in (0, 1):
    if prep:
        op.prepare(128)
    for subspace in (1, 0):
        ctx.set_seed(1, 0)
        s = rails_amd.Solver(ctx, op, B)
        s.set_parameters(PARAMS); s.set_option("verbose", 0); s.set_option("subspace", subspace)
        code, V, T = s.solve()
        k = V.shape[1]
        Vd = MV(ctx, data=V)
        print("prep", prep, "subspace", subspace, "code", code, "k", k, "trips", s.trips(), "V'V-I %.3e" % np.abs(Vd.dot(Vd) - np.eye(k)).max(), "fused", ctx.stats().get("update_gram_fused"), flush=True)
        s.close(); del Vd
