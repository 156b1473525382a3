"""diagnostic (not a test): generalized problem at a tight tolerance on the default back end"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rails_amd  # noqa: E402
from rails_amd import problems as P  # noqa: E402

m = int(os.environ.get("M", "200000"))
ctx = rails_amd.Context(device=0, seed=1)
A = P.banded_random(m, 27, 4096, seed=1)
M = P.mass_diag(m, seed=11)
B = P.rhs(m, 16, seed=8)
prm = {"Restart size": 200, "Reduced size": 128, "Expand size": 16, "Lanczos iterations": 20, "Tolerance": float(os.environ.get("TOL", "1e-8"))}
op = rails_amd.HipOperatorWrapper(ctx, *A)
mop = rails_amd.HipOperatorWrapper(ctx, *M)
s = rails_amd.Solver(ctx, op, B, M=mop)
assert s.set_parameters(prm) == 0
s.set_option("verbose", 1)
s.set_option("mass", 1)
s.set_option("max_trips", 400)
s.set_option("subspace", int(os.environ.get("SUB", "1")))
code, V, T = s.solve()
print("code", code, "trips", s.trips(), "V", V.shape, "T", T.shape, "backend", s.backend_stats())
print("relative residual", s.relative_residual())
