"""Diagnostic (under tests/ because it uses the CPU oracle): the projected matrices V'AV and V'BB'V of both HIP back ends, one trip from the
oracle's V after j trips of configs[1] at reduced size, against numpy (RAILS_DEBUG_DUMP_PROJECTED) -- how the round-2 deviation of the
coordinate-space back end was traced to its device basis.
    PYTHONPATH=. python tests/diag_projected.py        (on the GPU box)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rails_amd
from rails_amd import problems as P
from oracle.oracle import Oracle
import scipy.sparse as sp
orc = Oracle()
A = P.laplace7(12, 12, 10)
As = sp.csr_matrix((A[2], A[1], A[0]), shape=(1440, 1440))
B = P.rhs(1440, 8, seed=3)
params = {"Restart size": 64, "Reduced size": 32, "Expand size": 8, "Lanczos iterations": 20, "Tolerance": 1e-3}
seed = 4
def read(path):
    toks = open(path).read().split()
    k = int(toks[0]); d = np.array(toks[1:1 + 2 * k * k], dtype=float).reshape(k * k, 2)
    return d[:, 0].reshape(k, k).T, d[:, 1].reshape(k, k).T
for j in (3, 4, 6, 9):  # trips of the oracle run whose V is handed over
    head = orc.solve(A, B, orc.params({**params, "Maximum iterations": j, "rng_mode": 1, "seed": seed}))
    V0 = np.ascontiguousarray(head["V"])
    one = {**params, "Restart from solution": 1, "Maximum iterations": 1}
    res = {}
    for sub in (1, 0):
        path = "/tmp/proj_%d_%d.txt" % (j, sub)
        if os.path.exists(path): os.remove(path)
        os.environ["RAILS_DEBUG_DUMP_PROJECTED"] = path
        ctx = rails_amd.Context(device=0, seed=1)
        ctx.set_seed(seed + j, 0)
        op = rails_amd.HipOperatorWrapper(ctx, *A)
        s = rails_amd.Solver(ctx, op, B)
        s.set_parameters(one); s.set_option("verbose", 0); s.set_option("subspace", sub)
        code, V, T = s.solve(V0=V0)
        res[sub] = read(path) + (T,)
        s.close(); ctx.close()
    a_true = V0.T @ (As @ V0); b_true = (V0.T @ B) @ (B.T @ V0)
    for sub in (1, 0):
        a, b, T = res[sub]
        print("trip", j, "subspace", sub, "k", a.shape[0], "|a - V'AV| %.1e" % (np.abs(a - a_true).max() / np.abs(a_true).max()), "|b - V'BB'V| %.1e" % (np.abs(np.abs(b) - np.abs(b_true)).max() / np.abs(b_true).max()),
              "asym %.1e" % (np.abs(a - a.T).max() / np.abs(a).max()), flush=True)
    print("   T diff between back ends %.1e; a diff %.1e" % (np.abs(res[1][2] - res[0][2]).max() / np.abs(res[0][2]).max(), np.abs(res[1][0] - res[0][0]).max() / np.abs(res[0][0]).max()))
