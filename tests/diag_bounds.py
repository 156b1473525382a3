"""Round 3: the measured values behind two acceptance bounds: V'V - I of the direct back end at full size
(tests/test_gpu_fullsize.py) and the trip counts of the MOC solve on both back ends against the oracle (tests/test_gpu_moc.py)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rails_amd
from rails_amd import problems as P
from rails_amd.wrappers import HipMultiVectorWrapper as MV
from oracle.oracle import Oracle
M_ROWS = 1_000_000
PARAMS = {"Restart size": 200, "Reduced size": 128, "Expand size": 16, "Lanczos iterations": 20, "Tolerance": 1e-6}
ctx = rails_amd.Context(device=0, seed=1)
A = P.banded_random(M_ROWS, 27, 4096, seed=0)
op = rails_amd.HipOperatorWrapper(ctx, *A)
B = P.rhs(M_ROWS, 16, seed=7)
for seed in (1, 2, 3):
    for subspace in (1, 0):
        ctx.set_seed(seed, 0)
        s = rails_amd.Solver(ctx, op, B)
        s.set_parameters(PARAMS); s.set_option("verbose", 0); s.set_option("subspace", subspace)
        code, V, T = s.solve()
        k = V.shape[1]
        Vd = MV(ctx, data=V)
        print("fullsize seed", seed, "subspace", subspace, "code", code, "k", k, "trips", s.trips(), "V'V-I %.3e" % np.abs(Vd.dot(Vd) - np.eye(k)).max(), flush=True)
        s.close(); del Vd
del op
ctx.close()
from moc_problem import add_border, load, schur_dense
from rails_amd.schur import SchurOperator
A, mdiag, B = load()
A2, m2, B2 = add_border(A, mdiag, B)
PM = {"Maximum iterations": 1000, "Tolerance": 1e-3, "Expand size": 3, "Lanczos iterations": 10}
Sd, ms, BSd, i1, i2 = schur_dense(A2, m2, B2)
orc = Oracle()
m = Sd.shape[0]
out = orc.solve(P.dense_to_csr(Sd), BSd, orc.params({**PM, "rng_mode": 1, "seed": 1}), M=(np.arange(m + 1, dtype=np.int64), np.arange(m, dtype=np.int32), ms.copy()))
print("moc oracle trips", out["trips"], "ret", out["ret"], flush=True)
ho = out["res_hist"]
for subspace in (1, 0):
    ctx = rails_amd.Context(device=0, seed=1)
    S = SchurOperator(ctx, (A2.indptr.astype(np.int64), A2.indices.astype(np.int32), A2.data.astype(np.float64)), m2, tol=1e-12)
    BS = S.restrict(B2)
    Mop = rails_amd.HipOperatorWrapper(ctx, np.arange(S.m2 + 1, dtype=np.int64), np.arange(S.m2, dtype=np.int32), S.mass22)
    s = rails_amd.Solver(ctx, S.op, BS, M=Mop)
    s.set_parameters(PM); s.set_option("verbose", 0); s.set_option("mass", 1); s.set_option("subspace", subspace)
    code, V, T = s.solve()
    h = s.history()
    n = min(len(h), len(ho))
    rel = np.abs(h[:n] - ho[:n]) / np.abs(ho[:n])
    first_bad = int(np.argmax(rel > 1e-6)) if (rel > 1e-6).any() else n
    print("moc subspace", subspace, "code", code, "trips", s.trips(), "first trip whose estimate differs from the oracle's by more than 1e-6:", first_bad,
          "max rel diff over the first 20 trips %.2e" % rel[:20].max(), flush=True)
    s.close(); ctx.close()
