"""The MOC application problem (tests/test_gpu_moc.py) on the coordinate-space back end with RAILS_SUBSPACE_VERIFY=1: several hundred trips of
a stagnating generalized solve through the Schur-complement operator -- how orthonormal does the device basis stay, how well are blocks
represented?    PYTHONPATH=.:tests python scripts/probe_moc_verify.py"""
import os, sys
import numpy as np
os.environ["RAILS_SUBSPACE_VERIFY"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rails_amd
from rails_amd.schur import SchurOperator
from moc_problem import add_border, load

A, mdiag, B = load()
A2, m2, B2 = add_border(A, mdiag, B)
for overlap in ("1", "0"):
    os.environ["RAILS_SUBSPACE_OVERLAP"] = overlap
    ctx = rails_amd.Context(device=0, seed=1)
    S = SchurOperator(ctx, (A2.indptr.astype(np.int64), A2.indices.astype(np.int32), A2.data.astype(np.float64)), m2, tol=1e-12)
    BS = S.restrict(B2)
    Mop = rails_amd.HipOperatorWrapper(ctx, np.arange(S.m2 + 1, dtype=np.int64), np.arange(S.m2, dtype=np.int32), S.mass22)
    s = rails_amd.Solver(ctx, S.op, BS, M=Mop)
    s.set_parameters({"Maximum iterations": 1000, "Tolerance": 1e-3, "Expand size": 3, "Lanczos iterations": 10})
    s.set_option("verbose", 0); s.set_option("mass", 1); s.set_option("subspace", 1)
    code, V, T = s.solve()
    st = s.backend_stats()
    print("overlap", overlap, "code", code, "trips", s.trips(), "V'V - I %.1e" % np.abs(V.T @ V - np.eye(V.shape[1])).max(),
          {k: st[k] for k in ("dim", "absorb", "one_by_one", "dropped", "delicate_blocks", "reprojected_blocks", "replaced_columns", "compress", "overlapped_blocks", "verify_representation", "verify_orthonormality")}, flush=True)
    s.close(); ctx.close()
