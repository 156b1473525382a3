"""Worker of tests/test_distributed_cpu.py: one rank of a row-partitioned run on the CPU (gloo).

The partition plan and the two exchange hooks are the product's host logic (rails_amd.partition); the local
compute is the CPU oracle in row-partitioned mode.  Rank 0 checks the result against a single-rank solve."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch.distributed as dist

    from oracle.oracle import Oracle
    from rails_amd import partition, problems as P

    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    case = sys.argv[1]
    orc = Oracle()
    orc.set_num_threads(2)
    if case == "banded":
        A = P.banded_random(3000, 9, 150, seed=5)
        # Lanczos iterations <= rank of the residual operator at the first trip (2 + p): no near-breakdown, so the
        # trajectories of the partitioned and the single-rank run stay comparable at rounding level
        params = {"Restart size": 40, "Reduced size": 20, "Expand size": 4, "Lanczos iterations": 8, "Tolerance": 1e-4}
        p = 8
    else:
        A = P.laplace7(12, 10, 9)
        params = {"Restart size": 48, "Reduced size": 24, "Expand size": 3, "Lanczos iterations": 6, "Tolerance": 1e-5}
        p = 6
    m = A[0].size - 1
    B = P.rhs(m, p, seed=9)
    starts = partition.row_ranges(m, world)
    r0, r1 = int(starts[rank]), int(starts[rank + 1])
    rp, col, val = P.csr_rows(A, r0, r1)
    plan = partition.HaloPlan(starts, rank, col.astype(np.int64), partition.all_gather_object_fn())
    # the two exchanges, on host buffers over gloo
    orc.set_partition(partition.make_allreduce(on_device=False), partition.make_halo(plan, on_device=False), plan, m_global=m)
    # SpMM check: local rows of A*X against the global product
    X = np.random.default_rng(1).uniform(-1, 1, (m, 5))
    spmm_err = float(np.abs(orc.op_apply(rp, plan.col_local, val, X[r0:r1]) - orc.csr_spmm(*A, X)[r0:r1]).max())
    gram_err = float(np.abs(orc.dot(X[r0:r1], X[r0:r1]) - X.T @ X).max())
    prm = orc.params({**params, "rng_mode": 1, "seed": 21, "row0": r0})
    out = orc.solve((rp, plan.col_local, val), B[r0:r1], prm, vcap=params["Restart size"] + params["Expand size"])
    # gather V
    Vs = [None] * world
    dist.all_gather_object(Vs, out["V"])
    result = {"rank": rank, "ret": out["ret"], "trips": out["trips"], "spmm_err": spmm_err, "gram_err": gram_err}
    if rank == 0:
        V = np.vstack(Vs)
        orc.set_partition(None, None, None, 0)
        ref = orc.solve(A, B, orc.params({**params, "rng_mode": 1, "seed": 21}), vcap=params["Restart size"] + params["Expand size"])
        Xd, Xs = V @ out["T"] @ V.T, ref["V"] @ ref["T"] @ ref["V"].T
        n = min(6, len(out["res_hist"]), len(ref["res_hist"]))
        result.update(ref_ret=ref["ret"], ref_trips=ref["trips"], rel=float(np.linalg.norm(Xd - Xs) / np.linalg.norm(Xs)),
                      hist_err=float(np.max(np.abs(out["res_hist"][:n] - ref["res_hist"][:n]) / np.abs(ref["res_hist"][:n]))),
                      orth=float(np.abs(V.T @ V - np.eye(V.shape[1])).max()), n_ghost=plan.n_ghost, n_send=plan.n_send)
        print("RESULT " + json.dumps(result), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
