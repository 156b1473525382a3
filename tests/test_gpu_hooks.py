"""The collective hooks on a real GPU: a world_size = 1 RCCL group (the most this 1-GPU box allows) drives the
library's all-reduce hook through torch.distributed on device pointers and on the library's stream."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_allreduce_hook_over_rccl_single_rank(oracle):
    import torch
    import torch.distributed as dist

    import rails_amd
    from rails_amd import partition, problems as P

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        ts = torch.cuda.Stream(device=0)
        torch.cuda.set_stream(ts)
        ctx = rails_amd.Context(device=0, stream=ts.cuda_stream, seed=5)
        calls = []
        inner = partition.make_allreduce(on_device=True)

        def hook(ptr, n, stream):
            calls.append(n)
            return inner(ptr, n, stream)

        ctx.set_allreduce(hook)
        # wrap_buffer sees the library's device memory
        g = np.random.default_rng(0)
        Xh = g.uniform(-1, 1, (4000, 24))
        X = rails_amd.HipMultiVectorWrapper(ctx, data=Xh)
        t = partition.wrap_buffer(ctx.lib.rails_panel_device_ptr(X.panel.h), 4, True)
        ctx.sync()
        np.testing.assert_array_equal(t.cpu().numpy(), Xh[0, :4])
        # Gram through the hook (sum over one rank = identity)
        C = X.dot(X)
        np.testing.assert_allclose(C, Xh.T @ Xh, atol=1e-10)
        assert calls and calls[-1] == 24 * 24
        # a whole solve with the hook in every reduction (Gram, Lanczos sums, orthogonalisation)
        A = P.laplace7(12, 10, 8)
        m = A[0].size - 1
        B = P.rhs(m, 6, seed=3)
        params = {"Restart size": 80, "Reduced size": 40, "Expand size": 4, "Lanczos iterations": 8, "Tolerance": 1e-6}
        op = rails_amd.HipOperatorWrapper(ctx, *A)
        ctx.set_seed(11, 0)
        sv = rails_amd.Solver(ctx, op, B)
        assert sv.set_parameters(params) == 0
        sv.set_option("verbose", 0)
        sv.set_option("subspace", 0)  # direct back end: one all-reduce per Lanczos step + projections
        n0 = len(calls)
        code, V, T = sv.solve()
        assert code == 0 and len(calls) - n0 > 9 * sv.trips()
        # the default (coordinate-space) back end through the same hook: a handful of block reductions per trip
        ctx.set_seed(11, 0)
        sv.set_option("subspace", 1)
        n1 = len(calls)
        code, V2, T2 = sv.solve()
        assert code == 0 and sv.backend_stats()["absorb"] > 0 and 0 < len(calls) - n1 < 12 * sv.trips()
        X2 = V2 @ T2 @ V2.T
        Xd = V @ T @ V.T
        assert np.linalg.norm(X2 - Xd) / np.linalg.norm(Xd) < 1e-4
        out = oracle.solve(A, B, oracle.params({**params, "rng_mode": 1, "seed": 11}))
        Xg, Xo = V @ T @ V.T, out["V"] @ out["T"] @ out["V"].T
        assert np.linalg.norm(Xg - Xo) / np.linalg.norm(Xo) < 1e-4
        sv.close()
        ctx.close()
    finally:
        torch.cuda.set_stream(torch.cuda.default_stream(0))
        dist.destroy_process_group()


def test_native_rccl_single_rank(oracle):
    """The library's own RCCL communicator (rails_ctx_init_rccl) on a world of one rank -- the most this 1-GPU box allows: every
    reduction of a solve goes through ncclAllReduce on the context's stream, a zero-ghost plan through the grouped send/recv form."""
    import rails_amd
    from rails_amd import partition, problems as P

    ctx = rails_amd.Context(device=0, seed=5)
    try:
        assert ctx.rccl_size() == 0
        ctx.init_rccl(rails_amd.Context.rccl_unique_id(), 1, 0)
        assert ctx.rccl_size() == 1
        g = np.random.default_rng(0)
        Xh = g.uniform(-1, 1, (4000, 24))
        X = rails_amd.HipMultiVectorWrapper(ctx, data=Xh)
        n0 = ctx.stats()["allreduce"]
        C = X.dot(X)
        np.testing.assert_allclose(C, Xh.T @ Xh, atol=1e-10)
        assert ctx.stats()["allreduce"] == n0 + 1
        A = P.laplace7(12, 10, 8)
        m = A[0].size - 1
        B = P.rhs(m, 6, seed=3)
        params = {"Restart size": 80, "Reduced size": 40, "Expand size": 4, "Lanczos iterations": 8, "Tolerance": 1e-6}
        # a one-rank partition with its (empty) ghost plan installed without a hook: the product takes the RCCL exchange path
        plan = partition.HaloPlan(np.array([0, m]), 0, A[1].astype(np.int64), lambda obj: [obj])
        assert plan.n_ghost == 0 and plan.n_send == 0
        op = rails_amd.HipOperatorWrapper(ctx, A[0], plan.col_local, A[2], ncols_ext=m)
        op.set_halo(plan, None)
        Yh = op.apply(rails_amd.HipMultiVectorWrapper(ctx, data=Xh[:m, :8].copy())).to_host()
        ref = oracle.csr_spmm(*A, Xh[:m, :8])
        assert np.abs(Yh - ref).max() <= 1e-13 * np.abs(ref).max()
        for subspace in (0, 1):
            ctx.set_seed(11, 0)
            sv = rails_amd.Solver(ctx, op, B)
            assert sv.set_parameters(params) == 0
            sv.set_option("verbose", 0)
            sv.set_option("subspace", subspace)
            n1 = ctx.stats()["allreduce"]
            code, V, T = sv.solve()
            assert code == 0 and ctx.stats()["allreduce"] - n1 > 2 * sv.trips()
            out = oracle.solve(A, B, oracle.params({**params, "rng_mode": 1, "seed": 11}))
            Xg, Xo = V @ T @ V.T, out["V"] @ out["T"] @ out["V"].T
            assert np.linalg.norm(Xg - Xo) / np.linalg.norm(Xo) < 1e-4
            sv.close()
        # dropping the communicator (bench.py's fall-back to the hooks when another rank could not set it up)
        ctx.set_rccl(None)
        assert ctx.rccl_size() == 0
    finally:
        ctx.close()
