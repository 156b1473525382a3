"""The row-partitioned path of the HIP library on ONE GPU: every rank is a thread with its own context,
stream, CSR row block, ghost-row plan and solver; the all-reduce and halo hooks exchange through host memory between the
threads (ctypes releases the GIL inside the library, the hooks re-acquire it).  This drives exactly the code an N-GPU
run drives -- rails_ctx_set_partition, global-row RNG streams, packing + ghost gather in rails_spmm, the hook call sites of
every reduction, replicated host numerics -- with only torch.distributed/RCCL replaced (that side is covered by the gloo
world-size 2/3 tests of test_distributed_cpu.py and the world-size-1 RCCL test of test_gpu_hooks.py).

Checked against the oracle on the undivided problem (same seeds: the counter-based generator is keyed on global rows, so
every partition draws the same start vectors) and against the single-rank GPU run."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class Ranks:
    """In-process stand-in for the collectives: rank threads meet at a barrier and exchange numpy arrays."""

    def __init__(self, n):
        self.n = n
        self.barrier = threading.Barrier(n, timeout=120)
        self.slots = [None] * n
        self.plans = [None] * n
        self.errors = []

    def all_gather_object(self, rank):
        def fn(obj):
            self.slots[rank] = obj
            self.barrier.wait()
            out = list(self.slots)
            self.barrier.wait()
            return out
        return fn

    def allreduce(self, rank, ctx):
        import torch
        from rails_amd.partition import wrap_buffer

        def hook(ptr, n, stream):
            ctx.sync()  # the library's stream has produced the buffer
            t = wrap_buffer(ptr, n, True)
            self.slots[rank] = t.cpu().numpy().copy()
            self.barrier.wait()
            total = np.zeros(n)
            for r in range(self.n):  # fixed order: every rank gets the same bits
                total += self.slots[r]
            self.barrier.wait()
            t.copy_(torch.from_numpy(total))
            torch.cuda.synchronize()
            return 0
        return hook

    def halo(self, rank, ctx, plan):
        import torch
        from rails_amd.partition import wrap_buffer

        self.plans[rank] = plan

        def hook(send_ptr, recv_ptr, ncols, stream):
            ctx.sync()
            send = wrap_buffer(send_ptr, plan.n_send * ncols, True)
            self.slots[rank] = send.cpu().numpy().copy()
            self.barrier.wait()
            recv = np.zeros(plan.n_ghost * ncols)
            ro = 0
            for src in range(self.n):  # my ghosts are grouped by owner, ranks ascending
                nr = int(plan.recv_counts[src]) * ncols
                if nr:
                    ps = self.plans[src]
                    so = int(ps.send_counts[:rank].sum()) * ncols  # src's send buffer is grouped by destination
                    assert int(ps.send_counts[rank]) * ncols == nr
                    recv[ro:ro + nr] = self.slots[src][so:so + nr]
                ro += nr
            self.barrier.wait()
            if plan.n_ghost:
                wrap_buffer(recv_ptr, plan.n_ghost * ncols, True).copy_(torch.from_numpy(recv))
            torch.cuda.synchronize()
            return 0
        return hook

    def run(self, target):
        """target(rank) -> result, one thread per rank; re-raises the first failure."""
        results = [None] * self.n

        def body(r):
            try:
                results[r] = target(r)
            except BaseException as e:  # noqa: BLE001 - reported to the main thread
                self.errors.append((r, e))
                self.barrier.abort()
        threads = [threading.Thread(target=body, args=(r,)) for r in range(self.n)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(600)
        if self.errors:
            real = [e for e in self.errors if not isinstance(e[1], threading.BrokenBarrierError)] or self.errors
            raise real[0][1]
        return results


def _rank_setup(ranks, r, starts, A, seed):
    import rails_amd
    from rails_amd import partition
    from rails_amd import problems as P

    m = int(starts[-1])
    ctx = rails_amd.Context(device=0, seed=seed)
    ctx.set_partition(r, ranks.n, int(starts[r]), m)
    rowptr, colg, val = P.csr_rows(A, int(starts[r]), int(starts[r + 1]))
    plan = partition.HaloPlan(starts, r, colg, ranks.all_gather_object(r))
    op = rails_amd.HipOperatorWrapper(ctx, rowptr, plan.col_local, val, ncols_ext=plan.m_local + plan.n_ghost)
    op.set_halo(plan, ranks.halo(r, ctx, plan))
    ctx.set_allreduce(ranks.allreduce(r, ctx))
    return ctx, op, plan


@pytest.mark.parametrize("nranks", [2, 3])
def test_partitioned_spmm_and_reductions(oracle, nranks):
    from rails_amd import partition
    from rails_amd import problems as P
    from rails_amd.wrappers import HipMultiVectorWrapper as MV

    A = P.banded_random(5000, 27, 300, seed=1)
    m = 5000
    g = np.random.default_rng(3)
    X = g.uniform(-1, 1, (m, 24))
    starts = partition.row_ranges(m, nranks)
    ranks = Ranks(nranks)

    def work(r):
        ctx, op, plan = _rank_setup(ranks, r, starts, A, seed=5)
        r0, r1 = int(starts[r]), int(starts[r + 1])
        out = {}
        for nc in (24, 16, 3):
            Y = op.apply(MV(ctx, data=X[r0:r1, :nc]))
            out["Y%d" % nc] = Y.to_host()
            if nc == 16:
                out["kernel16"] = op.last_kernel()  # the in-loop width: the lean kernel in its ghost-row form (spmm.hip kernel 1c)
        Xl = MV(ctx, data=X[r0:r1])
        out["gram"] = Xl.view(0, 7).dot(Xl.view(8, 23))  # all-reduced over the ranks
        Rn = MV(ctx, m=r1 - r0, n=3, capacity=4)
        Rn.random()  # stream 0, global rows r0..r1
        out["rand"] = Rn.to_host()
        out["ghosts"] = plan.n_ghost
        ctx.close()
        return out

    res = ranks.run(work)
    for nc in (24, 16, 3):
        Y = np.vstack([res[r]["Y%d" % nc] for r in range(nranks)])
        ref = oracle.csr_spmm(*A, X[:, :nc])
        assert np.abs(Y - ref).max() <= 1e-13 * np.abs(ref).max()
    G = X[:, :8].T @ X[:, 8:]
    for r in range(nranks):
        assert res[r]["ghosts"] > 0 and res[r]["kernel16"].startswith("k_spmm_narrow")
        np.testing.assert_allclose(res[r]["gram"], G, atol=1e-11)
        assert np.array_equal(res[r]["gram"], res[0]["gram"])  # replicated small objects are bit-identical on every rank
    assert np.array_equal(np.vstack([res[r]["rand"] for r in range(nranks)]), oracle.random(m, 3, mode=1, seed=5, stream=0))


@pytest.mark.parametrize("nranks,projected", [(2, 0), (3, 0), (2, 1), (2, 2), (3, 2)])
def test_partitioned_solve_matches_oracle_and_single_rank(oracle, nranks, projected):
    import rails_amd
    from rails_amd import partition
    from rails_amd import problems as P

    A = P.laplace7(20, 20, 15)
    m = A[0].size - 1
    B = P.rhs(m, 8, seed=5)
    params = {"Restart size": 64, "Reduced size": 32, "Expand size": 8, "Lanczos iterations": 10, "Tolerance": 1e-3}
    seed = 3
    starts = partition.row_ranges(m, nranks)
    ranks = Ranks(nranks)

    def work(r):
        ctx, op, plan = _rank_setup(ranks, r, starts, A, seed=seed)
        r0, r1 = int(starts[r]), int(starts[r + 1])
        s = rails_amd.Solver(ctx, op, B[r0:r1], m_global=m)
        assert s.set_parameters(params) == 0
        s.set_option("verbose", 0)
        s.set_option("projected_lanczos", 1 if projected == 1 else 0)
        s.set_option("subspace", 1 if projected == 2 else 0)  # 2: the coordinate-space back end
        code, V, T = s.solve()
        out = dict(code=code, V=V, T=T, hist=s.history(), trips=s.trips(), rel=s.relative_residual(), stats=ctx.stats())
        s.close()
        ctx.close()
        return out

    res = ranks.run(work)
    out = oracle.solve(A, B, oracle.params({**params, "rng_mode": 1, "seed": seed}))
    V = np.vstack([res[r]["V"] for r in range(nranks)])
    T = res[0]["T"]
    for r in range(nranks):
        assert res[r]["code"] == 0 == out["ret"]
        assert res[r]["trips"] == res[0]["trips"]
        assert np.array_equal(res[r]["T"], T)  # replicated
        assert np.array_equal(res[r]["hist"], res[0]["hist"])
        assert res[r]["stats"]["allreduce"] > 0
        if projected == 1:
            assert res[r]["stats"]["lanczos_start"] > 0
    assert abs(res[0]["trips"] - out["trips"]) <= 1
    h, ho = res[0]["hist"], out["res_hist"]
    n = min(6, len(h), len(ho))
    np.testing.assert_allclose(h[:n], ho[:n], rtol=1e-6)
    Xg, Xo = V @ T @ V.T, out["V"] @ out["T"] @ out["V"].T
    assert np.linalg.norm(Xg - Xo) / np.linalg.norm(Xo) <= 1e-2
    assert np.abs(V.T @ V - np.eye(V.shape[1])).max() < 1e-10
    assert res[0]["rel"] < 5e-3
    # single-rank GPU run of the same problem
    ctx = rails_amd.Context(device=0, seed=seed)
    s = rails_amd.Solver(ctx, rails_amd.HipOperatorWrapper(ctx, *A), B)
    assert s.set_parameters(params) == 0
    s.set_option("verbose", 0)
    s.set_option("projected_lanczos", 1 if projected == 1 else 0)
    s.set_option("subspace", 1 if projected == 2 else 0)
    code, V1, T1 = s.solve()
    assert code == 0 and abs(s.trips() - res[0]["trips"]) <= 1
    X1 = V1 @ T1 @ V1.T
    assert np.linalg.norm(Xg - X1) / np.linalg.norm(X1) <= 1e-2
    s.close()
    ctx.close()


@pytest.mark.parametrize("kind,nranks", [("banded", 2), ("banded", 3), ("stencil27", 2), ("stencil27", 3), ("laplace7", 2)])
def test_halo_overlap_is_bitwise_the_serial_product(oracle, monkeypatch, kind, nranks):
    """The row-partitioned product with the interior rows on the second stream while the ghost rows travel (rails_spmm; the importer
    inside Epetra_CrsMatrix::Apply, src/Epetra_OperatorWrapper.cpp:75-91, overlaps the same way) against the serial order pack ->
    exchange -> one product over all rows: bit for bit the same panel, at the in-loop widths and at panel width; on z-slabs of a grid
    stencil (BASELINE configs[3]'s partition) the interior planes go to the plane-sweep kernel."""
    from rails_amd import partition
    from rails_amd import problems as P
    from rails_amd.wrappers import HipMultiVectorWrapper as MV

    if kind == "banded":
        A = P.banded_random(9000, 27, 300, seed=2)
    elif kind == "stencil27":
        A = P.stencil27(20, 18, 12 * nranks, random_values=True, seed=4)
    else:
        A = P.laplace7(16, 16, 24)
    m = A[0].size - 1
    g = np.random.default_rng(7)
    X = g.uniform(-1, 1, (m, 128))
    starts = partition.row_ranges(m, nranks)
    if kind != "banded":  # whole planes per rank
        plane = (20 * 18) if kind == "stencil27" else 256
        assert all(int(s) % plane == 0 for s in starts)
    results = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("RAILS_SPMM_HALO_OVERLAP", mode)
        ranks = Ranks(nranks)

        def work(r):
            ctx, op, plan = _rank_setup(ranks, r, starts, A, seed=5)
            r0, r1 = int(starts[r]), int(starts[r + 1])
            if mode == "0":
                op.set_variant(1)  # the serial order on the row kernels (in automatic mode it may pick the box kernel, which sums in its own order)
            out = {}
            for nc in (128, 32, 16, 6):
                Y = op.apply(MV(ctx, data=X[r0:r1, :nc]))
                out["Y%d" % nc] = Y.to_host()
                out["k%d" % nc] = op.last_kernel()
            out["stats"] = ctx.stats()
            ctx.close()
            return out

        results[mode] = ranks.run(work)
    for nc in (128, 32, 16, 6):
        Yo = np.vstack([results["1"][r]["Y%d" % nc] for r in range(nranks)])
        Ys = np.vstack([results["0"][r]["Y%d" % nc] for r in range(nranks)])
        assert np.array_equal(Yo, Ys), (nc, np.abs(Yo - Ys).max())
        ref = oracle.csr_spmm(*A, X[:, :nc])
        assert np.abs(Yo - ref).max() <= 4e-14 * np.sqrt(27) * np.abs(ref).max()
    for r in range(nranks):
        assert results["1"][r]["stats"]["spmm_halo_overlapped"] == 4 and results["0"][r]["stats"]["spmm_halo_overlapped"] == 0
        assert "halo overlapped" in results["1"][r]["k128"]
        if kind != "banded":
            assert "k_spmm_planes" in results["1"][r]["k128"] and "k_spmm_planes" in results["1"][r]["k16"]


def test_interior_rows_of_a_banded_block_take_the_sweep_kernel(monkeypatch):
    """At panel width the interior rows of a row-partitioned banded operator (no ghost columns: a rectangular operator over the local X
    rows) run as a sweep of their own beside the halo exchange -- at N > 1 the A*X of the roofline leg no longer falls back to the
    row-gather kernel -- and the panel is bitwise the serial row-kernel product."""
    from rails_amd import partition
    from rails_amd import problems as P
    from rails_amd.wrappers import HipMultiVectorWrapper as MV

    nranks, m = 3, 390000  # (the middle rank has ghost rows on both sides)
    A = P.banded_random(m, 27, 1000, seed=3)
    X = np.random.default_rng(8).uniform(-1, 1, (m, 128))
    starts = partition.row_ranges(m, nranks)
    results = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("RAILS_SPMM_HALO_OVERLAP", mode)
        ranks = Ranks(nranks)

        def work(r):
            ctx, op, plan = _rank_setup(ranks, r, starts, A, seed=5)
            r0, r1 = int(starts[r]), int(starts[r + 1])
            if mode == "0":
                op.set_variant(1)
            else:
                op.prepare(128)  # set-up: the schedule of the interior rows
            Y = op.apply(MV(ctx, data=X[r0:r1]))
            out = dict(Y=Y.to_host(), kernel=op.last_kernel(), stats=ctx.stats())
            ctx.close()
            return out

        results[mode] = ranks.run(work)
    Yo = np.vstack([results["1"][r]["Y"] for r in range(nranks)])
    Ys = np.vstack([results["0"][r]["Y"] for r in range(nranks)])
    assert np.array_equal(Yo, Ys), np.abs(Yo - Ys).max()
    for r in range(nranks):
        assert "k_spmm_sweep" in results["1"][r]["kernel"] and results["1"][r]["stats"]["spmm_sweep"] == 1
    rows = np.random.default_rng(1).choice(m, 3000, replace=False)
    rp, col, val = A
    ref = np.zeros((rows.size, 128))
    for k, i in enumerate(rows):
        ref[k] = val[rp[i]:rp[i + 1]] @ X[col[rp[i]:rp[i + 1]]]
    assert np.abs(Yo[rows] - ref).max() <= 4e-14 * np.sqrt(27) * np.abs(ref).max()
