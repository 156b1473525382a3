#!/usr/bin/env python3
"""bench.py -- RAILS iterations/sec + A*V SpMM HBM GB/s on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one trip of the RAILS loop (src/LyapunovSolver.hpp:136): operator apply on the new columns,
incremental projection, host projected Lyapunov solve, fused residual Lanczos, convergence test,
expansion + block orthogonalisation or restart.  Workload (config.workload): BASELINE.json configs[2]
as defined in SURVEY.md section 8(d): m = 1M rows PER GPU (weak scaling), 27 nnz/row banded-random CSR
(|j-i| <= 4096), B m x 16, Restart size 200, Reduced size 128, Expand size 16, Lanczos iterations 20,
synthetic seeded data, fp64.  The tolerance is set so that the timed window never hits convergence (the
cost of a trip does not depend on it); W warm-up trips run first, then EXACTLY K trips are timed between
barrier + synchronize on both sides, MAX over ranks.

One JSON line on rank 0 with `roofline` (the CSR x tall-skinny A*V kernel at k = 128 columns, timed live
with HIP events on the library's stream) and `cpu_baseline` (the CPU oracle, kind "port", on the same
full-size workload for a bounded number of trips timed one by one, rank 0 at N = 1 only); `config` also
carries the rate of the direct back end on the same workload (`direct_backend_it_s`, a short run of its own).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


# Rank 0 must print ONE JSON line on stdout.  Libraries underneath (RCCL prints a version banner at communicator creation) write
# to file descriptor 1 themselves, so fd 1 is pointed at stderr for the whole run and the result lines go to the saved descriptor.
_STDOUT_FD = None


def _protect_stdout():
    global _STDOUT_FD
    if _STDOUT_FD is None:
        sys.stdout.flush()
        _STDOUT_FD = os.dup(1)
        os.dup2(2, 1)


def emit(obj):
    data = (json.dumps(obj) + "\n").encode()
    if _STDOUT_FD is None:
        sys.stdout.write(data.decode())
        sys.stdout.flush()
    else:
        os.write(_STDOUT_FD, data)


def build_problem(args, rank, nranks):
    from rails_amd import problems as P

    ml = args.m
    mg = ml * nranks
    r0 = rank * ml
    if args.pattern == "banded":
        rowptr, colg, val = P.banded_random_block(mg, r0, r0 + ml, 27, args.bandwidth, seed=args.seed)
        desc = "banded-random |j-i|<=%d" % args.bandwidth
    elif args.pattern == "stencil27":
        n = round(ml ** (1.0 / 3.0))
        assert n * n * n == ml, "stencil27 needs a cubic number of rows per GPU"
        rowptr, colg, val = P.stencil27_block(n, n, n * nranks, rank * n, (rank + 1) * n, random_values=True, seed=args.seed)
        desc = "27-pt stencil %dx%dx%d random coefficients" % (n, n, n * nranks)
    elif args.pattern == "uniform":
        assert nranks == 1, "uniform-random columns are a single-GPU report-only variant"
        rowptr, col, val = P.uniform_random(ml, 27, seed=args.seed)
        colg = col.astype(np.int64)
        desc = "uniform-random columns"
    elif args.pattern == "laplace7":
        assert nranks == 1
        rowptr, col, val = P.laplace7(50, 50, ml // 2500)
        colg = col.astype(np.int64)
        desc = "7-pt Laplacian 50x50x%d" % (ml // 2500)
    else:
        raise SystemExit("unknown pattern " + args.pattern)
    B = P.rhs(ml, args.p, seed=args.seed + 7 + rank)
    return (rowptr, colg, val), B, mg, r0, desc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--warmup", type=int, default=14)
    ap.add_argument("--m", "--rows-per-gpu", dest="m", type=int, default=1000000, help="rows per GPU")
    ap.add_argument("--p", type=int, default=16, help="columns of B")
    ap.add_argument("--pattern", default="banded", choices=["banded", "stencil27", "uniform", "laplace7"])
    ap.add_argument("--bandwidth", type=int, default=4096)
    ap.add_argument("--restart", type=int, default=200)
    ap.add_argument("--reduced", type=int, default=128)
    ap.add_argument("--expand", type=int, default=16)
    ap.add_argument("--lanczos", type=int, default=20)
    ap.add_argument("--kspmm", type=int, default=128, help="columns of the A*V roofline measurement")
    ap.add_argument("--spmm-reps", type=int, default=20)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-trips", type=int, default=3, help="cpu_baseline: trips of the full-size CPU run that are timed")
    ap.add_argument("--cpu-warmup", type=int, default=5, help="cpu_baseline: trips of the full-size CPU run before the timed ones (at most --warmup)")
    ap.add_argument("--busy-steps", type=int, default=8, help="timed trips of the extra run with the GPU-busy meter on (0: skip it)")
    ap.add_argument("--cpu-one-thread", type=int, default=1, help="cpu_baseline: also time one full-size trip on a single thread (0: skip)")
    ap.add_argument("--direct-steps", type=int, default=8, help="timed trips of the extra run on the direct back end (0: skip it)")
    ap.add_argument("--spmm-variant", type=int, default=0)
    ap.add_argument("--no-kernel-legs", action="store_true", help="skip the per-kernel timings of roofline_kernels (only the two A*X legs remain)")
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="torch.distributed backend; gloo = rehearsal of the multi-process path "
                    "(collectives staged through host memory, all ranks may share one GPU with --one-device)")
    ap.add_argument("--one-device", action="store_true", help="every rank uses cuda:0 (rehearsal on a one-GPU box, at most 6 ranks)")
    ap.add_argument("--hooks", action="store_true", help="multi-GPU: exchange through the torch.distributed hooks instead of the library's own RCCL communicator")
    ap.add_argument("--force-hooks", action="store_true", help="single rank: still route every reduction through torch.distributed (RCCL, world size 1)")
    ap.add_argument("--projected-lanczos", type=int, default=0, help="1: coefficient-space residual Lanczos (rails/HipSolverOps.hpp)")
    ap.add_argument("--subspace", type=int, default=1, help="1 (default): coordinate-space back end (rails/SubspaceWrappers.hpp); 0: direct panels")
    ap.add_argument("--spmm-only", action="store_true", help="kernel experiment: only the A*X timing, for several column counts")
    ap.add_argument("--spmm-cols", default="128", help="comma list of column counts for --spmm-only")
    ap.add_argument("--spmm-variants", default="", help="--spmm-only: comma list of operator variants to time (default: 1, 2, 7, 9, or --spmm-variant)")
    ap.add_argument("--spmm-pad", type=int, default=0, help="--spmm-only: extra panel capacity (columns), i.e. a row stride that is not a power of two")
    args = ap.parse_args()
    _protect_stdout()

    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.one_device else int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log("bench.py: --gpus %d but WORLD_SIZE=%d; using WORLD_SIZE" % (args.gpus, world))
    nranks = world
    torch.cuda.set_device(local_rank)
    dist = None
    if nranks > 1 or args.force_hooks:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if nranks == 1:
            os.environ.setdefault("MASTER_PORT", "29571")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="gloo")

    import rails_amd
    from rails_amd import partition

    t_setup = time.time()
    (rowptr, colg, val), B, mg, r0, desc = build_problem(args, rank, nranks)
    ml = args.m
    # one non-default torch stream for everything: the library's kernels and torch.distributed's collectives (which
    # order themselves against the CURRENT torch stream) then serialise without host synchronisation
    tstream = torch.cuda.Stream(device=local_rank)
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream
    assert stream, "expected a non-default stream handle"
    ctx = rails_amd.Context(device=local_rank, stream=stream, seed=args.seed)
    ctx.set_partition(rank, nranks, r0, mg)
    # (the basis rotation of restarts runs on the library's own one-pass MFMA kernel, k_panel_gemm_wide: no vendor-library kernel on this path)
    collectives = "none (single GPU)"
    if nranks > 1:
        starts = np.arange(nranks + 1, dtype=np.int64) * ml
        plan = partition.HaloPlan(starts, rank, colg, partition.all_gather_object_fn())
        A = rails_amd.HipOperatorWrapper(ctx, rowptr, plan.col_local, val, ncols_ext=ml + plan.n_ghost)
        staged = args.backend != "nccl"
        native = args.backend == "nccl" and not args.hooks
        if native:
            # The library's own RCCL communicator: the unique id travels over the torch.distributed group that is up already.  Every
            # step that can fail on one rank alone is followed by an agreement (all-reduce MIN over torch.distributed) BEFORE the next
            # blocking collective, so that all ranks take the same branch: a rank that fell back on its own while the others sat in
            # broadcast_object_list or ncclCommInitRank would hang the job instead of falling back to the hooks.
            def agreed(ok):
                t = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cuda")
                dist.all_reduce(t, op=dist.ReduceOp.MIN)
                return int(t.item()) == 1

            uid, why = None, ""
            if rank == 0:
                try:
                    uid = rails_amd.Context.rccl_unique_id()
                except Exception as e:  # librccl missing, ncclGetUniqueId failed
                    why = str(e)
            if agreed(rank != 0 or uid is not None):
                box = [uid]
                dist.broadcast_object_list(box, src=0)
                ok = True
                try:
                    ctx.init_rccl(box[0], nranks, rank)  # blocking: every rank is here (agreed above)
                    A.set_halo(plan, None)
                except Exception as e:
                    ok, why = False, str(e)
                if not agreed(ok):
                    if ok:
                        ctx.set_rccl(None)
                    native = False
            else:
                native = False
            if native:
                collectives = "RCCL inside the library (ncclAllReduce of the projected blocks, ncclSend/ncclRecv of the ghost rows)"
            else:
                log("[rank %d] RCCL inside the library is not available on every rank%s: using the torch.distributed hooks" % (rank, (" (" + why + ")") if why else ""))
        if not native:
            A.set_halo(plan, partition.make_halo(plan, on_device=True, host_staged=staged))
            ctx.set_allreduce(partition.make_allreduce(on_device=True, host_staged=staged))
            collectives = "torch.distributed hooks (%s)" % args.backend
        halo_rows = plan.n_ghost
    else:
        A = rails_amd.HipOperatorWrapper(ctx, rowptr, colg.astype(np.int32), val)
        halo_rows = 0
        if args.force_hooks:
            ctx.set_allreduce(partition.make_allreduce(on_device=True, host_staged=args.backend != "nccl"))
    A.set_variant(args.spmm_variant)
    nnz_local = int(rowptr[-1])
    log("[rank %d] setup %.1fs: %s, m_local=%d nnz=%d ghosts=%d" % (rank, time.time() - t_setup, desc, ml, nnz_local, halo_rows))

    if args.spmm_only:
        for kk in [int(x) for x in args.spmm_cols.split(",")]:
            X = rails_amd.HipMultiVectorWrapper(ctx, m=ml, n=kk, capacity=kk + args.spmm_pad)
            Y = rails_amd.HipMultiVectorWrapper(ctx, m=ml, n=kk, capacity=kk + args.spmm_pad)
            X.random()
            variants = [int(v) for v in args.spmm_variants.split(",")] if args.spmm_variants else ((1, 2, 7, 9) if args.spmm_variant == 0 else (args.spmm_variant,))
            for variant in variants:
                A.set_variant(variant)
                A.prepare(kk)
                try:
                    for _ in range(3):
                        A.apply(X, Y)
                except rails_amd.RailsError as e:
                    log("variant %d k=%d: not applicable (%s)" % (variant, kk, str(e)[:80]))
                    continue
                ctx.sync()
                ctx.timer_start()
                for _ in range(args.spmm_reps):
                    A.apply(X, Y)
                ms = ctx.timer_stop() / args.spmm_reps
                ab = nnz_local * 12 + (ml + 1) * 4 + 2 * ml * kk * 8
                emit({"pattern": args.pattern, "kernel": A.last_kernel(), "variant": variant, "k": kk, "pad": args.spmm_pad, "ms": ms,
                      "alg_GBs": ab / ms / 1e6, "frac": ab / ms / 1e6 / HBM_PEAK_GBS})
            del X, Y
        return

    # ---- roofline leg: Y = A * X with k = 128 columns, HIP events on the library's stream -------------
    kk = args.kspmm
    X = rails_amd.HipMultiVectorWrapper(ctx, m=ml, n=kk, capacity=kk)
    Y = rails_amd.HipMultiVectorWrapper(ctx, m=ml, n=kk, capacity=kk)
    X.random()
    t_prep = time.time()
    A.prepare(kk)  # set-up for repeated products of this width (the sweep kernel's schedule on banded patterns; untimed, like the CSR upload)
    log("[rank %d] prepare(%d) %.2fs" % (rank, kk, time.time() - t_prep))
    for _ in range(3):
        A.apply(X, Y)
    ctx.sync()
    ctx.timer_start()
    for _ in range(args.spmm_reps):
        A.apply(X, Y)
    spmm_ms = ctx.timer_stop() / args.spmm_reps
    spmm_kernel = A.last_kernel()
    sweep_stats = A.sweep_stats(kk) if spmm_kernel.startswith("k_spmm_sweep") else None
    # algorithmic bytes per launch: nnz*(8+4) + (m+1)*4 + 2*m*k*8
    alg_bytes = nnz_local * 12 + (ml + 1) * 4 + 2 * ml * kk * 8
    achieved = alg_bytes / (spmm_ms * 1e-3) / 1e9
    del X, Y
    # the product the solver's loop does every trip: A * W at Expand size columns, W behind an odd-sized panel stride as in the solver
    ke = args.expand
    Xe = rails_amd.HipMultiVectorWrapper(ctx, m=ml, n=ke, capacity=ke + 1)
    Ye = rails_amd.HipMultiVectorWrapper(ctx, m=ml, n=ke, capacity=ke + 1)
    Xe.random()
    for _ in range(3):
        A.apply(Xe, Ye)
    ctx.sync()
    ctx.timer_start()
    for _ in range(args.spmm_reps):
        A.apply(Xe, Ye)
    inloop_ms = ctx.timer_stop() / args.spmm_reps
    inloop_bytes = nnz_local * 12 + (ml + 1) * 4 + 2 * ml * ke * 8
    inloop = {"kernel": A.last_kernel(), "columns": ke, "avg_ms": inloop_ms, "algorithmic_bytes": inloop_bytes,
              "frac": inloop_bytes / (inloop_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
    log("[rank %d] in-loop SpMM %s k=%d: %.3f ms (%.1f%% of %.0f GB/s)" % (rank, inloop["kernel"], ke, inloop_ms, 100 * inloop["frac"], HBM_PEAK_GBS))
    del Xe, Ye
    # ---- the other kernels a trip is made of, each timed live with HIP events on the library's stream at the shapes it has in the timed
    # solve below (m rows per GPU; basis of 352 columns = the middle of a restart cycle): `roofline_kernels` of the JSON line -----------
    roofline_kernels = [
        {"kernel": spmm_kernel, "role": "A*X at %d columns, the workload's matrix (the headline `roofline`)" % kk, "bound": "hbm", "algorithmic_bytes": alg_bytes,
         "avg_ms": spmm_ms, "frac": achieved / HBM_PEAK_GBS},
        {"kernel": inloop["kernel"], "role": "the in-loop A*W at Expand size %d" % ke, "bound": "hbm", "algorithmic_bytes": inloop_bytes, "avg_ms": inloop_ms,
         "frac": inloop["frac"]},
    ]
    if not args.no_kernel_legs and nranks == 1:
        import ctypes as _C

        from rails_amd._lib import check as _check
        from rails_amd.wrappers import _p, resid_lanczos

        lib = ctx.lib
        S8 = 8

        def timed(fn, reps=args.spmm_reps):
            # per-call HIP events on the library's stream, median: a call that makes the host stall once (a first-use set-up inside the
            # runtime: seen as one 78 ms sample of the 3.6 ms rotation) is not what the kernel takes
            fn()
            ctx.sync()
            samples = []
            for _ in range(reps):
                ctx.timer_start()
                fn()
                samples.append(ctx.timer_stop())
            return float(np.median(samples))

        def randomised(n, cap=None):
            v = rails_amd.HipMultiVectorWrapper(ctx, m=ml, n=n, capacity=cap or n)
            for j in range(0, n, 64):
                v.view(j, min(n, j + 64) - 1).random()
            return v

        g = np.random.default_rng(args.seed)
        kb = 352
        Pb = randomised(kb + 17, kb + 17 + 64)
        # the fused update + second projection of the block Gram-Schmidt (src/StlWrapper.cpp:314-344 as one pass): X -= P C1, C2 = P'X
        C1 = np.asfortranarray(g.uniform(-1, 1, (kb, 17)) * 1e-5)
        _check(lib.rails_deferred_reserve(ctx.h, 8, (kb + 64) * 32), "rails_deferred_reserve")
        ms = timed(lambda: _check(lib.rails_update_gram_deferred(ctx.h, -1.0, Pb.panel.h, 0, kb, _p(C1), kb, 17, Pb.panel.h, kb, 16, 0), "rails_update_gram_deferred"))
        by = (kb + 2 * 17) * ml * S8
        roofline_kernels.append({"kernel": "k_update_gram", "role": "block Gram-Schmidt: first update + second projection in one pass, k = %d, r = 17" % kb, "bound": "hbm",
                                 "algorithmic_bytes": by, "avg_ms": ms, "frac": by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS})
        # the first projection round [P | X]' X
        outg = np.zeros((kb + 17, 17), order="F")
        ms = timed(lambda: lib.rails_gram(ctx.h, Pb.panel.h, 0, kb + 17, Pb.panel.h, kb, 17, _p(outg), kb + 17))
        by = (kb + 17) * ml * S8
        roofline_kernels.append({"kernel": "k_gram_cols", "role": "block Gram-Schmidt: first projection round [P | X]' X, %d x 17 (incl. the copy of the result to the host)" % (kb + 17),
                                 "bound": "hbm", "algorithmic_bytes": by, "avg_ms": ms, "frac": by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS})
        # W = P Wc (materialise the expansion block)
        Wc = np.asfortranarray(g.uniform(-1, 1, (kb, 16)) * 1e-3)
        Wp = rails_amd.HipMultiVectorWrapper(ctx, m=ml, n=16, capacity=16)
        ms = timed(lambda: lib.rails_panel_gemm(ctx.h, 1.0, Pb.panel.h, 0, kb, _p(Wc), kb, 16, 0.0, Wp.panel.h, 0))
        by = (kb + 16) * ml * S8
        roofline_kernels.append({"kernel": "k_panel_gemm", "role": "W = P Wc (materialise the expansion block), k = %d, r = 16" % kb, "bound": "hbm", "algorithmic_bytes": by,
                                 "avg_ms": ms, "frac": by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS})
        # the restart rotation P2 = P Q: the one compute-bound product (fp64 MFMA, 78 TFLOP/s dense peak)
        kr, rr = 324, 268
        P2 = rails_amd.HipMultiVectorWrapper(ctx, m=ml, n=rr, capacity=rr)
        Q = np.asfortranarray(np.linalg.qr(g.standard_normal((kr, rr)))[0])
        ms = timed(lambda: _check(lib.rails_panel_gemm_wide(ctx.h, 1.0, Pb.panel.h, 0, kr, _p(Q), kr, rr, 0.0, P2.panel.h, 0), "rails_panel_gemm_wide"), reps=max(5, args.spmm_reps // 2))
        fl = 2.0 * ml * kr * rr
        roofline_kernels.append({"kernel": "k_panel_gemm_wide", "role": "restart rotation P2 = P Q, k = %d, r = %d" % (kr, rr), "bound": "mfma", "flops": fl, "avg_ms": ms,
                                 "achieved_TFLOPs": fl / (ms * 1e-3) / 1e12, "peak_TFLOPs": 78.0, "frac": fl / (ms * 1e-3) / 1e12 / 78.0})
        del P2, Wp
        # the fused residual Lanczos pass of the direct back end (src/LyapunovSolver.hpp:384-434 as one pass per step)
        kl, L = 200, args.lanczos
        Pb.resize(2 * kl)
        Vv, AVv = Pb.view(0, kl - 1), Pb.view(kl, 2 * kl - 1)
        Bv = randomised(args.p)
        Tm = g.uniform(-1, 1, (kl, kl)) * 1e-3
        Tm = np.asfortranarray(Tm + Tm.T)
        # (per pass: the difference of a run of L steps and a run of 2, so that what a call costs once -- uploads, the start pass, the
        # eigenvalues of the tridiagonal matrix on the host -- stays out)
        t_long = timed(lambda: resid_lanczos(ctx, AVv, Vv, Tm, Bv, L), reps=max(3, args.spmm_reps // 4))
        t_short = timed(lambda: resid_lanczos(ctx, AVv, Vv, Tm, Bv, 2), reps=max(3, args.spmm_reps // 4))
        ms = (t_long - t_short) / (L - 2)
        by = (2 * kl + args.p + 4) * ml * S8
        roofline_kernels.append({"kernel": "k_lanczos_pass", "role": "direct back end: one step of the residual Lanczos recurrence, k = %d, p = %d ((run of %d steps - run of 2) / %d: the pass with the two small kernels behind it)" % (kl, args.p, L, L - 2),
                                 "bound": "hbm", "algorithmic_bytes": by, "avg_ms": ms, "frac": by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS})
        del Pb, Bv, Vv, AVv
        # the plane-sweep kernel on the pattern of configs[1] / configs[3]: 27-point stencil with the same number of rows
        n3 = round(ml ** (1.0 / 3.0))
        if n3 * n3 * n3 == ml:
            from rails_amd import problems as P

            As = P.stencil27(n3, n3, n3, random_values=True, seed=args.seed)
            ops = rails_amd.HipOperatorWrapper(ctx, *As)
            for kc in (128, 32):
                Xs = randomised(kc)
                Ys = rails_amd.HipMultiVectorWrapper(ctx, m=ml, n=kc, capacity=kc)
                ms = timed(lambda: ops.apply(Xs, Ys))
                by = int(As[0][-1]) * 12 + (ml + 1) * 4 + 2 * ml * kc * S8
                roofline_kernels.append({"kernel": ops.last_kernel(), "role": "A*X at %d columns, 27-point stencil %d^3 (configs[3]'s pattern%s)" % (kc, n3, "" if kc == 128 else ": its Expand size"),
                                         "bound": "hbm", "algorithmic_bytes": by, "avg_ms": ms, "frac": by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS})
                del Xs, Ys
            del ops, As
        for rk in roofline_kernels:
            log("[rank %d] kernel %-20s %8.3f ms  frac %.3f  (%s)" % (rank, rk["kernel"], rk["avg_ms"], rk["frac"], rk["role"]))
    traffic = None
    traffic_stale = None
    tfile = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tfile):
        try:
            tj = json.load(open(tfile))
            traffic = tj.get("%s:%s" % (spmm_kernel, args.pattern))
            # the PMC passes behind that number ran the kernel at this average duration: a different one now means the kernel has changed
            # since (the committed number is then about another build)
            then_us = tj.get("%s:%s:kernel_us" % (spmm_kernel, args.pattern))
            if traffic is not None and then_us:
                traffic_stale = bool(abs(spmm_ms * 1e3 / then_us - 1.0) > 0.15)
        except Exception:
            traffic = None
    log("[rank %d] SpMM %s k=%d: %.3f ms, %.1f GB/s algorithmic (%.1f%% of %.0f GB/s)" % (rank, spmm_kernel, kk, spmm_ms, achieved,
                                                                                          100 * achieved / HBM_PEAK_GBS, HBM_PEAK_GBS))

    # ---- timed solve ---------------------------------------------------------------------------------
    W, K = args.warmup, args.steps
    params = {"Maximum iterations": 100000, "Tolerance": 1e-30, "Expand size": args.expand, "Lanczos iterations": args.lanczos,
              "Restart size": args.restart, "Reduced size": args.reduced, "Minimize solution space": 0}
    solver = rails_amd.Solver(ctx, A, B, m_global=mg)
    code = solver.set_parameters(params)
    assert code == 0
    solver.set_option("verbose", 1 if args.verbose else 0)
    solver.set_option("max_trips", W + K)
    solver.set_option("projected_lanczos", args.projected_lanczos)
    solver.set_option("subspace", args.subspace)
    marks = {}

    allocs = []  # device / pinned allocations made by the library so far, per trip
    stamps = []  # host time at the end of every trip (diagnostics: one slow trip shows up here)

    def on_trip(trip):
        if trip == W or trip == W + K:
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()
            marks[trip] = time.perf_counter()
        stamps.append(time.perf_counter())
        allocs.append(ctx.stats().get("device_allocations", 0))

    solver.set_trip_callback(on_trip)
    import gc

    gc.collect()
    gc.disable()  # the trip callback is the only Python in the loop: no collector pause inside the timed region
    try:
        code, _, _ = solver.solve(fetch=False)
    finally:
        gc.enable()
    assert solver.trips() == W + K, "solver stopped after %d trips (code %d)" % (solver.trips(), code)
    elapsed = marks[W + K] - marks[W]
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    its = K / elapsed
    import ctypes as _C

    _sm, _bs = _C.c_long(0), _C.c_long(0)
    rails_amd.load().rails_sb03md_counts(_C.byref(_sm), _C.byref(_bs))
    sb_counts = (int(_sm.value), int(_bs.value))
    _ext, _fresh = _C.c_long(0), _C.c_long(0)
    rails_amd.load().rails_sb03md_adi_counts(_C.byref(_ext), _C.byref(_fresh))
    hist = solver.history()
    log("[rank %d] host sections (s, whole solve incl. warm-up): %s" % (rank, json.dumps(solver.profile())))
    median_trip_ms = None
    if len(stamps) > W + 2:
        dts = np.diff(np.array(stamps))[W:]
        median_trip_ms = 1e3 * float(np.median(dts))
        log("[rank %d] timed trips: median %.2f ms, slowest %.2f ms (trip %d); library allocations inside the timed region: %d (at trips %s)" % (
            rank, 1e3 * float(np.median(dts)), 1e3 * float(dts.max()), W + 1 + int(dts.argmax()), allocs[-1] - allocs[W - 1],
            [i + 1 for i in range(W, len(allocs)) if allocs[i] != allocs[i - 1]]))
    log("[rank %d] counters: %s %s" % (rank, json.dumps(ctx.stats()), json.dumps(solver.backend_stats())))
    # steady-state rate: whole restart cycles, whatever --warmup / --steps cut out of them.  A restart trip stands out in the stamps
    # (shrink + re-basing: 2-3x a regular trip); the rate is taken from the first to the last restart trip of the run (warm-up included,
    # the first three trips -- library set-up -- excluded).
    steady_it_s, restart_trips = None, []
    if len(stamps) > 12:
        dall = np.diff(np.array(stamps))
        med = float(np.median(dall[3:]))
        restart_trips = [i for i in range(3, dall.size) if dall[i] > 1.6 * med]
        if len(restart_trips) >= 2:
            a, b = restart_trips[0], restart_trips[-1]
            steady_it_s = (b - a) / float(stamps[b + 1] - stamps[a + 1])
            log("[rank %d] steady state: restart trips %s -> %d trips in %.4f s = %.1f it/s over %d whole restart cycles" % (
                rank, [i + 1 for i in restart_trips], b - a, stamps[b + 1] - stamps[a + 1], steady_it_s, len(restart_trips) - 1))
    # where a trip's wall time goes on the host's side: sections in which the host computes (projected solve, residual Lanczos on
    # coordinates, restart algebra) against sections in which it mostly waits for the device (operator apply incl. the first projection)
    prof = solver.profile()
    ntr = max(1, solver.trips())
    host_keys = [k for k in prof if k in ("dense_solve", "Residual Lanczos", "Restart", "Compute VAV", "Apply B", "Expand", "Orthogonalize")]
    wait_keys = [k for k in prof if k in ("Apply A", "Apply M")]
    host_ms = 1e3 * sum(prof[k] for k in host_keys) / ntr
    device_critical_ms = 1e3 * sum(prof[k] for k in wait_keys) / ntr
    sections_ms = {k: round(1e3 * v / ntr, 4) for k, v in prof.items()}
    log("[rank %d] %d trips in %.3fs -> %.2f it/s; Lanczos estimates %.3e -> %.3e; V.N()=%d" % (rank, K, elapsed, its, hist[0], hist[-1], solver.k))

    # ---- GPU-busy fraction: a short run of its own on the same back end with the library's busy meter on (a pair of events around every
    # launch, a few microseconds each: not in the timed region above) -----------------------------------------------------------------
    gpu_busy_frac = None
    if args.busy_steps > 0:
        Wb, Kb = min(W, 6), args.busy_steps
        marks_b, busy_b = {}, {}
        solver.set_option("max_trips", Wb + Kb)
        ctx.set_meter(True)

        def on_trip_busy(trip):
            if trip == Wb or trip == Wb + Kb:
                ctx.sync()
                marks_b[trip] = time.perf_counter()
                busy_b[trip] = ctx.stats().get("gpu_busy_ms", 0.0)

        solver.set_trip_callback(on_trip_busy)
        gc.disable()
        try:
            solver.solve(fetch=False)
        finally:
            gc.enable()
        ctx.set_meter(False)
        if Wb in marks_b and Wb + Kb in marks_b:
            gpu_busy_frac = (busy_b[Wb + Kb] - busy_b[Wb]) * 1e-3 / (marks_b[Wb + Kb] - marks_b[Wb])
            log("[rank %d] busy meter: %d trips in %.3fs, GPU at work %.1f ms -> %.3f" % (rank, Kb, marks_b[Wb + Kb] - marks_b[Wb], busy_b[Wb + Kb] - busy_b[Wb], gpu_busy_frac))

    # ---- the direct back end (device panels for V and AV: fused one-pass Lanczos kernel, block orthogonalisation on MFMA -- the north
    # star's literal path) on the same workload, a short timed run of its own: config.direct_backend_it_s ---------------------------
    direct_its = None
    if args.subspace and args.direct_steps > 0:
        Wd, Kd = min(W, 4), args.direct_steps
        marks_d = {}
        solver.set_option("subspace", 0)
        solver.set_option("max_trips", Wd + Kd)

        def on_trip_direct(trip):
            if trip == Wd or trip == Wd + Kd:
                torch.cuda.synchronize()
                if dist is not None:
                    dist.barrier()
                torch.cuda.synchronize()
                marks_d[trip] = time.perf_counter()

        solver.set_trip_callback(on_trip_direct)
        gc.disable()
        try:
            solver.solve(fetch=False)
        finally:
            gc.enable()
        if Wd in marks_d and Wd + Kd in marks_d:
            dt = marks_d[Wd + Kd] - marks_d[Wd]
            if dist is not None:
                t = torch.tensor([dt], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt = float(t.item())
            direct_its = Kd / dt
            log("[rank %d] direct back end: %d trips in %.3fs -> %.2f it/s; counters %s" % (rank, Kd, dt, direct_its, json.dumps(ctx.stats())))

    # ---- cpu_baseline: the oracle (CPU restatement of the Stl path, OpenMP) on the SAME full-size workload, rank 0, N = 1: a bounded
    # number of trips, each timed on its own (no extrapolation) -------------------------------------------------------------------
    cpu = None
    if rank == 0 and nranks == 1 and not args.no_cpu:
        from oracle.oracle import Oracle
        from rails_amd import problems as P

        orc = Oracle()
        if args.pattern == "banded":
            As = P.banded_random(ml, 27, args.bandwidth, seed=args.seed)
        elif args.pattern == "stencil27":
            n = round(ml ** (1.0 / 3.0))
            As = P.stencil27(n, n, n, random_values=True, seed=args.seed)
        elif args.pattern == "laplace7":
            As = P.laplace7(50, 50, max(1, ml // 2500))
        else:
            As = P.uniform_random(ml, 27, seed=args.seed)
        Bs = P.rhs(ml, args.p, seed=args.seed + 7)
        Wc, Kc = min(W, args.cpu_warmup), args.cpu_trips
        out = orc.solve(As, Bs, orc.params({**params, "rng_mode": 1, "seed": args.seed, "max_trips": Wc + Kc}), vcap=args.restart + args.expand)
        ts = out["trip_seconds"]
        if ts.size >= Wc + Kc and Wc >= 1:
            dt = float(ts[Wc + Kc - 1] - ts[Wc - 1])
            cpu = {"value": Kc / dt, "unit": "iterations/s", "cores": orc.num_threads(), "kind": "port", "extrapolated": False,
                   "sample": "oracle (CPU restatement of the Stl path, OpenMP, %d threads of %d host CPUs) on the full workload (%d rows): trips %d..%d "
                             "timed one by one after %d warm-up trips (%.2f s per trip; V has %d columns at the end)" % (
                                 orc.num_threads(), os.cpu_count() or 0, ml, Wc + 1, Wc + Kc, Wc, dt / Kc, out["V"].shape[1])}
            log("[cpu] full size, %d threads: trips %d..%d in %.2fs -> %.3f it/s" % (orc.num_threads(), Wc + 1, Wc + Kc, dt, cpu["value"]))
            if args.cpu_one_thread:
                # the same workload on ONE thread: two more trips, continued from the basis the run above ended with (the reference's warm start), the
                # second one timed: a trip of the size of those timed above without repeating the warm-up on one thread
                nthreads = orc.num_threads()
                orc.set_num_threads(1)
                try:
                    t1 = time.perf_counter()
                    out1 = orc.solve(As, Bs, orc.params({**params, "Restart from solution": 1, "rng_mode": 1, "seed": args.seed + 1, "max_trips": 2}),
                                     V0=np.ascontiguousarray(out["V"]), vcap=args.restart + args.expand)
                    t1 = time.perf_counter() - t1
                    ts1 = out1["trip_seconds"]
                    if ts1.size >= 2:
                        d1 = float(ts1[1] - ts1[0])  # (the first trip of a warm start applies A to the whole basis: not a regular trip)
                        cpu["one_thread"] = {"value": 1.0 / d1, "unit": "iterations/s", "cores": 1,
                                             "sample": "the second trip of a run continued from the %d-column basis of the run above (%.2f s; the whole "
                                                       "continuation took %.1f s)" % (out["V"].shape[1], d1, t1)}
                        log("[cpu] full size, 1 thread: one trip in %.2fs -> %.3f it/s" % (d1, 1.0 / d1))
                finally:
                    orc.set_num_threads(nthreads)
        del out, As, Bs

    if rank == 0:
        line = {
            "metric": "RAILS iterations/sec (+ A*V SpMM HBM GB/s in roofline), m=1M rows/GPU, k=128, fp64",
            "value": its, "unit": "iterations/s", "n_gpus": nranks, "steps": K, "warmup": W, "ms_per_step": 1e3 * elapsed / K,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]: m=%d rows/GPU (global %d), 27 nnz/row %s CSR, B m x %d, Restart size %d, Reduced size %d, "
                                   "Expand size %d, Lanczos iterations %d" % (ml, mg, desc, args.p, args.restart, args.reduced, args.expand, args.lanczos),
                       "parallelism": "row-partition x%d, RCCL all-reduce of projected blocks" % nranks if nranks > 1 else "single GPU",
                       "collectives": collectives,
                       "spmm_columns": kk, "inloop_spmm": inloop, "residual_lanczos": ("the reference's recurrence on coordinate vectors (host)" if args.subspace else
                                            ("coefficient-space, Gram differences" if args.projected_lanczos else "fused one-pass-per-step kernel")),
                       "backend": "coordinates in a device-resident orthonormal basis" if args.subspace else "direct panels",
                       # one JOINT solve over n_gpus x m rows: under weak scaling the ideal is a constant iteration rate; the rate at which
                       # matrix rows are processed (rows x iterations / s) is the quantity that grows with the GPU count
                       "global_rows": int(mg), "row_iterations_per_s": its * mg,
                       # the same workload on the direct back end (fused Lanczos kernel + block orthogonalisation), a short run of its own
                       "direct_backend_it_s": direct_its,
                       # share of a trip during which the GPU is at work (the library's busy meter -- events around every launch -- over a short run
                       # of its own); the rest is host work with the GPU idle
                       "gpu_busy_frac": gpu_busy_frac,
                       "median_trip_ms": median_trip_ms,
                       # rate over whole restart cycles (first to last restart trip of the run, warm-up included): independent of --warmup / --steps
                       "steady_it_s": steady_it_s, "restart_trips": [i + 1 for i in restart_trips],
                       # per trip, averaged over the whole solve (warm-up included): host = sections in which the host computes (projected
                       # solve, Lanczos on coordinates, restart algebra); device_critical = sections in which it waits for the device (A * W:
                       # materialise, SpMM, first projection round); the rest of the device's work runs behind the host's
                       "host_ms": host_ms, "device_critical_ms": device_critical_ms, "sections_ms_per_trip": sections_ms,
                       "host_projected_solve_routes": {"smith_or_adi": sb_counts[0], "bartels_stewart": sb_counts[1],
                                                       # ADI calls that built on the call before (bordered matrix: shifts kept, inverses extended) / from scratch
                                                       "adi_extended": int(_ext.value), "adi_from_scratch": int(_fresh.value)}},
            "roofline": {"bound": "hbm", "kernel": spmm_kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_stale": traffic_stale, "algorithmic_bytes": alg_bytes, "avg_ms": spmm_ms,
                         "traffic_source": "profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of this kernel on this workload, corrected as profiles/README.md says)" if traffic else None,
                         "schedule": sweep_stats},
            "roofline_kernels": roofline_kernels,
            "cpu_baseline": cpu,
        }
        emit(line)
    solver.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
