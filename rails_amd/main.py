"""Command-line driver: solve A X M' + M X A' + B B' = 0 for X = V T V' with matrices from MatrixMarket files.

    python -m rails_amd.main [params.xml] [--dir DIR] [--A A.mtx] [--B B.mtx] [--M M.mtx] [--V V.mtx] [--T T.mtx]
    python -m torch.distributed.run --nproc-per-node N -m rails_amd.main ...      (one rank per GPU, rows partitioned)

File names, formats and the parameter file follow the reference's driver (src/main.cpp:57-68,111,123-126): `A.mtx`, `B.mtx`
and `M.mtx` in, `V.mtx` and `T.mtx` out, solver parameters from the "Lyapunov Solver" sublist of a Teuchos XML file.  A diagonal
mass matrix with zeros on its diagonal (a descriptor system) is handled as the reference's driver does (src/main.cpp:77-99): the
Lyapunov equation is solved on the Schur complement A22 - A21 A11^-1 A12 of the rows where M is nonzero (rails_amd/schur.py; one
rank), B is restricted to those rows and V has that many rows.  Otherwise M must be absent (identity) or symmetric positive
definite.
"""
import argparse
import os
import sys
import time

import numpy as np


def _log(rank, *a):
    if rank == 0:
        print(*a, flush=True)


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m rails_amd.main", description=__doc__.split("\n\n")[0])
    ap.add_argument("params", nargs="?", help="Teuchos XML (or JSON) parameter file; the 'Lyapunov Solver' sublist is used")
    ap.add_argument("--dir", default=".", help="directory of the input / output files")
    ap.add_argument("--A", default="A.mtx")
    ap.add_argument("--B", default="B.mtx")
    ap.add_argument("--M", default="M.mtx", help="mass matrix (used when the file exists unless --no-mass)")
    ap.add_argument("--no-mass", action="store_true", help="solve the standard equation (M = I) even if M.mtx exists")
    ap.add_argument("--V", default="V.mtx")
    ap.add_argument("--T", default="T.mtx")
    ap.add_argument("--warm-start", default=None, help="V.mtx of a previous solve (orthonormal columns): sets 'Restart from solution'")
    ap.add_argument("--set", action="append", default=[], metavar="NAME=VALUE", help="override one solver parameter, e.g. --set 'Tolerance=1e-6'")
    ap.add_argument("--direct", action="store_true", help="direct back end (device panels for V, AV) instead of the default coordinate-space back end")
    ap.add_argument("--projected-lanczos", action="store_true", help="direct back end with the coefficient-space residual Lanczos (M = I only)")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--quiet", action="store_true")
    args = ap.parse_args(argv)

    import torch

    import rails_amd
    from rails_amd import mmio, partition

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    def path(name):
        return name if os.path.isabs(name) else os.path.join(args.dir, name)

    params = mmio.read_parameters(args.params) if args.params else {}
    for kv in args.set:
        k, _, v = kv.partition("=")
        params[k.strip()] = float(v)

    _log(rank, "Loading matrices")
    t0 = time.time()
    m, n, rowptr, col, val = mmio.read_csr(path(args.A))
    if m != n:
        raise SystemExit("A must be square, got %d x %d" % (m, n))
    B = mmio.read_dense(path(args.B))
    if B.shape[0] != m:
        raise SystemExit("B has %d rows, A has %d" % (B.shape[0], m))
    Mcsr = None
    if not args.no_mass and os.path.exists(path(args.M)):
        mm, mn, mrp, mcol, mval = mmio.read_csr(path(args.M))
        if (mm, mn) != (m, m):
            raise SystemExit("M must be %d x %d" % (m, m))
        Mcsr = (mrp, mcol, mval)
    V0 = mmio.read_dense(path(args.warm_start)) if args.warm_start else None
    if V0 is not None:
        params["Restart from solution"] = 1
    _log(rank, "  A %d x %d, %d nonzeros; B %d x %d; %s; read in %.2f s" % (m, m, val.size, B.shape[0], B.shape[1],
                                                                          "M given" if Mcsr else "M = I", time.time() - t0))

    # one non-default torch stream for the library and the collectives (see bench.py)
    tstream = torch.cuda.Stream(device=local_rank)
    torch.cuda.set_stream(tstream)
    ctx = rails_amd.Context(device=local_rank, stream=tstream.cuda_stream, seed=args.seed)
    starts = partition.row_ranges(m, world)
    r0, r1 = int(starts[rank]), int(starts[rank + 1])
    ctx.set_partition(rank, world, r0, m)

    def operator(csr):
        rp, cg, vv = csr
        if world == 1:
            return rails_amd.HipOperatorWrapper(ctx, rp, cg.astype(np.int32), vv)
        p0, p1 = rp[r0], rp[r1]
        plan = partition.HaloPlan(starts, rank, cg[p0:p1], partition.all_gather_object_fn())
        op = rails_amd.HipOperatorWrapper(ctx, (rp[r0:r1 + 1] - p0).astype(np.int64), plan.col_local, vv[p0:p1], ncols_ext=plan.m_local + plan.n_ghost)
        op.set_halo(plan, partition.make_halo(plan, on_device=True))
        return op

    schur = None
    if Mcsr is not None:
        mrp, mcol, mval = Mcsr
        rows = np.repeat(np.arange(m), np.diff(mrp))
        diagonal = bool(np.all(rows == mcol))
        mdiag = np.zeros(m)
        mdiag[rows[rows == mcol]] = mval[rows == mcol]
        if diagonal and np.any(mdiag == 0.0):
            if world > 1:
                raise SystemExit("a singular mass matrix (Schur complement) runs on one rank, as in the reference (src/SchurOperator.cpp:226)")
            from rails_amd.schur import SchurOperator

            _log(rank, "Computing Schur complement")
            t0 = time.time()
            schur = SchurOperator(ctx, (rowptr, col, val), mdiag)
            _log(rank, "  %d algebraic rows eliminated, %d remain; A11 factorised in %.2f s" % (schur.m1, schur.m2, time.time() - t0))
            B = schur.restrict(B)
            if V0 is not None and V0.shape[0] == m:
                V0 = V0[schur.idx2]
            d2 = schur.mass22
            Mcsr = None if np.all(d2 == 1.0) else (np.arange(schur.m2 + 1, dtype=np.int64), np.arange(schur.m2, dtype=np.int64), d2)
            m = schur.m2
            r0, r1 = 0, m
            ctx.set_partition(0, 1, 0, m)
        elif not diagonal and np.any(mdiag == 0.0):
            raise SystemExit("M has zero diagonal entries but is not diagonal: not supported")

    _log(rank, "Creating solver")
    A = schur.op if schur is not None else operator((rowptr, col, val))
    Mop = operator(Mcsr) if Mcsr else None
    if world > 1:
        ctx.set_allreduce(partition.make_allreduce(on_device=True))
    solver = rails_amd.Solver(ctx, A, B[r0:r1], M=Mop, m_global=m)
    code = solver.set_parameters(params)
    if code != 0:
        raise SystemExit("set_parameters rejected the parameter set (code %d)" % code)
    solver.set_option("verbose", 0 if (args.quiet or rank != 0) else 1)
    if Mop is not None:
        solver.set_option("mass", 1)
    if args.direct or args.projected_lanczos:
        solver.set_option("subspace", 0)
    if args.projected_lanczos and Mop is None:
        solver.set_option("projected_lanczos", 1)

    _log(rank, "Performing solve")
    ctx.sync()
    t0 = time.perf_counter()
    code, V, T = solver.solve(V0=V0[r0:r1] if V0 is not None else None)
    ctx.sync()
    dt = time.perf_counter() - t0
    rel = solver.relative_residual()
    if schur is not None:
        _log(rank, "Amount of matrix-vector products after the solve: %d" % schur.applies)
    _log(rank, "solve returned %d after %d iterations in %.3f s: V is %d x %d, relative residual %.3e" % (code, solver.trips(), dt, m, solver.k, rel))
    if world > 1:
        parts = [None] * world
        dist.all_gather_object(parts, V)
        V = np.vstack(parts)
    if rank == 0:
        note = "rails_amd: X = V T V' solves A X M' + M X A' + B B' = 0; return code %d, relative residual %.3e" % (code, rel)
        mmio.write_array(path(args.V), V, comment=note)
        mmio.write_array(path(args.T), T, comment=note)
        print("wrote %s (%d x %d) and %s (%d x %d)" % (path(args.V), V.shape[0], V.shape[1], path(args.T), T.shape[0], T.shape[1]), flush=True)
    solver.close()
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0 if code == 0 else 3


if __name__ == "__main__":
    sys.exit(main())
