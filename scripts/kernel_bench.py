#!/usr/bin/env python3
"""Per-kernel microbenchmark at the C3 sizes (m = 1M rows, k up to 200, p = w = 16): HIP-event time per call through the
C ABI, algorithmic bytes / flops (SURVEY.md 8(d) formulas), achieved GB/s and TFLOP/s.  One JSON line per case.
Run under `rocprofv3 --kernel-trace --stats` for pure kernel durations and under `--pmc SQ_VALU_MFMA_BUSY_CYCLES ...` for the
MFMA utilisation of the projection kernels (scripts/gpu_kernels.sh).

    python scripts/kernel_bench.py [--m 1000000] [--reps 10] [--pattern banded]"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=1000000)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--pattern", default="banded")
    args = ap.parse_args()
    import rails_amd
    from rails_amd import problems as P
    from rails_amd.wrappers import HipMultiVectorWrapper as MV, resid_lanczos

    m, S = args.m, 8
    ctx = rails_amd.Context(device=0, seed=3)
    lib = ctx.lib

    def timed(name, fn, bytes_alg, flops, note=""):
        fn()
        ctx.sync()
        samples = []
        for _ in range(args.reps):  # per-call HIP events, median: one-off host hiccups (library loading, allocator) stay out
            ctx.timer_start()
            fn()
            samples.append(ctx.timer_stop())
        ms = float(np.median(samples))
        print(json.dumps({"case": name, "ms": round(ms, 4), "alg_GB": round(bytes_alg / 1e9, 3), "GBs": round(bytes_alg / ms / 1e6, 1),
                          "frac_hbm_8TBs": round(bytes_alg / ms / 1e6 / 8000.0, 3), "GFLOP": round(flops / 1e9, 2),
                          "TFLOPs": round(flops / ms / 1e9, 2), "note": note}), flush=True)

    def panel(n, cap=None):
        v = MV(ctx, m=m, n=n, capacity=cap or n)
        for j in range(0, n, 64):
            v.view(j, min(n, j + 64) - 1).random()
        return v

    V = panel(200, 216)
    AV = panel(200, 216)
    B = panel(16)
    W = panel(16)
    X128 = panel(128)
    Y128 = panel(128)
    # ---- achievable HBM bandwidth: streaming copy of a 128-column panel ---------------------------------------
    timed("copy 1M x 128 (panel_copy)", lambda: lib.rails_panel_copy(ctx.h, X128.panel.h, 0, 128, Y128.panel.h, 0), 2 * m * 128 * S, 0,
          "read + write; the achievable streaming bandwidth on this box")
    timed("scale 1M x 128 (panel_scale)", lambda: lib.rails_panel_scale(ctx.h, Y128.panel.h, 0, 128, 1.0000001), 2 * m * 128 * S, 0)
    # ---- projections (a4): Gram ---------------------------------------------------------------------------------
    for a, b, what in ((16, 200, "W'AV (:173)"), (200, 16, "V'AW (:187) / block CGS2"), (16, 16, "BW'BW, CholQR Gram"), (128, 128, "V'V wide"),
                       (200, 200, "AV'AV")):
        Xp, Yp = (V if a > 16 else W), (AV if b > 16 else B)
        out = np.zeros((a, b), order="F")
        timed("gram %dx%d  %s" % (a, b, what),
              lambda: lib.rails_gram(ctx.h, Xp.panel.h, 0, a, Yp.panel.h, 0, b, rails_amd.wrappers._p(out), a),
              (a + b) * m * S, 2.0 * m * a * b, "includes the D2H copy of the result + stream sync")
    # ---- updates (a5): panel GEMM -------------------------------------------------------------------------------
    rng = np.random.default_rng(1)
    for k, r, beta, what in ((200, 16, 1.0, "W -= V C (block CGS2)"), (16, 16, 0.0, "W R^-1 (CholQR, in place)"), (200, 128, 0.0, "V X (restart, in place)"),
                             (20, 16, 0.0, "Q v (expansion vectors)")):
        Cm = np.asfortranarray(rng.uniform(-1, 1, (k, r)) * 1e-3)
        if k == 16:
            src, dst, inplace = W, W, True
        elif r == 128:
            src, dst, inplace = V, V, True
        else:
            src, dst, inplace = V, W, False
        if k == 20:
            src = V
        timed("panel_gemm k=%d r=%d beta=%g  %s" % (k, r, beta, what),
              lambda: lib.rails_panel_gemm(ctx.h, 1.0 if beta == 0 else -1.0, src.panel.h, 0, k, rails_amd.wrappers._p(Cm), k, r, beta, dst.panel.h, 0),
              (k + (2 if beta else 1) * r) * m * S if not inplace else (k + r) * m * S, 2.0 * m * k * r)
    # re-randomise what the in-place updates scaled down
    for j in range(0, 200, 50):
        V.view(j, j + 49).random()
    W.random()
    # ---- residual Lanczos (a6) ----------------------------------------------------------------------------------
    T = rng.uniform(-1, 1, (200, 200))
    T = np.asfortranarray(T + T.T)
    L = 20
    timed("resid_lanczos k=200 p=16 L=20 (fused, one pass per step)", lambda: resid_lanczos(ctx, AV, V, T, B, L), (L + 1) * (2 * 200 + 16 + 2) * m * S, 0,
          "algorithmic bytes of THIS formulation: (L+1) passes over [AV V B] + the Lanczos vector; the reference's 4 passes per step would be 2x")
    sums = np.zeros(2 * 200 + 16 + 1)
    timed("lanczos_start k=200 p=16 (projected form: one pass)",
          lambda: lib.rails_lanczos_start(ctx.h, AV.panel.h, 0, V.panel.h, 0, 200, B.panel.h, 0, 16, rails_amd.wrappers._p(sums)), (2 * 200 + 16 + 1) * m * S, 0)
    # ---- orthogonalisation (a7) ---------------------------------------------------------------------------------
    Vo = panel(184, 216)
    Vo.orthogonalize()
    def orth():
        Vo.resize(184)
        Vo.orthogonalized = 184
        Vo.resize(200)
        Vo.view(184, 199).random()
        return Vo.orthogonalize()
    timed("orthogonalize k_old=184 w=16 (block CGS2 + CholQR2, incl. the random refill)", orth, (4 * 184 + 6 * 16) * m * S, 2.0 * m * (4 * 184 * 16 + 4 * 16 * 16))
    # ---- SpMM (a2) ----------------------------------------------------------------------------------------------
    if args.pattern == "banded":
        A = P.banded_random(m, 27, 4096, seed=1)
    else:
        n = round(m ** (1 / 3))
        A = P.stencil27(n, n, m // (n * n), random_values=True, seed=1)
    op = rails_amd.HipOperatorWrapper(ctx, *A)
    nnz = A[1].size
    for c, Xs, Ys in ((16, W, B), (128, X128, Y128)):
        timed("spmm %s %d columns (%s)" % (args.pattern, c, "in-loop A*W" if c == 16 else "warm start / headline"), lambda: op.apply(Xs, Ys),
              nnz * 12 + (m + 1) * 4 + 2 * m * c * S, 2.0 * nnz * c)
        print(json.dumps({"spmm_kernel": op.last_kernel()}))
    ctx.close()


if __name__ == "__main__":
    main()
