#!/usr/bin/env python3
"""Microbenchmark of the three panel kernels of one `A * W` of the coordinate-space back end at the C3 in-loop shapes: the basis
P has `dim` columns, the block X of w = 17 columns (A*W + the prefetched Lanczos start vector) sits right behind it.

    gram      [P | X]' X      (dim + w) x w      reads (dim + w) columns
    update    X -= P C        k = dim, r = w     reads dim + w columns, writes w
    material. W  = P Wc       k = dim, r = 16    reads dim columns, writes 16 (other panel)

    python scripts/absorb_bench.py [--m 1000000] [--dims 128,232,340] [--reps 10]"""
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
    ap.add_argument("--dims", default="128,232,340")
    ap.add_argument("--w", type=int, default=17)
    ap.add_argument("--reps", type=int, default=10)
    args = ap.parse_args()
    import rails_amd
    from rails_amd.wrappers import HipMultiVectorWrapper as MV, _p

    m, w = args.m, args.w
    ctx = rails_amd.Context(device=0, seed=3)
    lib = ctx.lib
    dims = [int(x) for x in args.dims.split(",")]
    cap = max(dims) + w + 47
    Pn = MV(ctx, m=m, n=cap, capacity=cap)
    for j in range(0, cap, 64):
        Pn.view(j, min(cap, j + 64) - 1).random()
    Ws = MV(ctx, m=m, n=16, capacity=16)
    rng = np.random.default_rng(1)

    def timed(name, fn, nbytes, flops):
        fn()
        ctx.sync()
        samples = []
        for _ in range(args.reps):
            ctx.timer_start()
            fn()
            samples.append(ctx.timer_stop())
        ms = float(np.median(samples))
        print(json.dumps({"case": name, "ms": round(ms, 4), "GB": round(nbytes / 1e9, 3), "GBs": round(nbytes / ms / 1e6, 1), "frac_hbm_8TBs": round(nbytes / ms / 8e9, 3),
                          "TFLOPs_useful": round(flops / ms / 1e9, 2)}), flush=True)

    out = np.zeros((w, w), order="F")
    timed("CholQR Gram %d x %d" % (w, w), lambda: lib.rails_gram(ctx.h, Pn.panel.h, 0, w, Pn.panel.h, 0, w, _p(out), w), w * m * 8, 2.0 * m * w * w)
    Rm = np.asfortranarray(np.triu(rng.uniform(0.5, 1.0, (w, w))))
    timed("CholQR update k=%d r=%d in place" % (w, w), lambda: lib.rails_panel_gemm(ctx.h, 1.0, Pn.panel.h, 0, w, _p(Rm), w, w, 0.0, Pn.panel.h, 0), 2 * w * m * 8, 2.0 * m * w * w)
    for dim in dims:
        out = np.zeros((dim + w, w), order="F")
        timed("gram (%d+%d) x %d" % (dim, w, w), lambda: lib.rails_gram(ctx.h, Pn.panel.h, 0, dim + w, Pn.panel.h, dim, w, _p(out), dim + w),
              (dim + w) * m * 8, 2.0 * m * (dim + w) * w)
        Cm = np.asfortranarray(rng.uniform(-1, 1, (dim, w)) * 1e-6)
        timed("update k=%d r=%d" % (dim, w), lambda: lib.rails_panel_gemm(ctx.h, -1.0, Pn.panel.h, 0, dim, _p(Cm), dim, w, 1.0, Pn.panel.h, dim),
              (dim + 2 * w) * m * 8, 2.0 * m * dim * w)
        Cw = np.asfortranarray(rng.uniform(-1, 1, (dim, 16)))
        timed("materialise k=%d r=16" % dim, lambda: lib.rails_panel_gemm(ctx.h, 1.0, Pn.panel.h, 0, dim, _p(Cw), dim, 16, 0.0, Ws.panel.h, 0),
              (dim + 16) * m * 8, 2.0 * m * dim * 16)


if __name__ == "__main__":
    main()
