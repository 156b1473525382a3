#!/usr/bin/env python3
"""Round 3 microbenchmark of the dense kernels a trip's device chain consists of, at the C3 sizes (m = 1M rows):
  * the restart rotation P2 = P Q (rails_panel_gemm_wide, k = 324, r = 268 -- src/LyapunovSolver.hpp:265,290 behind StlWrapper.cpp:168-187)
  * the fused update + second projection of the block Gram-Schmidt (rails_update_gram_deferred, k = 352, r = 17, r2 = 16)
  * the first projection [P | X]' X (rails_gram, 369 x 17) and the materialisation W = P Wc (rails_panel_gemm, k = 352, r = 16)
HIP-event medians through the C ABI, one JSON line per case, results checked against numpy on a sample of rows.

    python scripts/dense_bench.py [--m 1000000] [--reps 10]"""
import argparse
import ctypes as C
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
    ap.add_argument("--k", type=int, default=352)
    ap.add_argument("--skip-rotation", action="store_true", help="only the block Gram-Schmidt passes")
    args = ap.parse_args()
    import rails_amd
    from rails_amd._lib import check
    from rails_amd.wrappers import HipMultiVectorWrapper as MV, _p

    m, S = args.m, 8
    ctx = rails_amd.Context(device=0, seed=3)
    lib = ctx.lib
    rng = np.random.default_rng(1)

    def timed(name, fn, bytes_alg, flops, note=""):
        fn()
        ctx.sync()
        samples = []
        for _ in range(args.reps):
            ctx.timer_start()
            fn()
            samples.append(ctx.timer_stop())
        ms = float(np.median(samples))
        # the same calls back to back inside ONE pair of events (no host round trip between them)
        ctx.timer_start()
        for _ in range(args.reps):
            fn()
        burst = ctx.timer_stop() / args.reps
        note = (note + "; " if note else "") + "back to back: %.4f ms per call" % burst
        print(json.dumps({"case": name, "ms": round(ms, 4), "alg_GB": round(bytes_alg / 1e9, 3), "GBs": round(bytes_alg / ms / 1e6, 1),
                          "frac_hbm_8TBs": round(bytes_alg / ms / 1e6 / 8000.0, 3), "TFLOPs": round(flops / ms / 1e9, 2),
                          "frac_mfma_78TF": round(flops / ms / 1e9 / 78.0, 3), "note": note}), flush=True)

    def panel(n, cap=None):
        v = MV(ctx, m=m, n=n, capacity=cap or n)
        for j in range(0, n, 64):
            v.view(j, min(n, j + 64) - 1).random()
        return v

    rows = np.unique(np.concatenate([np.arange(40), np.arange(m - 40, m), rng.integers(0, m, 400)]))
    # ---- the restart rotation ---------------------------------------------------------------------------------
    k, r = 324, 268
    P1 = panel(k, 776)
    P2 = MV(ctx, m=m, n=r, capacity=400)
    Q = np.asfortranarray(np.linalg.qr(rng.standard_normal((k, r)))[0])
    if not args.skip_rotation:
        timed("panel_gemm_wide k=%d r=%d (restart rotation P2 = P Q)" % (k, r),
              lambda: check(lib.rails_panel_gemm_wide(ctx.h, 1.0, P1.panel.h, 0, k, _p(Q), k, r, 0.0, P2.panel.h, 0), "wide"), (k + r) * m * S, 2.0 * m * k * r)
        got = P2.to_host()[rows]
        want = P1.to_host()[rows] @ Q
        print(json.dumps({"check": "rotation", "max_abs_err": float(np.abs(got - want).max()), "scale": float(np.abs(want).max())}), flush=True)
        assert np.abs(got - want).max() <= 1e-12 * k
        for kk, rr in ((200, 128), (430, 290), (520, 400)):
            Pk = panel(kk, 776) if kk > k else P1
            Pr = MV(ctx, m=m, n=rr, capacity=400)
            Qk = np.asfortranarray(np.linalg.qr(rng.standard_normal((kk, rr)))[0])
            timed("panel_gemm_wide k=%d r=%d" % (kk, rr),
                  lambda: check(lib.rails_panel_gemm_wide(ctx.h, 1.0, Pk.panel.h, 0, kk, _p(Qk), kk, rr, 0.0, Pr.panel.h, 0), "wide"), (kk + rr) * m * S, 2.0 * m * kk * rr)
            got = Pr.to_host()[rows]
            want = Pk.to_host()[rows, :kk] @ Qk
            assert np.abs(got - want).max() <= 1e-12 * kk, (kk, rr, np.abs(got - want).max())
            del Pr
            if kk > k:
                del Pk
    del P2
    # ---- the block Gram-Schmidt passes of a trip ------------------------------------------------------------------
    k = args.k
    Pb = P1  # basis in columns [0, k), the A*W block behind it in [k, k + 17)
    Pb.resize(k + 17)
    out = np.zeros((k + 17, 17), order="F")
    timed("gram [P | X]' X  %d x 17 (first projection round)" % (k + 17),
          lambda: lib.rails_gram(ctx.h, Pb.panel.h, 0, k + 17, Pb.panel.h, k, 17, _p(out), k + 17), (k + 17) * m * S, 2.0 * m * (k + 17) * 17)
    out16 = np.zeros((k + 17, 16), order="F")
    timed("gram [P | X]' X  %d x 16 (the same without the 17th column)" % (k + 17),
          lambda: lib.rails_gram(ctx.h, Pb.panel.h, 0, k + 17, Pb.panel.h, k, 16, _p(out16), k + 17), (k + 17) * m * S, 2.0 * m * (k + 17) * 16)
    Wc = np.asfortranarray(rng.uniform(-1, 1, (k, 16)) * 1e-3)
    Wp = MV(ctx, m=m, n=16, capacity=16)
    timed("panel_gemm k=%d r=16 (materialise W = P Wc)" % k,
          lambda: lib.rails_panel_gemm(ctx.h, 1.0, Pb.panel.h, 0, k, _p(Wc), k, 16, 0.0, Wp.panel.h, 0), (k + 16) * m * S, 2.0 * m * k * 16)
    C1 = np.asfortranarray(rng.uniform(-1, 1, (k, 17)) * 1e-4)
    check(lib.rails_deferred_reserve(ctx.h, 8, (k + 64) * 32), "reserve")
    Xh_before = Pb.to_host()[rows]
    timed("update_gram k=%d r=17 r2=16 (X -= P C1, C2 = P'X in one pass)" % k,
          lambda: check(lib.rails_update_gram_deferred(ctx.h, -1.0, Pb.panel.h, 0, k, _p(C1), k, 17, Pb.panel.h, k, 16, 0), "update_gram"),
          (k + 2 * 17) * m * S, 2.0 * m * k * (17 + 16))
    # (the timed loop applied the update 1 + reps times)
    Xh_after = Pb.to_host()[rows]
    want = Xh_before[:, k:k + 17] - (args.reps + 1) * Xh_before[:, :k] @ C1
    print(json.dumps({"check": "update", "max_abs_err": float(np.abs(Xh_after[:, k:k + 17] - want).max())}), flush=True)
    assert np.abs(Xh_after[:, k:k + 17] - want).max() <= 1e-11
    C2 = np.zeros(k * 16)
    ctx.sync()
    check(lib.rails_deferred_fetch(ctx.h, 0, k * 16, _p(C2)), "fetch")
    G = np.zeros((k, 16), order="F")
    lib.rails_gram(ctx.h, Pb.panel.h, 0, k, Pb.panel.h, k, 16, _p(G), k)
    print(json.dumps({"check": "gram part", "max_abs_err": float(np.abs(C2.reshape(16, k).T - G).max()), "scale": float(np.abs(G).max())}), flush=True)
    C2s = np.asfortranarray(rng.uniform(-1, 1, (k, 16)) * 1e-6)
    timed("panel_gemm k=%d r=16 beta=1 (second update X -= P C2)" % k,
          lambda: lib.rails_panel_gemm(ctx.h, -1.0, Pb.panel.h, 0, k, _p(C2s), k, 16, 1.0, Pb.panel.h, k), (k + 2 * 16) * m * S, 2.0 * m * k * 16)
    ctx.close()


if __name__ == "__main__":
    main()
