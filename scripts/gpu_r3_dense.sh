#!/bin/bash
# round 3: the dense kernels of a trip's device chain and the restart rotation (own one-pass wide GEMM against the sliced kernel)
mkdir -p gpurun_out
out=gpurun_out/dense_bench.txt
rm -f $out
echo "== default (one-pass wide GEMM)" >> $out
PYTHONPATH=. timeout -k 10 400 python scripts/dense_bench.py >> $out 2>gpurun_out/dense_bench_err.txt || { tail -20 gpurun_out/dense_bench_err.txt; exit 1; }
echo "== RAILS_PANEL_GEMM_WIDE=0 (128-column slices of k_panel_gemm)" >> $out
RAILS_PANEL_GEMM_WIDE=0 PYTHONPATH=. timeout -k 10 400 python scripts/dense_bench.py 2>>gpurun_out/dense_bench_err.txt | grep wide >> $out
echo "== rocBLAS (opt-in)" >> $out
RAILS_WIDE_GEMM=rocblas PYTHONPATH=. timeout -k 10 400 python - >> $out 2>>gpurun_out/dense_bench_err.txt <<'PY'
import json, numpy as np, rails_amd
from rails_amd.wrappers import HipMultiVectorWrapper as MV, _p
ctx = rails_amd.Context(device=0, seed=3)
ctx.enable_library_gemm()
m, k, r = 1000000, 324, 268
P1 = MV(ctx, m=m, n=k, capacity=776)
for j in range(0, k, 64):
    P1.view(j, min(k, j + 64) - 1).random()
P2 = MV(ctx, m=m, n=r, capacity=400)
Q = np.asfortranarray(np.linalg.qr(np.random.default_rng(1).standard_normal((k, r)))[0])
f = lambda: ctx.lib.rails_panel_gemm_wide(ctx.h, 1.0, P1.panel.h, 0, k, _p(Q), k, r, 0.0, P2.panel.h, 0)
f(); f(); ctx.sync()
ts = []
for _ in range(10):
    ctx.timer_start(); f(); ts.append(ctx.timer_stop())
ms = float(np.median(ts))
print(json.dumps({"case": "rocblas dgemm k=324 r=268", "ms": ms, "TFLOPs": 2.0 * m * k * r / ms / 1e9, "ready": ctx.lib.rails_ctx_library_gemm_ready(ctx.h)}))
PY
cat $out
