#!/bin/bash
# round 3: plane-sweep kernel as the automatic choice: the SpMM test files, then timings (64-column chunks at 128 columns; very narrow panels)
mkdir -p gpurun_out
out=gpurun_out/planes_auto.txt
rm -f $out
timeout -k 10 900 python -m pytest tests/test_gpu_planes.py tests/test_gpu_kernels.py tests/test_gpu_sweep.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -8 || exit 1
for sh in 0 3; do
  echo "== shape $sh" >> $out
  RAILS_PLANES_SHAPE=$sh timeout -k 10 300 python bench.py --spmm-only --pattern stencil27 --spmm-cols 256,128 --spmm-variants 9 >> $out 2>gpurun_out/planes_auto_err.txt || exit 1
  RAILS_PLANES_SHAPE=$sh timeout -k 10 300 python bench.py --spmm-only --pattern laplace7 --spmm-cols 128 --spmm-variants 9 >> $out 2>gpurun_out/planes_auto_err.txt || exit 1
done
echo "== narrow" >> $out
timeout -k 10 300 python bench.py --spmm-only --pattern stencil27 --spmm-cols 8,4,2 --spmm-variants 9,1 >> $out 2>gpurun_out/planes_auto_err.txt || exit 1
timeout -k 10 300 python bench.py --spmm-only --pattern stencil27 --spmm-cols 16,32 --spmm-pad 1 --spmm-variants 0 >> $out 2>gpurun_out/planes_auto_err.txt || exit 1
grep -E "==|ms" $out | sed 's/"pad": 0, //; s/"alg_GBs.*frac/frac/'
