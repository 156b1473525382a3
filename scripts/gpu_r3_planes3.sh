#!/bin/bash
# round 3: the plane-sweep stencil kernel with lane groups for narrow panels: parity, then timings against the box and narrow kernels
mkdir -p gpurun_out
out=gpurun_out/planes_narrow.txt
rm -f $out
for shape in 0 1 2; do
  RAILS_PLANES_SHAPE=$shape timeout -k 10 300 python -m pytest tests/test_gpu_planes.py -x -q 2>&1 | tail -8 || exit 1
done
for pat in stencil27 laplace7; do
  echo "== $pat" >> $out
  timeout -k 10 300 python bench.py --spmm-only --pattern $pat --spmm-cols 128,64,32,16 --spmm-variants 9,2,1 >> $out 2>gpurun_out/planes_narrow_err.txt || exit 1
done
grep -E "==|ms" $out | sed 's/"pattern": "[a-z0-9]*", //; s/"pad": 0, //; s/"alg_GBs.*frac/frac/'
