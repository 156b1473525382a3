#!/bin/bash
# round 3: the plane-sweep stencil kernel: parity for every patch shape, then timings against the box kernel
mkdir -p gpurun_out
rm -f gpurun_out/planes_bench.txt
for shape in 0 1 2 3; do
  RAILS_PLANES_SHAPE=$shape timeout -k 10 300 python -m pytest tests/test_gpu_planes.py -x -q 2>&1 | tail -5 > gpurun_out/planes_tests_$shape.txt || { cat gpurun_out/planes_tests_$shape.txt; exit 1; }
  tail -2 gpurun_out/planes_tests_$shape.txt
done
for shape in 0 1 2 3; do
  echo "== shape $shape" >> gpurun_out/planes_bench.txt
  RAILS_PLANES_SHAPE=$shape timeout -k 10 200 python bench.py --spmm-only --pattern stencil27 --spmm-cols 128,64,32 --spmm-variants 9 >> gpurun_out/planes_bench.txt 2>gpurun_out/planes_bench_err_$shape.txt || exit 1
done
echo "== laplace7" >> gpurun_out/planes_bench.txt
timeout -k 10 200 python bench.py --spmm-only --pattern laplace7 --spmm-cols 128 --spmm-variants 9,2 >> gpurun_out/planes_bench.txt 2>gpurun_out/planes_bench_err_l7.txt
cat gpurun_out/planes_bench.txt
