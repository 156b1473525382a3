#!/bin/bash
# round 3: tuning runs of the plane-sweep stencil kernel (patch shapes, non-temporal stores / record loads, segment lengths)
mkdir -p gpurun_out
out=gpurun_out/planes_tune.txt
rm -f $out
for shape in 3 4; do
  RAILS_PLANES_SHAPE=$shape timeout -k 10 300 python -m pytest tests/test_gpu_planes.py -x -q 2>&1 | tail -3 || exit 1
done
run() { echo "== $*" >> $out; env "$@" timeout -k 10 200 python bench.py --spmm-only --pattern stencil27 --spmm-cols 128,32 --spmm-variants 9 >> $out 2>gpurun_out/planes_tune_err.txt || exit 1; }
run RAILS_PLANES_SHAPE=0
run RAILS_PLANES_SHAPE=0 RAILS_PLANES_FLAGS=1
run RAILS_PLANES_SHAPE=0 RAILS_PLANES_FLAGS=2
run RAILS_PLANES_SHAPE=0 RAILS_PLANES_FLAGS=3
run RAILS_PLANES_SHAPE=1 RAILS_PLANES_FLAGS=3
run RAILS_PLANES_SHAPE=3 RAILS_PLANES_FLAGS=0
run RAILS_PLANES_SHAPE=3 RAILS_PLANES_FLAGS=3
run RAILS_PLANES_SHAPE=4 RAILS_PLANES_FLAGS=3
run RAILS_PLANES_SHAPE=0 RAILS_PLANES_FLAGS=3 RAILS_PLANES_SEG=20
run RAILS_PLANES_SHAPE=0 RAILS_PLANES_FLAGS=3 RAILS_PLANES_SEG=25
run RAILS_PLANES_SHAPE=0 RAILS_PLANES_FLAGS=3 RAILS_PLANES_SEG=50
run RAILS_PLANES_SHAPE=0 RAILS_PLANES_FLAGS=3 RAILS_PLANES_SEG=100
run RAILS_PLANES_SHAPE=3 RAILS_PLANES_FLAGS=3 RAILS_PLANES_SEG=20
run RAILS_PLANES_SHAPE=3 RAILS_PLANES_FLAGS=3 RAILS_PLANES_SEG=50
grep -E "==|ms" $out | sed 's/"pattern": "stencil27", "kernel": "k_spmm_planes", "variant": 9, //; s/"pad": 0, //; s/"alg_GBs.*frac/frac/'
