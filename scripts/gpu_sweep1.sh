#!/bin/bash
# First GPU run of the sweep SpMM kernel: parity tests, then timings of the kernel and its ablations.
# the experiment builds of the sweep kernel are not in the shipped library: rebuild with them first (round 3)
mkdir -p gpurun_out; make -s -C rails_amd/csrc EXPERIMENTS=1 -B -j16 > gpurun_out/build_experiments.log 2>&1 || exit 1
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_sweep.py -x -q > gpurun_out/sweep1_tests.log 2>&1
rc=$?
tail -5 gpurun_out/sweep1_tests.log
if [ $rc -ne 0 ]; then echo "tests failed rc=$rc"; exit $rc; fi
timeout -k 10 300 python bench.py --spmm-only --spmm-cols 128 > gpurun_out/sweep1_spmm.jsonl 2> gpurun_out/sweep1_spmm.err || exit 1
cat gpurun_out/sweep1_spmm.jsonl
for ab in 1 2 3; do
  RAILS_SWEEP_ABLATE=$ab timeout -k 10 300 python bench.py --spmm-only --spmm-cols 128 --spmm-variant 7 > gpurun_out/sweep1_ablate$ab.jsonl 2> gpurun_out/sweep1_ablate$ab.err || exit 1
  echo "ablate $ab: $(cat gpurun_out/sweep1_ablate$ab.jsonl)"
done
timeout -k 10 300 python bench.py --spmm-only --spmm-cols 64 --spmm-variant 7 > gpurun_out/sweep1_spmm64.jsonl 2>> gpurun_out/sweep1_spmm.err
cat gpurun_out/sweep1_spmm64.jsonl
