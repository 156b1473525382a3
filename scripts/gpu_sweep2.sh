#!/bin/bash
# Sweep SpMM: parity tests, timing, and what each ingredient costs (experiment builds; their results are wrong by construction)
# the experiment builds of the sweep kernel are not in the shipped library: rebuild with them first (round 3)
mkdir -p gpurun_out; make -s -C rails_amd/csrc EXPERIMENTS=1 -B -j16 > gpurun_out/build_experiments.log 2>&1 || exit 1
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_sweep.py -x -q > gpurun_out/sweep2_tests.log 2>&1
rc=$?
tail -3 gpurun_out/sweep2_tests.log
if [ $rc -ne 0 ]; then echo "tests failed rc=$rc"; exit $rc; fi
: > gpurun_out/sweep2.jsonl
for ab in 0 1 4 48 96; do
  RAILS_SWEEP_ABLATE=$ab timeout -k 10 300 python bench.py --spmm-only --spmm-cols 128 --spmm-variant 7 > gpurun_out/sweep2_$ab.json 2> gpurun_out/sweep2_$ab.err || exit 1
  echo "ablate $ab: $(python -c "import json,sys; d=json.loads(open('gpurun_out/sweep2_$ab.json').readline()); print('%.3f ms' % d['ms'])")" | tee -a gpurun_out/sweep2.jsonl
done
RAILS_SWEEP_LAYOUT=1 timeout -k 10 300 python bench.py --spmm-only --spmm-cols 128 --spmm-variant 7 > gpurun_out/sweep2_layout1.json 2> gpurun_out/sweep2_layout1.err || exit 1
echo "layout 1: $(python -c "import json,sys; d=json.loads(open('gpurun_out/sweep2_layout1.json').readline()); print('%.3f ms' % d['ms'])")" | tee -a gpurun_out/sweep2.jsonl
