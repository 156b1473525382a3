#!/bin/bash
# Sweep SpMM timing experiments, all on one box: every argument is a comma-separated list of environment settings for one run, e.g.
#   bash scripts/gpu_sweep3.sh "" RAILS_SWEEP_LEVEL=0 RAILS_SWEEP_ABLATE=4,RAILS_SWEEP_LEVEL=0
# (results of RAILS_SWEEP_ABLATE builds are wrong by construction: timings only)
# the experiment builds of the sweep kernel are not in the shipped library: rebuild with them first (round 3)
mkdir -p gpurun_out; make -s -C rails_amd/csrc EXPERIMENTS=1 -B -j16 > gpurun_out/build_experiments.log 2>&1 || exit 1
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
: > gpurun_out/sweep3.txt
i=0
for setting in "$@"; do
  i=$((i+1))
  ( IFS=,; for kv in $setting; do export "$kv"; done
    timeout -k 10 300 python bench.py --spmm-only --spmm-cols 128 --spmm-variant 7 > gpurun_out/sweep3_$i.json 2> gpurun_out/sweep3_$i.err ) || exit 1
  echo "[$setting]: $(python -c "import json,sys; d=json.loads(open('gpurun_out/sweep3_$i.json').readline()); print('%.3f ms' % d['ms'], d.get('schedule',''))")" | tee -a gpurun_out/sweep3.txt
done
