#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x --timeout 600 > $O/gpu_tests.log 2>&1; echo exit=$? >> $O/gpu_tests.log; tail -8 $O/gpu_tests.log
grep -q "exit=0" $O/gpu_tests.log || exit 1
timeout -k 10 300 python bench.py --no-cpu > $O/bench_orth.json 2> $O/bench_orth.err && cut -c1-250 $O/bench_orth.json && tail -3 $O/bench_orth.err | cut -c1-420 &&
bash scripts/gpu_exp_cfg.sh
