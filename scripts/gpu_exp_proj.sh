#!/bin/bash
# projected-space Lanczos: diagnostics, bench with it, full GPU test-suite
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 400 python scripts/proj_diag.py > $O/proj_diag.log 2>&1; echo exit=$? >> $O/proj_diag.log; grep -v "res" $O/proj_diag.log | cut -c1-300
grep -q "exit=0" $O/proj_diag.log || exit 1
timeout -k 10 300 python bench.py --no-cpu --projected-lanczos 1 > $O/bench_proj.json 2> $O/bench_proj.err && cut -c1-260 $O/bench_proj.json && tail -3 $O/bench_proj.err | cut -c1-400 &&
timeout -k 10 900 python -m pytest tests -m gpu -q -x --timeout 600 > $O/gpu_tests.log 2>&1; echo exit=$? >> $O/gpu_tests.log; tail -8 $O/gpu_tests.log
