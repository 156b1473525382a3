#!/bin/bash
# all GPU tests + smoke + default bench (no profiles)
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 900 python -m pytest tests -q -m gpu --timeout 600 > $O/gpu_tests.log 2>&1; echo exit=$? >> $O/gpu_tests.log; tail -6 $O/gpu_tests.log
grep -q "exit=0" $O/gpu_tests.log || exit 1
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout -k 10 400 python bench.py --no-cpu > $O/bench_q.json 2> $O/bench_q.err; tail -3 $O/bench_q.err | cut -c1-700
timeout -k 10 400 python bench.py --no-cpu --pattern stencil27 > $O/bench_qs.json 2> $O/bench_qs.err; tail -1 $O/bench_qs.err | cut -c1-300
timeout -k 10 400 python bench.py --no-cpu --force-hooks > $O/bench_qh.json 2> $O/bench_qh.err; tail -1 $O/bench_qh.err | cut -c1-300
bash scripts/gpu_configs_both.sh
