#!/bin/bash
# full GPU validation + evidence for profiles/ (run through gpurun)
set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out
timeout -k 10 700 python -m pytest tests -q -m gpu > $O/gpu_tests.log 2>&1; echo exit=$? >> $O/gpu_tests.log; tail -4 $O/gpu_tests.log
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -4 $O/bench_default.err | cut -c1-300; cat $O/bench_default.json
bash scripts/gpu_profile.sh r01
