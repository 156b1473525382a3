#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 400 python scripts/proj_diag.py > $O/sub_diag.log 2>&1; echo exit=$? >> $O/sub_diag.log; grep -v "   res" $O/sub_diag.log | cut -c1-600
grep -q "exit=0" $O/sub_diag.log || exit 1
timeout -k 10 300 python bench.py --no-cpu --subspace 1 > $O/bench_sub.json 2> $O/bench_sub.err; tail -5 $O/bench_sub.err | cut -c1-600; cut -c1-200 $O/bench_sub.json
