#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out
for t in 1 2 4 8 16; do
  RAILS_LAPACK_THREADS=$t timeout -k 10 300 python bench.py --no-cpu > $O/bench_t$t.json 2> $O/bench_t$t.err || exit 1
  echo "threads=$t"; grep -E "host sections|trips in" $O/bench_t$t.err | cut -c1-420
done
