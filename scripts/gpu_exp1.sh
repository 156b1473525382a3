#!/bin/bash
# GPU experiment driver (run through gpurun): correctness first, then kernel experiments.
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout -k 10 400 python -m pytest tests/test_gpu_solver.py tests/test_gpu_kernels.py -q -m gpu -x > $O/t7.log 2>&1; echo exit=$? >> $O/t7.log; tail -4 $O/t7.log
rm -f $O/spmm2.jsonl
for pat in banded stencil27; do
 for xcd in 0 1; do
  for ch in 0 64 32 16; do
    echo "# pat=$pat xcd=$xcd chunk=$ch" >> $O/spmm2.jsonl
    RAILS_SPMM_XCD=$xcd RAILS_SPMM_CHUNK=$ch timeout -k 10 200 python bench.py --spmm-only --pattern $pat --spmm-cols 128 --spmm-variant 1 >> $O/spmm2.jsonl 2>> $O/spmm2.err
  done
 done
done
cat $O/spmm2.jsonl
for u in 1 2 4; do
  RAILS_LZ_UNROLL=$u timeout -k 10 300 python bench.py --no-cpu > $O/bench_u$u.json 2> $O/bench_u$u.err; grep -E "host sections|trips in" $O/bench_u$u.err
done
