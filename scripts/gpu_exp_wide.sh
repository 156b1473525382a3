#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -q -x --timeout 300 -k spmm > $O/wide_tests.log 2>&1; echo exit=$? >> $O/wide_tests.log; tail -5 $O/wide_tests.log
grep -q "exit=0" $O/wide_tests.log || exit 1
rm -f $O/spmm_wide.jsonl
for pat in stencil27 laplace7; do
  for w in 0 1; do
    echo "# $pat wide=$w" >> $O/spmm_wide.jsonl
    RAILS_SPMM_TILE_WIDE=$w timeout -k 10 200 python bench.py --spmm-only --pattern $pat --spmm-cols 128,64,32 --spmm-variant 2 >> $O/spmm_wide.jsonl 2>> $O/spmm_wide.err || exit 1
  done
done
cut -c1-200 $O/spmm_wide.jsonl
