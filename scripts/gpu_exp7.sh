#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "spmm" > $O/t12.log 2>&1; echo exit=$? >> $O/t12.log; tail -4 $O/t12.log
rm -f $O/spmm8.jsonl
for pat in stencil27 laplace7; do
 for mo in 1 0; do
  for xcd in 1 0; do
    echo "# pat=$pat morton=$mo xcd=$xcd" >> $O/spmm8.jsonl
    RAILS_SPMM_XCD=$xcd RAILS_SPMM_TILE_MORTON=$mo timeout -k 10 200 python bench.py --spmm-only --pattern $pat --spmm-cols 128,64,16 --spmm-variant 2 >> $O/spmm8.jsonl 2>> $O/spmm8.err
  done
 done
done
cat $O/spmm8.jsonl
