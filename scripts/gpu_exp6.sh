#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "spmm" > $O/t11.log 2>&1; echo exit=$? >> $O/t11.log; tail -4 $O/t11.log
rm -f $O/spmm7.jsonl
for pat in stencil27 laplace7; do
 for rows in 64 32; do
  for kc in 8 16; do
    echo "# pat=$pat rows=$rows kc=$kc" >> $O/spmm7.jsonl
    RAILS_SPMM_TILE_ROWS=$rows RAILS_SPMM_TILE_KC=$kc timeout -k 10 200 python bench.py --spmm-only --pattern $pat --spmm-cols 128,16 --spmm-variant 2 >> $O/spmm7.jsonl 2>> $O/spmm7.err
  done
 done
done
cat $O/spmm7.jsonl
