#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "spmm" > $O/t9.log 2>&1; echo exit=$? >> $O/t9.log; tail -6 $O/t9.log
rm -f $O/spmm4.jsonl
for pat in stencil27 laplace7; do
 for box in 1 0; do
 for rows in 64 128 256; do
  for kc in 8 16; do
    echo "# pat=$pat box=$box rows=$rows kc=$kc" >> $O/spmm4.jsonl
    RAILS_SPMM_TILE_BOX=$box RAILS_SPMM_TILE_ROWS=$rows RAILS_SPMM_TILE_KC=$kc timeout -k 10 200 python bench.py --spmm-only --pattern $pat --spmm-cols 128,16 --spmm-variant 2 >> $O/spmm4.jsonl 2>> $O/spmm4.err
  done
 done
 done
done
cat $O/spmm4.jsonl
tail -3 $O/spmm4.err
