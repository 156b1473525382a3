#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_hooks.py -q -m gpu -x -k "spmm or hook" > $O/t8.log 2>&1; echo exit=$? >> $O/t8.log; tail -6 $O/t8.log
rm -f $O/spmm3.jsonl
for pat in stencil27 banded; do
 for rows in 32 64 128 200 400; do
  for kc in 16 8; do
    echo "# pat=$pat rows=$rows kc=$kc" >> $O/spmm3.jsonl
    RAILS_SPMM_TILE_ROWS=$rows RAILS_SPMM_TILE_KC=$kc timeout -k 10 200 python bench.py --spmm-only --pattern $pat --spmm-cols 128,16 --spmm-variant 2 >> $O/spmm3.jsonl 2>> $O/spmm3.err
  done
 done
done
cat $O/spmm3.jsonl
tail -5 $O/spmm3.err
