#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out
rm -f $O/spmm5.jsonl
for pat in stencil27; do
 for rows in 64 128 256; do
  for kc in 8 16; do
    echo "# pat=$pat box=1 rows=$rows kc=$kc" >> $O/spmm5.jsonl
    RAILS_SPMM_TILE_ROWS=$rows RAILS_SPMM_TILE_KC=$kc timeout -k 10 200 python bench.py --spmm-only --pattern $pat --spmm-cols 128,16 --spmm-variant 2 >> $O/spmm5.jsonl 2>> $O/spmm5.err
  done
 done
done
echo "# laplace7 rowgather" >> $O/spmm5.jsonl
timeout -k 10 200 python bench.py --spmm-only --pattern laplace7 --spmm-cols 128,16,64 --spmm-variant 1 >> $O/spmm5.jsonl 2>> $O/spmm5.err
cat $O/spmm5.jsonl
