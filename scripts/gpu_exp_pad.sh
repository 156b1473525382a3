#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out
rm -f $O/spmm_pad.jsonl
for pat in stencil27 laplace7 banded; do
  for pad in 0 8 16 24 48; do
    for v in 2 6 3; do
      if [ $pat = banded ] && [ $v != 3 ]; then continue; fi
      timeout -k 10 200 python bench.py --spmm-only --pattern $pat --spmm-cols 128 --spmm-variant $v --spmm-pad $pad >> $O/spmm_pad.jsonl 2>> $O/spmm_pad.err || exit 1
    done
  done
done
cut -c1-220 $O/spmm_pad.jsonl
