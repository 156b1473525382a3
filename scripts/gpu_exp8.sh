#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "spmm" > $O/t13.log 2>&1; echo exit=$? >> $O/t13.log; tail -3 $O/t13.log
rm -f $O/spmm9.jsonl
for pat in stencil27 laplace7; do
    timeout -k 10 200 python bench.py --spmm-only --pattern $pat --spmm-cols 128,64,16 --spmm-variant 2 >> $O/spmm9.jsonl 2>> $O/spmm9.err
done
cat $O/spmm9.jsonl
