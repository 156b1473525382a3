#!/bin/bash
# chunked row-gather kernel: parity, then timings of the variants on the banded and stencil patterns
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -q -x --timeout 300 -k spmm > $O/cc_tests.log 2>&1; echo exit=$? >> $O/cc_tests.log; tail -5 $O/cc_tests.log
grep -q "exit=0" $O/cc_tests.log || exit 1
rm -f $O/spmm_cc.jsonl
for pat in banded stencil27 uniform; do
  for v in 3 1 4 5; do
    timeout -k 10 200 python bench.py --spmm-only --pattern $pat --spmm-cols 128 --spmm-variant $v >> $O/spmm_cc.jsonl 2>> $O/spmm_cc.err || exit 1
  done
done
cut -c1-200 $O/spmm_cc.jsonl
