#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "spmm or lanczos" > $O/t14.log 2>&1; echo exit=$? >> $O/t14.log; tail -3 $O/t14.log
rm -f $O/spmm10.jsonl
for pat in stencil27 laplace7; do
    timeout -k 10 200 python bench.py --spmm-only --pattern $pat --spmm-cols 128,64,16 --spmm-variant 2 >> $O/spmm10.jsonl 2>> $O/spmm10.err
done
cat $O/spmm10.jsonl
for u in 1 2 4; do
  RAILS_LZ_UNROLL=$u timeout -k 10 300 python bench.py --no-cpu > $O/bench_v$u.json 2> $O/bench_v$u.err; grep -E "host sections|trips in" $O/bench_v$u.err | cut -c1-330
done
P=$R/gpurun_out/pmc2; mkdir -p $P
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $P/sq1 -- python3 $R/bench.py --spmm-only --pattern stencil27 --spmm-cols 128 --spmm-variant 2 --spmm-reps 3 > $P/sq1.json 2> $P/sq1.err
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $P/fetch -- python3 $R/bench.py --spmm-only --pattern stencil27 --spmm-cols 128 --spmm-variant 2 --spmm-reps 3 > $P/fetch.json 2> $P/fetch.err
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $P/write -- python3 $R/bench.py --spmm-only --pattern stencil27 --spmm-cols 128 --spmm-variant 2 --spmm-reps 3 > $P/write.json 2> $P/write.err
ls $P
