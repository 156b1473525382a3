#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() { # name counters...
  name=$1; shift
  timeout -k 10 200 rocprofv3 --pmc "$@" --output-format csv -d $O/$name -- python3 $R/bench.py --spmm-only --pattern stencil27 --spmm-cols 128 --spmm-variant 2 --spmm-reps 3 > $O/$name.json 2> $O/$name.err
  tail -1 $O/$name.err
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run sq2 SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum
find $O -name "*counter_collection.csv" | head; du -sh $O
