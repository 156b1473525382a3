#!/bin/bash
# PMC study of an SpMM kernel on the banded-random bench pattern (separate passes: at most 8 SQ / 4 TCC counters each).
# Default: the sweep kernel at 128 columns; COLS=16 VAR=0 PAD=1 TAG=narrow_pmc looks at the in-loop 16-column product.
set -o pipefail
R=$GRAFT_REPO_ROOT
COLS=${COLS:-128}; VAR=${VAR:-7}; PAD=${PAD:-0}; TAG=${TAG:-sweep_pmc}; PATTERN=${PATTERN:-banded}
export PMC_TAG=$TAG
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() { # name counters...
  name=$1; shift
  timeout -k 10 200 rocprofv3 --pmc "$@" --output-format csv -d $O/$name -- python3 $R/bench.py --spmm-only --pattern $PATTERN --spmm-cols $COLS --spmm-variant $VAR --spmm-pad $PAD --spmm-reps 3 > $O/$name.json 2> $O/$name.err || { tail -3 $O/$name.err; return 1; }
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA &&
run sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM &&
run sq3 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_INSTS_SENDMSG SQ_WAVE_CYCLES SQ_BUSY_CYCLES &&
run fetch FETCH_SIZE &&
run write WRITE_SIZE &&
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum &&
run tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr || exit 1
python3 - <<'PY'
import csv, glob, os, collections
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/"+os.environ.get("PMC_TAG","sweep_pmc")
out=open(O+"/summary.txt","w")
for d in sorted(glob.glob(O+"/*")):
    if not os.path.isdir(d): continue
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "spmm" in r["Kernel_Name"]:
                acc[r["Kernel_Name"].split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in acc.items():
        line="%s %s %s" % (os.path.basename(d), k, {c: "%.5g (n=%d)"%(sum(x)/len(x),len(x)) for c,x in v.items()})
        print(line); out.write(line+"\n")
PY
