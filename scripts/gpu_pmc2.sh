#!/bin/bash
# PMC study of the row-gather kernel on the banded-random pattern: whole-width launch vs 32-column chunks
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc2
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() { # name counters...
  name=$1; shift
  timeout -k 10 200 rocprofv3 --pmc "$@" --output-format csv -d $O/$name -- python3 $R/bench.py --spmm-only --pattern banded --spmm-cols 128 --spmm-variant 1 --spmm-reps 3 > $O/$name.json 2> $O/$name.err || { tail -3 $O/$name.err; return 1; }
}
for ch in 0 32; do
  export RAILS_SPMM_CHUNK=$ch
  run c${ch}_sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM &&
  run c${ch}_fetch FETCH_SIZE &&
  run c${ch}_tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum &&
  run c${ch}_tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr || exit 1
done
python3 - <<'PY'
import csv, glob, os, collections
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/pmc2"
for d in sorted(glob.glob(O+"/c*_*")):
    if not os.path.isdir(d): continue
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "spmm" in r["Kernel_Name"]:
                acc[r["Kernel_Name"].split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in acc.items():
        print(os.path.basename(d), k, {c: "%.4g (n=%d)"%(sum(x)/len(x),len(x)) for c,x in v.items()})
PY
