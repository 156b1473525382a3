#!/bin/bash
# per-kernel evidence: HIP-event microbenchmark, rocprofv3 kernel stats of the same command, MFMA counters of the projections
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/kern
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 python3 $R/scripts/kernel_bench.py > $O/kernel_bench.jsonl 2> $O/kernel_bench.err || { tail -5 $O/kernel_bench.err; exit 1; }
cat $O/kernel_bench.jsonl
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/scripts/kernel_bench.py --reps 5 > $O/stats.jsonl 2> $O/stats.err || { tail -5 $O/stats.err; exit 1; }
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -- python3 $R/scripts/kernel_bench.py --reps 2 > $O/pmc_mfma.jsonl 2> $O/pmc_mfma.err || { tail -5 $O/pmc_mfma.err; exit 1; }
python3 - <<'PY'
import csv, glob, os, collections
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/kern"
for f in glob.glob(O+"/stats/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:25]:
        print("%-60s calls %6s avg %10.1f us  %5s%%" % (r["Name"].split("(")[0][-60:], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O+"/pmc_mfma/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0][-50:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items():
    print(k, {c: "%.4g" % (sum(x)/len(x)) for c,x in v.items()}, "n=%d" % len(next(iter(v.values()))))
PY
