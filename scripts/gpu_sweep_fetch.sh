#!/bin/bash
# Where the sweep kernel's HBM reads come from: FETCH_SIZE of the full kernel, without the LDS-DMA (stream + records only) and
# without the trips (X rows + records only).  One --pmc pass each.
# the experiment builds of the sweep kernel are not in the shipped library: rebuild with them first (round 3)
mkdir -p gpurun_out; make -s -C rails_amd/csrc EXPERIMENTS=1 -B -j16 > gpurun_out/build_experiments.log 2>&1 || exit 1
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/sweep_fetch
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for ab in ${@:-0 1 2}; do
  export RAILS_SWEEP_ABLATE=$ab
  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/ab$ab -- python3 $R/bench.py --spmm-only --pattern banded --spmm-cols 128 --spmm-variant 7 --spmm-reps 3 > $O/ab$ab.json 2> $O/ab$ab.err || { tail -3 $O/ab$ab.err; exit 1; }
done
python3 - <<'PY'
import csv, glob, os, collections
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/sweep_fetch"
out=open(O+"/summary.txt","w")
for d in sorted(glob.glob(O+"/ab*")):
    if not os.path.isdir(d): continue
    v=[]
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "spmm_sweep" in r["Kernel_Name"] and r["Counter_Name"]=="FETCH_SIZE":
                v.append(float(r["Counter_Value"]))
    line="%s FETCH_SIZE KiB mean %.5g n=%d  -> x2 x1024 = %.3f GB" % (os.path.basename(d), sum(v)/max(len(v),1), len(v), 2*1024*sum(v)/max(len(v),1)/1e9)
    print(line); out.write(line+"\n")
PY
