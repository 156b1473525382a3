#!/bin/bash
# PMC study of the block Gram-Schmidt kernels (k_gram_cols, k_update_gram, k_panel_gemm) at the C3 sizes
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/dense_pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$R
run() { # name counters...
  name=$1; shift
  timeout -k 10 200 rocprofv3 --pmc "$@" --output-format csv -d $O/$name -- python3 $R/scripts/dense_bench.py --skip-rotation --reps 3 > $O/$name.json 2> $O/$name.err || { tail -3 $O/$name.err; return 1; }
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS &&
run sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD &&
run fetch FETCH_SIZE &&
run write WRITE_SIZE GRBM_GUI_ACTIVE || exit 1
python3 - <<'PY'
import csv, glob, os, collections, re
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/dense_pmc"
for d in sorted(glob.glob(O+"/*")):
    if not os.path.isdir(d): continue
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n=r["Kernel_Name"]
            m=re.search(r"(k_gram_cols|k_update_gram|k_panel_gemm)<[^>]*>", n)
            if m:
                acc[m.group(0)][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in acc.items():
        print(os.path.basename(d), k, {c: "%.4g (n=%d)"%(sum(x)/len(x),len(x)) for c,x in v.items()})
PY
