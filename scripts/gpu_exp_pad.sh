#!/bin/bash
# row stride (panel capacity) x row-gather variant on the banded pattern: does a stride that is not a multiple of 1 KiB (4 L2
# channel granules) let the column-chunked kernel use all L2 channels?
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out
rm -f $O/spmm_pad.jsonl
for pat in banded stencil27; do
  for pad in 0 8 16 24; do
    for v in 3 4 5; do
      timeout -k 10 200 python bench.py --spmm-only --pattern $pat --spmm-cols 128 --spmm-variant $v --spmm-pad $pad >> $O/spmm_pad.jsonl 2>> $O/spmm_pad.err || exit 1
    done
  done
done
python3 - <<'PY'
import json
for l in open("gpurun_out/spmm_pad.jsonl"):
    d=json.loads(l); print(d["pattern"], "variant", d["variant"], "pad", d["pad"], "%.3f ms"%d["ms"], "%.3f"%d["frac"])
PY
