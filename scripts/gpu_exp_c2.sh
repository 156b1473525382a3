#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out
for r in 1 0 1 0; do
RAILS_ORTH_REPAIR=$r timeout -k 10 300 python scripts/run_configs.py c1 c2 c2 > $O/c2_$r.jsonl 2> $O/c2_$r.err || { tail -5 $O/c2_$r.err; exit 1; }
python3 - $r <<'PY'
import json,sys
for l in open("gpurun_out/c2_%s.jsonl"%sys.argv[1]):
    d=json.loads(l)
    if "config" in d: print("repair",sys.argv[1], d["config"][:20], "trips",d["trips"],"sec %.4f"%d["seconds"], {k: round(v,4) for k,v in d["host_sections"].items()})
PY
done
