#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out
RAILS_RUN_SUBSPACE=1 timeout -k 10 600 python scripts/run_configs.py c1 c2 c3 c3s c4slab > $O/cfg_sub.jsonl 2> $O/cfg_sub.err || { tail -5 $O/cfg_sub.err; exit 1; }
timeout -k 10 600 python scripts/run_configs.py c1 c2 c3 c3s c4slab > $O/cfg_fused.jsonl 2> $O/cfg_fused.err || { tail -5 $O/cfg_fused.err; exit 1; }
python3 - <<'PY'
import json
for f in ("gpurun_out/cfg_fused.jsonl","gpurun_out/cfg_sub.jsonl"):
    for l in open(f):
        d=json.loads(l)
        if "config" in d: print(f[-12:], d["config"][:44], "trips",d["trips"],"sec %.4f"%d["seconds"],"k",d["k_final"],"rel %.2e"%d["relative_residual"], {k: round(v,4) for k,v in d["host_sections"].items() if k in("Apply A","Residual Lanczos","dense_solve","Orthogonalize","Restart")}, d.get("backend",{}).get("dim"))
PY
