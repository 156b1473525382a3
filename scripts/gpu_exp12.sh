#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out
timeout -k 10 600 python -m pytest tests -q -m gpu -x > $O/t17.log 2>&1; echo exit=$? >> $O/t17.log; tail -3 $O/t17.log
timeout -k 10 900 python scripts/run_configs.py c1 c2 c3 c3s c4slab c5 > $O/configs.jsonl 2> $O/configs.err; python - <<'PY'
import json
for l in open('gpurun_out/configs.jsonl'):
    d=json.loads(l)
    if 'config' in d: print(d['config'], '| trips', d['trips'], '| s %.3f'%d['seconds'], '| it/s %.1f'%d['iterations_per_s'], '| k', d['k_final'], '| res %.2e'%d['relative_residual'], '| orth %.3f'%d['host_sections'].get('Orthogonalize',0), d['counters_cumulative'])
PY
tail -2 $O/configs.err
timeout -k 10 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err; grep -E "trips in|counters|cpu" $O/bench_default.err | cut -c1-300
bash scripts/gpu_profile.sh r01s --pattern stencil27 > $O/prof_r01s.log 2>&1; tail -2 $O/prof_r01s.log
bash scripts/gpu_profile.sh r01 > $O/prof_r01.log 2>&1; tail -2 $O/prof_r01.log
