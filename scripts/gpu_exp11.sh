#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out
timeout -k 10 600 python -m pytest tests -q -m gpu -x > $O/t16.log 2>&1; echo exit=$? >> $O/t16.log; tail -3 $O/t16.log
timeout -k 10 300 python bench.py --no-cpu > $O/bench_b.json 2> $O/bench_b.err; grep -E "host sections|trips in|counters|SpMM" $O/bench_b.err | cut -c1-330
timeout -k 10 300 python bench.py --no-cpu --force-hooks > $O/bench_h.json 2> $O/bench_h.err; grep -E "trips in|counters|Error|error" $O/bench_h.err | cut -c1-330
timeout -k 10 300 python bench.py --no-cpu --pattern stencil27 > $O/bench_s.json 2> $O/bench_s.err; grep -E "host sections|trips in|counters|SpMM|setup" $O/bench_s.err | cut -c1-330
timeout -k 10 900 python scripts/run_configs.py c1 c2 c3 c3s c4slab c5 > $O/configs.jsonl 2> $O/configs.err; cut -c1-600 $O/configs.jsonl; tail -3 $O/configs.err
