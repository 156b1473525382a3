#!/bin/bash
# full GPU validation + evidence for profiles/ (run through gpurun): tests, smoke, default bench (coordinate-space back end) and the
# direct back end beside it, rocprofv3 stats + PMC traffic for the default (banded) and the stencil workloads, per-kernel
# microbenchmark with MFMA counters, all configurations to convergence on both back ends
set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out
timeout -k 10 900 python -m pytest tests -q -m gpu --timeout 600 > $O/gpu_tests.log 2>&1; echo exit=$? >> $O/gpu_tests.log; tail -4 $O/gpu_tests.log
grep -q "exit=0" $O/gpu_tests.log || exit 1
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 && tail -1 $O/smoke.log &&
timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err && tail -4 $O/bench_default.err | cut -c1-300 && cat $O/bench_default.json &&
timeout -k 10 400 python bench.py --no-cpu --subspace 0 > $O/bench_direct.json 2> $O/bench_direct.err && cat $O/bench_direct.json &&
timeout -k 10 400 python bench.py --no-cpu --pattern stencil27 > $O/bench_stencil27.json 2> $O/bench_stencil27.err && cat $O/bench_stencil27.json &&
timeout -k 10 400 python bench.py --no-cpu --force-hooks > $O/bench_hooks.json 2> $O/bench_hooks.err && cat $O/bench_hooks.json &&
bash scripts/gpu_profile.sh r02 &&
bash scripts/gpu_profile.sh r02d --subspace 0 &&
bash scripts/gpu_profile.sh r02s --pattern stencil27 &&
bash scripts/gpu_kernels.sh > $O/gpu_kernels.log 2>&1 && tail -5 $O/gpu_kernels.log &&
cd $R && RAILS_RUN_SUBSPACE=1 timeout -k 10 600 python scripts/run_configs.py > $O/configs_default.jsonl 2> $O/configs_default.err &&
timeout -k 10 600 python scripts/run_configs.py c1 c2 c3 c3s c4slab > $O/configs_direct.jsonl 2> $O/configs_direct.err &&
RAILS_RUN_PROJECTED=1 timeout -k 10 600 python scripts/run_configs.py c1 c2 c3 c3s c4slab > $O/configs_projected.jsonl 2> $O/configs_projected.err && echo ALL-DONE
