#!/bin/bash
# Round-3 evidence of the final build for profiles/: un-profiled bench lines (default, the driver's arguments, 27-point pattern, configs[3]'s
# parameters, direct back end), rocprofv3 stats + PMC traffic (banded and stencil workloads), every single-GPU configuration to convergence.
set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out
mkdir -p $O
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 && tail -1 $O/smoke.log &&
timeout -k 10 400 python bench.py > $O/r03_bench_default.json 2> $O/bench_default.err && cut -c1-400 $O/r03_bench_default.json &&
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $O/r03_bench_driver_args.json 2> $O/bench_driver.err && cut -c1-200 $O/r03_bench_driver_args.json &&
timeout -k 10 400 python bench.py --no-cpu --pattern stencil27 > $O/r03_bench_stencil27.json 2> $O/bench_stencil27.err && cut -c1-200 $O/r03_bench_stencil27.json &&
timeout -k 10 400 python bench.py --no-cpu --pattern stencil27 --p 32 --restart 256 --reduced 128 --expand 32 --lanczos 40 > $O/r03_bench_c4_parameters.json 2> $O/bench_c4.err && cut -c1-200 $O/r03_bench_c4_parameters.json &&
timeout -k 10 400 python bench.py --no-cpu --subspace 0 > $O/r03_bench_direct.json 2> $O/bench_direct.err && cut -c1-200 $O/r03_bench_direct.json &&
bash scripts/gpu_profile.sh r03 &&
bash scripts/gpu_profile.sh r03s --pattern stencil27 &&
cd $R && RAILS_RUN_SUBSPACE=1 timeout -k 10 600 python scripts/run_configs.py > $O/r03_configs.jsonl 2> $O/configs_default.err && tail -3 $O/r03_configs.jsonl | cut -c1-300 && echo ALL-DONE
