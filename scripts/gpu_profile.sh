#!/bin/bash
# rocprofv3 evidence for profiles/: kernel-trace stats of the default bench command, then PMC passes (separate runs,
# --pmc only) for HBM traffic of the SpMM and Lanczos kernels.  usage: bash scripts/gpu_profile.sh <tag> [bench args]
set -o pipefail
TAG=${1:-r01}; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --no-cpu "$@" > $O/bench_stats.json 2> $O/bench_stats.err
tail -2 $O/bench_stats.err
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $ctr --output-format csv -d $O/pmc_$ctr -- python3 $R/bench.py --spmm-only --spmm-cols 128 --spmm-reps 3 "$@" > $O/pmc_$ctr.json 2> $O/pmc_$ctr.err
  tail -1 $O/pmc_$ctr.err
done
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $ctr --output-format csv -d $O/pmcsolve_$ctr -- python3 $R/bench.py --no-cpu --warmup 8 --steps 6 "$@" > $O/pmcsolve_$ctr.json 2> $O/pmcsolve_$ctr.err
  tail -1 $O/pmcsolve_$ctr.err
done
find $O -name "*.csv" | head -30
du -sh $O
