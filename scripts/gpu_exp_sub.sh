#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out
RAILS_SUBSPACE_PROFILE=1 timeout -k 10 300 python bench.py --no-cpu --subspace 1 > $O/bench_subp.json 2> $O/bench_subp.err; tail -3 $O/bench_subp.err | cut -c1-900
timeout -k 10 300 python bench.py --no-cpu --subspace 1 > $O/bench_sub.json 2> $O/bench_sub.err; tail -3 $O/bench_sub.err | cut -c1-900
RAILS_LAPACK_LIB=/opt/conda/lib/libmkl_rt.so timeout -k 10 300 python bench.py --no-cpu --subspace 1 > $O/bench_submkl.json 2> $O/bench_submkl.err; tail -3 $O/bench_submkl.err | cut -c1-500
