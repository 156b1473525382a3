#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "lanczos" > $O/t15.log 2>&1; echo exit=$? >> $O/t15.log; tail -3 $O/t15.log
timeout -k 10 300 python bench.py --no-cpu > $O/bench_b.json 2> $O/bench_b.err; grep -E "host sections|trips in|counters|SpMM" $O/bench_b.err | cut -c1-330
timeout -k 10 300 python bench.py --no-cpu --pattern stencil27 > $O/bench_s.json 2> $O/bench_s.err; grep -E "host sections|trips in|counters|SpMM|setup" $O/bench_s.err | cut -c1-330
bash scripts/gpu_profile.sh r01s --pattern stencil27
