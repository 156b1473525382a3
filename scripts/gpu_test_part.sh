#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_partition.py -q -x --timeout 300 > $O/part_tests.log 2>&1; echo exit=$? >> $O/part_tests.log; tail -30 $O/part_tests.log
