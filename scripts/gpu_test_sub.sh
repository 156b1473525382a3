#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out
RAILS_CONTRACT_VERBOSE=1 ./rails_amd/lib/wrapper_contract > $O/contract.log 2>&1; echo exit=$? >> $O/contract.log; tail -12 $O/contract.log
timeout -k 10 600 python -m pytest tests/test_gpu_solver.py tests/test_gpu_partition.py -q --timeout 300 -k "subspace or partitioned_solve" > $O/sub_tests.log 2>&1; echo exit=$? >> $O/sub_tests.log; tail -30 $O/sub_tests.log
