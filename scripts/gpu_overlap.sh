#!/bin/bash
# the overlapped block orthogonalisation of the coordinate-space back end: solver parity tests, then the bench with it on and off
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_solver.py tests/test_gpu_configs.py tests/test_gpu_fullsize.py tests/test_wrapper_contract.py -x -q > gpurun_out/overlap_tests.log 2>&1
rc=$?
tail -5 gpurun_out/overlap_tests.log
[ $rc -ne 0 ] && exit $rc
for ov in 1 0 1 0; do
  RAILS_SUBSPACE_OVERLAP=$ov timeout -k 10 300 python bench.py --no-cpu --direct-steps 0 --busy-steps 6 > gpurun_out/overlap_$ov.json 2> gpurun_out/overlap_$ov.err || exit 1
  echo "overlap $ov: $(python -c "import json; d=json.loads(open('gpurun_out/overlap_$ov.json').readline()); print('%.1f it/s, median trip %.2f ms, gpu busy %.2f' % (d['value'], d['config']['median_trip_ms'], d['config']['gpu_busy_frac']))")"
  grep "Lanczos estimates" gpurun_out/overlap_$ov.err | cut -c1-160
done
