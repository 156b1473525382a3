#!/bin/bash
# The slow trips of the driver-style bench (--steps 20 --warmup 5) after the OpenBLAS thread-start fix
set -o pipefail
mkdir -p gpurun_out
for r in 1 2 3 4; do
  RAILS_SOLVER_TRIP_TRACE=15 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu > gpurun_out/stall_fix$r.json 2> gpurun_out/stall_fix$r.err || exit 1
  echo "run $r:"; grep -E "rails trip|timed trips|it/s" gpurun_out/stall_fix$r.err | cut -c1-200
done
