#!/bin/bash
# rehearsal of the multi-process bench path on ONE GPU: N ranks (<= 6) share cuda:0, torch.distributed backend gloo with the
# collectives staged through host memory -- everything of `bench.py --gpus N` except RCCL itself.  usage: gpu_rehearse_ranks.sh [N] [rows per rank]
set -o pipefail
cd $GRAFT_REPO_ROOT
N=${1:-2}; M=${2:-250000}
O=gpurun_out
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus $N --backend gloo --one-device --rows-per-gpu $M --no-cpu --steps 12 --warmup 8 > $O/rehearse_$N.json 2> $O/rehearse_$N.err
echo "exit=$?"; wc -l $O/rehearse_$N.json; cut -c1-400 $O/rehearse_$N.json; grep -E "trips in|ghosts|counters" $O/rehearse_$N.err | cut -c1-330
