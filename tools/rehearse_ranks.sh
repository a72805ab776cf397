#!/bin/bash
# GPU-box helper: the sharded bench path with 1, 2 and 4 ranks on ONE GPU (gloo, host-staged reduce): the films must be identical.
cd "$(dirname "$0")/.."
for n in 1 2 4; do
  if [ $n = 1 ]; then
    timeout -k 10 300 python bench.py --rehearse-one-gpu --steps 1 --warmup 0 --no-cpu-baseline --res 256 --spp 16 > gpurun_out/rehearse_$n.json 2> gpurun_out/rehearse_$n.err || exit 1
  else
    timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus $n --rehearse-one-gpu --steps 1 --warmup 0 --no-cpu-baseline --res 256 --spp 16 > gpurun_out/rehearse_$n.json 2> gpurun_out/rehearse_$n.err || exit 1
  fi
  tail -1 gpurun_out/rehearse_$n.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$n ranks', d['film_checksum'], d['config']['rays_per_step'])"
done
