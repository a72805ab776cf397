#!/bin/bash
# GPU-box batch: new multi-GPU pieces (C-ABI RCCL world 1, C4 in eight shards, self-launched 2-rank rehearsal)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_distributed.py -x -q -k "c4_as_stated or rccl_communicator" > gpurun_out/b1_tests.log 2>&1; echo "tests rc=$?"; tail -15 gpurun_out/b1_tests.log
timeout -k 10 300 python bench.py --gpus 2 --rehearse-one-gpu --res 256 --spp 16 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/b1_rehearse2.json 2> gpurun_out/b1_rehearse2.err; echo "rehearse rc=$?"; cut -c1-600 gpurun_out/b1_rehearse2.json; tail -5 gpurun_out/b1_rehearse2.err
