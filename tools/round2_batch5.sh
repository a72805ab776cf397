#!/bin/bash
# texture tests first, then the whole GPU suite, then the m1 bench (one gpurun call)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_textures.py -q -m gpu -s > gpurun_out/r02e_textures.log 2>&1; echo "textures rc=$?" | tee -a gpurun_out/r02e_textures.log
tail -5 gpurun_out/r02e_textures.log
timeout -k 10 300 python bench.py --steps 3 --warmup 1 > gpurun_out/r02e_bench.json 2> gpurun_out/r02e_bench.err; echo "bench rc=$?"; tail -c 1500 gpurun_out/r02e_bench.json
