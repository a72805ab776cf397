#!/bin/bash
# GPU-box helper: sweep the refill threshold of the persistent traversal waves
cd "$(dirname "$0")/.."
for r in 8 16 24 32 40 48 56; do
  timeout -k 10 400 bash tools/sweep_build.sh "-DYAFGPU_REFILL=$r" --steps 3 --warmup 1 | cut -c1-150
done
