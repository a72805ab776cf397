#!/bin/bash
# GPU-box helper: sweep the traversal vote / burst parameters
cd "$(dirname "$0")/.."
for cfg in "1 1 2" "1 1 4" "1 2 2" "1 2 4" "1 3 4" "2 1 2" "1 1 1" "1 2 8"; do
  set -- $cfg
  timeout -k 10 400 bash tools/sweep_build.sh "-DYAFGPU_VOTE_NUM=$1 -DYAFGPU_VOTE_DEN=$2 -DYAFGPU_NODE_BURST=$3" --steps 3 --warmup 1 | cut -c1-200
done
