#!/bin/bash
# GPU-box helper: bench a list of extra-flag sets given one per line on stdin
cd "$(dirname "$0")/.."
while IFS= read -r flags; do
  timeout -k 10 400 bash tools/sweep_build.sh "$flags" --steps 3 --warmup 1 | cut -c1-160
done
