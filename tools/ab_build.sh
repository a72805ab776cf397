#!/bin/bash
# GPU-box helper: bench C2 / C3 with the host-built and the device-built tree.
for b in host device; do
  for w in "$@"; do
    echo "== YAFGPU_BUILD=$b workload=$w"
    YAFGPU_BUILD=$b timeout -k 10 300 python bench.py --steps 5 --warmup 2 --workload $w --no-cpu-baseline || exit 1
  done
done
