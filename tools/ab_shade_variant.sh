#!/bin/bash
# GPU-box helper: the scene-specialised shading kernel against the general one, alternating on the same box.
for rep in 1 2; do
  for v in auto general; do
    for w in "$@"; do
      echo "== variant=$v workload=$w rep=$rep"
      if [ $v = general ]; then export YAFGPU_SHADE_VARIANT=general; else unset YAFGPU_SHADE_VARIANT; fi
      timeout -k 10 300 python bench.py --steps 5 --warmup 2 --workload $w --no-cpu-baseline || exit 1
    done
  done
done
