#!/bin/bash
# GPU-box helper: pass pipelining at the whole frame with the persistent traversal launches on a part of the block slots
cd "$(dirname "$0")/.."
out="$1"; : > "$out"
for rep in 1 2; do
for wl in m1 c2; do
for v in "YAFGPU_PASS_PIPELINE=0 YAFGPU_BLOCKS_PER_CU=8" "YAFGPU_PASS_PIPELINE=1 YAFGPU_BLOCKS_PER_CU=8" "YAFGPU_PASS_PIPELINE=1 YAFGPU_BLOCKS_PER_CU=4" "YAFGPU_PASS_PIPELINE=1 YAFGPU_BLOCKS_PER_CU=5" "YAFGPU_PASS_PIPELINE=0 YAFGPU_BLOCKS_PER_CU=4" "YAFGPU_PASS_PIPELINE=1 YAFGPU_BLOCKS_PER_CU=3"; do
	line=$(env $v timeout -k 10 240 python3 bench.py --no-cpu-baseline --workload $wl --steps 12 --warmup 3 2>/dev/null | tail -1)
	python3 - "$wl $v" "$line" >> "$out" <<'PY'
import json, sys
tag, line = sys.argv[1:3]
try:
    d = json.loads(line)
    print(tag, "Mrays/s", d["value"], "ms_per_step", d["ms_per_step"])
except Exception as e:
    print(tag, "failed", line[:200])
PY
done; done; done
cat "$out"
