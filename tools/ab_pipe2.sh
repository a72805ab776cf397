#!/bin/bash
# GPU-box helper: passes in flight (YAFGPU_PASS_PIPELINE_DEPTH 2 / 3) over frame fractions, and the default (by size)
cd "$(dirname "$0")/.."
out="$1"; : > "$out"
for rep in 1 2; do
for wl in m1; do
for k in 2 4 8 16; do
for v in "YAFGPU_PASS_PIPELINE=0" "YAFGPU_PASS_PIPELINE_DEPTH=2" "YAFGPU_PASS_PIPELINE_DEPTH=3" "YAFGPU_NOP=1"; do
	line=$(env $v timeout -k 10 240 python3 bench.py --no-cpu-baseline --workload $wl --steps 12 --warmup 3 --emulate-shard $k 2>/dev/null | tail -1)
	python3 - "$wl 1/$k $v" "$line" >> "$out" <<'PY'
import json, sys
tag, line = sys.argv[1:3]
try:
    d = json.loads(line)
    print(tag, "Mrays/s", d["value"], "ms_per_step", d["ms_per_step"])
except Exception as e:
    print(tag, "failed", line[:200])
PY
done; done; done; done
cat "$out"
