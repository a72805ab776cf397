#!/bin/bash
# GPU-box helper: pass pipelining off / on over frame fractions (bench.py --emulate-shard K renders shard 0 of K: one rank's share)
cd "$(dirname "$0")/.."
out="$1"; : > "$out"
for rep in 1 2; do
for wl in m1 c2; do
for k in 1 2 4 8; do
for v in 0 1; do
	es=""; [ $k -gt 1 ] && es="--emulate-shard $k"
	line=$(env YAFGPU_PASS_PIPELINE=$v timeout -k 10 240 python3 bench.py --no-cpu-baseline --workload $wl --steps 12 --warmup 3 $es 2>/dev/null | tail -1)
	python3 - "$wl 1/$k pipeline=$v" "$line" >> "$out" <<'PY'
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
