#!/bin/bash
# GPU-box helper: A/B of two builds of the library (YAFARAY_LIBRARY) over bench workloads
# usage: tools/ab_lib.sh <outfile> "<workloads>" <name>=<path.so> ...
cd "$(dirname "$0")/.."
out="$1"; wls="$2"; shift 2
: > "$out"
for rep in 1 2; do
for wl in $wls; do
for spec in "$@"; do
	name="${spec%%=*}"; lib="${spec#*=}"
	st="--steps 6 --warmup 2"; [ "$wl" = c4 ] && st="--steps 3 --warmup 1"; [ "$wl" = c3 ] && st="--steps 2 --warmup 1"
	line=$(env YAFARAY_LIBRARY="$PWD/$lib" timeout -k 10 300 python3 bench.py --no-cpu-baseline --workload $wl $st 2>/dev/null | tail -1)
	python3 - "$wl $name" "$line" >> "$out" <<'PY'
import json, sys
tag, line = sys.argv[1:3]
try:
    d = json.loads(line); r = d["roofline"]
    print(tag, d["value"], d["ms_per_step"], r["pass_ms"])
except Exception as e:
    print(tag, "failed", line[:200])
PY
done; done; done
cat "$out"
