#!/bin/bash
# GPU-box helper: A/B of one environment switch over bench workloads.
# usage: tools/ab_env.sh <outfile> <VAR> "<values>" "<workloads>" [bench args]
cd "$(dirname "$0")/.."
out="$1"; var="$2"; vals="$3"; wls="$4"; shift 4
: > "$out"
for rep in 1 2; do
for wl in $wls; do
for v in $vals; do
	line=$(env "$var=$v" timeout -k 10 240 python3 bench.py --no-cpu-baseline --workload "$wl" --steps 6 --warmup 2 "$@" 2>/dev/null | tail -1)
	python3 - "$wl" "$var=$v" "$line" >> "$out" <<'PY'
import json, sys
wl, tag, line = sys.argv[1:4]
try:
    d = json.loads(line); r = d["roofline"]
    print(wl, tag, d["value"], d["ms_per_step"], r["pass_ms"], r.get("simt"))
except Exception as e:
    print(wl, tag, "failed", line[:200])
PY
done; done; done
cat "$out"
