#!/bin/bash
# GPU-box helper: sweep of the device kd builder's SAH parameters (YAFGPU_COST_RATIO, YAFGPU_EMPTY_BONUS) over bench workloads
# usage: tools/ab_tree.sh <outfile> "<workloads>" "<cost ratios>" "<empty bonuses>"
cd "$(dirname "$0")/.."
out="$1"; wls="$2"; crs="$3"; ebs="$4"
: > "$out"
for wl in $wls; do
for cr in $crs; do
for eb in $ebs; do
	line=$(env YAFGPU_COST_RATIO=$cr YAFGPU_EMPTY_BONUS=$eb timeout -k 10 240 python3 bench.py --no-cpu-baseline --workload "$wl" --steps 6 --warmup 2 2>/dev/null | tail -1)
	python3 - "$wl" "cost_ratio=$cr empty_bonus=$eb" "$line" >> "$out" <<'PY'
import json, sys
wl, tag, line = sys.argv[1:4]
try:
    d = json.loads(line); r = d["roofline"]; c = d["config"]
    print(wl, tag, d["value"], d["ms_per_step"], r["pass_ms"], r["per_ray"], "nodes", c["kd_nodes"], "leaf_refs", c["kd_leaf_refs"], "depth", c["kd_max_depth"], "build_s", c["tree_build_s"], "rays", c["rays_per_step"])
except Exception as e:
    print(wl, tag, "failed", line[:200])
PY
done; done; done
cat "$out"
