#!/bin/bash
# GPU-box helper: PMC passes for the render kernel (counters only; no trace domains besides kernel-trace)
# usage: tools/pmc.sh <outdir> [bench args]
cd "$(dirname "$0")/.."
repo="$PWD"
out="$repo/$1"; shift
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$repo"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$out/pass$i" -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 0 "$@" > "$out/pass$i.log" 2>&1 || echo "pass $i failed"
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections, re, json
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(set)
for f in glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "yafgpu" not in k: continue
        k = re.sub(r"\(.*", "", k).replace("void ", "").replace("yafgpu::", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "FETCH_SIZE": launches[k].add(r["Dispatch_Id"])
for k in sorted(agg):
    print("==", k, "launches", len(launches[k]))
    for c, v in sorted(agg[k].items()):
        print(f"   {c:26s} {v:.6g}")
# HBM-side traffic of the traversal kernels per launch: (FETCH_SIZE + WRITE_SIZE) KiB -> bytes.
# MI355X_MICROARCH.md: FETCH_SIZE under-reports wide coalesced streams by 2x on gfx950; the traversal's 8/16-byte
# gathers are an uncalibrated width, so the raw figure is reported.
tr = [k for k in agg if k.startswith("wf_trace") and ", false>" in k]
fetch = sum(agg[k].get("FETCH_SIZE", 0) for k in tr) * 1024
write = sum(agg[k].get("WRITE_SIZE", 0) for k in tr) * 1024
n = sum(len(launches[k]) for k in tr)
if n:
    json.dump({"kernel": "wf_trace", "launches": n, "fetch_bytes_per_launch": fetch / n, "write_bytes_per_launch": write / n,
               "traffic_bytes_per_launch": (fetch + write) / n, "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), raw"},
              open(out + "/traffic.json", "w"), indent=1)
    print("traffic per wf_trace launch: %.3f MB" % ((fetch + write) / n / 1e6))
PY
