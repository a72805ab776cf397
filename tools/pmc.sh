#!/bin/bash
# GPU-box helper: PMC passes for the render kernel (counters only; no trace domains besides kernel-trace)
# usage: tools/pmc.sh <outdir> [bench args]
cd "$(dirname "$0")/.."
repo="$PWD"
out="$repo/$1"; shift
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$repo"
export YAFGPU_OVERLAP=0      # one kernel on the GPU at a time: counters are per dispatch
i=0
for set in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$out/pass$i" -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 0 "$@" > "$out/pass$i.log" 2>&1 || echo "pass $i failed"
done
wl=m1; prev=""; for a in "$@"; do [ "$prev" = "--workload" ] && wl="$a"; prev="$a"; done
python3 - "$out" "$wl" <<'PY'
import csv, glob, sys, collections, re, json
out = sys.argv[1]
sys.path.insert(0, ".")
import bench
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
def ratios(keys):
    g = lambda c: sum(agg[k].get(c, 0.0) for k in keys)
    o = {}
    if g("SQ_ACTIVE_INST_VALU"): o["lane_utilisation"] = round(g("SQ_THREAD_CYCLES_VALU") / (64.0 * g("SQ_ACTIVE_INST_VALU")), 4)
    if g("SQ_INSTS_VALU"): o["salu_per_valu"] = round(g("SQ_INSTS_SALU") / g("SQ_INSTS_VALU"), 4)
    if g("TCC_HIT_sum") + g("TCC_MISS_sum"): o["l2_hit_rate"] = round(g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum")), 4)
    return o
# Memory-side traffic per launch: (FETCH_SIZE + WRITE_SIZE) KiB -> bytes.  MI355X_MICROARCH.md ("HBM"): FETCH_SIZE reports
# exactly half of a WIDE COALESCED streaming read (16 B/lane) and is uncalibrated for other widths; WRITE_SIZE is exact.
# The traversal reads are 8/16-byte gathers (uncalibrated width): `traffic` is the raw sum, `traffic_fetch_x2` the upper
# bound with the streaming correction applied to every fetched byte.  The shade kernel's record loads ARE 16 B/lane
# coalesced streams, so its corrected figure uses the x2.
doc = {"workload": sys.argv[2], "kernel_key": bench.kernel_key(),
       "source": "rocprofv3 --pmc (separate passes: FETCH_SIZE | WRITE_SIZE,TCC_HIT,TCC_MISS | SQ set 1 | SQ set 2), tools/pmc.sh",
       "fetch_correction": "wf_trace: none (8/16-B gathers are an uncalibrated width, MI355X_MICROARCH.md HBM section); x2 upper bound in traffic_fetch_x2"}
tr = [k for k in agg if k.startswith("wf_trace<") and ", false>" in k]
n = sum(len(launches[k]) for k in tr)
if n:
    fetch = sum(agg[k].get("FETCH_SIZE", 0) for k in tr) * 1024
    write = sum(agg[k].get("WRITE_SIZE", 0) for k in tr) * 1024
    doc.update({"kernel": "wf_trace", "launches": n, "fetch_bytes_per_launch_raw": fetch / n, "write_bytes_per_launch": write / n,
                "traffic_bytes_per_launch": (fetch + write) / n, "traffic_fetch_x2": (2 * fetch + write) / n})
    doc.update(ratios(tr))
    for name, sel in (("closest", "wf_trace<false"), ("any_hit", "wf_trace<true")):
        ks = [k for k in tr if k.startswith(sel)]
        m = sum(len(launches[k]) for k in ks)
        if m:
            doc[name] = dict(launches=m, fetch_bytes_per_launch_raw=sum(agg[k].get("FETCH_SIZE", 0) for k in ks) * 1024 / m,
                             write_bytes_per_launch=sum(agg[k].get("WRITE_SIZE", 0) for k in ks) * 1024 / m, **ratios(ks))
    print("traffic per wf_trace launch: %.3f MB raw" % ((fetch + write) / n / 1e6))
sh = [k for k in agg if "wf_shade" in k]
m = sum(len(launches[k]) for k in sh)
if m:
    fetch = sum(agg[k].get("FETCH_SIZE", 0) for k in sh) * 1024
    write = sum(agg[k].get("WRITE_SIZE", 0) for k in sh) * 1024
    doc["shade"] = dict(launches=m, fetch_bytes_per_launch_raw=fetch / m, write_bytes_per_launch=write / m,
                        traffic_bytes_per_launch_fetch_x2=(2 * fetch + write) / m, **ratios(sh))
json.dump(doc, open(out + "/traffic.json", "w"), indent=1)
PY
