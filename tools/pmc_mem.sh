#!/bin/bash
# GPU-box helper: memory-pipeline PMC passes (TA / TCP / TD / SQ VMEM levels) for the wavefront kernels.
# usage: tools/pmc_mem.sh <outdir> [bench args]
cd "$(dirname "$0")/.."
repo="$PWD"
out="$repo/$1"; shift
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$repo"
i=0
for set in \
 "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum GRBM_GUI_ACTIVE" \
 "TCP_GATE_EN1_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
 "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum" \
 "TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
 "SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
 "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum" \
 "TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum TA_BUFFER_TOTAL_CYCLES_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$out/pass$i" -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 0 "$@" > "$out/pass$i.log" 2>&1 || echo "pass $i failed"
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections, re
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(lambda: collections.defaultdict(set))
for f in glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "yafgpu" not in k: continue
        k = re.sub(r"\(.*", "", k).replace("void ", "").replace("yafgpu::", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        launches[k][r["Counter_Name"]].add(r["Dispatch_Id"])
for k in sorted(agg):
    print("==", k)
    for c, v in sorted(agg[k].items()):
        n = len(launches[k][c])
        print(f"   {c:40s} {v / n:14.6g} per launch ({n} launches)")
PY
