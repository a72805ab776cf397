#!/bin/bash
# GPU-box helper: instruction-mix / wait PMC passes (SQ block only) for the wavefront kernels.
# usage: tools/pmc_sq.sh <outdir> [bench args]
cd "$(dirname "$0")/.."
repo="$PWD"
out="$repo/$1"; shift
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$repo"
i=0
for set in \
 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
 "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" \
 "SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_MISC SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_IFETCH SQ_CYCLES"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$out/pass$i" -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 0 "$@" > "$out/pass$i.log" 2>&1 || echo "pass $i failed"
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
