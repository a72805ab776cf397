#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/b5_prof -- python3 tools/probe_sorted_rays.py 4000000 > gpurun_out/b5_probe.log 2>&1; echo "rc=$?"; tail -8 gpurun_out/b5_probe.log
f=$(find gpurun_out/b5_prof -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "trace_kernel" in r["Kernel_Name"]]
for r in rows:
    print(r["Kernel_Name"][:60], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, "ms")
PY
rm -rf gpurun_out/b5_prof
