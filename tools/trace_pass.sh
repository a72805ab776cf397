#!/bin/bash
# GPU-box helper: kernel timeline of the LAST pass of `bench.py <args>` (rocprofv3 --kernel-trace): start offset, duration, name
# usage: tools/trace_pass.sh <outfile> [bench args]
cd "$(dirname "$0")/.."
out="$1"; shift
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rm -rf gpurun_out/_trace
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/_trace -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" > gpurun_out/_trace.log 2>&1 || echo "rocprof failed"
f=$(find gpurun_out/_trace -name "*kernel_trace.csv" | head -1)
python3 - "$f" > "$out" <<'PY'
import csv, re, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "yafgpu" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# passes start with wf_generate; the roofline's extra passes come last: take the 3rd timed pass = 4th wf_generate overall (1 warmup + 3 steps)
gens = [i for i, r in enumerate(rows) if "wf_generate" in r["Kernel_Name"]]
import os
seg = int(os.environ.get("TRACE_SEG", "3"))      # which wf_generate-to-wf_generate stretch (a pass of a chunked or replayed render has several)
lo, hi = gens[seg], gens[seg + 1] if len(gens) > seg + 1 else len(rows)
t0 = int(rows[lo]["Start_Timestamp"])
end_prev = t0
for r in rows[lo:hi]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("yafgpu::", "")
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  gap {(s - end_prev) / 1e3:7.1f}  {name}  grid {r.get('Grid_Size', '')}")
    end_prev = max(end_prev, e)
print(f"pass total {(end_prev - t0) / 1e3:.1f} us")
PY
rm -rf gpurun_out/_trace
cat "$out"
