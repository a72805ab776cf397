#!/bin/bash
# GPU-box helper: the round's judged measurements -> gpurun_out/<tag>_*, to be copied into profiles/
#   m1 (BASELINE.json metric config): bench.py default line, rocprofv3 kernel stats of the same command, PMC passes (tools/pmc.sh)
#   c3 / c4: bench line + kernel stats
# usage: tools/round3_measure.sh <tag>
cd "$(dirname "$0")/.."
repo="$PWD"; tag="${1:-r03}"; out="$repo/gpurun_out"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$repo"
echo "== m1 bench (default command)"
timeout -k 10 600 python3 bench.py > "$out/${tag}_m1_bench.json" 2> "$out/${tag}_m1_bench.err" || echo "bench failed"
cut -c1-400 "$out/${tag}_m1_bench.json"
echo "== m1 rocprofv3 kernel stats (same command, no cpu baseline; YAFGPU_OVERLAP=0: one kernel on the GPU at a time, like the profiled pass the roofline's durations come from)"
YAFGPU_OVERLAP=0 timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${tag}_m1_prof" -- python3 bench.py --no-cpu-baseline > "$out/${tag}_m1_prof.log" 2>&1 || echo "rocprof failed"
f=$(find "$out/${tag}_m1_prof" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$out/${tag}_m1_kernel_stats.csv" && head -8 "$out/${tag}_m1_kernel_stats.csv"
echo "== m1 PMC passes"
bash tools/pmc.sh "gpurun_out/${tag}_m1_pmc" --workload m1 > "$out/${tag}_m1_pmc.txt" 2>&1; tail -3 "$out/${tag}_m1_pmc.txt"
for wl in c2; do
  timeout -k 10 600 python3 bench.py --workload $wl --no-cpu-baseline --steps 10 --warmup 2 > "$out/${tag}_${wl}_bench.json" 2> "$out/${tag}_${wl}_bench.err" || echo "bench $wl failed"
  cut -c1-300 "$out/${tag}_${wl}_bench.json"
done
echo "== what one rank of N does per pass (shard 0 of N of the metric frame on this one GPU; reduce not included)"
for n in 1 2 4 8; do
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 10 --warmup 2 --emulate-shard $n 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('m1 shard 0 of $n: ms per pass', d['ms_per_step'], 'Mrays/s of the shard', d['value'])
" | tee -a "$out/${tag}_emulate_shard.txt"
done
for wl in c3 c4; do
  echo "== $wl bench + kernel stats"
  timeout -k 10 600 python3 bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline > "$out/${tag}_${wl}_bench.json" 2> "$out/${tag}_${wl}_bench.err" || echo "bench $wl failed"
  cut -c1-300 "$out/${tag}_${wl}_bench.json"
  YAFGPU_OVERLAP=0 timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${tag}_${wl}_prof" -- python3 bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline > "$out/${tag}_${wl}_prof.log" 2>&1 || echo "rocprof $wl failed"
  f=$(find "$out/${tag}_${wl}_prof" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$out/${tag}_${wl}_kernel_stats.csv"
done
echo "== c4 PMC passes"
bash tools/pmc.sh "gpurun_out/${tag}_c4_pmc" --workload c4 > "$out/${tag}_c4_pmc.txt" 2>&1; tail -3 "$out/${tag}_c4_pmc.txt"
# the big raw trace directories stay on the box
rm -rf "$out/${tag}_m1_prof" "$out/${tag}_c3_prof" "$out/${tag}_c4_prof" "$out/${tag}_m1_pmc"/pass*/ "$out/${tag}_c4_pmc"/pass*/
ls -la "$out" | grep "$tag"
