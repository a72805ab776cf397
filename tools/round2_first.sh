#!/bin/bash
# GPU-box helper (round 2): GPU tests, then the default (m1) bench line, then rocprofv3 kernel stats of the same command
cd "$(dirname "$0")/.."
repo="$PWD"; tag="${1:-r02a}"; out="$repo/gpurun_out"
mkdir -p "$out"
timeout -k 10 900 python3 -m pytest tests -m gpu -q -x --durations=15 > "$out/${tag}_pytest.txt" 2>&1; rc=$?
tail -25 "$out/${tag}_pytest.txt"
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 > "$out/${tag}_bench_m1.json" 2> "$out/${tag}_bench_m1.err" || { tail -5 "$out/${tag}_bench_m1.err"; exit 1; }
cat "$out/${tag}_bench_m1.json"
cd /tmp && export TMPDIR=/tmp && cd "$repo"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${tag}_prof" -- python3 bench.py --no-cpu-baseline --steps 5 --warmup 1 > "$out/${tag}_prof.log" 2>&1
find "$out/${tag}_prof" -name "*kernel_stats.csv" -exec cp {} "$out/${tag}_kernel_stats.csv" \;
head -8 "$out/${tag}_kernel_stats.csv" | cut -c1-200
