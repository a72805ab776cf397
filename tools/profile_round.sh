#!/bin/bash
# GPU-box helper: the round's measurement set — GPU tests, bench lines for the three workloads, rocprofv3 kernel stats of the default bench
# usage: tools/profile_round.sh <tag>      (writes gpurun_out/<tag>_*)
cd "$(dirname "$0")/.."
repo="$PWD"; tag="$1"; out="$repo/gpurun_out"
mkdir -p "$out"
timeout -k 10 600 python3 -m pytest tests -m gpu -q -x > "$out/${tag}_pytest.txt" 2>&1 || { tail -5 "$out/${tag}_pytest.txt"; exit 1; }
tail -1 "$out/${tag}_pytest.txt"
timeout -k 10 600 python3 bench.py > "$out/${tag}_bench_c2.json" 2> "$out/${tag}_bench_c2.err" && tail -1 "$out/${tag}_bench_c2.json" | cut -c1-200
timeout -k 10 600 python3 bench.py --workload c3 --steps 2 --warmup 1 --no-cpu-baseline > "$out/${tag}_bench_c3.json" 2> "$out/${tag}_bench_c3.err" && tail -1 "$out/${tag}_bench_c3.json" | cut -c1-200
timeout -k 10 600 python3 bench.py --workload c4 --steps 2 --warmup 1 --no-cpu-baseline > "$out/${tag}_bench_c4.json" 2> "$out/${tag}_bench_c4.err" && tail -1 "$out/${tag}_bench_c4.json" | cut -c1-200
cd /tmp && export TMPDIR=/tmp && cd "$repo"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${tag}_prof" -- python3 bench.py --no-cpu-baseline --steps 5 --warmup 1 > "$out/${tag}_prof.log" 2>&1
find "$out/${tag}_prof" -name "*kernel_stats.csv" -exec cp {} "$out/${tag}_kernel_stats.csv" \;
head -8 "$out/${tag}_kernel_stats.csv" | cut -c1-160
