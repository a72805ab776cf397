#!/bin/bash
# GPU-box helper (round 2, batch 3): tests (incl. serial-state replay), fused-round variants on m1 and c2, c3 / c4 lines
cd "$(dirname "$0")/.."
repo="$PWD"; tag="${1:-r02c}"; out="$repo/gpurun_out"
mkdir -p "$out"
timeout -k 10 1000 python3 -m pytest tests -m gpu -q -x > "$out/${tag}_pytest.txt" 2>&1; rc=$?
tail -4 "$out/${tag}_pytest.txt"
[ $rc -ne 0 ] && { grep -n "Error\|assert\|FAILED\|differ" "$out/${tag}_pytest.txt" | head -30; }
echo "== m1"; bash tools/ab_variants.sh "$out/${tag}_ab_m1.txt" 1
echo "== c2"; bash tools/ab_variants.sh "$out/${tag}_ab_c2.txt" 1 --workload c2
echo "== c3"; bash tools/ab_variants.sh "$out/${tag}_ab_c3.txt" 1 --workload c3 --steps 2 --warmup 1
for wl in c4; do
  timeout -k 10 600 python3 bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline > "$out/${tag}_bench_$wl.json" 2> "$out/${tag}_bench_$wl.err"; tail -1 "$out/${tag}_bench_$wl.json" | cut -c1-400
  YAFGPU_SERIAL_REPLAY=0 timeout -k 10 600 python3 bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | cut -c1-300
done
exit $rc
