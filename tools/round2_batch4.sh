#!/bin/bash
# GPU-box helper (round 2, batch 4): full GPU tests, cost-ratio sweep of the tree builder on m1, refill thresholds, c4 with / without replay
cd "$(dirname "$0")/.."
repo="$PWD"; tag="${1:-r02d}"; out="$repo/gpurun_out"
mkdir -p "$out"
timeout -k 10 1000 python3 -m pytest tests -m gpu -q > "$out/${tag}_pytest.txt" 2>&1; rc=$?
tail -6 "$out/${tag}_pytest.txt"
[ $rc -ne 0 ] && { grep -n "Error\|assert \|FAILED\|differ" "$out/${tag}_pytest.txt" | head -40; }
echo "== m1 variants"; bash tools/ab_variants.sh "$out/${tag}_ab_m1.txt" 1
echo "== cost ratio sweep (m1, default library)"
for cr in 0.3 0.5 0.8 1.2; do
  YAFGPU_COST_RATIO=$cr timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('cost_ratio $cr', d['value'], d['ms_per_step'], r['pass_ms'], r['per_ray'], d['config']['kd_nodes'], d['config']['scene_device_MB'])
" | tee -a "$out/${tag}_cost.txt"
done
for wl in c4; do
  timeout -k 10 600 python3 bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline > "$out/${tag}_bench_$wl.json" 2> "$out/${tag}_bench_$wl.err"; tail -1 "$out/${tag}_bench_$wl.json" | cut -c1-300
  YAFGPU_SERIAL_REPLAY=0 timeout -k 10 600 python3 bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | cut -c1-300
done
exit $rc
