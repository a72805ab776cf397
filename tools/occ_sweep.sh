#!/bin/bash
# GPU-box helper: bench at several caps of resident blocks per CU (occupancy experiment for the persistent kernels)
cd "$(dirname "$0")/.."
for b in 8 6 4 2 1; do
  echo "== blocks/CU cap $b"
  YAFGPU_BLOCKS_PER_CU=$b timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python3 -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('   Mrays/s', d['value'], 'ms/step', d['ms_per_step'], r['pass_ms'])"
done
