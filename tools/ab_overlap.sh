#!/bin/bash
# GPU-box helper: bench with and without the two traversal launches overlapped
cd "$(dirname "$0")/.."
for v in 0 1 0 1; do
  if [ "$v" = "1" ]; then export YAFGPU_OVERLAP=1; else unset YAFGPU_OVERLAP; fi
  echo "== overlap=$v"
  for w in c2 c4; do
  timeout -k 10 300 python3 bench.py --workload $w --no-cpu-baseline --steps 4 --warmup 1 2>/dev/null | python3 -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('   ', '$w', 'Mrays/s', d['value'], 'ms/step', d['ms_per_step'])"
  done
done
