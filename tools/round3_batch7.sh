#!/bin/bash
# GPU-box batch: speculation with the serial-state replay (C4 as stated), the whole GPU suite, overlap at the full frame
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 1500 python -m pytest tests -q -m gpu -x > gpurun_out/b7_suite.log 2>&1; echo "suite rc=$?"; tail -4 gpurun_out/b7_suite.log
for sp in 0 1 0 1; do
  YAFGPU_SPECULATE=$sp timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 --workload c4 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('c4 speculate=$sp', d['value'], d['ms_per_step'], r['pass_ms'], d['config']['rays_per_step'])
" | tee -a gpurun_out/b7_ab_spec_c4.txt
done
for ov in 0 1 0 1; do
  YAFGPU_OVERLAP=$ov timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('m1 overlap=$ov', d['value'], d['ms_per_step'])
" | tee -a gpurun_out/b7_ab_overlap.txt
done
