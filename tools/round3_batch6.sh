#!/bin/bash
# GPU-box batch: the next segment parked beside a vertex's last shadow pair (WfArgs::speculate): parity, then A/B against YAFGPU_SPECULATE=0
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_integrator.py -x -q -k "not serial and not full_size_c4 and not xml" > gpurun_out/b6_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/b6_tests.log
for wl in m1 c2 c3; do
  for sp in 0 1 0 1; do
    YAFGPU_SPECULATE=$sp timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 10 --warmup 2 --workload $wl 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('$wl speculate=$sp', d['value'], d['ms_per_step'], r['pass_ms'], d['config']['rays_per_step'])
" | tee -a gpurun_out/b6_ab_spec.txt
  done
done
for n in 2 8; do for sp in 0 1; do
  YAFGPU_SPECULATE=$sp timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 10 --warmup 2 --emulate-shard $n 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('m1 shard 1/$n speculate=$sp ms_per_step', d['ms_per_step'])
" | tee -a gpurun_out/b6_ab_spec.txt
done; done
