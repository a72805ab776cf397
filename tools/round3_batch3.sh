#!/bin/bash
# GPU-box batch: A/B of the traversal's leaf placement (default = leaves apart; variants/leaf0.so = the old placement), integrator
# pins with the new kernel, the two fuzz seeds whose exemption needed the filter's reach
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_integrator.py tests/test_gpu_parity.py -x -q -k "integrators or ray_batches or render_matches_oracle or full_size_m1 or odd_geometry or transparent_shadows" > gpurun_out/b3_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/b3_tests.log
YAFGPU_SERIAL_FUZZ_FIRST=31 YAFGPU_SERIAL_FUZZ_SEEDS=32 timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -k "test_random_feature_mixes_with_serial_state" > gpurun_out/b3_fuzz31.log 2>&1; echo "fuzz31 rc=$?"
YAFGPU_SERIAL_FUZZ_FIRST=180 YAFGPU_SERIAL_FUZZ_SEEDS=181 timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -k "test_random_feature_mixes_with_serial_state" > gpurun_out/b3_fuzz180.log 2>&1; echo "fuzz180 rc=$?"
bash tools/ab_variants.sh gpurun_out/b3_ab_leaf_m1.txt 3 --workload m1
bash tools/ab_variants.sh gpurun_out/b3_ab_leaf_c2.txt 2 --workload c2
