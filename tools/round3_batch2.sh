#!/bin/bash
# GPU-box batch: the whole GPU suite, then the serial-state fuzz on seeds 12..199 (exemptions must be shown ties)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 1500 python -m pytest tests -q -m gpu -x > gpurun_out/b2_suite.log 2>&1; echo "suite rc=$?"; tail -5 gpurun_out/b2_suite.log
YAFGPU_SERIAL_FUZZ_FIRST=12 YAFGPU_SERIAL_FUZZ_SEEDS=200 timeout -k 10 1500 python -m pytest tests/test_gpu_parity.py -q -s -k "test_random_feature_mixes_with_serial_state" > gpurun_out/b2_fuzz.log 2>&1; echo "fuzz rc=$?"
grep -c "PASSED\|passed" gpurun_out/b2_fuzz.log; grep "exempt\|FAILED\|Error" gpurun_out/b2_fuzz.log | cut -c1-400 | head -20; tail -3 gpurun_out/b2_fuzz.log
