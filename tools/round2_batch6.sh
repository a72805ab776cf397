#!/bin/bash
# the whole GPU suite (one process), log under gpurun_out/
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -q -m gpu --durations=15 > gpurun_out/r02f_gpu_tests.log 2>&1; rc=$?
echo "gpu tests rc=$rc"; tail -30 gpurun_out/r02f_gpu_tests.log
exit $rc
