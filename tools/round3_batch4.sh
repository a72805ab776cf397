#!/bin/bash
# GPU-box batch: the top-of-tree copy (variants top1 = L1-resident, top2 = staged in LDS, pad = the LDS of top2 without it, nt = non-temporal
# triangle loads): parity of the variants, then the A/B
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for v in top1 top2; do
  YAFARAY_LIBRARY="$PWD/libyafaray_amd/variants/$v.so" timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_integrator.py -x -q -k "ray_batches or render_matches_oracle or integrators or odd_geometry or full_size_m1 or transparent_shadows or feature_mixes" > gpurun_out/b4_tests_$v.log 2>&1; echo "$v tests rc=$?"; tail -2 gpurun_out/b4_tests_$v.log
done
bash tools/ab_variants.sh gpurun_out/b4_ab_top_m1.txt 2 --workload m1
bash tools/ab_variants.sh gpurun_out/b4_ab_top_c4.txt 1 --workload c4 --spp 16
