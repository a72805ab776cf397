#!/bin/bash
# GPU-box helper (round 2, batch 2): tests with the new traversal kernel, A/B of its variants, chunk-size sweep, PMC of m1
cd "$(dirname "$0")/.."
repo="$PWD"; tag="${1:-r02b}"; out="$repo/gpurun_out"
mkdir -p "$out"
timeout -k 10 900 python3 -m pytest tests -m gpu -q -x > "$out/${tag}_pytest.txt" 2>&1; rc=$?
tail -4 "$out/${tag}_pytest.txt"
[ $rc -ne 0 ] && { grep -n "Error\|assert\|FAILED" "$out/${tag}_pytest.txt" | head -20; exit $rc; }
bash tools/ab_variants.sh "$out/${tag}_ab.txt" 1
echo "== chunk sweep (default library)"
for ch in 524288 1048576 2097152 4194304 8388608; do
  YAFGPU_WF_CHUNK=$ch timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('chunk $ch', d['value'], d['ms_per_step'], r['pass_ms'])
" | tee -a "$out/${tag}_chunks.txt"
done
bash tools/pmc.sh "gpurun_out/${tag}_pmc_m1" > "$out/${tag}_pmc_m1.txt" 2>&1
tail -30 "$out/${tag}_pmc_m1.txt"
