#!/bin/bash
# GPU-box helper: bench every library under libyafaray_amd/variants/ (and the default one) on one workload, alternating,
# and print one line per run: variant, Mrays/s, ms per kernel kind, lanes per instruction (stats pass)
# usage: tools/ab_variants.sh <outfile> <rounds> [bench args]
cd "$(dirname "$0")/.."
out="$1"; rounds="$2"; shift 2
: > "$out"
for r in $(seq 1 "$rounds"); do
  for lib in default $(ls libyafaray_amd/variants/*.so 2>/dev/null); do
    name="$(basename "$lib" .so)"
    if [ "$lib" = default ]; then unset YAFARAY_LIBRARY; else export YAFARAY_LIBRARY="$PWD/$lib"; fi
    timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 10 --warmup 2 "$@" 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('$name', d['value'], d['ms_per_step'], r['pass_ms'], r['simt'], r['per_ray'])
" >> "$out"
  done
done
unset YAFARAY_LIBRARY
cat "$out"
