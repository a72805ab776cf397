#!/bin/bash
# GPU-box helper: rebuild the library with the given extra flags and run the bench (no CPU baseline).
# usage: tools/sweep_build.sh "<extra hipcc flags>" [bench args...]
set -e
cd "$(dirname "$0")/.."
flags="$1"; shift
YAFGPU_EXTRA_FLAGS="$flags" bash libyafaray_amd/csrc/build.sh > /dev/null 2>&1
echo "== $flags"
python3 bench.py --no-cpu-baseline "$@" 2>/dev/null | python3 -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('   Mrays/s', d['value'], 'ms/step', d['ms_per_step'], r['pass_ms'], 'B/ray', r['bytes_per_ray'], r['per_ray'])"
