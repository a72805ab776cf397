#!/bin/bash
# Container helper: build A/B variants of the library (each a full build.sh run with extra -D flags) into
# libyafaray_amd/variants/<name>.so; they travel to the GPU box with the snapshot and are selected with YAFARAY_LIBRARY.
# usage: tools/build_variants.sh name1="-DFOO=1 -DBAR=2" name2="-DFOO=3" ...
cd "$(dirname "$0")/.."
mkdir -p libyafaray_amd/variants
for spec in "$@"; do
  name="${spec%%=*}"; flags="${spec#*=}"
  echo "== $name: $flags"
  # objects of a variant live in their own directory (kept between calls: only stale ones are rebuilt)
  YAFGPU_OBJ="$PWD/libyafaray_amd/csrc/obj_$name" YAFGPU_OUT="$PWD/libyafaray_amd/variants/$name.so" YAFGPU_EXTRA_FLAGS="$flags" bash libyafaray_amd/csrc/build.sh > /tmp/build_$name.log 2>&1 || { tail -20 /tmp/build_$name.log; exit 1; }
done
bash libyafaray_amd/csrc/build.sh > /tmp/build_default.log 2>&1 || { tail -20 /tmp/build_default.log; exit 1; }
ls -la libyafaray_amd/variants/
