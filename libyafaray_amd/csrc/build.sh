#!/bin/bash
# Builds libyafaray_gpu.so (HIP kernels + narrow ABI + Interface-shaped C API) for gfx950, in-tree.
# Objects are rebuilt only when their source, any header here or in include/, this script or the flags changed
# (YAFGPU_FORCE=1 rebuilds everything).
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="${YAFGPU_OUT:-$HERE/../libyafaray_gpu.so}"
OBJ="${YAFGPU_OBJ:-$HERE/obj}"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
# -ffp-contract=off and IEEE divide/sqrt: the shading arithmetic must round like the reference's
# expressions (discrete hit / lobe / shadow decisions decide image parity)
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -Wall -Wno-unused-function"
EXTRA="${YAFGPU_EXTRA_FLAGS:-}"
mkdir -p "$OBJ"
HEADERS=("$HERE"/*.h "$HERE"/../../include/*.h "$HERE/build.sh")
STAMP="$OBJ/.flags"
if [ ! -f "$STAMP" ] || [ "$(cat "$STAMP")" != "$FLAGS $EXTRA" ] || [ "${YAFGPU_FORCE:-0}" = "1" ]; then rm -f "$OBJ"/*.o; echo "$FLAGS $EXTRA" > "$STAMP"; fi

stale() {   # stale <object> <source>: true when the object must be rebuilt
  local o="$1" s="$2" h
  [ -f "$o" ] || return 0
  [ "$s" -nt "$o" ] && return 0
  for h in "${HEADERS[@]}"; do [ "$h" -nt "$o" ] && return 0; done
  return 1
}
PIDS=()
cc() {      # cc <object name> <source> [defines...]: compile in the background when stale
  local o="$OBJ/$1.o" s="$HERE/$2"; shift 2
  if stale "$o" "$s"; then "$HIPCC" $FLAGS "$@" -c "$s" -o "$o" $EXTRA & PIDS+=($!); fi
}
# the device unit and the scene-specialised shading kernels (yafgpu_shade_variant.hip) compile side by side
cc yafgpu_device yafgpu_device.hip
#   diffuse: shinydiffusemat + light_mat, no recursiveRaytrace (BASELINE configs C2, C3)
cc shade_diffuse yafgpu_shade_variant.hip -DYAFGPU_VARIANT_NAME=diffuse -DYAFGPU_MAT_MASK=0x5u -DYAFGPU_FEAT_RECURSE=0 -DYAFGPU_FEAT_TEXTURE=0 -DYAFGPU_FEAT_ANISO=0 -DYAFGPU_FEAT_MULTI=0
#   glossy: + glossy (as_diffuse), no recursiveRaytrace (C4)
cc shade_glossy yafgpu_shade_variant.hip -DYAFGPU_VARIANT_NAME=glossy -DYAFGPU_MAT_MASK=0x7u -DYAFGPU_FEAT_RECURSE=0 -DYAFGPU_FEAT_TEXTURE=0 -DYAFGPU_FEAT_ANISO=0 -DYAFGPU_FEAT_MULTI=0
#   ... the two with a second MIS pair per park (YAFGPU_FEAT_MULTI): scenes with several lights or several samples per light (C4)
cc shade_diffuse_mp yafgpu_shade_variant.hip -DYAFGPU_VARIANT_NAME=diffuse_mp -DYAFGPU_MAT_MASK=0x5u -DYAFGPU_FEAT_RECURSE=0 -DYAFGPU_FEAT_TEXTURE=0 -DYAFGPU_FEAT_ANISO=0 -DYAFGPU_FEAT_MULTI=1
cc shade_glossy_mp yafgpu_shade_variant.hip -DYAFGPU_VARIANT_NAME=glossy_mp -DYAFGPU_MAT_MASK=0x7u -DYAFGPU_FEAT_RECURSE=0 -DYAFGPU_FEAT_TEXTURE=0 -DYAFGPU_FEAT_ANISO=0 -DYAFGPU_FEAT_MULTI=1
#   ... and the two as the program of a serial-state replay's record pass (no light estimate, the vertex of a resume in registers)
cc shade_diffuse_rec yafgpu_shade_variant.hip -DYAFGPU_VARIANT_NAME=diffuse_rec -DYAFGPU_MAT_MASK=0x5u -DYAFGPU_FEAT_RECURSE=0 -DYAFGPU_FEAT_TEXTURE=0 -DYAFGPU_FEAT_ANISO=0 -DYAFGPU_FEAT_LIGHTS=0
cc shade_glossy_rec yafgpu_shade_variant.hip -DYAFGPU_VARIANT_NAME=glossy_rec -DYAFGPU_MAT_MASK=0x7u -DYAFGPU_FEAT_RECURSE=0 -DYAFGPU_FEAT_TEXTURE=0 -DYAFGPU_FEAT_ANISO=0 -DYAFGPU_FEAT_LIGHTS=0
#   full: every material type and recursiveRaytrace, no shader nodes / textures (the main unit's kernel has those too)
cc shade_full yafgpu_shade_variant.hip -DYAFGPU_VARIANT_NAME=full -DYAFGPU_MAT_MASK=0x3fu -DYAFGPU_FEAT_RECURSE=1 -DYAFGPU_FEAT_TEXTURE=0 -DYAFGPU_FEAT_ANISO=0 -DYAFGPU_FEAT_MULTI=0
cc kdtree_build kdtree_build.cpp
cc kdtree_build_device kdtree_build_device.hip
OBJS="$OBJ/yafgpu_device.o $OBJ/shade_diffuse.o $OBJ/shade_glossy.o $OBJ/shade_diffuse_mp.o $OBJ/shade_glossy_mp.o $OBJ/shade_diffuse_rec.o $OBJ/shade_glossy_rec.o $OBJ/shade_full.o $OBJ/kdtree_build.o $OBJ/kdtree_build_device.o"
for f in yafaray_c_api yafaray_xml yafaray_image yafaray_reduce; do
  if [ -f "$HERE/$f.cpp" ]; then cc "$f" "$f.cpp"; OBJS="$OBJS $OBJ/$f.o"; fi
done
for p in "${PIDS[@]:-}"; do [ -n "$p" ] && wait "$p"; done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$OUT" $OBJS -lpthread -lz ${YAFGPU_LINK_LIBS:-}
echo "built $OUT"
