#!/bin/bash
# Builds libyafaray_gpu.so (HIP kernels + narrow ABI + Interface-shaped C API) for gfx950, in-tree.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="${YAFGPU_OUT:-$HERE/../libyafaray_gpu.so}"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
# -ffp-contract=off and IEEE divide/sqrt: the shading arithmetic must round like the reference's
# expressions (discrete hit / lobe / shadow decisions decide image parity)
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -Wall -Wno-unused-function"
mkdir -p "$HERE/obj"
# the device unit and the scene-specialised shading kernels (yafgpu_shade_variant.hip) compile side by side
"$HIPCC" $FLAGS -c "$HERE/yafgpu_device.hip" -o "$HERE/obj/yafgpu_device.o" ${YAFGPU_EXTRA_FLAGS:-} &
PID_DEV=$!
#   diffuse: shinydiffusemat + light_mat, no recursiveRaytrace (BASELINE configs C2, C3)
"$HIPCC" $FLAGS -DYAFGPU_VARIANT_NAME=diffuse -DYAFGPU_MAT_MASK=0x5u -DYAFGPU_FEAT_RECURSE=0 -DYAFGPU_FEAT_TEXTURE=0 -DYAFGPU_FEAT_ANISO=0 -c "$HERE/yafgpu_shade_variant.hip" -o "$HERE/obj/shade_diffuse.o" ${YAFGPU_EXTRA_FLAGS:-} &
PID_V1=$!
#   glossy: + glossy (as_diffuse), no recursiveRaytrace (C4)
"$HIPCC" $FLAGS -DYAFGPU_VARIANT_NAME=glossy -DYAFGPU_MAT_MASK=0x7u -DYAFGPU_FEAT_RECURSE=0 -DYAFGPU_FEAT_TEXTURE=0 -DYAFGPU_FEAT_ANISO=0 -c "$HERE/yafgpu_shade_variant.hip" -o "$HERE/obj/shade_glossy.o" ${YAFGPU_EXTRA_FLAGS:-} &
PID_V2=$!
#   full: every material type and recursiveRaytrace, no shader nodes / textures (the main unit's kernel has those too)
"$HIPCC" $FLAGS -DYAFGPU_VARIANT_NAME=full -DYAFGPU_MAT_MASK=0x3fu -DYAFGPU_FEAT_RECURSE=1 -DYAFGPU_FEAT_TEXTURE=0 -DYAFGPU_FEAT_ANISO=0 -c "$HERE/yafgpu_shade_variant.hip" -o "$HERE/obj/shade_full.o" ${YAFGPU_EXTRA_FLAGS:-} &
PID_V3=$!
wait $PID_DEV; wait $PID_V1; wait $PID_V2; wait $PID_V3
"$HIPCC" $FLAGS -c "$HERE/kdtree_build.cpp" -o "$HERE/obj/kdtree_build.o"
"$HIPCC" $FLAGS -c "$HERE/kdtree_build_device.hip" -o "$HERE/obj/kdtree_build_device.o"
SRCS_CPP=""
for f in yafaray_c_api yafaray_xml yafaray_image; do
  if [ -f "$HERE/$f.cpp" ]; then "$HIPCC" $FLAGS -c "$HERE/$f.cpp" -o "$HERE/obj/$f.o"; SRCS_CPP="$SRCS_CPP $HERE/obj/$f.o"; fi
done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$OUT" "$HERE/obj/yafgpu_device.o" "$HERE/obj/shade_diffuse.o" "$HERE/obj/shade_glossy.o" "$HERE/obj/shade_full.o" "$HERE/obj/kdtree_build.o" "$HERE/obj/kdtree_build_device.o" $SRCS_CPP -lpthread -lz
echo "built $OUT"
