#!/bin/bash
# TEST SUPPORT: the host side of the C ABI with a stub in place of the device unit, under AddressSanitizer + UBSan (CPU only)
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
SRC="$HERE/../../libyafaray_amd/csrc"
g++ -std=c++17 -fsanitize=address,undefined -fno-sanitize-recover=undefined -g -O1 -shared -fPIC -w -I"$SRC" -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ \
    "$SRC/yafaray_c_api.cpp" "$SRC/yafaray_xml.cpp" "$SRC/yafaray_image.cpp" "$SRC/yafaray_reduce.cpp" "$SRC/kdtree_build.cpp" "$HERE/stub_device.cpp" \
    -o "$HERE/libyafaray_host_asan.so" -lpthread -L/opt/rocm/lib -lamdhip64 -lz -ldl -Wl,-rpath,/opt/rocm/lib
echo "built $HERE/libyafaray_host_asan.so"
