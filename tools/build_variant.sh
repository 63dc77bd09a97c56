#!/bin/bash
# Build a variant of libtfc with extra -D flags for tfc_conv.hip into matten_amd/libtfc_<tag>.so (experiments).
# usage: tools/build_variant.sh <tag> "<-D flags>"
set -e
tag=$1; flags=$2
cd "$(dirname "$0")/.."
mkdir -p build/var_$tag
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-unused-function $flags -x hip -c matten_amd/csrc/tfc_conv.hip -o build/var_$tag/tfc_conv.o 2>/dev/null
objs=$(ls build/obj/*.o | grep -v tfc_conv.hip.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o matten_amd/libtfc_$tag.so $objs build/var_$tag/tfc_conv.o
echo built matten_amd/libtfc_$tag.so
