#!/bin/bash
# Build a variant of libtfc with extra -D flags for one translation unit (default tfc_conv.hip) into
# matten_amd/libtfc_<tag>.so (experiments; load it with TFC_LIB_PATH).
# usage: tools/build_variant.sh <tag> "<-D flags>" [tfc_mlp.hip]
set -e
tag=$1; flags=$2; src=${3:-tfc_conv.hip}
cd "$(dirname "$0")/.."
mkdir -p build/var_$tag
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-unused-function $flags -x hip -c matten_amd/csrc/$src -o build/var_$tag/$src.o 2>/dev/null
objs=$(ls build/obj/*.o | grep -v "/$src.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o matten_amd/libtfc_$tag.so $objs build/var_$tag/$src.o
echo built matten_amd/libtfc_$tag.so
