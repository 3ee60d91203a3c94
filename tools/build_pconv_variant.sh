#!/bin/bash
# A/B variant of libagl.so with extra -D flags for pconv.hip: tools/build_pconv_variant.sh NAME "-DAGL_X=1 ..."
# -> attribute-guided-image-generation-from-layout_amd/agl/variants/libagl_NAME.so  (use with AGL_LIBRARY=...)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/attribute-guided-image-generation-from-layout_amd/csrc
OUT=$ROOT/attribute-guided-image-generation-from-layout_amd/agl/variants
mkdir -p $OUT /tmp/agl_variant_$1
hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wno-unused-result -Wno-pass-failed $2 -c $SRC/pconv.hip -o /tmp/agl_variant_$1/pconv.o 2>/dev/null
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libagl_$1.so $SRC/api.o $SRC/conv.o /tmp/agl_variant_$1/pconv.o $SRC/few.o $SRC/norm.o $SRC/pointwise.o $SRC/sn.o $SRC/loss.o $SRC/layout.o
echo built $OUT/libagl_$1.so
