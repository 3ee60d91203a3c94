#!/bin/bash
# A/B build of libagl.so with extra -D flags on one source: tools/ab_build.sh NAME SRC.hip "-DX=1 ..."  -> agl/ab/libagl_NAME.so
# (select at run time with AGL_LIBRARY=<path>; the .so files are git-ignored but travel to the GPU box)
set -e
cd "$(dirname "$0")/../attribute-guided-image-generation-from-layout_amd/csrc"
name=$1; src=$2; shift 2
mkdir -p ../agl/ab /tmp/ab_$name
make -s all
hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wno-unused-result -Wno-pass-failed "$@" -c $src -o /tmp/ab_$name/${src%.hip}.o
objs=""
for o in api.o conv.o pconv.o few.o norm.o pointwise.o sn.o loss.o layout.o; do
  if [ "$o" = "${src%.hip}.o" ]; then objs="$objs /tmp/ab_$name/$o"; else objs="$objs $o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o ../agl/ab/libagl_$name.so $objs
echo built agl/ab/libagl_$name.so
