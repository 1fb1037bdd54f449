#!/bin/bash
# build a variant of the library with extra compile flags: bash scripts/experiments/build_variant.sh <name> "<flags>" [files...]
# -> scripts/experiments/libdrn_hip_<name>.so (objects of the files named are rebuilt with the flags, the rest are the shipped objects)
set -e
NAME=$1; FLAGS=$2; shift 2
SRC=/root/repo/drn_amd/csrc
TMP=/tmp/variant_$NAME; rm -rf $TMP; mkdir -p $TMP
make -C $SRC -s >/dev/null
cp $SRC/*.o $TMP/
for f in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-unused-function $FLAGS -I$SRC -c $SRC/$f.hip -o $TMP/$f.o
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $TMP/*.o -o /root/repo/scripts/experiments/libdrn_hip_$NAME.so
ls -la /root/repo/scripts/experiments/libdrn_hip_$NAME.so
