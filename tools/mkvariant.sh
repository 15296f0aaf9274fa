#!/bin/bash
# Development build of the library that differs from lib/ in ONE translation unit:
#   tools/mkvariant.sh <name> <file.hip> [-DFLAG ...]   ->   <pkg>/lib_v<name>/libvfi_hip.so
# (the other objects are taken from lib/, which must be up to date: make -C <pkg>/csrc)
set -e
NAME=$1; SRC=$2; shift 2
R=$(cd "$(dirname "$0")/.." && pwd)
PKG=$R/video-frame-interpolation-based-on-deformable-kernel-region_amd
OUT=$PKG/lib_v$NAME
mkdir -p $OUT
STEM=${SRC%.hip}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -Wall -Wno-unused-function -Wno-pass-failed "$@" \
    -c $PKG/csrc/$SRC -o $OUT/$STEM.o
OBJS=""
for o in $PKG/lib/*.o; do
  b=$(basename $o)
  if [ "$b" = "$STEM.o" ]; then OBJS="$OBJS $OUT/$STEM.o"; else OBJS="$OBJS $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $OUT/libvfi_hip.so
echo "$OUT/libvfi_hip.so"
