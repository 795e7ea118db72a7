#!/bin/bash
# An experiment build of the library next to the product one: the named translation unit compiled
# with extra -D flags, linked with the product's other objects.
#   bash tools/build_variant.sh <name> <file.hip> -DCMX_...=.. ...   -> cartographer_amd/lib/var_<name>/libcartographer_mi355x.so
set -eu
NAME=$1; SRC=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
LIB=$ROOT/cartographer_amd/lib
OUT=$LIB/var_$NAME
mkdir -p "$OUT"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off "$@" \
  -c "$ROOT/cartographer_amd/csrc/$SRC" -o "$OUT/$SRC.o"
OBJS=""
for o in "$LIB"/*.hip.o; do
  if [ "$(basename "$o")" = "$SRC.o" ]; then OBJS="$OBJS $OUT/$SRC.o"; else OBJS="$OBJS $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libcartographer_mi355x.so" $OBJS
echo "$OUT/libcartographer_mi355x.so"
