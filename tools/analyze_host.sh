#!/bin/bash
# Clang static analyzer over the HOST side of every HIP translation unit of the product
# (leaks, null dereferences, uninitialised reads in the orchestration code around the kernels).
# Usage: bash tools/analyze_host.sh   -- prints the analyzer's warnings, if any.
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
CLANG=${CLANG:-/opt/rocm/lib/llvm/bin/clang++}
total=0
for src in "$REPO"/cartographer_amd/csrc/*.hip; do
  out=$("$CLANG" --analyze -x hip --cuda-host-only --offload-arch=gfx950 -std=c++17 \
        -I/opt/rocm/include -I"$REPO/include" -I"$REPO/cartographer_amd/csrc" \
        -Xclang -analyzer-output=text "$src" -o /dev/null 2>&1 | grep "warning:" || true)
  n=$(printf "%s" "$out" | grep -c "warning:" || true)
  echo "$(basename "$src"): $n warning(s)"
  [ -n "$out" ] && echo "$out"
  total=$((total + n))
done
echo "total: $total"
