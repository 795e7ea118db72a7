#!/bin/bash
# Runs the CPU test suite with every host-side C++ library (the oracle, the host fixture builders /
# synthetic worlds, and -- where /root/reference exists -- the reference translation units of
# oracle/_ref with their wrapper) rebuilt under -fsanitize=address,undefined, in a scratch copy of
# the repo.  ~20 minutes.  Usage: bash tools/sanitize_cpu.sh [scratch-dir]
set -eu
REPO=$(cd "$(dirname "$0")/.." && pwd)
SCRATCH=${1:-/tmp/cmx_sanitize}
rm -rf "$SCRATCH" && mkdir -p "$SCRATCH"
cp -r "$REPO" "$SCRATCH/repo" && cd "$SCRATCH/repo" && rm -rf .git gpurun_out
SAN="-fsanitize=address,undefined -fno-omit-frame-pointer -g -O1"
mkdir -p oracle/_build oracle/_ref cartographer_amd/lib
ORACLE_SRCS=$(grep "^SRCS :=" oracle/Makefile | sed 's/SRCS := //')      # every restatement source
(cd oracle && g++ $SAN -std=c++17 -fPIC -shared -ffp-contract=off -pthread -o _build/liboracle.so $ORACLE_SRCS)
g++ $SAN -std=c++17 -fPIC -shared -ffp-contract=off -pthread -o cartographer_amd/lib/libcmx_synth.so \
    cartographer_amd/csrc/host/probability_grid_builder.cc \
    cartographer_amd/csrc/host/hybrid_grid_builder.cc cartographer_amd/csrc/host/synth.cc \
    cartographer_amd/csrc/host/thread_driver.cc
if [ -d /root/reference/cartographer ]; then
  SRCS=$(make -pn -C oracle ref 2>/dev/null | grep "^REF_SRCS" | head -1 | sed 's/REF_SRCS := //' |
         sed 's#\$(REFERENCE)#/root/reference#g')
  (cd oracle && g++ $SAN -DNDEBUG -ffp-contract=off -std=c++17 -fPIC -shared -pthread \
      -Iref_shims -I/root/reference -o _ref/libref.so ref_wrapper.cc ref_wrapper_rt3d_mt.cc $SRCS)
  CERES_SRCS=$(make -pn -C oracle ref_ceres 2>/dev/null | grep "^REF_CERES_SRCS" | head -1 | sed 's/REF_CERES_SRCS := //' |
         sed 's#\$(REFERENCE)#/root/reference#g')
  (cd oracle && g++ $SAN -DNDEBUG -ffp-contract=off -std=c++17 -fPIC -shared -pthread \
      -Iref_shims -I/root/reference -o _ref/libref_ceres.so ref_ceres_wrapper.cc $CERES_SRCS)
fi
touch oracle/_build/liboracle.so cartographer_amd/lib/libcmx_synth.so oracle/_ref/libref.so oracle/_ref/libref_ceres.so 2>/dev/null || true
export LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)"
export ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1
# (the ABI / adapter / bench-contract tests load the HIP library or spawn subprocesses: left out)
python -m pytest tests -q -m "not gpu" -p no:cacheprovider --deselect tests/test_bench_contract.py \
    --deselect tests/test_adapter.py --deselect tests/test_abi.py > "$SCRATCH/log.txt" 2>&1 || true
tail -n 2 "$SCRATCH/log.txt"
echo "sanitizer reports: $(grep -c 'runtime error\|AddressSanitizer' "$SCRATCH/log.txt" || true)"
