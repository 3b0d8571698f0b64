#!/bin/bash
# Builds toypathtracer_amd/lib/libtoypathtracer_hip.so for gfx950 (MI355X).  hipcc cross-compiles
# without a GPU.  -ffp-contract=off (host AND device) is part of the numerical contract: no FMA
# contraction anywhere a branch can depend on the result (see tpt_math.h).
# -fvisibility=hidden: the library exports exactly what include/tpt_hip.h and include/tpt_test_api.h declare.
# A second build of the same sources with -DTPT_TEST_HOOKS (libtoypathtracer_hip_hooks.so) adds the unit-test / profiling
# entry points of include/tpt_test_hooks.h; only the GPU test suite and tools/ load it.  TPT_SKIP_HOOKS=1 skips it.
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
OUT=${TPT_OUT_DIR:-$HERE/../lib}
mkdir -p "$OUT"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="$TPT_EXTRA_FLAGS --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wall -Wno-unused-function"
build_one() { # suffix, extra flags
  $HIPCC $FLAGS $2 -c "$HERE/tpt_kernels.hip" -o "$OUT/tpt_kernels$1.o"
  $HIPCC $FLAGS $2 -x hip -c "$HERE/tpt_host.cpp" -o "$OUT/tpt_host$1.o"
  $HIPCC --offload-arch=gfx950 -shared -fPIC -Wl,--version-script="$HERE/exports.map" -o "$OUT/libtoypathtracer_hip$1.so" "$OUT/tpt_kernels$1.o" "$OUT/tpt_host$1.o"
  rm -f "$OUT/tpt_kernels$1.o" "$OUT/tpt_host$1.o"
  echo "built $OUT/libtoypathtracer_hip$1.so"
}
build_one "" "" &
P1=$!
if [ -z "$TPT_SKIP_HOOKS" ]; then build_one "_hooks" "-DTPT_TEST_HOOKS" & P2=$!; fi
wait $P1
if [ -n "$P2" ]; then wait $P2; fi
