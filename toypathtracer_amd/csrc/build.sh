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
HOST_UNITS="tpt_host tpt_host_pipeline tpt_host_draw tpt_host_shard tpt_host_hooks" # (tpt_context.h says what each holds)
build_one() { # suffix, extra flags
  local objs="$OUT/obj$1.tpt_kernels.o" pids=""
  $HIPCC $FLAGS $2 -c "$HERE/tpt_kernels.hip" -o "$OUT/obj$1.tpt_kernels.o" & pids="$!"
  for u in $HOST_UNITS; do
    $HIPCC $FLAGS $2 -x hip -c "$HERE/$u.cpp" -o "$OUT/obj$1.$u.o" & pids="$pids $!"
    objs="$objs $OUT/obj$1.$u.o"
  done
  for p in $pids; do wait $p || exit 1; done
  $HIPCC --offload-arch=gfx950 -shared -fPIC -Wl,--version-script="$HERE/exports.map" -o "$OUT/libtoypathtracer_hip$1.so" $objs
  rm -f $objs
  echo "built $OUT/libtoypathtracer_hip$1.so"
}
build_one "" "" &
P1=$!
if [ -z "$TPT_SKIP_HOOKS" ]; then build_one "_hooks" "-DTPT_TEST_HOOKS" & P2=$!; fi
wait $P1
if [ -n "$P2" ]; then wait $P2; fi
