#!/bin/bash
# Builds toypathtracer_amd/lib/libtoypathtracer_hip.so for gfx950 (MI355X).  hipcc cross-compiles
# without a GPU.  -ffp-contract=off (host AND device) is part of the numerical contract: no FMA
# contraction anywhere a branch can depend on the result (see tpt_math.h).
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
OUT=${TPT_OUT_DIR:-$HERE/../lib}
mkdir -p "$OUT"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="$TPT_EXTRA_FLAGS --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wall -Wno-unused-function"
$HIPCC $FLAGS -c "$HERE/tpt_kernels.hip" -o "$OUT/tpt_kernels.o"
$HIPCC $FLAGS -x hip -c "$HERE/tpt_host.cpp" -o "$OUT/tpt_host.o"
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT/libtoypathtracer_hip.so" "$OUT/tpt_kernels.o" "$OUT/tpt_host.o"
rm -f "$OUT/tpt_kernels.o" "$OUT/tpt_host.o"
echo "built $OUT/libtoypathtracer_hip.so"
