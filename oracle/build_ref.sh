#!/bin/bash
# oracle/build_ref.sh -- TEST INFRASTRUCTURE ONLY.
#
# Builds the *pristine reference* CPU path (Cpp/Source/Test.cpp + Maths.cpp + vendored enkiTS)
# from the sources where they lie under /root/reference into oracle/_ref/ (git-ignored, but it
# travels to the GPU box).  Nothing from /root/reference is copied into the repo: Test.cpp is
# streamed through sed to the compiler's stdin only to turn the compile-time macro
# DO_SAMPLES_PER_PIXEL (Config.h:22) into the runtime variable g_tpt_ref_spp (ref_shim.cpp); every
# other line is compiled as is.  With g_tpt_ref_spp == 4 the build is the reference as shipped.
#
#   libtpt_ref.so        SIMD path, oracle flags : -O2 -msse4.1 -ffp-contract=off  (bit-stable)
#   libtpt_ref_scalar.so SCALAR path (the parity target), same flags + -D__EMSCRIPTEN__ (Config.h:9-13)
#   libtpt_ref_fast.so  "as shipped" : -O3 -ffast-math -mavx2 -mfma (≈ -march=native, portable to the GPU box host; mirrors /fp:fast, GCC_FAST_MATH)
#
# -include string.h : Test.cpp:379 uses memcpy without including <string.h>.
# -msse4.1          : MathSimd.h:17 includes <smmintrin.h>.
set -e
REF=${TPT_REFERENCE_DIR:-/root/reference}
S=$REF/Cpp/Source
HERE=$(cd "$(dirname "$0")" && pwd)
OUT=$HERE/_ref
if [ ! -f "$S/Test.cpp" ]; then
    echo "build_ref.sh: reference sources not found under $S (expected on the GPU box); keeping prebuilt files" >&2
    exit 0
fi
mkdir -p "$OUT"
TMP=$(mktemp -d)
trap 'rm -rf "$TMP"' EXIT
# REDEF="MACRO value;MACRO value": further Config.h switches re-defined the same way (they are plain #defines in Config.h,
# used by Test.cpp only, so -D cannot reach them): variants of the scalar path for pinning the oracle's run-time switches.
build() {  # name, flags...
    local name=$1; shift
    local common="-std=c++11 -fPIC -DNDEBUG -Wno-attributes -include string.h -I$S"
    local extra=""
    if [ -n "$REDEF" ]; then
        local IFS=';'
        for kv in $REDEF; do extra="$extra\\n#undef ${kv%% *}\\n#define $kv"; done
        unset IFS
    fi
    # PERPIXEL=1: the reference's own GPU seed (ComputeShader.hlsl:380, Shaders.metal:401: one RNG stream per pixel and frame)
    # put in front of the pixel body of TraceRowJob (Test.cpp:281-282), again only in the stream the compiler reads.  This is
    # the seed mode the product runs by default; everything else of the build stays the reference's scalar CPU path.
    local seed=""
    if [ -n "$PERPIXEL" ]; then
        seed='s/for (int x = 0; x < data.screenWidth; ++x)/& if ((state = ((uint32_t)x * 1973 + y * 9277 + (uint32_t)data.frameCount * 26699) | 1), true)/'
        grep -q 'for (int x = 0; x < data.screenWidth; ++x)' "$S/Test.cpp" || { echo "build_ref.sh: Test.cpp:281 not found" >&2; exit 1; }
    fi
    ( cd "$TMP" && sed -e "/#include <atomic>/a #undef DO_SAMPLES_PER_PIXEL\\n#define DO_SAMPLES_PER_PIXEL g_tpt_ref_spp\\nextern int g_tpt_ref_spp;$extra" -e "$seed" "$S/Test.cpp" \
        | g++ $common "$@" -x c++ -c - -o "$TMP/$name.Test.o" )
    g++ $common "$@" -c "$S/Maths.cpp" -o "$TMP/$name.Maths.o"
    g++ $common "$@" -c "$S/enkiTS/TaskScheduler.cpp" -o "$TMP/$name.ts.o"
    g++ $common "$@" -c "$S/enkiTS/TaskScheduler_c.cpp" -o "$TMP/$name.tsc.o"
    g++ $common "$@" -c "$HERE/ref_shim.cpp" -o "$TMP/$name.shim.o"
    g++ -shared -o "$OUT/$name.so" "$TMP/$name.Test.o" "$TMP/$name.Maths.o" "$TMP/$name.ts.o" "$TMP/$name.tsc.o" "$TMP/$name.shim.o" -lpthread
}
build libtpt_ref      -O2 -msse4.1 -ffp-contract=off
# The reference's own SCALAR path (float3 and HitSpheres without SSE): Config.h:9-13 switches SIMD off
# when __EMSCRIPTEN__ is defined (that is how its WebAssembly build runs); __EMSCRIPTEN_PTHREADS__
# keeps the enkiTS threads (Config.h:15-19).  No source is touched.  This is "the reference CPU
# scalar path" BASELINE.json's north_star names as the parity target.
build libtpt_ref_scalar -O2 -ffp-contract=off -D__EMSCRIPTEN__ -D__EMSCRIPTEN_PTHREADS__
build libtpt_ref_fast -O3 -msse4.1 -mavx2 -mfma -ffast-math
# the reference's other compile-time switches (Config.h:23-25), scalar path: what the oracle's run-time switches are pinned to
REDEF="DO_LIGHT_SAMPLING 0" build libtpt_ref_nols -O2 -ffp-contract=off -D__EMSCRIPTEN__ -D__EMSCRIPTEN_PTHREADS__
REDEF="DO_MITSUBA_COMPARE 1" build libtpt_ref_mitsuba -O2 -ffp-contract=off -D__EMSCRIPTEN__ -D__EMSCRIPTEN_PTHREADS__
REDEF="DO_ANIMATE_SMOOTHING 0.5f" build libtpt_ref_smooth05 -O2 -ffp-contract=off -D__EMSCRIPTEN__ -D__EMSCRIPTEN_PTHREADS__
# the scalar path with the reference's GPU seed formula: pins the product's default seed mode to reference-compiled code
PERPIXEL=1 build libtpt_ref_perpixel -O2 -ffp-contract=off -D__EMSCRIPTEN__ -D__EMSCRIPTEN_PTHREADS__
echo "built $OUT/libtpt_ref.so $OUT/libtpt_ref_scalar.so $OUT/libtpt_ref_fast.so + Config.h variants (nols, mitsuba, smooth05) + libtpt_ref_perpixel.so"
