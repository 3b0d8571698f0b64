#!/bin/bash
# Experimental builds of the same library with different compile-time constants: tools/build_variant.sh NAME "-DTPT_BLOCK=64 ..."
HERE=$(cd "$(dirname "$0")" && pwd)
TPT_SKIP_HOOKS=1 TPT_EXTRA_FLAGS="-DTPT_TEST_HOOKS $2" TPT_OUT_DIR="$HERE/_variants/$1" bash "$HERE/../toypathtracer_amd/csrc/build.sh"
