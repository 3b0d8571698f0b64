#!/bin/bash
# Profiling build of the SAME library with block-entry counters compiled in (-DTPT_STATS). Not shipped.
HERE=$(cd "$(dirname "$0")" && pwd)
TPT_EXTRA_FLAGS="-DTPT_STATS" TPT_OUT_DIR="$HERE/_stats" bash "$HERE/../toypathtracer_amd/csrc/build.sh"
