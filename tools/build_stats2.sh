#!/bin/bash
# Light profiling build (-DTPT_STATS=2): a handful of event counters only (claims, frame transitions, idle polls, wave
# lifetimes), so that it runs at full speed.  tools/_stats2/
HERE=$(cd "$(dirname "$0")" && pwd)
TPT_SKIP_HOOKS=1 TPT_EXTRA_FLAGS="-DTPT_TEST_HOOKS -DTPT_STATS=2" TPT_OUT_DIR="$HERE/_stats2" bash "$HERE/../toypathtracer_amd/csrc/build.sh"
