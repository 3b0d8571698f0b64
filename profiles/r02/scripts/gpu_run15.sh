#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for h in 3; do
cd /tmp && TPT_TABLE=device TPT_HELP=$h timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$R/gpurun_out/burst5_h$h" -o t -- python "$R/tools/burst_trace.py" 20 > /dev/null 2>&1
cd "$R"; echo "== TPT_TABLE=device TPT_HELP=$h"; python tools/burst_trace.py --analyse gpurun_out/burst5_h$h | head -70
done
