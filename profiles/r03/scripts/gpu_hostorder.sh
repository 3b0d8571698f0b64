#!/bin/bash
# r03: host-pointer DrawTest: trace-ahead launches before the copies are enqueued (TPT_HOST_TRACE_FIRST=1) or beside the DMA; in the tool and as bench.py measures it
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for tf in 0 1; do
  echo "== TPT_HOST_TRACE_FIRST=$tf"
  TPT_HOST_TRACE_FIRST=$tf FILLS= AHEADS=2 timeout 60 python tools/host_drawtest_rate.py 2>&1 | grep -v amdgpu.ids | tail -6
done
