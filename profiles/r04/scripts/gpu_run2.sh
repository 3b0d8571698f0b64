#!/bin/bash
# r04 run 2: the GPU suite on the new layout -- product library with exactly its ABI exported, unit-test hooks in a second
# build, the context's own stream a blocking stream (no torch.zeros monkeypatch, no explicit synchronises after fills), stream
# batching on by default -- then the suite again with stream batching off; the ordering test 20x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
echo "== full GPU suite (stream batching on = default)"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "$F" | tail -8
echo "== ordering test x20"; for i in 1 2 3 4; do timeout 300 python -m pytest tests/test_gpu_api.py -m gpu -x -q -k "default_stream_fill" 2>&1 | grep -v "$F" | tail -1; done
echo "== full GPU suite, TPT_FORCE_STREAM_BATCH=0"; TPT_FORCE_STREAM_BATCH=0 timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "$F" | tail -4
echo "== which .so files were mapped"; python - <<'PY'
import os, sys
sys.path.insert(0, '.')
from toypathtracer_amd import api
api.InitializeTest()
with api.using_hooks():
    pass
print(sorted(set(l.split()[-1] for l in open('/proc/self/maps') if 'toypathtracer' in l)))
PY
