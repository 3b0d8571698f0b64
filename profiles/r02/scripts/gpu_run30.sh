#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
echo "== probe across re-initialisations"
timeout 120 python - <<'PY'
from toypathtracer_amd import api
for k in range(5):
    api.InitializeTest(); print(k, api.pipeline_info(), flush=True); api.ShutdownTest()
PY
for i in 1 2; do echo "== full gpu suite $i"; timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -12 | cut -c1-300; done
