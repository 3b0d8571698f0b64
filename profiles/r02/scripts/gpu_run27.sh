#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
timeout 900 python -X faulthandler -m pytest tests/test_gpu_api.py -m gpu -q -x -k "drawtest or cxx_host" 2>&1 | head -60
