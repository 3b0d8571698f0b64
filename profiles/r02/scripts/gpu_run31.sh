#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
timeout 900 python -m pytest tests/test_gpu_api.py -m gpu -q -x -k "shards or sharded or two_ranks" 2>&1 | tail -25 | cut -c1-300
