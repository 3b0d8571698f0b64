#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
for i in 1 2; do echo "== full gpu suite $i"; timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -12 | cut -c1-300; done
