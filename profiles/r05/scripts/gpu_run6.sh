#!/bin/bash
# Round 5, GPU call 6: is the CHECKER deterministic on the GPU box's host?  (no GPU work: oracle/tpt_oracle.c only)
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
lscpu | grep "Model name\|^CPU(s)\|Thread\|Socket\|Hypervisor\|Virtualization" 
uname -r
timeout 500 python tools/oracle_determinism.py 400 0 32 1 2>&1 | tail -30
