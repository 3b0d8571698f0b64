#!/bin/bash
# Round 6, GPU call 37: soak of the grouped traversal across scenes (tools/grouped_soak.py: fields of 1000 / 4096 / 20 000 spheres, two
# clouds of spheres spread through a volume) -- the shipped build (half-line bounds) against the line-only build, the flat filter and the
# lane-refill kernel: 48 frames each, ray counts and image hashes must be the same in every column.
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
mkdir -p gpurun_out/soak
timeout 900 python tools/grouped_soak.py 48 0 2>&1 | grep -v "$F" > gpurun_out/soak/shipped.txt
TPT_LIB_DIR=$PWD/tools/_variants/r6_nohalf timeout 900 python tools/grouped_soak.py 48 0 2>&1 | grep -v "$F" > gpurun_out/soak/line_only.txt
timeout 1500 python tools/grouped_soak.py 48 3 2>&1 | grep -v "$F" > gpurun_out/soak/flat_filter.txt
timeout 1500 python tools/grouped_soak.py 48 2 2>&1 | grep -v "$F" > gpurun_out/soak/no_groups.txt
cat gpurun_out/soak/shipped.txt
for v in line_only flat_filter no_groups; do echo "== shipped vs $v: $(diff <(cut -c1-26,40- gpurun_out/soak/shipped.txt) <(cut -c1-26,40- gpurun_out/soak/$v.txt) | grep -c '^[<>]') differing lines of $(wc -l < gpurun_out/soak/$v.txt)"; done
