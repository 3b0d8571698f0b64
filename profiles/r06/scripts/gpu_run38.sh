#!/bin/bash
# Round 6, GPU call 38: call 37's two cloud scenes were not grouped (radii over a decade: too many members beyond 64 radii from their
# group's centre); again with radii over a factor of 4 -- grouped -- shipped build against line-only, flat filter and no groups.  Then: how
# many of the dealt candidates lie wholly beyond the hit the ray already holds (profiling build).
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
mkdir -p gpurun_out/soak
timeout 900 python tools/grouped_soak.py 48 0 cloud 2>&1 | grep -v "$F" > gpurun_out/soak/cloud_shipped.txt
TPT_LIB_DIR=$PWD/tools/_variants/r6_nohalf timeout 900 python tools/grouped_soak.py 48 0 cloud 2>&1 | grep -v "$F" > gpurun_out/soak/cloud_line_only.txt
timeout 1500 python tools/grouped_soak.py 48 3 cloud 2>&1 | grep -v "$F" > gpurun_out/soak/cloud_flat_filter.txt
timeout 1500 python tools/grouped_soak.py 48 2 cloud 2>&1 | grep -v "$F" > gpurun_out/soak/cloud_no_groups.txt
cat gpurun_out/soak/cloud_shipped.txt
for v in line_only flat_filter no_groups; do echo "== shipped vs $v: $(diff <(cut -c1-26,40- gpurun_out/soak/cloud_shipped.txt) <(cut -c1-26,40- gpurun_out/soak/cloud_$v.txt) | grep -c '^[<>]') differing lines of $(wc -l < gpurun_out/soak/cloud_$v.txt)"; done
echo "== beyond the hit already held (stats2 build; the line printed as 'behind the origin' counts BEYOND here)"; N=6 timeout 300 python tools/stats2_c5.py 2>&1 | grep -v "$F" | tail -3
