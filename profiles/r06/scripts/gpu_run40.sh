#!/bin/bash
# Round 6, GPU call 40: longer soaks of the closing tree: 1000 sets x 3 frames of the 4096-sphere scene in a time-sliced child process
# (32 queues + 16 streams) against the oracle's hashes; the same 900 renders in a plain process; 5 grouped scenes x 200 frames, twice
# (the two passes must print the same hashes).
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
echo "== time-sliced child, 1000 sets of 3 frames"; timeout 1200 python tests/c5_timeslice_child.py 1000 0 2>&1 | grep -v "$F" | tail -1 | cut -c1-260
echo "== plain process, 300 sets"; GPU_MAX_HW_QUEUES=20 timeout 900 python tools/c5_timeslice.py 300 3 2>&1 | grep -v "$F" | grep "c5_timeslice:" | cut -c1-220
mkdir -p gpurun_out/soak
timeout 1500 python tools/grouped_soak.py 200 0 2>&1 | grep -v "$F" > gpurun_out/soak/long_a.txt
timeout 1500 python tools/grouped_soak.py 200 0 2>&1 | grep -v "$F" > gpurun_out/soak/long_b.txt
echo "== 5 scenes x 200 frames, two passes: $(diff gpurun_out/soak/long_a.txt gpurun_out/soak/long_b.txt | grep -c '^[<>]') differing lines of $(wc -l < gpurun_out/soak/long_a.txt)"
tail -3 gpurun_out/soak/long_a.txt
