#!/bin/bash
# r03 run 8: the blend chain -- grid-stride resolve (cap sweep) and a high-priority context stream: rocprofv3 kernel stats of the
# steady-state command + rank 0 of N loopback, frame by frame
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
for cfg in "100000 0" "512 0" "256 0" "128 0" "512 1" "100000 1"; do set -- $cfg
  echo "=== TPT_RESOLVE_BLOCKS=$1 TPT_RESOLVE_PRIO=$2"
  export TPT_RESOLVE_BLOCKS=$1 TPT_RESOLVE_PRIO=$2
  cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/res_$1_$2" -o c2 -- python3 "$R/bench.py" --steps 200 --warmup 20 --no-cpu-baseline --no-extras --parity-frames 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  bench %8.1f Mray/s' % d['value'])"
  cd "$R"; grep "Resolve\|TraceQueue" gpurun_out/res_$1_$2/c2_kernel_stats.csv | cut -d, -f1-4,6-7 | cut -c1-200
  TPT_EMU_N=1,4,8 TPT_EMU_FRAMES=300 timeout 300 python tools/shard_loopback.py 2>&1 | grep "^N="
done
