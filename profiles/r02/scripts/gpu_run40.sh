#!/bin/bash
# r02 run 40: host pacing (host waits for the slot's resolve before enqueueing the next trace on it) for small tiles
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for env in "TPT_HOST_PACE=0" "TPT_HOST_PACE=1" "TPT_HOST_PACE=1 TPT_SHARD_CAP=12" "TPT_HOST_PACE=1 TPT_SHARD_CAP=16" "TPT_HOST_PACE=0 TPT_SHARD_CAP=16"; do
  echo "== $env"; env $env TPT_EMU_N=1,4,8 timeout 200 python tools/shard_loopback.py 2>&1 | grep "^N="
done
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.4f ms/step  launch %.3f ms grid %d' % (d['value'], d['ms_per_step'], d['trace_launch_ms_avg'], d['config']['grid_blocks']))"; }
for env in "TPT_HOST_PACE=0" "TPT_HOST_PACE=1"; do for args in "--steps 200 --warmup 20" "--workload c1 --steps 400 --warmup 40"; do echo "-- $env $args"; env $env timeout 300 python bench.py --no-cpu-baseline --no-extras $args 2>/dev/null | tail -1 | summ; done; done
