#!/bin/bash
# r04 run 1: the divide proof (tools/exhaustive/exhaustive_div), the CU-mask bit -> CU map, this box's baseline, and the
# first sweep of CU-masked trace streams (TPT_RESERVE_CUS) against the blend chain / host-pointer DrawTest
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
echo "== exhaustive divide"; timeout 400 tools/exhaustive/exhaustive_div 2>&1 | tail -6
echo "== CU mask probe"; timeout 120 tools/probes/cumask_probe 2>&1 | tail -20
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.4f ms/step  launch %.3f ms grid %d parity %s host %s sync %s' % (d['value'], d['ms_per_step'], d['trace_launch_ms_avg'], d['config']['grid_blocks'], d.get('parity_ok'), d.get('drawtest_host_ms'), d.get('sync_device_caller_ms')))"; }
for rc in "0 0" "4 0" "8 0" "8 1" "16 0" "16 1" "32 1"; do
  set -- $rc
  echo "== TPT_RESERVE_CUS=$1 mode $2"
  export TPT_RESERVE_CUS=$1 TPT_RESERVE_MODE=$2
  echo "-- driver cmd"; timeout 300 python bench.py --no-cpu-baseline --extras host,sync --steps 20 --warmup 5 2>&1 | tail -1 | summ
  echo "-- steady"; timeout 300 python bench.py --no-cpu-baseline --no-extras --parity-frames 0 --steps 200 --warmup 20 2>&1 | tail -1 | summ
done
unset TPT_RESERVE_CUS TPT_RESERVE_MODE
for rc in 0 8; do
  echo "== kernel stats, steady state, TPT_RESERVE_CUS=$rc"
  (cd /tmp && TPT_RESERVE_CUS=$rc TPT_RESERVE_MODE=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04_prof_rc$rc -o p -- python $R/bench.py --no-cpu-baseline --no-extras --parity-frames 0 --steps 200 --warmup 20 > /dev/null 2>&1)
  f=$(find gpurun_out/r04_prof_rc$rc -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cut -d, -f1-7 "$f" | head -8
done
echo "== loopback N=8 frame by frame, reserve 0 / 8"
for rc in 0 8; do TPT_RESERVE_CUS=$rc TPT_RESERVE_MODE=1 TPT_EMU_BATCH=1 TPT_EMU_N=1,8 TPT_EMU_FRAMES=320 timeout 300 python tools/shard_loopback.py 2>&1 | grep "^N="; done
