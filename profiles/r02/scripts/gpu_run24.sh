#!/bin/bash
# r02 run 24: PC sampling of the default bench command (rocprofv3 beta): which configuration does gfx950 accept?
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
rocprofv3 -L 2>/dev/null | grep -i -A12 "pc.sampl" | head -40
for cfg in "stochastic cycles 1048576" "stochastic cycles 65536" "host_trap time 10" "host_trap time 100" "host_trap time 1000"; do
  set -- $cfg
  echo "== method $1 unit $2 interval $3"
  timeout 150 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit $2 --pc-sampling-method $1 --pc-sampling-interval $3 --kernel-trace --output-format csv -d "$R/gpurun_out/pcs_$1_$3" -o p -- python "$R/bench.py" --no-cpu-baseline --no-extras --steps 100 --warmup 20 > "$R/gpurun_out/pcs_bench.log" 2>&1
  echo "rc=$?"; grep -i "not supported\|error" "$R/gpurun_out/pcs_bench.log" | head -3
  for f in $(find "$R/gpurun_out/pcs_$1_$3" -name "*pc_sampling*.csv" 2>/dev/null | head -2); do wc -l $f; head -3 $f; done
done
