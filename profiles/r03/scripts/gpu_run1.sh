#!/bin/bash
# r03 run 1: baseline of the round-2 tree + what the box offers (partition modes, PC sampling)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
O=$R/gpurun_out/r03_run1; mkdir -p $O
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.4f ms/step  launch %.3f ms grid %d' % (d['value'], d['ms_per_step'], d['trace_launch_ms_avg'], d['config']['grid_blocks']))"; }
echo "== box"; rocm-smi --showcomputepartition --showmemorypartition 2>&1 | head -20; rocminfo | grep -c "gfx950"; nproc
echo "== baseline bench"
for args in "--steps 20 --warmup 5" "--steps 200 --warmup 20"; do echo "-- $args"; timeout 300 python bench.py --no-cpu-baseline --no-extras $args 2>&1 | tail -1 | summ; done
echo "== pc sampling available?"
cd /tmp && timeout 120 rocprofv3 -L 2>&1 | grep -i -A3 "pc.samp" | head -20
timeout 120 rocprofv3 --help 2>&1 | grep -i "pc-samp" | head
echo "== try host-trap pc sampling"
cd /tmp && ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1 timeout 300 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method host_trap --pc-sampling-unit time --pc-sampling-interval 100 --output-format csv -d $O/pcs -o pcs -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras 2>&1 | tail -5
ls -la $O/pcs 2>/dev/null | head; find $O/pcs -name "*.csv" | head
echo "== try stochastic"
cd /tmp && ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1 timeout 300 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method stochastic --pc-sampling-unit cycles --pc-sampling-interval 65536 --output-format csv -d $O/pcs2 -o pcs -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras 2>&1 | tail -5
find $O/pcs2 -name "*.csv" | head
cd $R
echo "== SQ counters (baseline)"; bash tools/gpu_pmc.sh "--no-extras" base 2>&1 | tail -30
