# usage: tools/qsweep.sh -- path-queue kernel: compile-time variants (tools/_variants/*) x grid divisor
R=$GRAFT_REPO_ROOT
run() { echo "$1 overlap $2 griddiv $3: $(TPT_LIB=$4 GPU_MAX_HW_QUEUES=24 TPT_GRID_DIV=$3 timeout 100 python bench.py --no-cpu-baseline --persistent 3 --overlap $2 2>&1 | tail -1 | cut -c1-60)"; }
for d in 4 6 8 12 16; do run base 16 $d ""; done
for v in fuse32 fuse1 fuse58 p2048w16 p512w4 p2048w8; do run $v 16 8 $R/tools/_variants/$v/libtoypathtracer_hip.so; done
