R=$GRAFT_REPO_ROOT; cd $R
run() { echo "$1: $(TPT_LIB=$2 timeout 200 python bench.py --no-cpu-baseline --no-extras --parity-frames 0 --steps 200 --warmup 20 $3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.4f ms/step grid %d' % (d['value'], d['ms_per_step'], d['config']['grid_blocks']))")"; }
run base "" ""
for v in "$@"; do run $v $R/tools/_variants/$v/libtoypathtracer_hip.so ""; run "$v c3" $R/tools/_variants/$v/libtoypathtracer_hip.so "--workload c3 --steps 20 --warmup 10"; done
run base "" ""
