#!/bin/bash
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.4f ms/step  occ %d grid %d lds %d' % (d['value'], d['ms_per_step'], d['config']['blocks_per_cu'], d['config']['grid_blocks'], d['config']['lds_bytes_per_block']))"; }
for ov in 1 2 3 4; do echo "-- overlap $ov"; timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --overlap $ov 2>&1 | tail -1 | summ; done
for ov in 1 2; do echo "-- fold1 overlap $ov"; timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --fold 1 --overlap $ov 2>&1 | tail -1 | summ; done
for b in b64 b128; do for ov in 1 2 3; do echo "-- $b overlap $ov"; TPT_LIB=tools/_variants/$b/libtoypathtracer_hip.so timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --overlap $ov 2>&1 | tail -1 | summ; done; done
echo "-- b64 fold1 overlap 2"; TPT_LIB=tools/_variants/b64/libtoypathtracer_hip.so timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --overlap 2 --fold 1 2>&1 | tail -1 | summ
echo "-- c3 overlap 2"; timeout 300 python bench.py --workload c3 --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | summ
