#!/bin/bash
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
summ() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %.4f ms/step  occ %d grid %d lds %d' % (d['value'], d['ms_per_step'], d['config']['blocks_per_cu'], d['config']['grid_blocks'], d['config']['lds_bytes_per_block']))"; }
for fold in 0 1; do for m in 4 6 8 10 12 14 16; do echo "-- fold $fold maxblocks $m"; TPT_MAX_BLOCKS_PER_CU=$m timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --fold $fold 2>&1 | tail -1 | summ; done; done
for c in 64 128 256; do echo "-- chunk $c"; TPT_CHUNK=$c timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | summ; done
