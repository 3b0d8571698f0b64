#!/bin/bash
# r04 soak: long streams through the path queues (50 000 frames of C2, 200 000 of C1 with 8 frames per launch, 1 500 of C5), each
# TWICE: every run must finish (the queue deadlock of r04_run10 was a once-in-millions-of-pops event) and both runs of a workload
# must end with the same image hash and ray total
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %d steps  image %s  rays/step %.3f' % (d['value'], d['steps'], d.get('image_fnv'), d['rays_per_step']))"; }
for rep in 1 2; do
  echo "== c2 x 50000 ($rep)"; timeout 200 python bench.py --no-cpu-baseline --no-extras --parity-frames 1 --steps 50000 --warmup 20 2>/dev/null | grep '^{"metric"' | tail -1 | line
  echo "== c1 x 200000 ($rep)"; timeout 200 python bench.py --no-cpu-baseline --no-extras --parity-frames 1 --workload c1 --steps 200000 --warmup 20 2>/dev/null | grep '^{"metric"' | tail -1 | line
  echo "== c5 x 1500 ($rep)"; timeout 200 python bench.py --no-cpu-baseline --no-extras --parity-frames 1 --workload c5 --steps 1500 --warmup 20 2>/dev/null | grep '^{"metric"' | tail -1 | line
done
