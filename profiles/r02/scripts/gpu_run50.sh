#!/bin/bash
# r02 run 50: VALU instructions per frame, one frame per launch vs four (final binary; 64 workgroups per launch); look-ahead 3
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
pmc() { # bench args, tag
  cd /tmp && TPT_GRID_DIV=8 timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d "$R/gpurun_out/pmc_$2" -o p -- python "$R/bench.py" --no-cpu-baseline --no-extras --overlap 1 $1 > /dev/null 2>&1
  cd "$R"; python - <<PY
import csv, glob, collections
for f in glob.glob('gpurun_out/pmc_$2/**/*counter_collection.csv', recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'TraceQueue' in r['Kernel_Name']: acc[(r['Kernel_Name'][:48], r['Counter_Name'])].append(float(r['Counter_Value']))
    for k, v in sorted(acc.items()): print('%-50s %-22s mean %16.1f  n %d' % (k[0], k[1], sum(v)/len(v), len(v)))
PY
}
echo "== one frame per launch"; pmc "--steps 10 --warmup 2" f1
echo "== four frames per launch (counts are per LAUNCH = 4 frames)"; pmc "--batch 4 --steps 16 --warmup 4 --prime 4" f4
for la in 2 3; do echo "== TPT_HOST_LOOKAHEAD=$la"; TPT_HOST_LOOKAHEAD=$la python - <<'PY'
import json, subprocess, sys
out = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--steps", "50", "--warmup", "10"], capture_output=True, text=True).stdout.strip().splitlines()[-1]
d = json.loads(out)
print("sync caller %.3f ms %.0f Mray/s" % (d["sync_device_caller_ms"], d["sync_device_caller_Mray_s"]), "host %.3f ms" % d["drawtest_host_ms"])
PY
done
