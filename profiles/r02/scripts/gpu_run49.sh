#!/bin/bash
# r02 run 49: look-ahead for synchronous tptDrawDevice callers: parity, rate; regression check of the streaming path
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_api.py tests/test_gpu_parity.py -q -x -m gpu -k "synchronous or lookahead or torch_tile or overlap_is_bit or pipelined or batch or flags_animate or alpha" 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -6
python - <<'PY'
import json, subprocess, sys
for args in (["--steps", "200", "--warmup", "20"], ["--steps", "20", "--warmup", "5"], ["--workload", "c1", "--steps", "400", "--warmup", "40"]):
    out = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline"] + args, capture_output=True, text=True).stdout.strip().splitlines()[-1]
    d = json.loads(out)
    print(args, "%.1f Mray/s" % d["value"], "sync caller %.3f ms %.0f Mray/s" % (d["sync_device_caller_ms"], d["sync_device_caller_Mray_s"]), "host %.3f" % d["drawtest_host_ms"])
PY
TPT_HOST_LOOKAHEAD=0 python - <<'PY'
import json, subprocess, sys
out = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--steps", "50", "--warmup", "10"], capture_output=True, text=True).stdout.strip().splitlines()[-1]
d = json.loads(out)
print("look-ahead off:", "sync caller %.3f ms %.0f Mray/s" % (d["sync_device_caller_ms"], d["sync_device_caller_Mray_s"]), "host %.3f" % d["drawtest_host_ms"])
PY
