"""Kernel timeline of one pipelined burst under rocprofv3 --kernel-trace.
usage: rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python tools/burst_trace.py [FRAMES]
       python tools/burst_trace.py --analyse DIR"""
import os, sys, glob, csv
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "--analyse":
    for f in glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True):
        rows = [r for r in csv.DictReader(open(f))]
        ks = sorted([(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows], key=lambda x: x[0])
        # the last burst: everything after the largest gap between consecutive trace starts
        tr = [k for k in ks if "TraceQueue" in k[2]]
        gaps = [(tr[i + 1][0] - tr[i][0], i) for i in range(len(tr) - 1)]
        cut = tr[max(gaps)[1] + 1][0] if gaps else tr[0][0]
        t0 = cut
        print("burst (times in us from the first trace start):")
        for s, e, name in ks:
            if s >= t0 - 200000:
                short = "trace" if "TraceQueue" in name else "resolve" if "Resolve" in name else "publish" if "Publish" in name else name[:20]
                if short in ("trace", "resolve", "publish"):
                    print("  %-8s start %9.1f  end %9.1f  dur %8.1f" % (short, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
    sys.exit(0)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "20")
import torch
from toypathtracer_amd import api
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
api.InitializeTest()
w, h = 1280, 720
tile = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
for rep in range(2):
    for f in range(n):
        api.UpdateTest(0.0, f, w, h, 2); api.draw_device(0.0, f, w, h, tile.data_ptr(), 2)
    api.synchronize()
    torch.cuda.synchronize()
    import time; time.sleep(0.3)
api.ShutdownTest()
