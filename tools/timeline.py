"""Render N small frames under rocprofv3 --kernel-trace and analyse the kernel timeline (concurrency, gaps).
usage: rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python tools/timeline.py W H SPP OVERLAP FRAMES
       python tools/timeline.py --analyse DIR"""
import os, sys, glob, csv
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if sys.argv[1] == "--analyse":
    for f in glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True):
        rows = [r for r in csv.DictReader(open(f))]
        tr = sorted([(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "Trace" in r["Kernel_Name"]])
        rs = sorted([(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "Resolve" in r["Kernel_Name"]])
        tr, rs = tr[len(tr) // 4:], rs[len(rs) // 4:]  # skip the warm-up quarter
        dur = [e - s for s, e in tr]
        span = tr[-1][1] - tr[0][0]
        busy = sum(dur)
        print("%s: %d trace kernels, mean duration %.1f us, period %.1f us, mean concurrency %.2f" % (
            os.path.basename(os.path.dirname(f)), len(tr), sum(dur) / len(dur) / 1e3, span / len(tr) / 1e3, busy / span))
        gaps = [rs[i + 1][0] - rs[i][1] for i in range(len(rs) - 1)]
        rd = [e - s for s, e in rs]
        print("   resolve: mean duration %.1f us, mean gap between consecutive resolves %.1f us" % (sum(rd) / len(rd) / 1e3, sum(gaps) / len(gaps) / 1e3))
        # delay from the end of trace(f) to the start of resolve(f) (kernels are in frame order in both lists)
        tr_by_end = sorted(tr, key=lambda x: x[0])
        n = min(len(tr_by_end), len(rs))
        d = [rs[i][0] - tr_by_end[i][1] for i in range(n)]
        print("   end of trace(f) -> start of resolve(f): mean %.1f us, min %.1f us" % (sum(d) / n / 1e3, min(d) / 1e3))
    sys.exit(0)
import torch
from toypathtracer_amd import api
w, h, spp, ov, n = [int(v) for v in sys.argv[1:6]]
api.InitializeTest()
api.set_samples_per_pixel(spp)
api.set_frame_overlap(ov)
tile = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
for f in range(n):
    api.UpdateTest(0.0, f, w, h, 2); api.draw_device(0.0, f, w, h, tile.data_ptr(), 2)
api.synchronize()
api.ShutdownTest()
