"""ROW_SERIAL seeds (the reference's exact image): frame by frame through DrawTest(host buffer) and batched through
tptDrawDeviceBatch (frames x rows lanes per launch), with what the pipeline looked like for each."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "20")
import torch
from toypathtracer_amd import api
import bench
api.InitializeTest()
w, h = 1280, 720
for label, prep in (("cold", None), ("after a 40-frame per-pixel stream", "stream"), ("frame overlap 16 set explicitly", "ov")):
    if prep == "stream":
        tile = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
        for f in range(40):
            api.UpdateTest(0.0, f, w, h, 2)
            api.draw_device(0.0, f, w, h, tile.data_ptr(), 2)
        api.synchronize()
    if prep == "ov":
        api.set_frame_overlap(16)
    for per in (32,):
        ms, mr = bench.row_serial_batched_rate(api, torch, w, h, per_launch=per, launches=8)
        print("%-36s batched %d: %.3f ms/frame %.1f Mray/s grid %d %s" % (label, per, ms, mr, api.launch_info()["grid_blocks"], api.pipeline_info()), flush=True)
ms, mr = bench.row_serial_rate(api, w, h)
print("row serial DrawTest host: %.2f ms %.1f Mray/s" % (ms, mr))
api.ShutdownTest()
