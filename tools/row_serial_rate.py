import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
import torch, numpy as np
from toypathtracer_amd import api
sys.path.insert(0, "tests")
import bench
api.InitializeTest()
for per in (32, 16):
    ms, mr = bench.row_serial_batched_rate(api, torch, 1280, 720, per_launch=per, launches=3)
    print("row serial batched %d: %.3f ms/frame %.1f Mray/s grid %d" % (per, ms, mr, api.launch_info()["grid_blocks"]))
ms, mr = bench.row_serial_rate(api, 1280, 720)
print("row serial DrawTest host: %.2f ms %.1f Mray/s" % (ms, mr))
api.ShutdownTest()
