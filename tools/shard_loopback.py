"""Rank 0 of an N-way sharded run on ONE GPU through the C ABI alone (tptCommInitLoopback: same tile, snapshot ring, events and
assemble kernel as tptCommInit, a device copy instead of ncclGather).  Prints what this GPU sustains as rank 0 and the
aggregate N x that -- the render + exchange pipeline apart from RCCL itself.  TPT_EMU_N=1,2,4,8 TPT_EMU_FRAMES=300 TPT_EMU_EVERY=0|1|k"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from toypathtracer_amd import api

api.InitializeTest()
api.set_stream_batching(os.environ.get("TPT_STREAM_BATCH", "1") == "1")  # several frames per launch behind a frame-by-frame caller's back (the library's default; TPT_STREAM_BATCH=0: off)
w, h = [int(v) for v in os.environ.get("TPT_EMU_SIZE", "1280x720").split("x")]
frames, warm = int(os.environ.get("TPT_EMU_FRAMES", "300")), 40
batch = int(os.environ.get("TPT_EMU_BATCH", "1"))  # frames per launch and per exchange (tptDrawShardedBatch)
frames -= frames % batch; warm -= warm % batch
image = torch.zeros(h, w, 4, dtype=torch.float32, device="cuda")
torch.cuda.synchronize()
base = None
for n in [int(v) for v in os.environ.get("TPT_EMU_N", "1,2,4,8").split(",")]:
    api.comm_init_loopback(n, 8)
    api.set_shard_exchange_interval(int(os.environ.get("TPT_EMU_EVERY", "0")))  # 0: automatic (deferred batches of 2 / 4 / 8 frames for small tiles), 1: frame by frame
    if os.environ.get("TPT_EMU_OV"):
        api.set_frame_overlap(int(os.environ["TPT_EMU_OV"]))
    for f in range(0, warm, batch):
        api.UpdateTest(0.0, f, w, h, 2)
        api.draw_sharded_batch(0.0, f, batch, w, h, image.data_ptr(), 2)
    best = None
    for rep in range(2):  # best of two timed passes (a one-off stall now and then would otherwise decide the figure)
        r0 = api.sharded_finish()
        t0 = time.perf_counter()
        first = warm + rep * frames
        for f in range(first, first + frames, batch):
            api.UpdateTest(0.0, f, w, h, 2)
            api.draw_sharded_batch(0.0, f, batch, w, h, image.data_ptr(), 2)
        t_enq = time.perf_counter() - t0
        rays = api.sharded_finish() - r0
        dt = time.perf_counter() - t0
        if os.environ.get("TPT_EMU_VERBOSE"):
            print("  N=%d pass %d: %.3f ms/frame" % (n, rep, dt / frames * 1e3), flush=True)
        if best is None or dt < best[0]:
            best = (dt, rays, t_enq)
    dt, rays, t_enq = best
    agg = rays / dt / 1e9 * n
    base = base or agg
    print("N=%d: %.3f ms/frame  rank 0 %.2f Gray/s  aggregate %.1f Gray/s  efficiency %.0f %%  (batch %d; in flight %d; host enqueue %.3f ms/frame)" % (
        n, dt / frames * 1e3, rays / dt / 1e9, agg, 100 * agg / (base * n), batch, api.pipeline_info()["overlap_effective"], t_enq / frames * 1e3), flush=True)
    api.comm_destroy()
api.ShutdownTest()
