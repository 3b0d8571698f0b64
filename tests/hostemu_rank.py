"""TEST INFRASTRUCTURE: one RANK of a multi-process run of the library's own multi-GPU path (tptCommInit / tptDrawSharded /
tptDrawShardedBatch / tptShardedFinish, csrc/tpt_host.cpp) on the host-emulation build, with tests/hostemu/fake_rccl.cpp standing in for
librccl.so.1.  usage: hostemu_rank.py RANK NRANKS DIR [STRIPE_ROWS]   (started N times by tests/test_host_logic.py)"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
from common import oracle_frames  # noqa: E402
from oracle_lib import FLAG_PROGRESSIVE, SEED_PER_PIXEL, Oracle  # noqa: E402

assert "hostemu" in os.environ.get("TPT_LIB", "")
sys.modules["torch"] = None  # api.py loads torch first when it is there -- and with it the real librccl, which would answer the library's dlopen
from toypathtracer_amd import api as tpt  # noqa: E402

rank, n, d = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
stripe = int(sys.argv[4]) if len(sys.argv) > 4 else 8
W, H, SPP = 72, 52, 2  # 6.5 stripes of 8 rows: uneven over the ranks, the last one partial
tpt.InitializeTest()
tpt.set_samples_per_pixel(SPP)
uid_file = os.path.join(d, "uid.bin")
if rank == 0:
    uid = tpt.comm_get_unique_id()
    with open(uid_file + ".tmp", "wb") as f:
        f.write(bytes(uid))
    os.rename(uid_file + ".tmp", uid_file)
else:
    for _ in range(30000):
        if os.path.exists(uid_file):
            break
        time.sleep(0.002)
    uid = open(uid_file, "rb").read()
tpt.comm_init(uid, n, rank, stripe)
assert tpt.comm_info()[:2] == (n, rank), tpt.comm_info()
img = np.zeros((H, W, 4), np.float32) if rank == 0 else None
ptr = img.ctypes.data if rank == 0 else 0
f = 0
for k in [1, 1, 1, 3, 1, 2, 1, 1]:  # frame by frame and in batches: 11 frames
    tpt.UpdateTest(0.0, f, W, H, FLAG_PROGRESSIVE)
    if k == 1:
        tpt.draw_sharded(0.0, f, W, H, ptr, FLAG_PROGRESSIVE)
    else:
        tpt.draw_sharded_batch(0.0, f, k, W, H, ptr, FLAG_PROGRESSIVE)
    f += k
total = tpt.sharded_finish()
own = tpt.ray_counter_read()
if rank == 0:
    o = Oracle.get()
    want_total, want, _ = oracle_frames(o, W, H, SPP, f, seed_mode=SEED_PER_PIXEL)
    assert img.tobytes() == want.tobytes(), "rank 0's assembled image differs from the 1-GPU render"
    assert total == want_total, (total, want_total)
    print("OK rank 0 of %d: image and ray total of %d frames equal the oracle's (%d rays, %d of them traced here)" % (n, f, total, own), flush=True)
else:
    assert total == own, (total, own)
    print("OK rank %d of %d: %d rays" % (rank, n, own), flush=True)
tpt.comm_destroy()
tpt.ShutdownTest()
