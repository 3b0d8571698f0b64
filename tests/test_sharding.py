"""Row-stripe sharding: index logic, and the N>1 exchange step (gather + counter reduce) with
world_size 2 over gloo on CPU.  The tile renderer here is the oracle (there is no GPU); the product's
tptSetRowShard / tptLocalRowCount mirror toypathtracer_amd.sharding (checked on the GPU in
test_gpu_parity.py::test_sharded_equals_unsharded)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from toypathtracer_amd import sharding


@pytest.mark.parametrize("h,stripe,parts", [(360, 8, 8), (720, 4, 8), (117, 8, 2), (100, 16, 4), (7, 8, 2), (64, 64, 1)])
def test_partition_is_a_permutation(h, stripe, parts):
    rows = np.concatenate([sharding.local_to_global_rows(h, stripe, parts, p) for p in range(parts)])
    assert sorted(rows.tolist()) == list(range(h))
    for p in range(parts):
        assert sharding.local_row_count(h, stripe, parts, p) == len(sharding.local_to_global_rows(h, stripe, parts, p))


def test_product_row_mapping_matches_python():
    from toypathtracer_amd import api
    lib = api.load_library()
    for (h, stripe, parts) in [(360, 8, 8), (117, 8, 2), (100, 16, 4), (7, 8, 2)]:
        for p in range(parts):
            lib.tptSetRowShard(stripe, parts, p)
            rows = sharding.local_to_global_rows(h, stripe, parts, p)
            assert lib.tptLocalRowCount(h) == len(rows)
            assert [lib.tptLocalRowToGlobal(i) for i in range(len(rows))] == rows.tolist()
    lib.tptSetRowShard(0, 1, 0)


def _worker(rank, world, port, w, h, stripe, q, mirror=False):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from oracle_lib import Oracle, SEED_PER_PIXEL
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    o = Oracle.get()
    s, m = o.default_scene()
    cam = o.default_camera(w, h)
    sf = sharding.ShardedFrame(w, h, stripe, rank, world, torch.device("cpu"), dist)
    total = None
    import ctypes
    for frame in range(6 if mirror else 2):
        if mirror:
            # what bench.py does on the GPU: ask for the snapshot addresses BEFORE the draw, let the library's resolve
            # kernel fill them (tptSetTileMirror), then exchange(snapshot_done=True)
            sf.begin_frame()
            mirror_ptr, counter_ptr = sf.mirror_pointers()
        # render this rank's rows (progressive accumulation stays in the rank's own tile)
        full = np.zeros((h, w, 4), np.float32)
        rows = sharding.local_to_global_rows(h, stripe, world, rank)
        full[rows] = sf.tile[: len(rows)].numpy()
        rays = 0
        for y in rows:
            r, _ = o.render(s, m, cam, w, h, 2, frame, seed_mode=SEED_PER_PIXEL, backbuffer=full, y0=int(y), y1=int(y) + 1, threads=1)
            rays += r
        sf.tile[: len(rows)] = torch.from_numpy(full[rows])
        sf.ray_counter += rays          # the kernels' atomic adds (tptSetRayCounter)
        if mirror:
            # stand-in for tptResolveMirrorKernel: raw writes through the two addresses
            ctypes.memmove(mirror_ptr, sf.tile.data_ptr(), sf.tile.numel() * 4)
            ctypes.memmove(counter_ptr, sf.ray_counter.data_ptr(), 8)
            sf.exchange(snapshot_done=True)
        else:
            sf.exchange()
    img, total = sf.finish()
    if rank == 0:
        q.put((img.numpy().copy(), total))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,h,stripe,mirror", [(2, 42, 4, False), (3, 50, 8, False), (2, 42, 4, True), (8, 100, 8, True)],
                         ids=["2ranks", "3ranks_uneven", "2ranks_mirrored_snapshot", "8ranks_uneven_mirrored_snapshot"])
def test_gather_reassembles_the_single_process_image(oracle, world, h, stripe, mirror):
    from oracle_lib import SEED_PER_PIXEL
    w = 64
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, w, h, stripe, q, mirror)) for r in range(world)]
    for p in procs:
        p.start()
    img, total = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sc, m = oracle.default_scene()
    cam = oracle.default_camera(w, h)
    bb = np.zeros((h, w, 4), np.float32)
    rays_sum = 0
    for frame in range(6 if mirror else 2):  # 6 frames: the ring of 4 send buffers wraps
        r, _ = oracle.render(sc, m, cam, w, h, 2, frame, seed_mode=SEED_PER_PIXEL, backbuffer=bb)
        rays_sum += r
    assert total == rays_sum
    assert img.tobytes() == bb.tobytes()


@pytest.mark.parametrize("world,h,stripe,w", [(2, 42, 4, 7), (3, 50, 8, 5), (8, 100, 8, 3), (8, 2160, 8, 2), (8, 720, 8, 2), (4, 1080, 8, 2), (5, 33, 3, 4), (2, 8, 8, 4), (8, 5, 8, 4), (1, 17, 8, 3)])
def test_cabi_exchange_index_math_equals_the_python_twin(emu, world, h, stripe, w):
    """The C-ABI exchange a Test.h host drives (tptDrawSharded / tptShardedFinish) has never run with more than one rank on
    hardware; its index arithmetic -- csrc/tpt_shard.h, the very functions tpt_host.cpp and the kernels call: rows per rank,
    the kernel's local <-> global row maps, padRows, the [padRows + 1][w] snapshot with the counter row, the rank-major
    receive buffer, tptAssembleKernel's inverse map, the ring slot, the counter reads of tptShardedFinish -- is replayed on
    the CPU for EVERY rank (tests/lane_emu.cpp: emu_shard_exchange) and held against toypathtracer_amd/sharding.py, whose
    gather the gloo tests above run for real at world sizes 2 / 3 / 8.  Uneven heights, ranks without rows, 1 rank included."""
    import ctypes as C
    from toypathtracer_amd import sharding
    fn = emu.emu_shard_exchange
    fn.restype = C.c_longlong
    fn.argtypes = [C.c_int] * 5 + [C.c_ulonglong, C.c_void_p, C.c_void_p, C.c_void_p]
    image = np.full(h * w, -7, np.int64)
    info = np.zeros(4, np.int64)
    rowmap = np.zeros(h, np.int64)
    frames = 5
    assert fn(w, h, stripe, world, 4, frames, image.ctypes.data, info.ctypes.data, rowmap.ctypes.data) == 0
    assert np.array_equal(image, np.arange(h * w))                       # every pixel found its way home
    pad = int(info[0])                                                   # the equal-count gather's tile height: whole stripes,
    tallest = sharding.padded_rows(h, stripe, world)                     # at least the tallest rank's rows (the twin pads to exactly that)
    assert pad % stripe == 0 and tallest <= pad < tallest + stripe
    assert int(info[1]) == sum((r + 1) * 1000003 + frames + (1 << 40) for r in range(world))  # 64-bit counters, exact
    assert int(info[2]) == frames % 4                                    # snapshot ring slot
    want = np.empty(h, np.int64)                                         # ShardedFrame.rowmap: row of the flattened receive buffer
    for p in range(world):
        g = sharding.local_to_global_rows(h, stripe, world, p)
        assert len(g) == sharding.local_row_count(h, stripe, world, p)
        want[g] = p * (pad + 1) + np.arange(len(g))
    assert np.array_equal(rowmap, want)
