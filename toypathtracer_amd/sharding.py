"""Row-stripe sharding of one frame across ranks + the single gather that reassembles it.

New relative to the reference (which is single-device; its only fan-out is the CPU row task set,
Test.cpp:357-361, 4-row granules): rows are dealt out in stripes of `stripe_rows`, round-robin over
the ranks (cost is not uniform in y: sky rows are cheap, the sphere-covered bottom is not), each
rank renders its stripes into a compact tile that stays resident in its HBM, and ONE gather
(RCCL over xGMI: torch.distributed "nccl" backend on ROCm) brings the tiles to rank 0, plus one
8-byte sum-reduce of the ray counters.  Seeds depend on global (x, y) only, so the assembled image
is bit-identical to a 1-GPU render.

The exchange is software-pipelined against rendering: frame f's tile is snapshotted into one of `depth`
(default 4) send buffers on the render stream and gathered on a communication stream while frame f+1 renders
(the accumulation tile itself is read-modify-written in place by the kernel, so it cannot be the
gather source).  With CPU tensors (gloo, the CPU test suite) the same code runs without streams.

Pure index logic (mirrors tptLocalRowCount / tptLocalRowToGlobal in tpt_host.cpp) is kept in plain
numpy so the CPU test suite can exercise it.
"""
import numpy as np


def local_row_count(height, stripe_rows, num_parts, part):
    if num_parts <= 1 or stripe_rows <= 0:
        return height
    stride, off = stripe_rows * num_parts, stripe_rows * part
    full, rem = divmod(height, stride)
    rows = full * stripe_rows
    extra = rem - off
    if extra > 0:
        rows += min(extra, stripe_rows)
    return rows


def local_to_global_rows(height, stripe_rows, num_parts, part):
    """int64 array: global row of every local row of `part`."""
    n = local_row_count(height, stripe_rows, num_parts, part)
    ly = np.arange(n, dtype=np.int64)
    if num_parts <= 1 or stripe_rows <= 0:
        return ly
    return (ly // stripe_rows) * (stripe_rows * num_parts) + stripe_rows * part + (ly % stripe_rows)


def padded_rows(height, stripe_rows, num_parts):
    """Tile height every rank pads to, so a plain (equal-count) gather can be used."""
    return max(local_row_count(height, stripe_rows, num_parts, p) for p in range(num_parts))


def assemble(tiles, height, stripe_rows, num_parts, out=None, row_index=None):
    """tiles[p]: [>=local_rows(p), width, 4] array/tensor of rank p -> full [height, width, 4] image."""
    import torch  # plumbing only

    first = tiles[0]
    is_torch = isinstance(first, torch.Tensor)
    if out is None:
        shape = (height,) + tuple(first.shape[1:])
        out = torch.empty(shape, dtype=first.dtype, device=first.device) if is_torch else np.empty(shape, first.dtype)
    for p in range(num_parts):
        if row_index is not None:
            idx = row_index[p]
        else:
            rows = local_to_global_rows(height, stripe_rows, num_parts, p)
            idx = torch.as_tensor(rows, device=first.device) if is_torch else rows
        out[idx] = tiles[p][: len(idx)]
    return out


class ShardedFrame:
    """One process per GPU.  The caller renders this rank's rows into `self.tile` (on `render_stream` when
    on a GPU) and then calls `exchange()`; `finish()` drains the pipeline and returns what rank 0 holds.

    Per frame there is exactly ONE collective: a gather of `pad_rows + 1` rows per rank -- the tile plus one
    extra row whose first 8 bytes carry the rank's 64-bit ray counter (exact integer, bit-cast) -- and on
    rank 0 one index_select that de-interleaves the row stripes into the image."""

    def __init__(self, width, height, stripe_rows, rank, world, device, dist=None, depth=4):
        import torch

        self.torch, self.dist = torch, dist
        self.width, self.height, self.stripe_rows = width, height, stripe_rows
        self.rank, self.world = rank, world
        self.device = torch.device(device)
        self.on_gpu = self.device.type == "cuda"
        self.rows = local_row_count(height, stripe_rows, world, rank)
        self.pad_rows = padded_rows(height, stripe_rows, world)
        z = dict(dtype=torch.float32, device=self.device)
        self.tile = torch.zeros((self.pad_rows, width, 4), **z)      # accumulation tile, resident across frames
        self.ray_counter = torch.zeros(1, dtype=torch.int64, device=self.device)  # kernels add to it (tptSetRayCounter)
        self.image = torch.zeros((height, width, 4), **z) if rank == 0 else None
        self.total_rays = torch.zeros(1, dtype=torch.int64, device=self.device)
        self.steps = 0
        self.depth = depth = max(1, int(depth))  # gathers that may be in flight behind the renderer
        if world > 1:
            assert width >= 1
            self.send = [torch.zeros((self.pad_rows + 1, width, 4), **z) for _ in range(depth)]
            if rank == 0:
                self.recv = [torch.zeros((world, self.pad_rows + 1, width, 4), **z) for _ in range(depth)]
                self.recv_list = [[r[i] for i in range(world)] for r in self.recv]
                # image row y comes from row (p * (pad_rows + 1) + ly) of the flattened receive buffer
                rowmap = np.empty(height, np.int64)
                for p in range(world):
                    g = local_to_global_rows(height, stripe_rows, world, p)
                    rowmap[g] = p * (self.pad_rows + 1) + np.arange(len(g))
                self.rowmap = torch.as_tensor(rowmap, device=self.device)
            else:
                self.recv = self.recv_list = None
        if self.on_gpu:
            # Two streams = two more hardware queues, alive as long as the process (torch hands out pooled streams): create
            # ONE ShardedFrame per process.  The library already holds 16 trace queues; past ~32 the runtime time-slices
            # them and everything slows down (eight ShardedFrames in one process: 0.17 -> 0.62 ms per frame).
            self.render_stream = torch.cuda.Stream(device=self.device)
            self.comm_stream = torch.cuda.Stream(device=self.device)
            self.ev_ready = [torch.cuda.Event() for _ in range(depth)]   # send buffer filled (render stream)
            self.ev_free = [torch.cuda.Event() for _ in range(depth)]    # send buffer consumed (comm stream)
            # the buffers above were filled on torch's current stream: both streams wait for those fills (a stream dependency, not a
            # device synchronise; the library works on render_stream once the caller passes it to tptSetStream)
            cur = torch.cuda.current_stream(self.device)
            self.render_stream.wait_stream(cur)
            self.comm_stream.wait_stream(cur)
        else:
            self.render_stream = self.comm_stream = None

    def _fill_send(self, k):
        send = self.send[k]
        send[: self.pad_rows].copy_(self.tile, non_blocking=True)
        # the 64-bit ray counter rides in the first 8 bytes of the extra row (bit-cast, not converted)
        send[self.pad_rows, 0, :2].view(self.torch.int64).copy_(self.ray_counter, non_blocking=True)

    def mirror_pointers(self):
        """Device addresses (snapshot tile, 8-byte counter slot) of the send buffer the NEXT exchange() will use, for
        tptSetTileMirror: the library's resolve kernel then fills the snapshot itself and exchange(snapshot_done=True)
        skips the two copy kernels -- two fewer kernels in every frame's dependency chain.  None when not sharded."""
        if self.world <= 1:
            return None
        send = self.send[self.steps % self.depth]
        return send.data_ptr(), send[self.pad_rows].data_ptr()

    def begin_frame(self):
        """With mirroring: call BEFORE the frame's draw is enqueued -- makes the render stream wait until the collective
        that last read the send buffer about to be overwritten has finished."""
        if self.world > 1 and self.on_gpu and self.steps >= self.depth:
            # always the stream wait, never an event query as a shortcut: a query on a re-recorded event has been seen to
            # answer "done" before the new record's work was, and a blend that overwrites a snapshot the
            # gather is still reading would corrupt the exchanged image once in a long while
            self.render_stream.wait_event(self.ev_free[self.steps % self.depth])

    def exchange(self, snapshot_done=False):
        """Call after this frame's render has been enqueued on render_stream."""
        torch = self.torch
        k = self.steps % self.depth
        self.steps += 1
        if self.world <= 1:
            return
        if self.on_gpu:
            with torch.cuda.stream(self.render_stream):
                if not snapshot_done:
                    if self.steps > self.depth:
                        self.render_stream.wait_event(self.ev_free[k])    # gather of frame f-depth has read this buffer
                    self._fill_send(k)
                self.ev_ready[k].record(self.render_stream)
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(self.ev_ready[k])
                self._collect(k)
                self.ev_free[k].record(self.comm_stream)
        else:
            self._fill_send(k)
            self._collect(k)

    def _collect(self, k):
        torch = self.torch
        self.dist.gather(self.send[k], self.recv_list[k] if self.rank == 0 else None, dst=0)  # the one exchange step
        if self.rank == 0:
            flat = self.recv[k].view(self.world * (self.pad_rows + 1), self.width, 4)
            torch.index_select(flat, 0, self.rowmap, out=self.image)
            self.last = k

    def finish(self):
        """Drain both streams.  Returns (image, cumulative rays over all ranks) on rank 0, (None, None) elsewhere."""
        torch = self.torch
        if self.on_gpu:
            self.render_stream.synchronize()
            self.comm_stream.synchronize()
        if self.world <= 1:
            self.image.copy_(self.tile[: self.height])
            self.total_rays.copy_(self.ray_counter)
        if self.rank != 0:
            return None, None
        if self.world > 1 and self.steps > 0:
            counters = self.recv[self.last][:, self.pad_rows, 0, :2].contiguous().view(torch.int64)
            self.total_rays[0] = counters.sum()
        return self.image, int(self.total_rays.item())
