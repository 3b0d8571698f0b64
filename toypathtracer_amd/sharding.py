"""Row-stripe sharding of one frame across ranks: the index arithmetic, and a CPU MODEL of the one gather that reassembles it.

On GPUs the exchange is the library's own (csrc/tpt_host_shard.cpp: tptCommInit / tptDrawSharded / tptShardedFinish over RCCL; api.comm_init,
api.draw_sharded) -- there is ONE multi-GPU implementation, and this module is not it.  What lives here is what can run without a GPU: the
row <-> rank index logic (the same functions as csrc/tpt_shard.h, checked against each other by tests/test_sharding.py) and `ShardedFrame`,
the exchange protocol restated on CPU tensors over a `torch.distributed` group (gloo), which is how the N > 1 path is exercised in this
repository's CPU test suite at world sizes 2, 3 and 8.

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
    """One process per rank (CPU tensors).  The caller renders this rank's rows into `self.tile` and then calls `exchange()`;
    `finish()` returns what rank 0 holds.

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
            raise RuntimeError("ShardedFrame is the CPU model of the exchange (gloo tests); on GPUs use the library's own: "
                               "api.comm_init / api.draw_sharded / api.sharded_finish (csrc/tpt_host_shard.cpp)")
        self.render_stream = self.comm_stream = None

    def _fill_send(self, k):
        send = self.send[k]
        send[: self.pad_rows].copy_(self.tile, non_blocking=True)
        # the 64-bit ray counter rides in the first 8 bytes of the extra row (bit-cast, not converted)
        send[self.pad_rows, 0, :2].view(self.torch.int64).copy_(self.ray_counter, non_blocking=True)

    def mirror_pointers(self):
        """Addresses (snapshot tile, 8-byte counter slot) of the send buffer the NEXT exchange() will use: the library's blend kernel
        writes the snapshot itself (tptSetTileMirror; csrc/tpt_host_shard.cpp does the same with its ring of 4), and
        exchange(snapshot_done=True) then sends it as it is.  None when not sharded."""
        if self.world <= 1:
            return None
        send = self.send[self.steps % self.depth]
        return send.data_ptr(), send[self.pad_rows].data_ptr()

    def begin_frame(self):
        """(the library waits here for the gather that last read the send buffer about to be overwritten; nothing is asynchronous on the CPU)"""

    def exchange(self, snapshot_done=False):
        """Call after this frame's rows have been rendered into self.tile."""
        k = self.steps % self.depth
        self.steps += 1
        if self.world <= 1:
            return
        if not snapshot_done:
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
        """Returns (image, cumulative rays over all ranks) on rank 0, (None, None) elsewhere."""
        torch = self.torch
        if self.world <= 1:
            self.image.copy_(self.tile[: self.height])
            self.total_rays.copy_(self.ray_counter)
        if self.rank != 0:
            return None, None
        if self.world > 1 and self.steps > 0:
            counters = self.recv[self.last][:, self.pad_rows, 0, :2].contiguous().view(torch.int64)
            self.total_rays[0] = counters.sum()
        return self.image, int(self.total_rays.item())
