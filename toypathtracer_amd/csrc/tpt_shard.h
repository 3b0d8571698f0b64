// tpt_shard.h -- index arithmetic of the row-stripe sharding and of the one exchange step (tptDrawSharded), in one place.
//
// What the reference does with a task set over rows and a join (Cpp/Source/Test.cpp:357-361), across GPUs: the image's rows are
// dealt out in stripes of S rows, round-robin over N ranks; rank r keeps its stripes in a compact tile; per frame every rank
// sends `padRows + 1` rows -- its blended tile, padded to the tallest rank's height, plus one row whose first 8 bytes carry
// its 64-bit ray counter -- to rank 0 with one ncclGather, and rank 0 de-interleaves the gathered tiles into the image.
// Plain integer functions shared by the host runtime (tpt_host.cpp), the kernels (tpt_kernels.hip: mapItem, tptAssembleKernel)
// and the CPU-only test harness (tests/lane_emu.cpp -> tests/test_sharding.py, which holds them against
// toypathtracer_amd/sharding.py for world sizes 2 / 3 / 8 -- no GPU, no second rank needed).
#pragma once
#include "tpt_math.h"

namespace tpt {

// rows of an h-row image that land on rank `part` (stripes of S rows, N ranks); N <= 1 or S <= 0: all of them
TPT_HD int shardLocalRows(int h, int S, int N, int part)
{
    if (N <= 1 || S <= 0) return h;
    const int stride = S * N, off = S * part;
    const int full = h / stride, rem = h % stride;
    int rows = full * S;
    const int extra = rem - off;
    if (extra > 0) rows += extra < S ? extra : S;
    return rows;
}
// image row of local tile row ly of rank `part`
TPT_HD int shardLocalToGlobal(int ly, int S, int N, int part)
{
    if (N <= 1 || S <= 0) return ly;
    return (ly / S) * S * N + S * part + (ly % S);
}
// local tile row of image row gy on the rank that owns it (inverse of shardLocalToGlobal)
TPT_HD int shardGlobalToLocal(int gy, int S, int N)
{
    if (N <= 1 || S <= 0) return gy;
    const int stripe = gy / S;
    return (stripe / N) * S + (gy - stripe * S);
}
TPT_HD int shardOwner(int gy, int S, int N) { return (N <= 1 || S <= 0) ? 0 : (gy / S) % N; }
// the same two maps as the kernels evaluate them, from the launch's precomputed stripe constants (KernelArgs: stripeRows = S,
// stripeStride = S N, stripeOffset = S part; an unsharded launch passes stripeRows = stripeStride = h, stripeOffset = 0)
TPT_HD int shardKernelLocalToGlobal(int ly, int stripeRows, int stripeStride, int stripeOffset)
{
    return (ly / stripeRows) * stripeStride + stripeOffset + (ly % stripeRows);
}
TPT_HD int shardKernelGlobalToLocal(int gy, int stripeRows, int stripeStride, int stripeOffset) // rows of this rank only
{
    const int q = gy / stripeStride;
    return q * stripeRows + (gy - q * stripeStride - stripeOffset);
}
// tile height every rank pads to, so that an equal-count gather can be used: rank 0 owns the most stripes; whole stripes
TPT_HD int shardPadRows(int h, int S, int N)
{
    const int stripes = (h + S - 1) / S;
    return ((stripes + N - 1) / N) * S;
}
// the exchange buffers: a rank's snapshot is [padRows + 1][w] float4 -- rows 0 .. padRows - 1 the tile, row padRows the
// counter row; rank 0's receive buffer is [N][padRows + 1][w] float4.  Offsets in float4 units.
TPT_HD size_t shardSnapshotPixels(int padRows, int w) { return (size_t)(padRows + 1) * (size_t)w; }
TPT_HD size_t shardCounterPixel(int padRows, int w) { return (size_t)padRows * (size_t)w; } // its first 8 bytes = the 64-bit ray counter, bit-cast
// where pixel (x, gy) of the image sits in rank 0's receive buffer (what tptAssembleKernel reads)
TPT_HD size_t shardGatheredPixel(int x, int gy, int w, int S, int N, int padRows)
{
    return ((size_t)shardOwner(gy, S, N) * (size_t)(padRows + 1) + (size_t)shardGlobalToLocal(gy, S, N)) * (size_t)w + (size_t)x;
}
// which of the `ring` snapshot buffers frame number `frames` (0-based count of sharded frames so far) writes
TPT_HD int shardRingSlot(unsigned long long frames, int ring) { return (int)(frames % (unsigned long long)ring); }

} // namespace tpt
