"""Time the torch ops of ShardedFrame's exchange on an idle GPU (which one is slow?)."""
import torch, time, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from toypathtracer_amd.sharding import local_to_global_rows, padded_rows
W, H, S = 1280, 720, 8
dev = torch.device("cuda")
def bench(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for N in (2, 4, 8):
    pad = padded_rows(H, S, N)
    tile = torch.zeros((pad, W, 4), device=dev)
    send = torch.zeros((pad + 1, W, 4), device=dev)
    recv = torch.zeros((N, pad + 1, W, 4), device=dev)
    image = torch.zeros((H, W, 4), device=dev)
    ctr = torch.zeros(1, dtype=torch.int64, device=dev)
    rowmap = np.empty(H, np.int64)
    for p in range(N):
        g = local_to_global_rows(H, S, N, p)
        rowmap[g] = p * (pad + 1) + np.arange(len(g))
    rowmap = torch.as_tensor(rowmap, device=dev)
    flat = recv.view(N * (pad + 1), W, 4)
    print("N=%d  snapshot copy %.1f us | counter copy %.1f us | index_select %.1f us | permuted copy %.1f us" % (
        N, bench(lambda: send[:pad].copy_(tile, non_blocking=True)),
        bench(lambda: send[pad, 0, :2].view(torch.int64).copy_(ctr, non_blocking=True)),
        bench(lambda: torch.index_select(flat, 0, rowmap, out=image)),
        bench(lambda: image.view(H // (S * N), N, S, W, 4).copy_(recv[:, :pad].view(N, pad // S, S, W, 4).permute(1, 0, 2, 3, 4), non_blocking=True)) if H % (S * N) == 0 else float("nan")))
