import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from toypathtracer_amd import api
api.InitializeTest()
w, h = 1280, 720
tile = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
for f in range(40):
    api.UpdateTest(0.0, f, w, h, 2); api.draw_device(0.0, f, w, h, tile.data_ptr(), 2)
cost, order = api.debug_chunk_order()
print('n', len(cost), 'cost min/mean/max', cost.min(), cost.mean(), cost.max(), 'nonzero', (cost > 0).sum())
print('order first 10', order[:10], 'cost of those', cost[order[:10]])
print('order last 10', order[-10:], 'cost of those', cost[order[-10:]])
print('is permutation', sorted(order.tolist()) == list(range(len(order))))
api.ShutdownTest()
