"""MFMA pre-filter sweep alone: TFLOP/s (24 flops per (vertex, query) pair) for a few shapes (diagnostic, GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from reak_amd import lib
ctx = lib.Context(0); ev = bench.HipEvents()
D = 12
for n, B in ((1 << 20, 384), (1 << 20, 128), (1 << 20, 1024), (1 << 18, 384), (1 << 16, 384), (1 << 20, 64)):
    nn = lib.HipNeighborSearch(ctx, D, n); nn.fill_uniform(n, seed=7); nn.set_coord_bound(1.0)
    q = torch.rand(B, D, dtype=torch.float64, device="cuda")
    idx = torch.zeros(B, dtype=torch.int32, device="cuda"); dist = torch.zeros(B, dtype=torch.float64, device="cuda")
    for _ in range(3): nn.nearest_async(q.data_ptr(), B, idx.data_ptr(), dist.data_ptr())
    ctx.synchronize()
    pairs = [(ev.create(), ev.create()) for _ in range(10)]
    for a, b in pairs: nn.nearest_async(q.data_ptr(), B, idx.data_ptr(), dist.data_ptr(), events=(a, b))
    ctx.synchronize()
    ms = sum(ev.elapsed_ms(a, b) for a, b in pairs) / 10
    print("n=%d B=%d %s %.3f ms  %.1f TFLOP/s  %.0f GB/s algorithmic" % (n, B, nn.kernel_name(), ms, n * B * 24 / ms / 1e9, n * 96 / ms / 1e6), flush=True)
    nn.close()
