"""Register-direct NN sweep (B <= 8): GB/s for a few shapes, and a check against the tiled kernel (diagnostic, GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from reak_amd import lib
ctx = lib.Context(0); ev = bench.HipEvents()
for D in (12, 6):
    for n, B in ((4 << 20, 8), (4 << 20, 4), (4 << 20, 1), (16 << 20, 8), (1 << 20, 8), (100000, 8), (1000, 3)):
        nn = lib.HipNeighborSearch(ctx, D, n); nn.fill_uniform(n, seed=7)
        q = torch.rand(max(B, 16), D, dtype=torch.float64, device="cuda")
        idx = torch.zeros(16, dtype=torch.int32, device="cuda"); dist = torch.zeros(16, dtype=torch.float64, device="cuda")
        idx2 = torch.zeros(16, dtype=torch.int32, device="cuda"); dist2 = torch.zeros(16, dtype=torch.float64, device="cuda")
        for _ in range(3): nn.nearest_async(q.data_ptr(), B, idx.data_ptr(), dist.data_ptr())
        ctx.synchronize()
        name = nn.kernel_name()
        pairs = [(ev.create(), ev.create()) for _ in range(10)]
        for a, b in pairs: nn.nearest_async(q.data_ptr(), B, idx.data_ptr(), dist.data_ptr(), events=(a, b))
        ctx.synchronize()
        ms = sum(ev.elapsed_ms(a, b) for a, b in pairs) / 10
        nn.nearest_async(q.data_ptr(), 16, idx2.data_ptr(), dist2.data_ptr())  # 16 queries: the tiled kernel
        ctx.synchronize()
        same = bool(torch.equal(idx[:B], idx2[:B]) and torch.equal(dist[:B], dist2[:B]))
        print("D=%d n=%d B=%d %s %.4f ms  %.0f GB/s  same as %s: %s" % (D, n, B, name, ms, n * D * 8 / ms / 1e6, nn.kernel_name(), same), flush=True)
        nn.close()
