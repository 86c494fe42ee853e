"""k-NN sweep alone (star_neighborhood k and radius) on large trees: microseconds per batch (diagnostic, GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from reak_amd import lib
ctx = lib.Context(0); ev = bench.HipEvents()
for D, n, B in ((6, 1 << 20, 8), (6, 1 << 20, 1), (12, 1 << 20, 8), (6, 1 << 22, 8), (6, 1 << 17, 1), (7, 1 << 20, 8)):
    nn = lib.HipNeighborSearch(ctx, D, n); nn.fill_uniform(n, seed=3)
    logn = int(np.floor(np.log2(n))) + 1
    k, radius = 4 * logn, 3.0 * (logn / n) ** (1.0 / D)
    q = torch.rand(B, D, dtype=torch.float64, device="cuda")
    idx = torch.zeros(B, k, dtype=torch.int32, device="cuda"); dist = torch.zeros(B, k, dtype=torch.float64, device="cuda")
    cnt = torch.zeros(B, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    call = lambda: lib._check(nn.lib.rkh_nn_queryk_async(nn.h, q.data_ptr(), B, k, float(radius), idx.data_ptr(), dist.data_ptr(), cnt.data_ptr()))
    for _ in range(3): call()
    ctx.synchronize()
    a, b = ev.create(), ev.create()
    ev.record(a, ctx.stream)
    for _ in range(20): call()
    ev.record(b, ctx.stream); ctx.synchronize()
    us = ev.elapsed_ms(a, b) * 1e3 / 20
    DP = next(d for d in (2, 4, 6, 8, 12, 16) if D <= d)
    print("D=%d n=%d B=%d k=%d: %.1f us per batch, %.0f GB/s algorithmic (n * Dp * 8), found %s" % (D, n, B, k, us, n * DP * 8 / us / 1e3, cnt.cpu().numpy()[:3]), flush=True)
    nn.close()
