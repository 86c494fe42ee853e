"""Many-queries sweep in few dimensions: bf16 against f32-input matrix instructions (RKH_NN_BF16_MIN_DIMS=2 | 99)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from reak_amd import lib
ctx = lib.Context(0); ev = bench.HipEvents()
for D, n, B in ((6, 1 << 20, 1024), (6, 25000, 1000), (6, 1 << 17, 384), (3, 1 << 20, 1024), (3, 1 << 16, 384), (4, 1 << 18, 384)):
    nn = lib.HipNeighborSearch(ctx, D, n); nn.fill_uniform(n, seed=7); nn.set_coord_bound(1.0)
    q = torch.rand(B, D, dtype=torch.float64, device="cuda")
    idx = torch.zeros(B, dtype=torch.int32, device="cuda"); dist = torch.zeros(B, dtype=torch.float64, device="cuda")
    for _ in range(3): nn.nearest_async(q.data_ptr(), B, idx.data_ptr(), dist.data_ptr())
    ctx.synchronize()
    pairs = [(ev.create(), ev.create()) for _ in range(10)]
    for a, b in pairs: nn.nearest_async(q.data_ptr(), B, idx.data_ptr(), dist.data_ptr(), events=(a, b))
    ctx.synchronize()
    ms = sum(ev.elapsed_ms(a, b) for a, b in pairs) / 10
    print("D=%d n=%d B=%d %s %.1f us" % (D, n, B, nn.kernel_name(), ms * 1e3), flush=True)
    nn.close()
