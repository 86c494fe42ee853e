"""One shape of the matrix-core NN sweep, a few launches (for rocprofv3 --pmc passes): python tests/diag_nn_mfma_one.py [n] [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from reak_amd import lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
ctx = lib.Context(0)
nn = lib.HipNeighborSearch(ctx, 12, n); nn.fill_uniform(n, seed=7); nn.set_coord_bound(1.0)
q = torch.rand(B, 12, dtype=torch.float64, device="cuda")
idx = torch.zeros(B, dtype=torch.int32, device="cuda"); dist = torch.zeros(B, dtype=torch.float64, device="cuda")
for _ in range(5): nn.nearest_async(q.data_ptr(), B, idx.data_ptr(), dist.data_ptr())
ctx.synchronize()
print(nn.kernel_name())
