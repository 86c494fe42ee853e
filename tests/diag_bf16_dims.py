"""Many-queries sweep against numpy for several dimensions (diagnostic, GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from reak_amd import lib
ctx = lib.Context(0)
for D in (2, 3, 4, 6, 7, 8, 12, 16):
    rng = np.random.default_rng(D)
    n, B = 3000, 130
    pts = rng.uniform(-1, 1, size=(n, D)); q = rng.uniform(-1, 1, size=(B, D))
    nn = lib.HipNeighborSearch(ctx, D, n); nn.added_vertices(pts); nn.set_coord_bound(1.0)
    idx, dist = nn.nearest(q)
    d2 = ((q[:, None, :] - pts[None, :, :]) ** 2).sum(-1)
    ref = d2.argmin(1)
    print("D", D, nn.kernel_name(), "mismatches", int((idx != ref).sum()), "none", int((idx == 0xFFFFFFFF).sum()), flush=True)
    nn.close()
