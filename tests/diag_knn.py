"""k-NN sweep of BASELINE config C3 (6-D tree of 1 M vertices, k = 4(floor(log2 n)+1) = 80, star radius) and the
published NN configuration of BASELINE.md (6-D, N = 25 000, 1000 queries): time per query per vertex.
Diagnostic, run on the GPU box."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch

import oracle_lib as O
from bench import HipEvents
from reak_amd import lib as L

ctx = L.Context(0)
ev = HipEvents()
stream = ctx.stream


def timed(fn, reps=10):
    for _ in range(2):
        fn()
    ctx.synchronize()
    a, b = ev.create(), ev.create()
    ev.record(a, stream)
    for _ in range(reps):
        fn()
    ev.record(b, stream)
    ctx.synchronize()
    return ev.elapsed_ms(a, b) / reps


D = 6
for n in (1_000_000, 8_000_000):
    nn = L.HipNeighborSearch(ctx, D, n)
    nn.fill_uniform(n, seed=3)
    logn = int(np.floor(np.log2(n))) + 1
    k = 4 * logn
    radius = 3.0 * (logn / n) ** (1.0 / D)  # gamma = 3 * unit distance
    for B in (1, 8, 64):
        q = torch.rand(B, D, dtype=torch.float64, device="cuda")
        idx = torch.zeros(B, k, dtype=torch.int32, device="cuda")
        dist = torch.zeros(B, k, dtype=torch.float64, device="cuda")
        cnt = torch.zeros(B, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        ms = timed(lambda: L._check(nn.lib.rkh_nn_queryk_async(nn.h, q.data_ptr(), B, k, float(radius), idx.data_ptr(),
                                                               dist.data_ptr(), cnt.data_ptr())))
        gb = n * D * 8 / 1e9
        print(f"kNN n={n} D={D} k={k} B={B}: {ms*1e3:.1f} us per batch, {gb/(ms*1e-3):.0f} GB/s algorithmic "
              f"(one pass over the rows; the kernel makes two), mean neighbours {float(cnt.float().mean()):.1f}", flush=True)
    nn.close()

# the reference's published NN configuration (dox/results/test_vp_tree/dvp_umap_vecS_6.dat): 6-D, N = 25 000, 1000 queries
n, B = 25000, 1000
nn = L.HipNeighborSearch(ctx, D, n)
nn.fill_uniform(n, seed=5)
q = torch.rand(B, D, dtype=torch.float64, device="cuda")
idx = torch.zeros(B, dtype=torch.int32, device="cuda")
dist = torch.zeros(B, dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
ms = timed(lambda: nn.nearest_async(q.data_ptr(), B, idx.data_ptr(), dist.data_ptr()), reps=50)
print(f"1-NN linear search 6-D N=25000, 1000 queries: {ms*1e3:.1f} us per batch = {ms*1e3/B/n:.3e} us per query per vertex "
      f"(reference, CPU of 2012: linear search 8.2e-3, DVP-tree arity 4 1.59e-4)", flush=True)
pts = np.random.default_rng(0).random((n, D)); qq = np.random.default_rng(1).random((B, D))
t0 = time.time(); O.nn1(qq, pts, fast=True); dt = time.time() - t0
print(f"CPU oracle linear search (this box, 1 core, -O3): {dt*1e6/B/n:.3e} us per query per vertex", flush=True)
ti, td, tb, tq = O.vptree_nn1(qq, pts, fast=True)
print(f"CPU static vantage-point tree (this box, 1 core, -O3): {tq*1e6/B/n:.3e} us per query per vertex "
      f"({tq*1e6/B:.1f} us per query, build {tb*1e3:.1f} ms)", flush=True)
