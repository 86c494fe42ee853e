"""Latency of a quasi-static edge walk launch against the number of interpolation points (diagnostic, GPU box).
Few edges per launch -- a device step of one graph planner -- so the launch waits for its passes: the slope between
the rows is the cost of one pass of edge_check_kernel, the intercept the call's fixed cost (two copies + sync)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from reak_amd import lib, scenarios

ctx = lib.Context(0)
scn = scenarios.make_c3(1)
sc = lib.Scene(ctx, scn)
n = sc.n
lo, hi, mi = scn.meta["lower"], scn.meta["upper"], scn.meta["min_interval"]
rng = np.random.default_rng(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
# free configurations: keep the ones whose tiny edge completes
cand = rng.uniform(-1.0, 1.0, size=(4096, n))
d = np.zeros(n); d[n - 1] = 1.0       # the wrist joint turns: the arm's shapes barely move
_, chk = sc.move_position_toward(lo, hi, mi, cand, cand + d * 0.26)
a = cand[chk >= 5][:B]
assert len(a) == B
for npts in (0, 1, 8, 30, 31, 32, 33, 62, 64, 96, 126):
    b = a + d * (npts * mi + 0.5 * mi)
    for _ in range(20):
        out, chk = sc.move_position_toward(lo, hi, mi, a, b)
    t0 = time.perf_counter()
    reps = 300
    for _ in range(reps):
        sc.move_position_toward(lo, hi, mi, a, b)
    dt = (time.perf_counter() - t0) / reps
    print(f"edges {B:5d} points {npts:4d} checked {chk.min()}..{chk.max()}  {dt * 1e6:8.1f} us per call", flush=True)
