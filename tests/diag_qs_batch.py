"""Quasi-static batch planner timing (C1 and C3 worlds), for A/B of the edge-walk kernels (diagnostic, GPU box).
usage: diag_qs_batch.py [problems] [max_vertices]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from reak_amd import lib, scenarios

P = int(sys.argv[1]) if len(sys.argv) > 1 else 64
mv = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
ctx = lib.Context(0)
for name, scn in (("C1", scenarios.make_c1(world_seed=1)), ("C3", scenarios.make_c3(1))):
    sc = lib.Scene(ctx, scn)
    lo, hi, mi = scn.meta["lower"], scn.meta["upper"], scn.meta["min_interval"]
    for p in (1, P):
        prms = [scn.rrt_params(seed=1 + i, max_vertices=mv) for i in range(p)]
        best = None
        for rep in range(3):
            pl = lib.RrtPlanner(sc, prms if p > 1 else prms[0], qs=lib.make_qs_space(sc.n, lo, hi, mi))
            t0 = time.perf_counter()
            pl.solve_planning_query()
            dt = time.perf_counter() - t0
            best = dt if best is None or dt < best else best
            if rep < 2:
                pl.close()
        sts = list(pl.all_stats)
        nv = sum(s.num_vertices for s in sts)
        print(f"{name} quasi-static RRT P={p} mv={mv}: {best * 1e3:8.1f} ms  {nv / best:10.0f} vertices/s  edges_checked {sum(s.edges_checked for s in sts)}", flush=True)
