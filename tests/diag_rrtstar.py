import sys, time; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
from reak_amd import lib, scenarios
import oracle_lib
ctx = lib.Context(0)
for name, scn, nd, mv in (("C1", scenarios.make_c1(1), 3, 5000), ("C3", scenarios.make_c3(1), 6, 5000)):
    sc = lib.Scene(ctx, scn); lo, hi, mi = scn.meta["lower"], scn.meta["upper"], scn.meta["min_interval"]
    qs = lib.make_qs_space(nd, lo, hi, mi)
    for P in (1, 16):
        pl = lib.RrtStarPlanner(sc, [scn.rrt_params(seed=s+1, max_vertices=mv) for s in range(P)], qs)
        t0 = time.perf_counter(); pl.solve_planning_query(); t = time.perf_counter() - t0
        it = sum(int(s.loop_iterations) for s in pl.all_stats); ed = sum(int(s.edges_checked) for s in pl.all_stats)
        print(f"{name} RRT* P={P} mv={mv}: {t:.2f}s  {it/t:.0f} iterations/s  {ed/t:.0f} edges/s  rewires {pl.stats.rewires} sol {pl.stats.num_solutions} best {pl.stats.best_cost:.4f}")
    osc = oracle_lib.OracleScene(scn, fast=True)
    rc, out, g = osc.rrtstar_qs(lo, hi, mi, scn.rrt_params(seed=1, max_vertices=mv))
    print(f"{name} RRT* CPU oracle: {out.seconds:.2f}s {out.loop_iterations/out.seconds:.0f} iterations/s {out.edges_checked/out.seconds:.0f} edges/s rewires {out.rewires}")
