"""PRM throughput: GPU batch driver vs the CPU oracle (diagnostic, run on the GPU box)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle_lib as O
from reak_amd import lib as L, scenarios as S

ctx = L.Context(0)
for name, mk, nd, mv in (("C1", S.make_c1, 3, 3000), ("C3", S.make_c3, 6, 3000), ("C4", S.make_c4, 12, 2000)):
    scn = mk()
    sc = L.Scene(ctx, scn)
    lo, hi, mi = scn.meta["lower"], scn.meta["upper"], scn.meta["min_interval"]
    qs = L.make_qs_space(nd, lo, hi, mi)
    for P in (1, 16, 64):
        prms = [scn.prm_params(seed=1 + i, max_vertices=mv, sampling_radius=(1.5 if nd == 12 else 1.0)) for i in range(P)]
        pl = L.PrmPlanner(sc, prms, qs)
        t0 = time.time(); pl.solve_planning_query(); dt = time.time() - t0
        it = sum(int(s.loop_iterations) for s in pl.all_stats); ed = sum(int(s.edges_checked) for s in pl.all_stats)
        print(f"{name} PRM P={P} mv={mv}: {dt:.2f}s  {it/dt:.0f} iterations/s  {ed/dt:.0f} edges/s  steps {pl.stats.device_steps}", flush=True)
        pl.close()
    osc = O.OracleScene(scn, fast=True)
    rc, out, g = osc.prm_qs(lo, hi, mi, prms[0])
    print(f"{name} PRM CPU oracle: {out.seconds:.2f}s {out.loop_iterations/out.seconds:.0f} iterations/s {out.edges_checked/out.seconds:.0f} edges/s", flush=True)
