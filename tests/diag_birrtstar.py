"""Bidirectional RRT*: GPU vs oracle, field by field (diagnostic, GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import oracle_lib as O
from reak_amd import lib as L, scenarios as S
ctx = L.Context(0)
c1 = S.make_c1(world_seed=1)
sc, osc = L.Scene(ctx, c1), O.OracleScene(c1)
lo, hi, mi = c1.meta["lower"], c1.meta["upper"], c1.meta["min_interval"]
for nv in (50, 200, 1200):
    prm = c1.rrt_params(seed=1, max_vertices=nv)
    rc, rout, rg = osc.birrtstar_qs(lo, hi, mi, prm)
    pl = L.BiRrtStarPlanner(sc, prm, L.make_qs_space(3, lo, hi, mi))
    st = pl.solve_planning_query()
    g = pl.graph()
    print(nv, "edges_checked", st.edges_checked, rout.edges_checked, "samples", st.samples, rout.samples, "iters", st.loop_iterations, rout.loop_iterations)
    for key in ("near_pred", "near_succ", "pred", "succ", "pos", "dist", "fwd_dist"):
        same = g[key].shape == rg[key].shape and np.array_equal(g[key], rg[key])
        print("  ", key, same, "" if same else np.flatnonzero(np.any(np.atleast_2d(g[key] != rg[key]).reshape(len(g[key]), -1), axis=1))[:5] if g[key].shape == rg[key].shape else (g[key].shape, rg[key].shape))
    pl.close()
