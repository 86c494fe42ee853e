"""BASELINE config C3 at scale: RRT* on the 6-DOF chain / 50 obstacles, quasi-static space, tens of thousands of
vertices; GPU batch vs the CPU oracle on seed 1 (graph equality checked).  Diagnostic, run on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import oracle_lib as O
from reak_amd import lib as L, scenarios as S

mv = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
P = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ctx = L.Context(0)
scn = S.make_c3()
sc = L.Scene(ctx, scn)
lo, hi, mi = scn.meta["lower"], scn.meta["upper"], scn.meta["min_interval"]
qs = L.make_qs_space(6, lo, hi, mi)
prms = [scn.rrt_params(seed=1 + i, max_vertices=mv) for i in range(P)]
pl = L.RrtStarPlanner(sc, prms, qs)
t0 = time.time(); pl.solve_planning_query(); dt = time.time() - t0
it = sum(int(s.loop_iterations) for s in pl.all_stats); ed = sum(int(s.edges_checked) for s in pl.all_stats)
print(f"C3 RRT* GPU P={P} mv={mv}: {dt:.1f}s  {it/dt:.0f} iterations/s  {ed/dt:.0f} edges/s  rewires {pl.stats.rewires} "
      f"solutions {pl.stats.num_solutions} best {pl.stats.best_cost:.4f}", flush=True)
g = pl.graph(0)
if mv > 60000:  # the oracle's linear k-NN makes it impractical here: check the cost invariants of the result instead
    pred, dist, pos = g["pred"].astype(np.int64), g["dist"], g["pos"]
    v = np.arange(1, len(pred))
    ok = pred[v] != 0xFFFFFFFF
    w = np.sqrt(((pos[v[ok]] - pos[pred[v[ok]]]) ** 2).sum(axis=1))
    # an edge's weight is the distance travelled along it, and can_be_connected (planning_visitors.hpp:385-395) accepts a
    # walk that stops up to 5 % short of its target, so dist[v] - dist[pred] lies in [|edge| / 1.05, |edge|]
    dw = dist[v[ok]] - dist[pred[v[ok]]]
    print(f"vertices {len(pred)}, connected {ok.sum()}, edges with weight != length: {(np.abs(dw - w) > 1e-9).sum()}, "
          f"all within [|e|/1.05, |e|]: {bool(((dw <= w + 1e-9) & (dw >= w / 1.05 - 1e-9)).all())}, "
          f"costs increase along the tree: {bool((dw > 0).all())}", flush=True)
    sys.exit(0)
osc = O.OracleScene(scn, fast=False)
rc, out, rg = osc.rrtstar_qs(lo, hi, mi, prms[0])
print(f"C3 RRT* CPU oracle seed 1: {out.seconds:.1f}s {out.loop_iterations/out.seconds:.0f} iterations/s; "
      f"graphs equal: pred {np.array_equal(g['pred'], rg['pred'])} dist {np.array_equal(g['dist'], rg['dist'])} "
      f"rewires {out.rewires} vs {pl.stats.rewires}", flush=True)
