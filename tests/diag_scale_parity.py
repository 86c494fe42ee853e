"""Parity at bench scale: P problems x mv vertices in the planner's automatic mode; a few problems are re-run by the
oracle and compared (counts, NN sequence, accept bits, topology).  Diagnostic, run on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import oracle_lib as O
from reak_amd import lib as L, scenarios as S

P = int(sys.argv[1]) if len(sys.argv) > 1 else 128
mv = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
ctx = L.Context(0); c2 = S.make_c2(1); sc = L.Scene(ctx, c2); osc = O.OracleScene(c2, fast=True)
prms = [c2.rrt_params(seed=500 + i, max_vertices=mv) for i in range(P)]
pl = L.RrtPlanner(sc, prms)
t0 = time.time(); pl.solve_planning_query(); dt = time.time() - t0
print(f"GPU: P={P} mv={mv}: {dt:.1f}s, {sum(int(s.num_vertices) - 1 for s in pl.all_stats)/dt:.0f} expansions/s", flush=True)
for i in (0, P // 2, P - 1):
    rc, ro, rt = osc.rrt_dyn(prms[i])
    st, tr = pl.all_stats[i], pl.tree(i)
    ok = ((st.num_vertices, st.iterations, st.edges_checked) == (ro.num_vertices, ro.iterations, ro.edges_checked)
          and np.array_equal(tr["nn_seq"], rt["nn_seq"]) and np.array_equal(tr["accept"], rt["accept"])
          and np.array_equal(tr["parent"], rt["parent"]) and np.allclose(tr["pos"], rt["pos"], rtol=1e-10, atol=1e-12))
    print(f"problem {i}: oracle {ro.seconds:.1f}s, {ro.num_vertices} vertices, {ro.iterations} iterations; identical: {ok}", flush=True)
