import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from reak_amd import lib, scenarios
ctx = lib.Context(0); c2 = scenarios.make_c2(1); sc = lib.Scene(ctx, c2)
rng = np.random.default_rng(3)
lo = np.array([c2.dyn.lower[i] for i in range(12)]); hi = np.array([c2.dyn.upper[i] for i in range(12)])
x = rng.uniform(lo, hi, size=(5000, 12))
print(sc.proximity_counts(x))
print(sc.proximity_counts(x[:32]))
print(sc.proximity_counts(x[:1]))
pl = lib.RrtPlanner(sc, c2.rrt_params(seed=7000, max_vertices=20000)); pl.solve_planning_query()
st = pl.tree()["pos"]; print(st.shape, st[:2])
print(sc.proximity_counts(st)); print(sc.proximity_counts(st[:5000]))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
import oracle_lib
osc = oracle_lib.OracleScene(c2); d = osc.min_distance(st[:5000]); print("oracle min distance quantiles", np.quantile(d, [0, 0.01, 0.1, 0.5]))
