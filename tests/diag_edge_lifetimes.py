"""How long do the steer edges of the headline workload live?  One C2 problem is solved to 20 000 vertices; then random
samples are steered from their nearest tree vertex (the candidates of a round) and tree vertices toward the goal (the
probes), and the collision-free steps of each edge are counted (diagnostic, GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from reak_amd import lib, scenarios
ctx = lib.Context(0)
scn = scenarios.make_c2(world_seed=1)
scene = lib.Scene(ctx, scn)
pl = lib.RrtPlanner(scene, scn.rrt_params(seed=5000, max_vertices=20000))
pl.solve_planning_query()
t = pl.tree()
pos = t["pos"][:, :scene.D]
pl.close()
rng = np.random.default_rng(3)
lo = np.array([scn.dyn.lower[i] for i in range(scene.D)]); hi = np.array([scn.dyn.upper[i] for i in range(scene.D)])
q = rng.uniform(lo, hi, size=(8192, scene.D))
nn = lib.HipNeighborSearch(ctx, scene.D, len(pos)); nn.added_vertices(pos)
idx, _ = nn.nearest(q)
S = scn.dyn.steps_per_edge
_, steps_c, _ = scene.steer_position_toward(pos[idx], q)
goal = np.asarray(scn.goal, dtype=np.float64).reshape(1, -1)
src = pos[rng.integers(1, len(pos), size=8192)]
_, steps_p, _ = scene.steer_position_toward(src, np.repeat(goal, len(src), axis=0))
for name, st in (("candidates", steps_c), ("goal probes", steps_p)):
    st = np.minimum(st.astype(np.int64), S)
    print("%s: mean collision-free steps %.2f of %d, full length %.1f %%, dead within 5 steps %.1f %%, lane-steps used %.1f %%"
          % (name, st.mean(), S, 100.0 * (st >= S).mean(), 100.0 * (st < 5).mean(), 100.0 * st.mean() / S))
print("histogram of candidate steps:", np.bincount(np.minimum(steps_c, S), minlength=S + 1))
print("histogram of probe steps:    ", np.bincount(np.minimum(steps_p, S), minlength=S + 1))
