"""Does the edge's second lane detect collisions?  Obstacles placed ON a link (diagnostic, GPU box)."""
import os, sys, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import oracle_lib
from reak_amd import lib, scenarios, types as T
ctx = lib.Context(0); scn = scenarios.make_c2(1); osc = oracle_lib.OracleScene(scn)
rng = np.random.default_rng(3)
lo = np.array([scn.dyn.lower[i] for i in range(12)]); hi = np.array([scn.dyn.upper[i] for i in range(12)])
x = rng.uniform(lo, hi, size=(1, 12)) * 0.5; t = rng.uniform(lo, hi, size=(1, 12))
robot = [s for s in scn.shapes if s.anchor >= 0]
fr = osc.fk(x)[0]
for k in (1, 2, 3, 4, 5):
    p = fr[2 * k + 1][:3]   # joint k end frame position = base of link k's capsule
    for kind, dims in ((T.SHAPE_SPHERE, [0.08, 0, 0]), (T.SHAPE_BOX, [0.15, 0.2, 0.1]), (T.SHAPE_CCYLINDER, [0.2, 0.05, 0])):
        ob = T.Shape(kind=kind, anchor=-1); ob.pose = T.make_pose(tuple(p), (1.0, 0.0, 0.0, 0.0)); ob.dims[:] = dims
        for rs in ([0, k], [k, 0]):
            s2 = copy.copy(scn); s2.shapes = [robot[i] for i in rs] + [ob]
            sc = lib.Scene(ctx, s2); r = {}
            for lanes in ("64", "1", "2"):
                os.environ["RKH_LANES_PER_EDGE"] = lanes
                r[lanes] = int(sc.steer_position_toward(x, t)[1][0])
            print("link", k, "obstacle kind", kind, "robot", rs, r, flush=True)
