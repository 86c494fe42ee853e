"""Isolate proximity-verdict differences between steer kernel mappings (diagnostic, GPU box)."""
import os, sys, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import oracle_lib
from reak_amd import lib, scenarios
ctx = lib.Context(0); scn = scenarios.make_c2(1)
rng = np.random.default_rng(0)
lo = np.array([scn.dyn.lower[i] for i in range(12)]); hi = np.array([scn.dyn.upper[i] for i in range(12)])
a = rng.uniform(lo, hi, size=(8192, 12)) * 0.6
b = rng.uniform(lo, hi, size=(8192, 12))
robot = [s for s in scn.shapes if s.anchor >= 0]; env = [s for s in scn.shapes if s.anchor < 0]
def verdicts(s2, x, tgt):
    sc = lib.Scene(ctx, s2); r = {}
    for lanes in ("64", "1", "2"):
        os.environ["RKH_LANES_PER_EDGE"] = lanes
        r[lanes] = sc.steer_position_toward(x, tgt)[1]
    return r
sel = [65, 92, 164, 172]
x, t = a[sel], b[sel]
print("full scene", verdicts(scn, x, t), flush=True)
for e, oi in zip(sel, (17, 25, 36, 3)):
    for rs in ([0, 4], [4, 0], [0, 0, 5], [0, 0, 0, 5], [0, 5, 0, 0]):
        for es in ([oi],):
            s2 = copy.copy(scn); s2.shapes = [robot[i] for i in rs] + [env[i] for i in es]
            v = verdicts(s2, a[e:e+1], b[e:e+1])
            print("edge", e, "robot", rs, "env", "only %d" % oi if len(es) == 1 else "all", {k: int(x[0]) for k, x in v.items()}, flush=True)
