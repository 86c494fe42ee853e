import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from reak_amd import lib, scenarios
ctx = lib.Context(0); scn = scenarios.make_c2(1); sc = lib.Scene(ctx, scn)
rng = np.random.default_rng(0)
B = 256
x = rng.uniform(-1, 1, size=(B, 12)); u = rng.uniform(-10, 10, size=(B, 6))
os.environ.pop("RKH_LANES_PER_EDGE", None)
c = sc.diag_feval_cycles(x, u, iters=50).astype(np.float64) / 50
print("one wave  ", " ".join("%s=%.0f" % (n, v) for n, v in zip(["sincos", "fwd", "tcm", "bwd", "M", "chol", "prox", "total"], np.median(c, axis=0))))
os.environ["RKH_LANES_PER_EDGE"] = "128"
c = sc.diag_feval_cycles(x, u, iters=50).astype(np.float64) / 50
a, b = c[0::2], c[1::2]
print("duo wave 0", " ".join("%s=%.0f" % (n, v) for n, v in zip(["sincos", "frames", "columns", "M", "factor", "wait", "solve", "total"], np.median(a, axis=0))))
print("duo wave 1", " ".join("%s=%.0f" % (n, v) for n, v in zip(["sincos", "vel", "beam", "force", "-", "wait", "tail", "total"], np.median(b, axis=0))))
