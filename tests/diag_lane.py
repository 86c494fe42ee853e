"""Per-phase cycle counts of the two-lanes-per-edge steer kernel (diagnostic, run on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RKH_LANES_PER_EDGE"] = "1"
import numpy as np
from reak_amd import lib, scenarios
ctx = lib.Context(0); scn = scenarios.make_c2(1); sc = lib.Scene(ctx, scn)
rng = np.random.default_rng(0)
names = ["frames", "jac", "M", "bwd", "chol", "pFK", "pCull", "pExact"]
for B in (32, 32 * 1024, 32 * 4096):
    x = rng.uniform(-1, 1, size=(B, 12)); u = rng.uniform(-10, 10, size=(B, 6))
    c = sc.diag_feval_cycles(x, u, iters=20).astype(np.float64) / 20
    c = c[: B // 32]
    print("waves=%d" % (B // 32), " ".join("%s=%.0f" % (n, v) for n, v in zip(names, np.median(c, axis=0))), flush=True)
