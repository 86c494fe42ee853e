"""Two-lanes-per-edge steer kernels side by side (diagnostic, run on the GPU box): per-phase cycles of one f-eval +
proximity test per wave, and edges/s of a large steer batch.  RKH_LANES_PER_EDGE: 1 = LDS-resident kernel
(propagate_lane.hip), 2 = registers + DPP, two waves per SIMD (propagate_pair.hip)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from reak_amd import lib, scenarios
ctx = lib.Context(0); scn = scenarios.make_c2(1); sc = lib.Scene(ctx, scn)
rng = np.random.default_rng(0)
names = ["frames", "jac", "M", "bwd", "chol", "pFK", "pCull", "pExact"]
lo = np.array([scn.dyn.lower[i] for i in range(12)]); hi = np.array([scn.dyn.upper[i] for i in range(12)])
B = int(os.environ.get("DIAG_EDGES", 262144))
a = rng.uniform(lo, hi, size=(B, 12)) * 0.6
b = rng.uniform(lo, hi, size=(B, 12))
ref = None
for lanes in ("1", "2"):
    os.environ["RKH_LANES_PER_EDGE"] = lanes
    for nb in (32, 32 * 2048, 32 * 8192):
        x = rng.uniform(-1, 1, size=(nb, 12)); u = rng.uniform(-10, 10, size=(nb, 6))
        c = sc.diag_feval_cycles(x, u, iters=20).astype(np.float64) / 20
        c = c[: nb // 32]
        med = np.median(c, axis=0)
        print("lanes=%s waves=%d" % (lanes, nb // 32), " ".join("%s=%.0f" % (n, v) for n, v in zip(names, med)),
              "feval=%.0f prox=%.0f" % (med[:5].sum(), med[5:].sum()), flush=True)
    out = sc.steer_position_toward(a[:4096], b[:4096])  # warm-up
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter(); out = sc.steer_position_toward(a, b); dt = time.perf_counter() - t0
        best = min(best, dt)
    print("lanes=%s steer %d edges: %.1f ms -> %.2f M edges/s (incl. copies), mean free steps %.2f" %
          (lanes, B, best * 1e3, B / best / 1e6, out[1].mean()), flush=True)
    if ref is None: ref = out
    else: print("bit-identical to lanes=1:", np.array_equal(ref[0], out[0]) and np.array_equal(ref[1], out[1]), flush=True)
