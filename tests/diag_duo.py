"""Two waves per edge (state_derivative_duo) against one wave per edge: identical results, time per launch (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from reak_amd import lib, scenarios

ctx = lib.Context(0); scn = scenarios.make_c2(1); sc = lib.Scene(ctx, scn)
rng = np.random.default_rng(0)
lo = np.array([scn.dyn.lower[i] for i in range(12)]); hi = np.array([scn.dyn.upper[i] for i in range(12)])
for B in (8, 256):
    a = rng.uniform(lo, hi, size=(B, 12)) * 0.5
    b = rng.uniform(lo, hi, size=(B, 12))
    res = {}
    for lanes in ("64", "128"):
        os.environ["RKH_LANES_PER_EDGE"] = lanes
        out, steps, rec = sc.steer_position_toward(a, b, record=True)
        t0 = time.perf_counter()
        for _ in range(20):
            sc.steer_position_toward(a, b)
        res[lanes] = (out, steps, rec, (time.perf_counter() - t0) / 20)
    o64, o128 = res["64"], res["128"]
    print(f"B={B}: steps equal {np.array_equal(o64[1], o128[1])}  states equal {np.array_equal(o64[0], o128[0])}  "
          f"records equal {np.array_equal(o64[2], o128[2])}  mean steps {o64[1].mean():.1f}  "
          f"64: {o64[3] * 1e3:.3f} ms  128: {o128[3] * 1e3:.3f} ms", flush=True)
