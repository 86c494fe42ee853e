"""Which edges differ between the steer kernel mappings (diagnostic, GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import oracle_lib
from reak_amd import lib, scenarios
ctx = lib.Context(0); scn = scenarios.make_c2(1); sc = lib.Scene(ctx, scn); osc = oracle_lib.OracleScene(scn)
rng = np.random.default_rng(0)
lo = np.array([scn.dyn.lower[i] for i in range(12)]); hi = np.array([scn.dyn.upper[i] for i in range(12)])
B = int(os.environ.get("DIAG_EDGES", 8192))
a = rng.uniform(lo, hi, size=(B, 12)) * 0.6
b = rng.uniform(lo, hi, size=(B, 12))
res = {}
for lanes in ("64", "1", "2"):
    os.environ["RKH_LANES_PER_EDGE"] = lanes
    res[lanes] = sc.steer_position_toward(a, b)
    os.environ["RKH_LANES_PER_EDGE"] = lanes
    again = sc.steer_position_toward(a, b)
    print("lanes", lanes, "repeatable:", np.array_equal(res[lanes][0], again[0]) and np.array_equal(res[lanes][1], again[1]), flush=True)
nchk = min(B, 2048)
rc, rout, rsteps, _ = osc.steer(a[:nchk], b[:nchk])
for lanes in ("64", "1", "2"):
    out, steps = res[lanes][0], res[lanes][1]
    bad_s = np.nonzero(steps[:nchk] != rsteps)[0]
    bad_x = np.nonzero(~np.isclose(out[:nchk], rout, rtol=1e-9, atol=1e-11).all(axis=1))[0]
    print("lanes", lanes, "vs oracle: steps differ at", bad_s[:10], len(bad_s), "states differ at", bad_x[:10], len(bad_x), flush=True)
for lanes in ("1", "2"):
    ds = np.nonzero(res[lanes][1] != res["64"][1])[0]
    dx = np.nonzero((res[lanes][0] != res["64"][0]).any(axis=1))[0]
    print("lanes", lanes, "vs 64: steps differ", len(ds), ds[:10], "states differ", len(dx), dx[:10], flush=True)
    for e in dx[:4]:
        print("  edge", e, "steps", res["64"][1][e], res[lanes][1][e], "maxdiff", np.abs(res[lanes][0][e] - res["64"][0][e]).max(),
              "lane-in-wave", e % 32, flush=True)
