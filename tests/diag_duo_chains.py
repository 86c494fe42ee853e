"""Two waves per edge on the random chains of other sizes: results against one wave per edge (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from reak_amd import lib, scenarios
ctx = lib.Context(0)
for n in [int(a) for a in sys.argv[1:]] or [2, 3, 4, 7]:
    scn = scenarios.make_random_chain(n, seed=n); sc = lib.Scene(ctx, scn)
    rng = np.random.default_rng(100 + n)
    lo = np.array([scn.dyn.lower[i] for i in range(2 * n)]); hi = np.array([scn.dyn.upper[i] for i in range(2 * n)])
    a = rng.uniform(lo, hi, size=(40, 2 * n)) * 0.5; b = rng.uniform(lo, hi, size=(40, 2 * n))
    res = {}
    for lanes in ("64", "128"):
        os.environ["RKH_LANES_PER_EDGE"] = lanes
        print("n", n, "lanes", lanes, "...", flush=True)
        res[lanes] = sc.steer_position_toward(a, b, record=True)
    print("n", n, "steps equal", np.array_equal(res["64"][1], res["128"][1]), "states equal", np.array_equal(res["64"][0], res["128"][0]),
          "records equal", np.array_equal(res["64"][2], res["128"][2]), "steps", res["64"][1][:10], res["128"][1][:10], flush=True)
