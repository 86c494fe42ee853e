"""One CPU worker of bench.py's multi-core baseline: the oracle's sequential RRT loop (C2 world, dynamics) for one seed,
started on a given tree (rows of a .npy file; bench.py cpu_baseline_all_cores) and run for about `seconds` seconds.
TEST INFRASTRUCTURE (oracle); prints one JSON line.  usage: python tests/cpu_worker.py <seed> <tree.npy> <seconds>"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import oracle_lib  # noqa: E402
from reak_amd import scenarios  # noqa: E402

seed, path, seconds = int(sys.argv[1]), sys.argv[2], float(sys.argv[3])
scn = scenarios.make_c2(world_seed=1)
osc = oracle_lib.OracleScene(scn, fast=True)
warm = np.load(path)
prm = scn.rrt_params(seed=seed, max_vertices=2 ** 31 - 1)
rc, o = osc.rrt_dyn_warm(prm, warm, 64)
iters = int(max(64, min(50000, seconds / max(o.seconds / 64, 1e-6))))
rc, o = osc.rrt_dyn_warm(prm, warm, iters)
print(json.dumps({"rc": rc, "vertices_added": int(o.num_vertices) - 1 - len(warm), "iterations": int(o.iterations),
                  "edges": int(o.edges_checked), "seconds": float(o.seconds)}))
