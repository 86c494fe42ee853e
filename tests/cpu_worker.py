"""One CPU worker of bench.py's multi-core baseline: the oracle's sequential RRT (C2 world, dynamics) for one seed.
TEST INFRASTRUCTURE (oracle); prints one JSON line.  usage: python tests/cpu_worker.py <seed> <max_vertices>"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import oracle_lib  # noqa: E402
from reak_amd import scenarios  # noqa: E402

seed, nv = int(sys.argv[1]), int(sys.argv[2])
scn = scenarios.make_c2(world_seed=1)
osc = oracle_lib.OracleScene(scn, fast=True)
rc, out, _ = osc.rrt_dyn(scn.rrt_params(seed=seed, max_vertices=nv))
print(json.dumps({"rc": rc, "vertices": int(out.num_vertices), "edges": int(out.edges_checked), "seconds": float(out.seconds)}))
