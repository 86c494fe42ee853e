"""Single-problem RRT rate (C2, 20 000 vertices) for the batch factor given in RKH_BATCH_FACTOR (diagnostic, GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from reak_amd import lib, scenarios
P = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ctx = lib.Context(0)
scn = scenarios.make_c2(world_seed=1)
scene = lib.Scene(ctx, scn)
bench.single_problem_rate(lib, scene, scn, P, 2000)
r = bench.single_problem_rate(lib, scene, scn, P, 20000)
print("factor", os.environ.get("RKH_BATCH_FACTOR", "default"), "max", os.environ.get("RKH_BATCH_MAX", "default"), "P", P, "%.0f expansions/s  %.0f edges/s  %.3f s" % (r["value"], r["edges_collision_checked_per_s"], r["seconds"]), flush=True)
