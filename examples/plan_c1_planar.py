"""BASELINE config C1 end to end on the GPU: the planar 3R arm (reference 2D classes), 10 rectangles, quasi-static RRT
to 5000 vertices, then the report files a ReaK run leaves behind (vlist_sbmp_report, least_cost_sbmp_report) and the
best solution's way-points.  Run on a machine with an MI355X:  python examples/plan_c1_planar.py [out_dir] [seed]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reak_amd import lib, reports, scenarios  # noqa: E402

out_dir = sys.argv[1] if len(sys.argv) > 1 else "c1_out"
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
os.makedirs(out_dir, exist_ok=True)
scn = scenarios.make_c1_planar(world_seed=1)
ctx = lib.Context(0)
scene = lib.Scene(ctx, scn)
lo, hi, mi = scn.meta["lower"], scn.meta["upper"], scn.meta["min_interval"]
planner = lib.RrtPlanner(scene, scn.rrt_params(seed=seed, max_vertices=5000), qs=lib.make_qs_space(3, lo, hi, mi))
st = planner.solve_planning_query()
tree = planner.tree()
prefix = os.path.join(out_dir, "c1_")
files = reports.write_rrt_progress(prefix, tree["pos"], progress_interval=1000)
path, cost = planner.solution()
if len(path):
    reports.write_solution_path(prefix + "solution_000_%s" % reports.cxx_double(cost), tree["pos"], path)
print("vertices %d, iterations %d, edges checked %d, solutions %d, best cost %s" %
      (st.num_vertices, st.iterations, st.edges_checked, st.num_solutions, reports.cxx_double(st.best_cost)))
print("wrote", ", ".join(os.path.basename(f) for f in files), "+ solution path (%d way-points)" % len(path))
