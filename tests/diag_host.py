import sys, time, os; sys.path.insert(0,'.')
import numpy as np
from reak_amd import lib, scenarios
P = int(sys.argv[1]) if len(sys.argv) > 1 else 32
MV = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
ctx = lib.Context(0); scn = scenarios.make_c2(1); sc = lib.Scene(ctx, scn)
pl = lib.RrtPlanner(sc, [scn.rrt_params(seed=s+1, max_vertices=MV) for s in range(P)])
pl.enqueue(0)
t_enq = t_sync = 0.0; n_enq = 0
t0 = time.perf_counter()
while True:
    a = time.perf_counter(); pl.enqueue(16); n_enq += 16
    b = time.perf_counter(); pl.sync()
    c = time.perf_counter()
    t_enq += b-a; t_sync += c-b
    if pl.done: break
tot = time.perf_counter()-t0
nodes = sum(int(s.num_vertices)-1 for s in pl.all_stats); spec = sum(int(s.edges_speculated) for s in pl.all_stats)
print(f"P={P} MV={MV} GL={os.environ.get('RKH_LANES_PER_EDGE','auto')} total {tot:.3f}s enqueue {t_enq:.3f}s sync {t_sync:.3f}s rounds {pl.stats.rounds} nodes/s {nodes/tot:.0f} edges_prop/s {spec/tot:.0f}")
