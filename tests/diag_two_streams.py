"""One planner with P problems vs G planners with P/G problems each on their own streams, driven from G host threads
(diagnostic, GPU box): do the tails of one group's steer kernel overlap the other group's NN sweep?"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from reak_amd import lib as L, scenarios as S

P = int(sys.argv[1]) if len(sys.argv) > 1 else 256
mv = int(sys.argv[2]) if len(sys.argv) > 2 else 30000
ctx = L.Context(0); c2 = S.make_c2(1); sc = L.Scene(ctx, c2)
for G in (1, 2, 1, 2, 4):
    planners = [L.RrtPlanner(sc, [c2.rrt_params(seed=900 + g * P + i, max_vertices=mv) for i in range(P // G)]) for g in range(G)]
    ctx.synchronize()
    t0 = time.time()
    th = [threading.Thread(target=pl.solve_planning_query) for pl in planners]
    for t in th: t.start()
    for t in th: t.join()
    dt = time.time() - t0
    nodes = sum(int(s.num_vertices) - 1 for pl in planners for s in pl.all_stats)
    print(f"G={G}: {dt:.2f}s {nodes / dt:.0f} expansions/s", flush=True)
    for pl in planners: pl.close()
