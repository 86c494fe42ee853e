import sys; sys.path.insert(0,'.')
import bench, torch
from reak_amd import lib
ctx = lib.Context(0); ev = bench.HipEvents()
for n, B in [(4*1024*1024, 1), (4*1024*1024, 8), (4*1024*1024, 16), (1024*1024, 8), (16*1024*1024, 8), (4*1024*1024, 64)]:
    r = bench.nn_sweep_microbench(lib, ctx, ev, n, B, 20)
    print(f"n={n} B={B}: {r['ms_per_sweep']*1e3:.1f} us  {r['achieved']:.0f} GB/s  frac {r['frac']:.3f}  queries/s {r['queries_per_s']:.0f}")
