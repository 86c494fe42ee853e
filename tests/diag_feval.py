import sys; sys.path.insert(0,'.')
import numpy as np
from reak_amd import lib, scenarios
ctx = lib.Context(0); scn = scenarios.make_c2(1); sc = lib.Scene(ctx, scn)
rng = np.random.default_rng(0)
for B in (1, 256, 2048, 8192):
    x = rng.uniform(-1,1,size=(B,12)); u = rng.uniform(-10,10,size=(B,6))
    c = sc.diag_feval_cycles(x,u,iters=50).astype(np.float64)/50
    names = ["sincos","fwd","tcm","bwd","M","chol","prox","total"]
    print("B=%d"%B, " ".join("%s=%.0f"%(n,v) for n,v in zip(names, np.median(c,axis=0))))
