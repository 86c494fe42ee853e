"""NN-sweep microbenchmark alone (HBM-bound regime), for rocprofv3 --pmc / --kernel-trace runs:
   rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc -- python tests/bench_nn_only.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from reak_amd import lib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4 * 1024 * 1024
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
bound = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
ctx = lib.Context(0)
print(json.dumps(bench.nn_sweep_microbench(lib, ctx, bench.HipEvents(), n, B, 10, coord_bound=bound)))
