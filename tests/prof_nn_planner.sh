#!/bin/bash
# planner-regime NN sweep with different seed strides (diagnostic, GPU box)
set -o pipefail
out=gpurun_out/r02_s_nn
mkdir -p $out
for s in 8 0 4 16; do
  RKH_NN_SEED_STRIDE=$s timeout -k 10 300 python bench.py --no-microbench --no-cpu-baseline > $out/bench_seed$s.json 2> $out/bench_seed$s.err || { tail -5 $out/bench_seed$s.err; exit 1; }
  python - <<PY
import json
d = json.load(open("$out/bench_seed$s.json"))
r, m = d["roofline"], d["nn_sweep_mfma_timed"]
print("seed stride $s: value %.3fM  ms/step %.0f  nn avg_launch_us %.1f  %.1f TF  %.0f GB/s  steer share %.3f" % (d["value"] / 1e6, d["ms_per_step"], r["avg_launch_us"], m["achieved"], r["achieved"], d["steer_kernels"]["share_of_step_time"]), flush=True)
PY
done
