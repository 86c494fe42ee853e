#!/bin/bash
# many-queries NN sweeps side by side (diagnostic, GPU box): the split-bf16 estimate on the matrix cores against the
# f32-input matrix instructions (RKH_NN_BF16=0): parity tests, the microbenchmark on large trees, the planner's timed region.
set -o pipefail
out=gpurun_out/nn_variants
mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "nn1 or coordinate or knn or removed or prefilter" > $out/tests.log 2>&1 || { tail -30 $out/tests.log; exit 1; }
tail -2 $out/tests.log
for w in 1 0; do
  echo "== RKH_NN_BF16=$w"
  RKH_NN_BF16=$w timeout -k 10 200 python tests/diag_nn_mfma.py > $out/mfma_bf16_$w.log 2>&1 || { tail -5 $out/mfma_bf16_$w.log; exit 1; }
  grep TFLOP $out/mfma_bf16_$w.log
  RKH_NN_BF16=$w timeout -k 10 300 python bench.py --no-cpu-baseline --no-microbench > $out/bench_bf16_$w.json 2> $out/bench_bf16_$w.err || { tail -5 $out/bench_bf16_$w.err; exit 1; }
  python - <<PY
import json
d = json.load(open("$out/bench_bf16_$w.json"))
print("value %.3fM  ms/step %.0f  nn timed: %s %.1f TF-equivalent" % (d["value"] / 1e6, d["ms_per_step"], d["nn_sweep_mfma_timed"]["kernel"], d["nn_sweep_mfma_timed"]["achieved"]))
PY
done
