#!/bin/bash
# NN sweeps after a change: parity tests, microbenchmarks, HBM traffic of the planner regime (GPU box)
set -o pipefail
out=gpurun_out/$1
mkdir -p $out
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "nn1 or coordinate or knn" > $out/tests.log 2>&1 || { tail -30 $out/tests.log; exit 1; }
tail -2 $out/tests.log
timeout -k 10 200 python tests/diag_nn_stream.py > $out/stream.log 2>&1 && grep "n=4194304\|n=16777216" $out/stream.log
timeout -k 10 200 python tests/diag_nn_mfma.py > $out/mfma.log 2>&1 && grep TFLOP $out/mfma.log
bash tests/prof_pmc_planner.sh $1 | tail -4
