#!/bin/bash
# NN sweep variants side by side (diagnostic, GPU box): register-direct stream kernel with / without the LDS transpose,
# matrix-core sweep with different seed strides.
set -o pipefail
out=gpurun_out/r02_r_nn
mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "nn1 or coordinate or knn" > $out/tests.log 2>&1 || { tail -30 $out/tests.log; exit 1; }
tail -3 $out/tests.log
echo "== stream, direct rows"; timeout -k 10 200 python tests/diag_nn_stream.py > $out/stream_direct.log 2>&1 && grep "n=4194304\|n=16777216" $out/stream_direct.log
echo "== stream, transposed"; RKH_NN_XPOSE=1 timeout -k 10 200 python tests/diag_nn_stream.py > $out/stream_xpose.log 2>&1 && grep "n=4194304\|n=16777216" $out/stream_xpose.log
for s in 0 4 8 16; do
  echo "== mfma, seed stride $s"; RKH_NN_SEED_STRIDE=$s timeout -k 10 200 python tests/diag_nn_mfma.py > $out/mfma_seed$s.log 2>&1 && cat $out/mfma_seed$s.log | grep TFLOP
done
