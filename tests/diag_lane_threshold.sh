#!/bin/bash
# single-/few-problem rate against the edge count above which a round goes to the two-lanes-per-edge kernel (GPU box)
for P in ${PROBLEMS:-1 4 16}; do
  for T in ${THRESHOLDS:-256 1024 4500}; do
    RKH_LANE_THRESHOLD=$T timeout -k 10 200 python tests/diag_single.py $P > /tmp/dlt.log 2>&1 || { tail -5 /tmp/dlt.log; exit 1; }
    grep expansions /tmp/dlt.log | sed "s/^/threshold $T: /"
  done
done
