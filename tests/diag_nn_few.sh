#!/bin/bash
# few-queries sweeps over large trees: matrix-core pre-filter (coordinate bound declared) against the fp64 sweeps (GPU box)
for n in ${ROWS:-4194304 262144 65536}; do
  for B in ${QUERIES:-1 4 8 16 32}; do
    for bound in 0 1; do
      timeout -k 10 120 python tests/bench_nn_only.py $n $B $bound > /tmp/nf.log 2>&1 || { tail -5 /tmp/nf.log; exit 1; }
      python - <<PY
import json
d = json.loads([l for l in open('/tmp/nf.log') if l.startswith('{')][-1])
print("n %8d B %2d bound $bound: %-22s %.4f ms  %.0f GB/s" % (d['n'], d['queries_per_sweep'], d['kernel'], d['ms_per_sweep'], d['achieved']), flush=True)
PY
    done
  done
done
