#!/bin/bash
# the headline workload at other batch shapes and batch factors (diagnostic, GPU box)
run() { timeout -k 10 300 python bench.py --no-microbench --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys,os; d=json.loads(sys.stdin.read()); print('factor %s $*: %.3f M/s  %.0f ms/step  rounds %d  spec_eff %.3f' % (os.environ.get('RKH_BATCH_FACTOR','default'), d['value']/1e6, d['ms_per_step'], d['rounds'], d['speculation_efficiency']))" || exit 1; }
for f in ${FACTORS:-default 1.25 2 3 4}; do
  if [ "$f" = default ]; then unset RKH_BATCH_FACTOR; else export RKH_BATCH_FACTOR=$f; fi
  run --problems 64 --max-vertices 100000
  run --problems 128 --max-vertices 20000 --steps 3
  run --problems 32 --max-vertices 100000
done
