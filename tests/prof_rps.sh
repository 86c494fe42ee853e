#!/bin/bash
# headline vs rounds enqueued per host sync (diagnostic, GPU box)
for r in 16 32 64; do
  timeout -k 10 300 python bench.py --no-microbench --no-cpu-baseline --rounds-per-sync $r > gpurun_out/r02_y_rps$r.json 2> gpurun_out/r02_y_rps$r.err || { tail -3 gpurun_out/r02_y_rps$r.err; exit 1; }
  python -c "
import json; d=json.load(open('gpurun_out/r02_y_rps$r.json')); print('rounds/sync $r: %.3f M/s  %.0f ms/step  rounds %d spec_eff %.3f' % (d['value']/1e6, d['ms_per_step'], d['rounds'], d['speculation_efficiency']))"
done
