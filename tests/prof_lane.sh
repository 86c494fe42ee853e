#!/bin/bash
# rocprofv3 kernel stats of one bench configuration -> gpurun_out/ (run on the GPU box)
# usage: tests/prof_lane.sh <lanes> <problems> <max_vertices> <tag>
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
RKH_LANES_PER_EDGE=$1 timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$4 -o $4 -- python $ROOT/bench.py --steps 1 --warmup 0 --problems $2 --max-vertices $3 --no-cpu-baseline --no-microbench > $OUT/$4_bench.log 2>&1
F=$(find /tmp/prof_$4 -name "*kernel_stats.csv" < /dev/null | head -1)
if [ -n "$F" ]; then cp "$F" $OUT/$4_kernel_stats.csv; cut -c1-160 "$F" | head -12; else echo "no kernel_stats file"; find /tmp/prof_$4 -type f < /dev/null | head; fi
tail -1 $OUT/$4_bench.log | cut -c1-300
