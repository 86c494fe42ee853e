#!/bin/bash
# default bench.py run + rocprofv3 kernel stats of the same command -> gpurun_out/ (run on the GPU box)
# usage: tests/prof_default.sh <tag>
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout -k 10 900 python bench.py > $OUT/$1_bench.json 2> $OUT/$1_bench.err
echo "bench rc=$?"; tail -c 600 $OUT/$1_bench.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$1 -o $1 -- python $ROOT/bench.py > $OUT/$1_bench_under_rocprof.log 2>&1
echo "rocprof rc=$?"
F=$(find /tmp/prof_$1 -name "*kernel_stats.csv" < /dev/null | head -1)
if [ -n "$F" ]; then cp "$F" $OUT/$1_kernel_stats.csv; cut -c1-150 "$F" | head -14; else echo "no kernel_stats file"; fi
