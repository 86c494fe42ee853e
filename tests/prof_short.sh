#!/bin/bash
# rocprofv3 kernel stats of one timed bench step (no warm-up, no microbenchmarks, no CPU baseline) -> gpurun_out/
# usage: tests/prof_short.sh <tag> [extra bench.py arguments]
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o $TAG -- python $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-microbench "$@" > $OUT/${TAG}_bench_under_rocprof.log 2>&1
echo "rocprof rc=$?"; tail -c 400 $OUT/${TAG}_bench_under_rocprof.log
F=$(find /tmp/prof_$TAG -name "*kernel_stats.csv" < /dev/null | head -1)
if [ -n "$F" ]; then cp "$F" $OUT/${TAG}_kernel_stats.csv; cut -c1-160 "$F" | head -16; else echo "no kernel_stats file"; fi
