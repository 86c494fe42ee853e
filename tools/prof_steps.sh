#!/bin/bash
# rocprofv3 kernel trace of one bench step + per-launch durations of the steer kernels of some mid-run rounds
# usage (GPU box): tools/prof_steps.sh <tag> [extra bench args]
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o $TAG -- python $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-microbench "$@" > $OUT/bench_under_rocprof.log 2>&1
echo "rocprof rc=$?"
F=$(find /tmp/prof_$TAG -name "*kernel_stats.csv" < /dev/null | head -1)
T=$(find /tmp/prof_$TAG -name "*kernel_trace.csv" < /dev/null | head -1)
[ -n "$F" ] && cp "$F" $OUT/kernel_stats.csv && cut -c1-160 "$F" | head -12
[ -n "$T" ] && python $ROOT/tools/trace_rounds.py "$T" > $OUT/trace_rounds.txt && head -80 $OUT/trace_rounds.txt
