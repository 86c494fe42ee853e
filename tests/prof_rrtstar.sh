#!/bin/bash
# where a device step of the graph planners goes: kernel stats of a single-problem RRT* run (GPU box)
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d /tmp/prof_rs -o rs -- python $ROOT/tests/diag_rrtstar_large.py 20000 1 > $OUT/r02_rs.log 2>&1
echo rc=$?; grep "C3 RRT" $OUT/r02_rs.log
F=$(find /tmp/prof_rs -name "*kernel_stats.csv" < /dev/null | head -1); cut -c1-150 "$F" | head -14
G=$(find /tmp/prof_rs -name "*memory_copy_stats.csv" < /dev/null | head -1); [ -n "$G" ] && cut -c1-150 "$G" | head -6
