#!/bin/bash
# per-kernel time of one single-problem solve (C2, 20 000 vertices): which kernel bounds the round latency (GPU box)
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_single
rm -rf $out && mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o single -- python3 $GRAFT_REPO_ROOT/tests/diag_single.py ${1:-1} > $out/run.log 2>&1 || { tail -5 $out/run.log; exit 1; }
f=$(find $out -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] || { echo "no kernel_stats.csv"; ls -R $out; exit 1; }
cut -c1-200 "$f" < /dev/null | head -14
grep expansions $out/run.log
