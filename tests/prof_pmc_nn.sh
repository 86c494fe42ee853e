#!/bin/bash
# rocprofv3 --pmc passes (one counter group per run, no tracing flags) over the matrix-core NN sweep alone -> gpurun_out/
# usage: tests/prof_pmc_nn.sh <tag> "<counters>" [n] [B]
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc $2 --output-format csv -d /tmp/pmc_$1 -o $1 -- python $ROOT/tests/diag_nn_mfma_one.py ${3:-1048576} ${4:-1024} > $OUT/$1_pmc.log 2>&1
echo "rc=$?"
F=$(find /tmp/pmc_$1 -name "*counter_collection.csv" < /dev/null | head -1)
if [ -n "$F" ]; then
  python3 - "$F" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][:48]
    if "mfma" not in k: continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k in acc:
    for c, v in acc[k].items():
        print(k, c, "%.4g per dispatch" % (v / cnt[(k, c)]), cnt[(k, c)])
PY
else echo "no counter file"; tail -5 $OUT/$1_pmc.log; fi
