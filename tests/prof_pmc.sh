#!/bin/bash
# rocprofv3 --pmc pass (own run, no tracing flags) over a short bench -> gpurun_out/ (run on the GPU box)
# usage: tests/prof_pmc.sh <tag> "<counters>"
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --pmc $2 --output-format csv -d /tmp/pmc_$1 -o $1 -- python $ROOT/bench.py --steps 1 --warmup 0 --problems ${PMC_PROBLEMS:-128} --max-vertices ${PMC_VERTICES:-6000} --no-cpu-baseline --no-microbench > $OUT/$1_pmc.log 2>&1
echo "rc=$?"
F=$(find /tmp/pmc_$1 -name "*counter_collection.csv" < /dev/null | head -1)
if [ -n "$F" ]; then
  python3 - "$F" "$OUT/$1_pmc_summary.csv" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][:60]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
with open(sys.argv[2], "w") as f:
    w = csv.writer(f); w.writerow(["kernel", "counter", "sum", "dispatches"])
    for k in acc:
        for c, v in acc[k].items():
            w.writerow([k, c, v, cnt[(k, c)]]); print(k, c, v, cnt[(k, c)])
PY
else echo "no counter file"; find /tmp/pmc_$1 -type f < /dev/null | head; fi
