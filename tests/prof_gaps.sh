#!/bin/bash
# kernel timeline of one bench step -> idle gaps in front of each kernel type (run on the GPU box)
# usage: tests/prof_gaps.sh <tag> [extra bench.py arguments]
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/gaps_$TAG -o $TAG -- python $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-microbench "$@" > $OUT/${TAG}_gaps_run.log 2>&1
echo "rocprof rc=$?"
F=$(find /tmp/gaps_$TAG -name "*kernel_trace.csv" < /dev/null | head -1)
M=$(find /tmp/gaps_$TAG -name "*memory_copy_trace.csv" < /dev/null | head -1)
python3 - "$F" "$M" <<'PY' | tee $OUT/${TAG}_gaps.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in rows)
busy = 0; cur_end = t0; gap_before = collections.defaultdict(float); dur = collections.defaultdict(float); cnt = collections.Counter()
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"]); k = r["Kernel_Name"].split("(")[0].replace("void rkh::", "").replace("rkh::", "")[:40]
    if s > cur_end: gap_before[k] += s - cur_end
    busy += max(0, e - max(s, cur_end)); cur_end = max(cur_end, e); dur[k] += e - s; cnt[k] += 1
print("wall %.1f ms, GPU busy (union of kernels) %.1f ms, idle %.1f ms" % ((t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6))
for k in sorted(dur, key=lambda k: -dur[k]):
    print("%-42s n=%6d  dur %9.1f ms  idle-before %8.1f ms  (%.1f us each)" % (k, cnt[k], dur[k] / 1e6, gap_before[k] / 1e6, gap_before[k] / 1e3 / cnt[k]))
if len(sys.argv) > 2 and sys.argv[2]:
    m = list(csv.DictReader(open(sys.argv[2])))
    tot = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in m)
    by = collections.Counter(); byb = collections.Counter()
    for r in m: by[r.get("Direction", "?")] += 1; byb[r.get("Direction", "?")] += int(r.get("Size", 0) or 0)
    print("memory copies:", dict(by), "bytes", dict(byb), "total copy time %.1f ms" % (tot / 1e6))
PY
