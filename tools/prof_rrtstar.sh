#!/bin/bash
# kernel trace of a single-problem C3 RRT* run: stats + the launches of a few consecutive device steps (GPU box)
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r03_rs}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d /tmp/prof_$TAG -o rs -- python $ROOT/tests/diag_rrtstar_large.py 20000 ${2:-1} > $OUT/run.log 2>&1
echo rc=$?; grep "RRT" $OUT/run.log
F=$(find /tmp/prof_$TAG -name "*kernel_stats.csv" < /dev/null | head -1); cp "$F" $OUT/kernel_stats.csv; cut -c1-150 "$F" | head -14
T=$(find /tmp/prof_$TAG -name "*kernel_trace.csv" < /dev/null | head -1)
M=$(find /tmp/prof_$TAG -name "*memory_copy_trace.csv" < /dev/null | head -1)
python - "$T" "$M" <<'EOF' > $OUT/steps.txt
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("rkh::", "")[:50]))
try:
    for r in csv.DictReader(open(sys.argv[2])):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")))
except Exception as e:
    print("no copy trace", e)
rows.sort()
mid = len(rows) // 2
prev = rows[mid][0]
for s, e, n in rows[mid:mid + 60]:
    print(f"{n:52s} {(e - s) / 1e3:8.1f} us  gap {(s - prev) / 1e3:8.1f}")
    prev = e
EOF
head -62 $OUT/steps.txt
