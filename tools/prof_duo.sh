#!/bin/bash
# kernel durations of the one-wave and the two-waves-per-edge steer kernels on the same edges (GPU box)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_duo -o duo -- python $ROOT/tests/diag_duo.py > $ROOT/gpurun_out/r03_duo_run.log 2>&1
F=$(find /tmp/prof_duo -name "*kernel_stats.csv" < /dev/null | head -1)
python3 - "$F" <<'EOF'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:4]:
    print(r["Name"].split("(")[0][-60:], "calls", r["Calls"], "avg us %.1f" % (float(r["AverageNs"]) / 1e3), "min %.1f max %.1f" % (float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
EOF
