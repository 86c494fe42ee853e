#!/bin/bash
# A/B of bench.py variants on the GPU box: tools/ab_bench.sh <tag> "<ENV=.. ENV=.. | bench args>" ...
# each variant string: environment assignments first, then bench arguments after a '|'
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
i=0
for v in "$@"; do
  envs=${v%%|*}; args=${v#*|}
  f=$OUT/v$i.json
  env $envs timeout -k 10 300 python bench.py --no-cpu-baseline --no-microbench $args > $f 2> $OUT/v$i.err || { echo "variant $i failed"; tail -3 $OUT/v$i.err; exit 1; }
  python - "$f" "$v" <<'EOF'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
sk = d.get("steer_kernels", {})
print(f"{sys.argv[2]:70s} {d['value']/1e6:7.3f} M/s  {d['ms_per_step']:8.1f} ms/step  rounds {d['rounds']}  steer/round {sk.get('avg_round_ms', 0):.3f} ms  nn {d['roofline'].get('avg_launch_us') or 0:.0f} us  spec_eff {d['speculation_efficiency']:.3f}")
EOF
  i=$((i+1))
done
