#!/bin/bash
# HBM traffic of the planner-regime NN sweep: two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; the TCC block
# cannot hold both) over one timed bench step of the default workload -> gpurun_out/<tag>_nn_planner_pmc.json
# usage: tests/prof_pmc_planner.sh <tag> [problems] [max_vertices]
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
TAG=$1; P=${2:-256}; NV=${3:-100000}
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 900 rocprofv3 --pmc $C --output-format csv -d /tmp/pmcp_${TAG}_$C -o $TAG -- python $ROOT/bench.py --steps 1 --warmup 0 --problems $P --max-vertices $NV --no-cpu-baseline --no-microbench > $OUT/${TAG}_pmc_$C.log 2>&1
  echo "$C rc=$?"
done
python3 - "$TAG" "$P" "$NV" "$OUT" <<'PY'
import csv, glob, json, sys, collections
tag, P, NV, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"/tmp/pmcp_{tag}_{c}/**/*counter_collection.csv", recursive=True)
    tot = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] != c: continue
        k = r["Kernel_Name"].split("(")[0]
        tot[k] += float(r["Counter_Value"]); n[k] += 1
    for k in tot:
        if "nn1_sweep_bf16" in k: res[c] = (tot[k], n[k])
    top = sorted(tot.items(), key=lambda kv: -kv[1])[:6]
    print(c, [(k[-40:], round(v / 1e6, 2), n[k]) for k, v in top])
fetch_kb, nf = res["FETCH_SIZE"]; write_kb, nw = res["WRITE_SIZE"]
# rocprofv3 reports both counters in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request of a wide coalesced stream:
# x 2 (MI355X_MICROARCH.md, HBM section)
per_launch = (2.0 * fetch_kb / nf + write_kb / nw) * 1024.0
rec = {"kernel": "nn1_sweep_bf16_kernel", "problems_per_gpu": P, "max_vertices": NV, "dispatches": nf,
       "FETCH_SIZE_KiB_sum": fetch_kb, "WRITE_SIZE_KiB_sum": write_kb, "fetch_correction": 2.0,
       "hbm_bytes_per_launch": per_launch,
       "how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate runs of `bench.py --steps 1 --warmup 0 "
              "--no-cpu-baseline --no-microbench`; mean over all dispatches of the kernel"}
json.dump(rec, open(f"{out}/{tag}_nn_planner_pmc.json", "w"), indent=1)
print(json.dumps(rec))
PY
