#!/bin/bash
# HBM traffic of the planner-regime NN search (nn_mirror.hip): two separate rocprofv3 --pmc passes (FETCH_SIZE,
# WRITE_SIZE; the TCC block cannot hold both) over one timed bench step of the default workload
# -> gpurun_out/<tag>/nn_planner_pmc.json.  usage (GPU box): tools/prof_pmc_nn.sh <tag> [problems] [max_vertices]
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; P=${2:-512}; NV=${3:-100000}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 900 rocprofv3 --pmc $C --output-format csv -d /tmp/pmcp_${TAG}_$C -o $TAG -- python $ROOT/bench.py --steps 1 --warmup 0 --problems $P --max-vertices $NV --no-cpu-baseline --no-microbench > $OUT/pmc_$C.log 2>&1
  echo "$C rc=$?"
done
python3 - "$TAG" "$P" "$NV" "$OUT" <<'PY'
import csv, glob, json, sys, collections
tag, P, NV, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
per = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"/tmp/pmcp_{tag}_{c}/**/*counter_collection.csv", recursive=True)
    tot = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] != c: continue
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        tot[k] += float(r["Counter_Value"]); n[k] += 1
    per[c] = {k: (tot[k], n[k]) for k in tot if "nn1_mirror" in k}
    top = sorted(tot.items(), key=lambda kv: -kv[1])[:8]
    print(c, [(k[-44:], round(v / 1e6, 2), n[k]) for k, v in top])
# rocprofv3 reports both counters in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request of a wide coalesced stream:
# x 2 (MI355X_MICROARCH.md, HBM section).  One "launch" of the NN search = its five kernels of one round.
kernels = sorted(set(per["FETCH_SIZE"]) | set(per["WRITE_SIZE"]))
rounds = max(v[1] for v in per["FETCH_SIZE"].values())
by_kernel, total = {}, 0.0
for k in kernels:
    f, nf = per["FETCH_SIZE"].get(k, (0.0, 1)); w, nw = per["WRITE_SIZE"].get(k, (0.0, 1))
    b = (2.0 * f / nf + w / nw) * 1024.0
    by_kernel[k] = b
    total += b
rec = {"kernel": "nn1_mirror_kernel (prep + pass 1 + thresholds + pass 2 + resolve)", "problems_per_gpu": P,
       "max_vertices": NV, "rounds": rounds, "fetch_correction": 2.0, "hbm_bytes_per_launch": total,
       "hbm_bytes_per_launch_by_kernel": by_kernel,
       "how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate runs of `bench.py --steps 1 --warmup 0 "
              "--no-cpu-baseline --no-microbench`; per kernel the mean over its dispatches, summed over the five kernels of a round"}
json.dump(rec, open(f"{out}/nn_planner_pmc.json", "w"), indent=1)
print(json.dumps(rec))
PY
