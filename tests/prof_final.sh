#!/bin/bash
# Round-end measurement set (GPU box): full bench line, kernel stats of one timed step, HBM traffic (PMC) of the planner-regime
# and of the HBM-regime NN sweep.  usage: tests/prof_final.sh <tag>   -> gpurun_out/<tag>_*
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
TAG=$1
mkdir -p $OUT
cd $ROOT
timeout -k 10 900 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err || { echo "bench failed"; tail -5 $OUT/${TAG}_bench.err; exit 1; }
python - <<PY
import json
d = json.load(open("$OUT/${TAG}_bench.json"))
print("value %.3fM  ms/step %.0f" % (d["value"] / 1e6, d["ms_per_step"]))
for k in ("roofline", "nn_sweep_mfma_timed", "nn_sweep_hbm", "nn_sweep_mfma"):
    print(k, {a: d[k][a] for a in ("achieved", "frac") if a in d[k]})
print("steer", d["steer_kernels"]["achieved"], d["steer_kernels"]["frac"])
PY
bash tests/prof_short.sh ${TAG} | tail -3
bash tests/prof_pmc_planner.sh ${TAG} | tail -2
# HBM-regime sweep (4 Mi x 12, 8 queries, unit-cube bound declared): FETCH_SIZE / WRITE_SIZE, separate passes
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d /tmp/pmcn_${TAG}_$C -o $TAG -- python $ROOT/tests/bench_nn_only.py 4194304 8 1.0 > $OUT/${TAG}_nnpmc_$C.log 2>&1
  echo "$C rc=$?"
done
python3 - "$TAG" "$OUT" <<'PY'
import csv, glob, json, sys, collections
tag, out = sys.argv[1], sys.argv[2]
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"/tmp/pmcn_{tag}_{c}/**/*counter_collection.csv", recursive=True)
    tot = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] != c: continue
        k = r["Kernel_Name"].split("(")[0]
        tot[k] += float(r["Counter_Value"]); n[k] += 1
    for k in tot:
        if "nn1_few_mfma_kernel" in k: res[c] = (k, tot[k], n[k])
kname, fetch_kb, nf = res["FETCH_SIZE"]; _, write_kb, nw = res["WRITE_SIZE"]
per_launch = (2.0 * fetch_kb / nf + write_kb / nw) * 1024.0
rec = {"kernel": kname.replace("void rkh::", ""), "n_rows": 4194304, "dims": 12, "queries_per_sweep": 8,
       "algorithmic_bytes": 4194304 * 12 * 8, "FETCH_SIZE_KiB_per_launch": fetch_kb / nf, "WRITE_SIZE_KiB_per_launch": write_kb / nw,
       "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request of a wide coalesced stream -> x2 (MI355X_MICROARCH.md, HBM); WRITE_SIZE exact",
       "hbm_bytes_per_launch": per_launch, "traffic_over_algorithmic": per_launch / (4194304 * 12 * 8),
       "coord_bound": 1.0,
       "command": "rocprofv3 --pmc FETCH_SIZE ... -- python tests/bench_nn_only.py 4194304 8 1.0 (and a second pass with --pmc WRITE_SIZE); mean over the launches of the kernel"}
json.dump(rec, open(f"{out}/{tag}_nn_sweep_pmc.json", "w"), indent=1)
print(json.dumps(rec))
PY
