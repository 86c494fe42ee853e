#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric on BASELINE config C2.

Workload (config.workload = "C2"): 6-DOF revolute KTE chain, RRT with RK4 forward-dynamics propagation
(dt = 1e-3, 20 steps/edge), 50 convex obstacles, goal probe per added vertex, run to --max-vertices.
One *step* = one complete rrt_planner::solve_planning_query pass for the rank's independent problems
(--problems seeds per GPU, each bit-for-bit the sequential planner on its seed).  `value` = valid node
expansions per second over all ranks (vertices added / wall time, max over ranks); edges-collision-checked/s
is reported next to it.  Inputs (scene, sample stream chunks) are device-resident before the timed region.

Multi-GPU: independent seeds shard across ranks ("scaling": "weak"), no data-path collective; after the timed
region one all-reduce(min) of the best solution cost and all-reduce(sum) of the counters.

Extra objects on the JSON line:
  roofline      NN-sweep kernel of the timed region against HBM, exactly as SURVEY.md 8(d) defines it: achieved =
                sweeps * n * D * 8 bytes / kernel time (HIP events on the planner stream), peak 8 TB/s; `traffic` = HBM bytes
                per launch from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the same workload
                (profiles/r02_nn_planner_pmc.json).  In the planner a sweep serves hundreds of queries per tree, so the
                kernel is bound by the (vertex, query) pair arithmetic, not by these bytes: see nn_sweep_mfma_timed.
  nn_sweep_mfma_timed  the same launches against the matrix pipes: algorithmic rate (24 flops per pair) against the dense
                f32-input MFMA peak, and the bf16 instructions the split estimate actually issues against the bf16 peak
  nn_sweep_hbm  the sweep in its HBM-bound regime (tree larger than the 256 MiB Infinity Cache, 8 queries per sweep, the
                cloud's hyperbox declared: matrix-core pre-filter in front of the exact fp64 test), measured outside the
                timed region; beside it 1 / 4 / 32 queries per sweep and the all-fp64 sweep without a declared bound
  nn_sweep_mfma the matrix-core sweep alone on one 1 Mi-row tree
  steer_kernels the dominant kernels of the timed region: share of the step time, edges/s inside the kernel, fp64
                operation rate against the no-FMA VALU peak with the EXACT operation count of one f-eval as the restated
                reference performs it (oracle/flop_count.cpp; `useful` = without its products over structural zeros)
  collide       the proximity test inside the steer kernels: (robot shape, obstacle) pair tests per second
                before culling, and the fraction each stage lets through (static reach, bounding cull, closed forms)
  single_problem  P = 1 and P = 16 (one / sixteen problems per GPU, 20 000 vertices each), outside the timed region
  cpu_baseline  the CPU oracle (restatement of the reference planner, -O3 -march=native) on a bounded sample of the
                same workload, single thread like ReaK itself; host CPU model and core count stated
  cpu_baseline_all_cores  the same sample on every host core (one oracle process per core, independent seeds)
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402


def hip_runtime():
    for name in ("libamdhip64.so", "/opt/rocm/lib/libamdhip64.so"):
        try:
            return C.CDLL(name)
        except OSError:
            continue
    raise RuntimeError("libamdhip64.so not found")


class HipEvents:
    """HIP events recorded on an explicit stream (torch.cuda.Event only sees torch's current stream)."""

    def __init__(self):
        self.hip = hip_runtime()
        self.hip.hipEventCreate.argtypes = [C.POINTER(C.c_void_p)]
        self.hip.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
        self.hip.hipEventSynchronize.argtypes = [C.c_void_p]
        self.hip.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]

    def create(self):
        ev = C.c_void_p()
        assert self.hip.hipEventCreate(C.byref(ev)) == 0
        return ev

    def record(self, ev, stream):
        assert self.hip.hipEventRecord(ev, C.c_void_p(stream)) == 0

    def elapsed_ms(self, a, b):
        self.hip.hipEventSynchronize(b)
        ms = C.c_float()
        assert self.hip.hipEventElapsedTime(C.byref(ms), a, b) == 0
        return float(ms.value)


def nn_sweep_microbench(lib, ctx, events, n_rows, B, reps, coord_bound=0.0):
    """HBM-bound regime of the NN sweep: n_rows x 12 fp64 (> 256 MiB), B queries per sweep.  coord_bound > 0: the cloud's
    hyperbox is declared (the unit cube here), which lets the sweep run its matrix-core pre-filter (same answers)."""
    D = 12
    nn = lib.HipNeighborSearch(ctx, D, n_rows)
    nn.fill_uniform(n_rows, seed=7)
    if coord_bound > 0.0:
        nn.set_coord_bound(coord_bound)
    import torch

    q = torch.rand(B, D, dtype=torch.float64, device="cuda")
    idx = torch.zeros(B, dtype=torch.int32, device="cuda")
    dist = torch.zeros(B, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    for _ in range(3):
        nn.nearest_async(q.data_ptr(), B, idx.data_ptr(), dist.data_ptr())
    ctx.synchronize()
    pairs = [(events.create(), events.create()) for _ in range(reps)]
    for a, b in pairs:  # each sweep kernel is bracketed by its own event pair on the launch stream
        nn.nearest_async(q.data_ptr(), B, idx.data_ptr(), dist.data_ptr(), events=(a, b))
    ctx.synchronize()
    ms = sum(events.elapsed_ms(a, b) for a, b in pairs) / reps
    bytes_per_sweep = n_rows * D * 8
    gbps = bytes_per_sweep / (ms * 1e-3) / 1e9
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "r02_nn_sweep_pmc.json")
    if os.path.exists(pmc):  # HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (same kernel, same n)
        rec = json.load(open(pmc))
        if rec.get("n_rows") == n_rows and rec.get("queries_per_sweep") == B and rec["kernel"].startswith(nn.kernel_name()):
            traffic = rec["hbm_bytes_per_launch"]
    name = nn.kernel_name()
    nn.close()
    return {"kernel": name, "bound": "hbm", "n": n_rows, "dims": D, "queries_per_sweep": B, "ms_per_sweep": ms, "achieved": gbps,
            "peak": 8000.0, "unit": "GB/s", "frac": gbps / 8000.0, "queries_per_s": B / (ms * 1e-3),
            "algorithmic_bytes": bytes_per_sweep, "traffic": traffic}


def mfma_view(kernel, tflops_algorithmic, dims, pairs_per_launch=None):
    """The many-queries NN sweep against the matrix pipes.  `achieved` stays the ALGORITHMIC rate -- 2 * Dp flops per (vertex,
    query) pair, the rank-Dp product the estimate is -- against the dense f32-input MFMA peak, the precision the estimate
    has to bound and the rate round 1 priced it at.  Since round 2 the kernel runs that product as a split-bf16 estimate
    on v_mfma_f32_32x32x16_bf16 (three k-slots per coordinate + three for |x|^2, 8 per lane half and instruction):
    `executed` is what the pipe actually does, against the dense bf16 peak."""
    dp = next(d for d in (2, 4, 6, 8, 12, 16, 24, 32) if dims <= d)
    out = {"kernel": kernel, "bound": "mfma", "achieved": tflops_algorithmic, "peak": 157.3, "unit": "TFLOP/s",
           "frac": tflops_algorithmic / 157.3, "flops_per_pair_algorithmic": 2 * dp}
    if pairs_per_launch is not None:
        out["pairs_per_launch"] = pairs_per_launch
    if "bf16" in kernel:
        instr = (3 * (dp // 2) + 3 + 7) // 8                 # matrix instructions per 32 x 32 block
        per_pair = instr * 32 * 32 * 16 * 2 / 1024.0          # executed bf16 flops per pair
        ex = tflops_algorithmic / (2 * dp) * per_pair
        out["executed"] = {"instruction": "v_mfma_f32_32x32x16_bf16", "per_32x32_block": instr, "flops_per_pair": per_pair,
                           "achieved": ex, "peak": 2500.0, "frac": ex / 2500.0}
        out["note"] = ("split-bf16 estimate on the matrix cores (bit-identical answers: the exact fp64 recheck of the "
                       "survivors is not counted); frac = algorithmic rate / dense f32-input MFMA peak, executed = the "
                       "bf16 instructions actually issued against the dense bf16 peak")
    else:
        out["note"] = "2 * Dp flops per pair on v_mfma_f32_32x32x2_f32; the survivors' exact fp64 recheck not counted"
    return out


def nn_mfma_microbench(lib, ctx, events, n_rows=1 << 20, B=1024, reps=10):
    """The matrix-core pre-filter sweep alone on one large tree (exclusive use of the GPU, outside the timed region)."""
    import torch

    D = 12
    nn = lib.HipNeighborSearch(ctx, D, n_rows)
    nn.fill_uniform(n_rows, seed=7)
    nn.set_coord_bound(1.0)
    q = torch.rand(B, D, dtype=torch.float64, device="cuda")
    idx = torch.zeros(B, dtype=torch.int32, device="cuda")
    dist = torch.zeros(B, dtype=torch.float64, device="cuda")
    for _ in range(3):
        nn.nearest_async(q.data_ptr(), B, idx.data_ptr(), dist.data_ptr())
    ctx.synchronize()
    pairs = [(events.create(), events.create()) for _ in range(reps)]
    for a, b in pairs:
        nn.nearest_async(q.data_ptr(), B, idx.data_ptr(), dist.data_ptr(), events=(a, b))
    ctx.synchronize()
    ms = sum(events.elapsed_ms(a, b) for a, b in pairs) / reps
    name = nn.kernel_name()
    nn.close()
    tf = n_rows * B * 24.0 / (ms * 1e-3) / 1e12
    out = mfma_view(name, tf, D)
    out.update({"n": n_rows, "dims": D, "queries_per_sweep": B, "ms_per_sweep": ms})
    return out


def nn_published_config(lib, ctx, events, n_rows=25000, D=6, B=1000, reps=20):
    """the reference's NN-search table is quoted at 6-D, N = 25 000 vertices, in us per query per vertex
    (linear search 8.2e-3, DVP-tree 1.6e-4 on the authors' 2012 desktop).  The same figure for the GPU sweep, and for the
    oracle's linear search and its restated DVP-tree (oracle/dvp_tree.hpp) on THIS host's CPU, same points and queries."""
    import torch
    import oracle_lib

    rng = np.random.default_rng(5)
    pts = rng.uniform(-1.0, 1.0, size=(n_rows, D))
    q = rng.uniform(-1.0, 1.0, size=(B, D))
    nn = lib.HipNeighborSearch(ctx, D, n_rows)
    nn.added_vertices(pts)
    nn.set_coord_bound(1.0)  # the cloud's hyperbox: lets the sweep run its single-precision pre-filter (same answers)
    dq = torch.from_numpy(q).cuda()
    idx = torch.zeros(B, dtype=torch.int32, device="cuda")
    dist = torch.zeros(B, dtype=torch.float64, device="cuda")
    for _ in range(3):
        nn.nearest_async(dq.data_ptr(), B, idx.data_ptr(), dist.data_ptr())
    ctx.synchronize()
    pairs = [(events.create(), events.create()) for _ in range(reps)]
    for a, b in pairs:
        nn.nearest_async(dq.data_ptr(), B, idx.data_ptr(), dist.data_ptr(), events=(a, b))
    ctx.synchronize()
    ms = sum(events.elapsed_ms(a, b) for a, b in pairs) / reps
    name = nn.kernel_name()
    gpu_idx = idx.cpu().numpy().astype(np.int64)
    nn.close()
    per = lambda seconds: seconds * 1e6 / B / n_rows
    t0 = time.perf_counter()
    lin_idx, _ = oracle_lib.nn1(q, pts, fast=True)
    t_lin = time.perf_counter() - t0
    out = {"n": n_rows, "dims": D, "queries": B, "unit": "us per query per vertex",
           "gpu_sweep": {"kernel": name, "value": per(ms * 1e-3), "us_per_batch": ms * 1e3},
           "cpu_linear_search": {"value": per(t_lin)},
           "reference_2012_desktop": {"linear_search": 8.2e-3, "dvp_tree": 1.6e-4, "source": "BASELINE.md 1"},
           "same_answers": bool(np.array_equal(gpu_idx, lin_idx.astype(np.int64))), "host": host_description()}
    for arity in (2, 4):
        for inc in (False, True):
            t_idx, _, info = oracle_lib.dvptree(q, pts, arity=arity, incremental=inc, seed=1, fast=True)
            out[f"cpu_dvp_tree_arity{arity}_{'incremental' if inc else 'bulk'}"] = {
                "value": per(info["query_s"]), "build_s": info["build_s"],
                "distance_evaluations_per_query": info["dist_evals"] / B,
                "same_answers": bool(np.array_equal(t_idx.astype(np.int64), lin_idx.astype(np.int64)))}
    return out


def cpu_quota():
    """CPUs this process may actually use: the affinity mask capped by the cgroup's CPU quota (a GPU box hands one GPU's
    share of the host, 16 CPUs, to a job whose affinity mask still lists every core of the machine)."""
    n = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per) + 0.5)))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, int(q / per + 0.5)))
        except (OSError, ValueError):
            pass
    return n


def host_description():
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"cpu_model": model, "nproc": os.cpu_count(), "affinity_cores": len(os.sched_getaffinity(0)),
            "cgroup_cpu_quota": cpu_quota()}


def cpu_baseline(lib, scene, scn, max_vertices, seconds_target=18.0):
    """Oracle (kind 'port': the reference cannot be built here) on a bounded sample of the SAME workload,
    measured at the benchmark's own tree sizes.  The sequential planner's cost per iteration grows with the tree (linear-
    search NN), so a from-scratch run of a few seconds only sees a young tree.  Instead one tree is grown to max_vertices
    on the GPU (same world, its own seed) and the oracle's loop is started on prefixes of it (orc_rrt_dyn_warm: a prefix
    of an RRT is the tree at that moment): the rate at 1 k, 10 k, 30 k and max_vertices vertices is MEASURED, the rate of
    a whole run to max_vertices is their integral sum(dn / rate(n)) -- `value`."""
    import oracle_lib

    osc = oracle_lib.OracleScene(scn, fast=True)
    pl = lib.RrtPlanner(scene, scn.rrt_params(seed=424242, max_vertices=max_vertices))
    pl.solve_planning_query()
    pos = pl.tree()["pos"]
    pl.close()
    sizes = [n for n in (1000, 10000, 30000) if n < max_vertices] + [max_vertices]
    share = seconds_target / len(sizes)
    pts, it_tot, sec_tot, edges_tot = [], 0, 0.0, 0
    for n in sizes:
        prm = scn.rrt_params(seed=7 + n, max_vertices=2 ** 31 - 1)
        rc, o = osc.rrt_dyn_warm(prm, pos[1:n + 1], 64)  # calibration: cost of an iteration at this size
        iters = int(max(64, min(20000, share / max(o.seconds / 64, 1e-6))))
        rc, o = osc.rrt_dyn_warm(prm, pos[1:n + 1], iters)
        assert rc == 0
        added = int(o.num_vertices) - 1 - n
        pts.append({"tree_size": n, "iterations": int(o.iterations), "vertices_added": added, "seconds": float(o.seconds),
                    "expansions_per_s": added / o.seconds, "edges_checked_per_s": int(o.edges_checked) / o.seconds})
        it_tot += int(o.iterations)
        sec_tot += float(o.seconds)
        edges_tot += int(o.edges_checked)
    # time of a whole run: trapezoid of 1 / rate over the tree size (rate falls monotonically with n)
    ns = [0] + [p_["tree_size"] for p_ in pts]
    inv = [1.0 / pts[0]["expansions_per_s"]] + [1.0 / p_["expansions_per_s"] for p_ in pts]
    t_run = sum(0.5 * (inv[i] + inv[i + 1]) * (ns[i + 1] - ns[i]) for i in range(len(ns) - 1))
    whole = max_vertices / t_run
    at_full = pts[-1]
    return {"value": whole, "unit": "valid node expansions/s", "cores": 1, "kind": "port",
            "edges_checked_per_s": whole * at_full["edges_checked_per_s"] / at_full["expansions_per_s"],
            "at_final_tree_size": at_full, "by_tree_size": pts,
            "sample": f"the same C2 world; the oracle's sequential RRT loop (-O3 -march=native, 1 thread: ReaK is single-"
                      f"threaded) MEASURED on a tree of {max_vertices} vertices ({at_full['iterations']} iterations, "
                      f"{at_full['seconds']:.1f} s) and on its prefixes of {', '.join(str(n) for n in sizes[:-1])} vertices "
                      f"({it_tot} iterations, {sec_tot:.1f} s in all); value = {max_vertices} / sum(dn / rate(n)), the rate of a "
                      f"whole run to {max_vertices} vertices",
            "host": host_description(),
            "note": "ReaK planner, CPU restatement (reference binary unavailable: needs Boost + BGL-Extra); the tree the "
                    "loop starts on was grown by the GPU planner on another seed of the same world"}


def cpu_baseline_all_cores(lib, scene, scn, max_vertices, seconds_target=15.0):
    """The final-tree-size leg of cpu_baseline on every CPU the job may use (cgroup quota): one oracle process per CPU,
    independent seeds (the reference's own evaluation mode is independent Monte-Carlo runs,
    planner_exec_engines.hpp:139-206), each on the same 100 000-vertex tree."""
    import subprocess
    import tempfile

    cores = cpu_quota()
    pl = lib.RrtPlanner(scene, scn.rrt_params(seed=424243, max_vertices=max_vertices))
    pl.solve_planning_query()
    pos = pl.tree()["pos"]
    pl.close()
    worker = os.path.join(ROOT, "tests", "cpu_worker.py")
    with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as td:
        path = os.path.join(td, "tree.npy")
        np.save(path, pos[1:max_vertices + 1])
        t0 = time.perf_counter()
        procs = [subprocess.Popen([sys.executable, worker, str(1 + i), path, str(seconds_target)], stdout=subprocess.PIPE,
                                  stderr=subprocess.DEVNULL) for i in range(cores)]
        outs = [json.loads(p.communicate()[0].decode().strip().splitlines()[-1]) for p in procs]
        wall = time.perf_counter() - t0
    busy = max(o["seconds"] for o in outs)
    nodes = sum(o["vertices_added"] for o in outs)
    return {"value": nodes / busy, "unit": "valid node expansions/s", "cores": cores, "kind": "port",
            "edges_checked_per_s": sum(o["edges"] for o in outs) / busy,
            "sample": f"{cores} processes (the job's CPU quota; affinity mask {len(os.sched_getaffinity(0))} of "
                      f"{os.cpu_count()} cores) x seeds 1..{cores}, each the sequential loop on a tree of {max_vertices} "
                      f"vertices ({busy:.1f} s planner time, {wall:.1f} s wall incl. process start), oracle -O3 -march=native",
            "tree_size": max_vertices, "host": host_description()}


def feval_ops(scn, samples=32):
    """Exact fp64 operation count of one x' = f(x, u) of the restated reference (mean over random states of the C2 box)."""
    import oracle_lib

    rng = np.random.default_rng(11)
    lo = np.array([scn.dyn.lower[i] for i in range(2 * scn.n_dof)])
    hi = np.array([scn.dyn.upper[i] for i in range(2 * scn.n_dof)])
    acc = {}
    for _ in range(samples):
        c = oracle_lib.feval_op_count(scn, rng.uniform(lo, hi), rng.uniform(-50.0, 50.0, size=scn.n_dof))
        for k, v in c.items():
            acc[k] = acc.get(k, 0) + v
    return {k: v / samples for k, v in acc.items()}


def steer_occupancy():
    """Resource usage of the steer kernels as the built code objects state it (tools/kernel_resources.py reads the
    AMDGPU metadata of reak_amd/librkh.so; tests/test_kernel_resources.py pins the same figures)."""
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import kernel_resources as kr

        res = kr.kernel_resources()
        out = {}
        for k in ("rkh::propagate_pair_step_kernel<6>", "rkh::propagate_pair_kernel<6>", "rkh::propagate_kernel<6, 64, false, false>",
                  "rkh::propagate_kernel<6, 64, false, true>"):
            d = res[k]
            lds_waves = (160 * 1024 // d["group_segment_fixed_size"]) if d["group_segment_fixed_size"] else 32
            out[k.replace("rkh::", "")] = {"vgpr": d["vgpr_count"], "vgpr_spills": d["vgpr_spill_count"],
                                           "scratch_bytes": d["private_segment_fixed_size"],
                                           "lds_bytes": d["group_segment_fixed_size"],
                                           "waves_per_simd": min(kr.waves_per_simd(d), max(1, lds_waves // 4))}
        out["mapping"] = ("32 edges per wave (two adjacent lanes per edge); every launch of the step-wise form packs the "
                          "live edges of all problems into full waves")
        return out
    except Exception as e:  # the llvm tools are part of the ROCm image; without them the figures are simply not quoted
        return {"unavailable": repr(e)}


def nn_planner_traffic(problems, max_vertices):
    """HBM bytes per launch of the planner-regime NN sweep from the committed PMC passes, if they match this run."""
    path = os.path.join(ROOT, "profiles", "r03_nn_planner_pmc.json")
    if not os.path.exists(path):
        return None
    rec = json.load(open(path))
    if rec.get("problems_per_gpu") == problems and rec.get("max_vertices") == max_vertices:
        return rec.get("hbm_bytes_per_launch")
    return None


def single_problem_rate(lib, scene, scn, P, max_vertices):
    """P problems per GPU (P = 1: the latency-bound case), wall clock of a whole solve."""
    prms = [scn.rrt_params(seed=5000 + i, max_vertices=max_vertices) for i in range(P)]
    pl = lib.RrtPlannerPool(scene, prms, groups=1)
    pl.enqueue(0)
    t0 = time.perf_counter()
    pl.solve_planning_query()
    dt = time.perf_counter() - t0
    nodes = sum(int(st.num_vertices) - 1 for st in pl.all_stats)
    edges = sum(int(st.edges_checked) for st in pl.all_stats)
    pl.close()
    return {"problems": P, "max_vertices": max_vertices, "value": nodes / dt, "unit": "valid node expansions/s",
            "edges_collision_checked_per_s": edges / dt, "seconds": dt}


def collide_counts(lib, scene, scn, configs_per_s):
    """SURVEY 8(d) `collide`: pair tests per second of the steer kernels' proximity test and what each stage of it culls.
    The stage counts come from a diagnostic launch of the same proximity code (rkh_diag_proximity_counts) on the vertices
    of a 20 000-vertex tree of this world -- collision-free states, where the kernels spend their tests; the rate is the
    configurations the timed region tested (one per executed RK4 step) x the scene's proxy pairs."""
    pl = lib.RrtPlanner(scene, scn.rrt_params(seed=7000, max_vertices=20000))
    pl.solve_planning_query()
    states = pl.tree()["pos"]
    pl.close()
    c = scene.proximity_counts(states)
    per = float(c["states"]) * c["pairs_per_state"]
    # the same counts on states drawn uniformly from the state box: what the cull lets through when the arm is anywhere
    rng = np.random.default_rng(3)
    lo = np.array([scn.dyn.lower[i] for i in range(2 * scn.n_dof)])
    hi = np.array([scn.dyn.upper[i] for i in range(2 * scn.n_dof)])
    u = scene.proximity_counts(rng.uniform(lo, hi, size=(65536, 2 * scn.n_dof)))
    uper = float(u["states"]) * u["pairs_per_state"]
    uniform = {"states": u["states"], "fraction_past_bounding_cull": u["pairs_past_cull"] / uper,
               "fraction_closed_form": u["closed_forms"] / uper, "fraction_golden_section": u["golden_section"] / uper,
               "states_in_collision": u["states_in_collision"] / float(u["states"])}
    return {"pair_tests_per_s": configs_per_s * c["pairs_per_state"], "configurations_per_s": configs_per_s,
            "uniform_states": uniform,
            "pairs_per_configuration": c["pairs_per_state"],
            "fraction_within_static_reach": c["pairs_in_static_reach"] / max(1, c["pairs_per_state"]),
            "fraction_past_bounding_cull": c["pairs_past_cull"] / per, "fraction_closed_form": c["closed_forms"] / per,
            "fraction_golden_section": c["golden_section"] / per, "culled_fraction": 1.0 - c["pairs_past_cull"] / per,
            "sample": "%d vertices of a 20000-vertex tree of the same world (%d of them in collision by the diagnostic "
                      "launch)" % (c["states"], c["states_in_collision"]),
            "note": "pair tests = (robot shape, obstacle) pairs of the scene x configurations tested by the steer kernels in "
                    "the timed region, before any culling (SURVEY 8(d)); stages: static reach prefix -> fp32 bounding cull -> "
                    "closed forms (capped cylinder / box pairs: separating-axis screen, then the golden-section search)"}


def c3_rrtstar_rate(lib, ctx, events, P=16, max_vertices=20000, knn_n=1 << 20):
    """BASELINE config C3 beside the headline: RRT* (quasi-static space of the same 6-DOF chain, star_neighborhood k-NN
    rewiring) for P problems, and the k-NN sweep alone on a 1 Mi-vertex tree (k = 4 (floor(log2 n) + 1), star radius),
    HIP-event timed, against the HBM peak by SURVEY 8(d): n * D * 8 algorithmic bytes per query batch."""
    import torch

    from reak_amd import scenarios

    scn = scenarios.make_c3()
    sc = lib.Scene(ctx, scn)
    lo, hi, mi = scn.meta["lower"], scn.meta["upper"], scn.meta["min_interval"]
    qs = lib.make_qs_space(6, lo, hi, mi)
    pl = lib.RrtStarPlanner(sc, [scn.rrt_params(seed=9000 + i, max_vertices=max_vertices) for i in range(P)], qs)
    t0 = time.perf_counter()
    pl.solve_planning_query()
    dt = time.perf_counter() - t0
    it = sum(int(s.loop_iterations) for s in pl.all_stats)
    ed = sum(int(s.edges_checked) for s in pl.all_stats)
    rw = sum(int(s.rewires) for s in pl.all_stats)
    pl.close()
    def knn_sweep(D):
        nn = lib.HipNeighborSearch(ctx, D, knn_n)
        nn.fill_uniform(knn_n, seed=3)
        logn = int(np.floor(np.log2(knn_n))) + 1
        k, radius = 4 * logn, 3.0 * (logn / knn_n) ** (1.0 / D)
        B = 8
        q = torch.rand(B, D, dtype=torch.float64, device="cuda")
        idx = torch.zeros(B, k, dtype=torch.int32, device="cuda")
        dist = torch.zeros(B, k, dtype=torch.float64, device="cuda")
        cnt = torch.zeros(B, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        call = lambda: lib._check(nn.lib.rkh_nn_queryk_async(nn.h, q.data_ptr(), B, k, float(radius), idx.data_ptr(),
                                                             dist.data_ptr(), cnt.data_ptr()))
        for _ in range(3):
            call()
        ctx.synchronize()
        a, b = events.create(), events.create()
        events.record(a, ctx.stream)
        reps = 20
        for _ in range(reps):
            call()
        events.record(b, ctx.stream)
        ctx.synchronize()
        ms = events.elapsed_ms(a, b) / reps
        nn.close()
        gbps = knn_n * D * 8 / (ms * 1e-3) / 1e9
        return {"n": knn_n, "dims": D, "k": k, "queries_per_sweep": B, "us_per_batch": ms * 1e3, "bound": "hbm",
                "achieved": gbps, "peak": 8000.0, "unit": "GB/s", "frac": gbps / 8000.0,
                "tree_bytes": knn_n * D * 8,
                "note": "algorithmic bytes n * D * 8 per batch (the kernel makes two passes over the rows: bound sweep + "
                        "collect sweep)"}

    return {"workload": "C3: 6-DOF chain, RRT* (quasi-static space), star_neighborhood k-NN rewiring", "problems": P,
            "max_vertices": max_vertices, "iterations_per_s": it / dt, "edges_collision_checked_per_s": ed / dt,
            "rewires": rw, "seconds": dt,
            # 6-D: the quasi-static space the RRT* above plans in (48 MB tree, Infinity-Cache resident); 12-D: the state
            # space of the same chain (q, qd) -- the 96 MB tree SURVEY 8(a3)/(d) sizes C3 at
            "knn_sweep": knn_sweep(6), "knn_sweep_d12": knn_sweep(12)}


def c4_prm_rate(lib, ctx, P=16, max_vertices=1500):
    """BASELINE config C4 as written: 12-DOF dual arm, PRM, 200 convex MESH obstacles (batched GJK)."""
    from reak_amd import scenarios

    scn = scenarios.make_c4(world_seed=1, meshes=True)
    sc = lib.Scene(ctx, scn)
    lo, hi, mi = scn.meta["lower"], scn.meta["upper"], scn.meta["min_interval"]
    qs = lib.make_qs_space(12, lo, hi, mi)
    prms = [scn.prm_params(seed=9100 + i, max_vertices=max_vertices, sampling_radius=1.5) for i in range(P)]
    pl = lib.PrmPlanner(sc, prms, qs)
    t0 = time.perf_counter()
    pl.solve_planning_query()
    dt = time.perf_counter() - t0
    it = sum(int(s.loop_iterations) for s in pl.all_stats)
    ed = sum(int(s.edges_checked) for s in pl.all_stats)
    nv = sum(int(s.num_vertices) for s in pl.all_stats)
    pl.close()
    return {"workload": "C4: 12-DOF dual arm, PRM (quasi-static space), 200 convex mesh obstacles of 12-32 vertices, GJK",
            "problems": P, "max_vertices": max_vertices, "iterations_per_s": it / dt,
            "edges_collision_checked_per_s": ed / dt, "roadmap_vertices_per_s": nv / dt, "seconds": dt}


def run_c5(args, rank, world, local_rank, dist, reduce_device, lib, scenarios, dist_utils, torch):
    """BASELINE config C5: independent RRT* seeds (C3 world), --c5-problems per GPU, --c5-vertices each, sharded over the
    ranks with no data-path collective; afterwards all-reduce(min) of the best solution cost and all-reduce(sum) of the
    counters.  A step = one complete solve of the rank's seeds."""
    ctx = lib.Context(local_rank)
    scn = scenarios.make_c3()
    scene = lib.Scene(ctx, scn)
    lo, hi, mi = scn.meta["lower"], scn.meta["upper"], scn.meta["min_interval"]
    qs = lib.make_qs_space(6, lo, hi, mi)
    P = args.c5_problems

    def run_step(step_index):
        seeds = dist_utils.seeds_for_rank(step_index, rank, world, P)
        pl = lib.RrtStarPlanner(scene, [scn.rrt_params(seed=s, max_vertices=args.c5_vertices) for s in seeds], qs)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        if dist is None:
            pl.solve_planning_query()
        else:
            # RRT* resumes where it stopped: plan in slices of loop iterations and reduce tree sizes / edges (sum) and the
            # best cost (min) over the ranks after each; done ranks keep answering until all are
            done, budget = False, 0
            while True:
                if not done:
                    budget += args.c5_report_iterations
                    pl.solve_planning_query(max_loop_iterations=budget)
                    done = all(int(s.num_vertices) >= args.c5_vertices + 2 for s in pl.all_stats)
                all_done, *_ = dist_utils.progress_reduce(
                    dist, done, sum(int(s.num_vertices) for s in pl.all_stats), sum(int(s.edges_checked) for s in pl.all_stats),
                    min(float(s.best_cost) for s in pl.all_stats), reduce_device)
                if all_done:
                    break
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        r = {"seconds": dt, "nodes": sum(int(s.num_vertices) for s in pl.all_stats),
             "edges": sum(int(s.edges_checked) for s in pl.all_stats),
             "iters": sum(int(s.loop_iterations) for s in pl.all_stats),
             "best": min(float(s.best_cost) for s in pl.all_stats)}
        pl.close()
        return r

    for w in range(args.warmup):
        run_step(1000 + w)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t_begin = time.perf_counter()
    tot = {"nodes": 0, "edges": 0, "iters": 0}
    best = float("inf")
    for k in range(args.steps):
        r = run_step(k)
        for key in tot:
            tot[key] += r[key]
        best = min(best, r["best"])
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t_begin
    elapsed, nodes_all, edges_all, iters_all, best = dist_utils.reduce_results(dist, elapsed, tot["nodes"], tot["edges"],
                                                                               tot["iters"], best, reduce_device)
    if rank == 0:
        print(json.dumps({
            "metric": "RRT* loop iterations/sec (+ edges-collision-checked/sec), config C5", "value": iters_all / elapsed,
            "unit": "RRT* iterations/s", "edges_collision_checked_per_s": edges_all / elapsed,
            "vertices_per_s": nodes_all / elapsed, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / max(1, args.steps) * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "C5: 6-DOF chain, RRT* (quasi-static space, star_neighborhood k-NN rewiring), "
                                   "independent seeds sharded one batch per GPU, best-cost all-reduce(min)",
                       "vertices_per_seed": args.c5_vertices, "seeds_per_gpu": P,
                       "parallelism": f"{world} x {P} independent RRT* planners"},
            "best_solution_cost": None if best == float("inf") else best}))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--problems", type=int, default=int(os.environ.get("RKH_BENCH_PROBLEMS", "512")))
    ap.add_argument("--max-vertices", type=int, default=100000)
    ap.add_argument("--groups", type=int, default=int(os.environ.get("RKH_BENCH_GROUPS", "1")),
                    help="planner handles (HIP streams) the problems of a GPU are split over; 2 overlaps one group's "
                         "steer tail with the other's NN sweep, which paid before the steer waves were packed and the "
                         "batches fitted to whole passes of the machine; now one group is faster (DESIGN.md section 5)")
    ap.add_argument("--rounds-per-sync", type=int, default=16)
    ap.add_argument("--report-interval", type=int, default=4,
                    help="N > 1: all-reduce {vertices, edges} (sum) and the best cost (min) every this many syncs")
    ap.add_argument("--workload", choices=["c2", "c5"], default="c2",
                    help="c2 (default, BASELINE.json's metric): RRT with RK4 dynamics; c5: BASELINE config C5 -- independent "
                         "RRT* seeds sharded over the ranks (one planner batch per GPU), best-cost all-reduce (min)")
    ap.add_argument("--c5-vertices", type=int, default=1000000, help="vertices per RRT* seed of --workload c5")
    ap.add_argument("--c5-problems", type=int, default=1, help="RRT* seeds per GPU of --workload c5")
    ap.add_argument("--c5-report-iterations", type=int, default=20000,
                    help="--workload c5, N > 1: RRT* loop iterations between two report-interval reductions")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-microbench", action="store_true")
    args = ap.parse_args()

    # `python bench.py --gpus N` without a launcher: start the N ranks here (one process per GPU, RCCL rendezvous on
    # 127.0.0.1) before anything touches the GPU, and relay their output; under torchrun the world must match --gpus.
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        import socket
        import subprocess

        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and not (world == 1 and args.gpus <= 1):
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} "
                 f"(or run `python bench.py --gpus {args.gpus}` without a launcher)")
    import torch

    # RKH_BENCH_BACKEND=gloo + RKH_BENCH_SHARE_GPU=1: rehearsal of the N > 1 path on a box with fewer GPUs than ranks
    # (ranks share the card, reductions go through gloo on the CPU); the driver's runs use RCCL, one rank per GPU.
    backend = os.environ.get("RKH_BENCH_BACKEND", "nccl")
    if os.environ.get("RKH_BENCH_SHARE_GPU") == "1":
        local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    reduce_device = torch.device("cuda", local_rank) if backend == "nccl" else torch.device("cpu")

    from reak_amd import dist_utils, lib, scenarios

    if args.workload == "c5":
        return run_c5(args, rank, world, local_rank, dist, reduce_device, lib, scenarios, dist_utils, torch)
    os.environ.setdefault("RKH_PROFILE_NN", "1")
    ctx = lib.Context(local_rank)
    events = HipEvents()
    scn = scenarios.make_c2(world_seed=1)
    scene = lib.Scene(ctx, scn)
    P = args.problems

    nn_kernel = ["nn1_sweep_kernel"]
    interval_reports = [0]

    def run_step(step_index, timed):
        seeds = dist_utils.seeds_for_rank(step_index, rank, world, P)
        pl = lib.RrtPlannerPool(scene, [scn.rrt_params(seed=s, max_vertices=args.max_vertices) for s in seeds],
                                groups=args.groups)
        pl.enqueue(0)  # sample chunks resident before the clock starts
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n_sync = 0
        while True:
            if not pl.done:
                pl.enqueue(args.rounds_per_sync)
                pl.sync()
            n_sync += 1
            if dist is None:
                if pl.done:
                    break
            elif n_sync % args.report_interval == 0:
                # the report-interval reductions of SURVEY.md 8(e), inside the timed region: tree sizes / edges (sum) and
                # best solution cost (min) over all ranks; a rank that is done keeps answering until every rank is
                all_done, *_ = dist_utils.progress_reduce(
                    dist, pl.done, sum(int(st.num_vertices) - 1 for st in pl.all_stats),
                    sum(int(st.edges_checked) for st in pl.all_stats), min(float(st.best_cost) for st in pl.all_stats),
                    reduce_device)
                interval_reports[0] += 1
                if all_done:
                    break
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        nodes = sum(int(st.num_vertices) - 1 for st in pl.all_stats)
        edges = sum(int(st.edges_checked) for st in pl.all_stats)
        spec = sum(int(st.edges_speculated) for st in pl.all_stats)
        rounds = max(int(st.rounds) for st in pl.all_stats)
        best = min(float(st.best_cost) for st in pl.all_stats)
        prof = [pl.nn_profile()]
        nn_pairs = pl.nn_pairs()
        steer_ms, steer_launches = pl.steer_profile()
        steer_steps = pl.steer_steps()
        nn_kernel[0] = lib.load().rkh_nn_kernel_name().decode()
        pl.close()
        return {"seconds": t1 - t0, "nodes": nodes, "edges": edges, "spec": spec, "rounds": rounds, "best": best,
                "nn_ms": sum(p[0] for p in prof), "nn_bytes": sum(p[1] for p in prof), "nn_launches": sum(p[2] for p in prof),
                "nn_pairs": nn_pairs, "steer_ms": steer_ms, "steer_launches": steer_launches, "steer_steps": steer_steps}

    for w in range(args.warmup):
        run_step(1000 + w, False)
    tot = {"seconds": 0.0, "nodes": 0, "edges": 0, "spec": 0, "rounds": 0, "nn_ms": 0.0, "nn_bytes": 0, "nn_launches": 0,
           "nn_pairs": 0, "steer_ms": 0.0, "steer_launches": 0, "steer_steps": 0}
    best = float("inf")
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t_begin = time.perf_counter()
    for k in range(args.steps):
        r = run_step(k, True)
        for key in tot:
            tot[key] += r[key]
        best = min(best, r["best"])
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t_begin

    # tree-size / edge counters (sum) and best solution cost (min) over all seeds: the only collectives
    elapsed, nodes_all, edges_all, spec_all, best = dist_utils.reduce_results(
        dist, elapsed, tot["nodes"], tot["edges"], tot["spec"], best, reduce_device)

    if rank == 0:
        nn_gbps = (tot["nn_bytes"] / (tot["nn_ms"] * 1e-3) / 1e9) if tot["nn_ms"] > 0 else 0.0
        nn_tflops = (tot["nn_pairs"] * 24.0 / (tot["nn_ms"] * 1e-3) / 1e12) if tot["nn_ms"] > 0 else 0.0
        nn_traffic = nn_planner_traffic(P, args.max_vertices)
        out = {
            "metric": "valid RRT node-expansions/sec (+ edges-collision-checked/sec)",
            "value": nodes_all / elapsed,
            "unit": "valid node expansions/s",
            "edges_collision_checked_per_s": edges_all / elapsed,
            "edges_propagated_per_s": spec_all / elapsed,
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / max(1, args.steps) * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "C2: 6-DOF revolute KTE chain, RRT + RK4 dynamics (dt=1e-3, 20 steps/edge), 50 convex "
                                   "obstacles, goal probe per vertex, LINEAR_SEARCH_KNN semantics",
                       "max_vertices": args.max_vertices, "problems_per_gpu": P, "seeds": "independent per problem",
                       "parallelism": f"{world} x {P} independent planners",
                       "planner_groups_per_gpu": args.groups},
            "rounds": tot["rounds"],
            "interval_reductions": interval_reports[0],
            "speculation_efficiency": (tot["edges"] / tot["spec"]) if tot["spec"] else None,
            "best_solution_cost": None if best == float("inf") else best,
            # NN sweep of the timed region (rank 0) by SURVEY.md 8(d): algorithmic bytes = n * D * 8 per swept tree, against
            # the 8 TB/s HBM peak.  Every launch sweeps the trees of all problems for a whole speculative batch of queries
            # each (a few hundred per tree), which makes the kernel compute-bound: the HBM fraction is low by design and
            # the matrix-core view of the same launches is nn_sweep_mfma_timed.
            "roofline": {"kernel": nn_kernel[0], "bound": "hbm", "achieved": nn_gbps, "peak": 8000.0, "unit": "GB/s",
                         "frac": nn_gbps / 8000.0, "traffic": nn_traffic, "launches": tot["nn_launches"],
                         "algorithmic_bytes_per_launch": (tot["nn_bytes"] / tot["nn_launches"]) if tot["nn_launches"] else None,
                         "avg_launch_us": (tot["nn_ms"] * 1e3 / tot["nn_launches"]) if tot["nn_launches"] else None,
                         "queries_per_launch": tot["spec"] / max(1, tot["nn_launches"]),
                         "note": "algorithmic bytes (sum over the swept trees of n * D * 8, SURVEY 8(d)) / HIP-event kernel "
                                 "time on the planner stream; traffic = HBM bytes per launch, FETCH_SIZE x 2 + WRITE_SIZE of "
                                 "separate --pmc passes of the same code (tools/prof_pmc_nn.sh -> profiles/"
                                 "r03_nn_planner_pmc.json; null if not collected for this configuration); a launch = the "
                                 "five kernels of one round's NN search (B operands, pass 1, thresholds, pass 2, exact "
                                 "resolution), HIP events around all five"},
            "nn_sweep_mfma_timed": mfma_view(nn_kernel[0], nn_tflops, scn.n_dof * 2,
                                             (tot["nn_pairs"] / tot["nn_launches"]) if tot["nn_launches"] else None),
        }
        # the dominant kernels of the timed region (rank 0): the two steer mappings, fp64 VALU bound.  Work per propagated
        # edge: 20 RK4 steps x 4 f-evals, each the EXACT operation count of the restated reference's get_state_derivative
        # (oracle/flop_count.cpp, mean over sampled states) + 20 proximity tests (not counted).  The reference's operation
        # order forbids FMA contraction, so the usable peak is one operation per lane and cycle = half of the 78.6 TFLOP/s
        # FMA figure.
        # the per-round HIP events stop after 8192 rounds per planner (planner.hip kProfRounds): beyond that the NN / steer
        # figures would cover a prefix of the run only while the work counters cover all of it -- say so instead
        profiled_all = tot["steer_launches"] >= tot["rounds"]
        out["profile_coverage"] = {"rounds": tot["rounds"], "profiled_rounds": tot["steer_launches"], "complete": profiled_all}
        if not profiled_all:
            for k in ("roofline", "nn_sweep_mfma_timed"):
                out[k]["note"] = "INCOMPLETE: only the first profiled_rounds rounds were timed (profile_coverage); " + out[k]["note"]
        if tot["steer_ms"] > 0 and profiled_all:
            fe = feval_ops(scn)
            n_steps = 20
            # EXECUTED work: the kernels count the RK4 steps they integrate (an edge stops at its first state that is not
            # free, MEAQR_topology.hpp:550-559: 11-12 of its 20 steps on average), 4 f-evals each.  `launched` prices every
            # propagated edge at its full 20 steps -- what rounds 1-2 reported as `achieved`.
            rate = tot["steer_steps"] * 4 * fe["useful"] / (tot["steer_ms"] * 1e-3) / 1e12
            rate_launched = (tot["spec"] + tot["nodes"]) * n_steps * 4 * fe["useful"] / (tot["steer_ms"] * 1e-3) / 1e12
            out["steer_kernels"] = {"kernels": "propagate_pair_step_kernel (one launch per RK4 step over the live edges of "
                                               "all problems; large rounds) + propagate_pair_kernel / propagate_kernel "
                                               "(whole-edge launches; small rounds)",
                                    "bound": "fp64_valu", "share_of_step_time": tot["steer_ms"] * 1e-3 / tot["seconds"],
                                    "avg_round_ms": tot["steer_ms"] / max(1, tot["steer_launches"]),
                                    "edges_per_s_in_kernel": tot["spec"] / (tot["steer_ms"] * 1e-3),
                                    "executed_steps": tot["steer_steps"],
                                    "executed_steps_per_edge": tot["steer_steps"] / max(1, tot["spec"] + tot["nodes"]),
                                    "f_eval_ops": fe,
                                    "achieved": rate, "peak": 39.3, "unit": "Tops/s (fp64, no FMA)", "frac": rate / 39.3,
                                    "launched": {"achieved": rate_launched, "frac": rate_launched / 39.3,
                                                 "note": "every propagated edge (candidates + goal probes) priced at 20 steps: "
                                                         "not executed work; rounds 1-2 quoted this for the candidates"},
                                    "frac_counting_the_reference_dense_products": rate / 39.3 * fe["all"] / fe["useful"],
                                    "occupancy": steer_occupancy(),
                                    "note": "executed f-eval operations only (proximity tests excluded; edges = candidates + "
                                            "goal probes); rank-0 launches of the timed region; `useful` operations of the "
                                            "reference's f-eval, i.e. without its dense products over structural zeros"}
        # the microbenchmarks and the CPU baselines are single-GPU extras: rank 0 at N = 1 only
        if not args.no_microbench and world == 1:
            # the cloud's hyperbox (the unit cube) is declared, as a planner's topology does: from 5 queries per sweep on
            # that selects the matrix-core pre-filter (same answers); up to 4 the register-direct fp64 sweep runs
            rows = 4 * 1024 * 1024
            out["nn_sweep_hbm"] = nn_sweep_microbench(lib, ctx, events, rows, 8, 20, coord_bound=1.0)
            keep = ("kernel", "ms_per_sweep", "achieved", "frac")
            out["nn_sweep_hbm"]["by_queries_per_sweep"] = {
                str(b): {k: v for k, v in nn_sweep_microbench(lib, ctx, events, rows, b, 20, coord_bound=1.0).items()
                         if k in keep} for b in (1, 4, 32)}
            # without a declared bound every sweep is exact fp64 arithmetic (8 queries: 288 operations per row)
            out["nn_sweep_hbm"]["no_coordinate_bound"] = {
                k: v for k, v in nn_sweep_microbench(lib, ctx, events, rows, 8, 20).items() if k in keep}
            out["nn_sweep_mfma"] = nn_mfma_microbench(lib, ctx, events)
        if not args.no_microbench and world == 1:
            out["single_problem"] = [single_problem_rate(lib, scene, scn, 1, 20000),
                                     single_problem_rate(lib, scene, scn, 16, 20000)]
            if tot["steer_ms"] > 0:
                out["collide"] = collide_counts(lib, scene, scn, tot["steer_steps"] / (tot["steer_ms"] * 1e-3))
            out["c3_rrtstar"] = c3_rrtstar_rate(lib, ctx, events)
            out["c4_prm_meshes"] = c4_prm_rate(lib, ctx)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(lib, scene, scn, args.max_vertices)
            out["cpu_baseline_all_cores"] = cpu_baseline_all_cores(lib, scene, scn, args.max_vertices)
            out["nn_published_config"] = nn_published_config(lib, ctx, events)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
