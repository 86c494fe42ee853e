"""The C++ adaptors of include/rkh_adaptors.hpp (NNFinder, KNN synchro, steerable C_free topology, proximity pair,
planner entry) compiled by g++ against librkh.so.  CPU: they compile and link.  GPU: tests/cpp/abi_smoke.cpp grows an
RRT one query at a time through the sockets and through the batched planner entry; both must be the oracle's tree."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

from reak_amd import scenarios

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "abi_smoke.cpp")
EXE = os.path.join(ROOT, "tests", "cpp", "abi_smoke")


def build_smoke():
    from reak_amd import lib as L

    L.build()
    lib_dir = os.path.join(ROOT, "reak_amd")
    newer = [SRC, os.path.join(ROOT, "include", "rkh_adaptors.hpp"), os.path.join(ROOT, "include", "rkh.h")]
    if os.path.exists(EXE) and all(os.path.getmtime(EXE) >= os.path.getmtime(f) for f in newer):
        return EXE
    subprocess.run(["g++", "-std=c++17", "-O2", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), SRC,
                    "-L", lib_dir, "-lrkh", "-Wl,-rpath," + lib_dir, "-o", EXE], check=True)
    return EXE


def test_adaptors_compile_and_link():
    assert os.path.exists(build_smoke())


def _write_scene(f, scn):
    ops, shapes = scn.ops_array(), scn.shapes_array()
    f.write(np.int32(len(scn.ops)).tobytes())
    f.write(bytes(ops))
    f.write(bytes(scn.base))
    f.write(np.int32(len(scn.shapes)).tobytes())
    f.write(bytes(shapes))


@pytest.mark.gpu
def test_planners_through_the_cpp_sockets_are_the_sequential_planners(tmp_path, oracle):
    """abi_smoke.cpp (g++ against librkh.so): (1) RRT grown one query at a time through the NNFinder / synchro / steerable
    topology / proximity sockets, (2) the same query through hip_rrt_planner::solve_planning_query(Query&) with a
    point-to-point query object and the progress / solution hooks, (3) both are one tree = the oracle's, (4) RRT, RRT*, PRM
    and the bidirectional RRT over the quasi-static C1 scene through their adaptors: graphs, solution costs and the calls
    the query object received are the oracle's sequential planners'."""
    from reak_amd import lib as L

    exe = build_smoke()
    scn = scenarios.make_c2(world_seed=1)
    prm = scn.rrt_params(seed=6, max_vertices=160)
    blob = tmp_path / "scene.bin"
    with open(blob, "wb") as f:
        _write_scene(f, scn)
        f.write(bytes(scn.dyn))
        f.write(bytes(prm))
    c1 = scenarios.make_c1(world_seed=1)
    lo, hi, mi = c1.meta["lower"], c1.meta["upper"], c1.meta["min_interval"]
    qprm = c1.rrt_params(seed=2, max_vertices=700)
    radius = 1.0
    qblob = tmp_path / "qs_scene.bin"
    with open(qblob, "wb") as f:
        _write_scene(f, c1)
        f.write(bytes(L.make_qs_space(3, lo, hi, mi)))
        f.write(bytes(qprm))
        f.write(np.float64(radius).tobytes())
    run = subprocess.run([exe, str(blob), str(qblob)], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stdout + run.stderr
    out = json.loads(run.stdout.strip().splitlines()[-1])
    assert out["same_tree"] and out["knn_sorted"] and out["bad_arg_throws"] and out["proxy_disagree"] == 0
    assert out["reports_ok"] and out["topo_ok"] and out["stale_rng_throws"] and out["rng_advanced"] and out["qs_topo_ok"]
    osc = oracle.OracleScene(scn)
    rc, ro, rtree = osc.rrt_dyn(prm)
    assert rc == 0
    assert (out["vertices"], out["iterations"]) == (ro.num_vertices, ro.iterations)
    assert (out["planner_vertices"], out["planner_iterations"]) == (ro.num_vertices, ro.iterations)
    assert out["parents"] == [int(v) for v in rtree["parent"][1:]]
    assert out["rrt_solutions"] == ro.num_solutions
    # ---- quasi-static scene: every planner against its oracle twin
    o1 = oracle.OracleScene(c1)
    rc, ro, rt = o1.rrt_qs(lo, hi, mi, qprm)
    q = out["qs_rrt"]
    assert (q["vertices"], q["iterations"], q["solutions"]) == (ro.num_vertices, ro.iterations, ro.num_solutions)
    assert q["parent"][1:] == [int(v) for v in rt["parent"][1:]] and q["parent"][0] == -1
    assert (ro.num_solutions == 0 and q["best"] == -1.0) or q["best"] == ro.best_cost
    rc, ro, rg = o1.rrtstar_qs(lo, hi, mi, qprm)
    q = out["qs_rrtstar"]
    assert (q["vertices"], q["rewires"]) == (ro.num_vertices, ro.rewires) and q["reports"] == ro.num_vertices // 100
    want = [-1 if (v == 0xFFFFFFFF or i == 0) else int(v) for i, v in enumerate(rg["pred"])]
    assert q["pred"] == want
    if ro.num_solutions:
        assert q["solutions"] == 1 and q["best"] == ro.best_cost and q["path_len"] >= 2
    pp = c1.prm_params(seed=2, max_vertices=700, sampling_radius=radius)
    rc, ro, rg = o1.prm_qs(lo, hi, mi, pp)
    q = out["qs_prm"]
    assert (q["vertices"], q["edges"], q["components"]) == (ro.num_vertices, ro.num_edges, ro.num_components)
    assert q["weight_sum"] == float(np.asarray(rg["edge_w"], dtype=np.float64).cumsum()[-1])  # the same left-to-right sum
    assert q["register_calls"] == 0  # prm_planner registers nothing (density_plan_visitor), rkh.h
    rc, ro, rtr = o1.birrt_qs(lo, hi, mi, qprm)
    q = out["qs_birrt"]
    assert (q["vertices_1"], q["vertices_2"], q["solutions"]) == (ro.n1, ro.n2, ro.num_solutions)
    if ro.num_solutions:
        assert q["best"] == ro.best_cost and abs(q["query_best"] - ro.best_cost) <= 1e-12 * max(1.0, ro.best_cost)
