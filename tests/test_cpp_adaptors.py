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


@pytest.mark.gpu
def test_rrt_through_the_cpp_sockets_is_the_sequential_planner(tmp_path, oracle):
    exe = build_smoke()
    scn = scenarios.make_c2(world_seed=1)
    prm = scn.rrt_params(seed=6, max_vertices=160)
    blob = tmp_path / "scene.bin"
    with open(blob, "wb") as f:
        ops, shapes = scn.ops_array(), scn.shapes_array()
        f.write(np.int32(len(scn.ops)).tobytes())
        f.write(bytes(ops))
        f.write(bytes(scn.base))
        f.write(np.int32(len(scn.shapes)).tobytes())
        f.write(bytes(shapes))
        f.write(bytes(scn.dyn))
        f.write(bytes(prm))
    run = subprocess.run([exe, str(blob)], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stdout + run.stderr
    out = json.loads(run.stdout.strip().splitlines()[-1])
    assert out["same_tree"] and out["knn_sorted"] and out["bad_arg_throws"] and out["proxy_disagree"] == 0
    osc = oracle.OracleScene(scn)
    rc, ro, rtree = osc.rrt_dyn(prm)
    assert rc == 0
    assert (out["vertices"], out["iterations"]) == (ro.num_vertices, ro.iterations)
    assert (out["planner_vertices"], out["planner_iterations"]) == (ro.num_vertices, ro.iterations)
    assert out["parents"] == [int(v) for v in rtree["parent"][1:]]
