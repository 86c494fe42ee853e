"""ctypes access to the CPU oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by reak_amd/."""
import ctypes as C
import os
import subprocess

import numpy as np

from reak_amd import types as T

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ORACLE_DIR = os.path.join(_ROOT, "oracle")


class RrtOut(C.Structure):
    _fields_ = [
        ("num_vertices", C.c_uint64),
        ("iterations", C.c_uint64),
        ("num_solutions", C.c_uint64),
        ("edges_checked", C.c_uint64),
        ("states_checked", C.c_uint64),
        ("f_evals", C.c_uint64),
        ("pair_tests", C.c_uint64),
        ("best_cost", C.c_double),
        ("seconds", C.c_double),
    ]


class RrtStarOut(C.Structure):
    _fields_ = [("num_vertices", C.c_uint64), ("samples", C.c_uint64), ("loop_iterations", C.c_uint64),
                ("num_solutions", C.c_uint64), ("rewires", C.c_uint64), ("edges_checked", C.c_uint64),
                ("states_checked", C.c_uint64), ("best_cost", C.c_double), ("seconds", C.c_double)]


class BiRrtStarOut(C.Structure):
    _fields_ = [("num_vertices", C.c_uint64), ("samples", C.c_uint64), ("loop_iterations", C.c_uint64),
                ("rewires", C.c_uint64), ("fwd_rewires", C.c_uint64), ("joins", C.c_uint64), ("edges_checked", C.c_uint64),
                ("states_checked", C.c_uint64), ("best_join_cost", C.c_double), ("seconds", C.c_double)]


class PrmOut(C.Structure):
    _fields_ = [("num_vertices", C.c_uint64), ("num_edges", C.c_uint64), ("samples", C.c_uint64),
                ("rejected", C.c_uint64), ("loop_iterations", C.c_uint64), ("num_components", C.c_uint64),
                ("publish_calls", C.c_uint64), ("merged_at_vertex", C.c_int64), ("edges_checked", C.c_uint64),
                ("states_checked", C.c_uint64), ("seconds", C.c_double)]


class BiRrtOut(C.Structure):
    _fields_ = [("n1", C.c_uint64), ("n2", C.c_uint64), ("loop_iterations", C.c_uint64), ("samples", C.c_uint64),
                ("num_solutions", C.c_uint64), ("joins", C.c_uint64), ("edges_checked", C.c_uint64),
                ("best_cost", C.c_double), ("seconds", C.c_double)]


def build():
    subprocess.run(["make", "-s", "-C", _ORACLE_DIR], check=True)


_libs = {}


def feval_op_count(scn, x, u):
    """Exact fp64 operation counts of ONE x' = f(x, u) evaluation by the restated reference (oracle/flop_count.cpp):
    dict with add / mul / div / sqrt / trig / cmp, the structure-aware add_useful / mul_useful, and the totals
    `all` (every operation the reference's dense code performs) and `useful` (without its products over structural zeros)."""
    path = os.path.join(_ORACLE_DIR, "liboracle_flops.so")
    if not os.path.exists(path):
        build()
    lib = C.CDLL(path)
    lib.oracle_feval_op_count.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                          C.POINTER(C.c_uint64)]
    ops = scn.ops_array()
    x = np.ascontiguousarray(x, dtype=np.float64).ravel()
    u = np.ascontiguousarray(u, dtype=np.float64).ravel()
    cnt = (C.c_uint64 * 8)()
    rc = lib.oracle_feval_op_count(C.cast(ops, C.c_void_p), len(scn.ops), C.cast(C.byref(scn.base), C.c_void_p),
                                   T.dptr(x), T.dptr(u), cnt)
    if rc != 0:
        raise RuntimeError("oracle_feval_op_count: singular mass matrix")
    names = ["add", "mul", "div", "sqrt", "trig", "cmp", "add_useful", "mul_useful"]
    r = dict(zip(names, [int(v) for v in cnt]))
    r["all"] = r["add"] + r["mul"] + r["div"] + r["sqrt"] + r["trig"]
    r["useful"] = r["add_useful"] + r["mul_useful"] + r["div"] + r["sqrt"] + r["trig"]
    return r


def load(fast=False):
    name = "liboracle_fast.so" if fast else "liboracle.so"
    if name in _libs:
        return _libs[name]
    path = os.path.join(_ORACLE_DIR, name)
    if not os.path.exists(path):
        build()
    lib = C.CDLL(path)
    d, dp, u32p = C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_uint32)
    lib.orc_mt19937_nth.restype = C.c_uint32
    lib.orc_mt19937_nth.argtypes = [C.c_uint32, C.c_uint32]
    lib.orc_sample_hyperbox.argtypes = [C.c_uint32, dp, dp, C.c_int, C.c_int, dp]
    lib.orc_euclid.restype = d
    lib.orc_euclid.argtypes = [dp, dp, C.c_int]
    lib.orc_nn1.argtypes = [dp, C.c_int, dp, C.c_uint64, C.c_int, u32p, dp]
    lib.orc_knn.argtypes = [dp, C.c_int, dp, C.c_uint64, C.c_int, C.c_uint32, d, u32p, dp, u32p]
    lib.orc_star_neighborhood.argtypes = [C.c_uint64, d, d, C.POINTER(C.c_uint64), dp]
    lib.orc_highest_set_bit.restype = C.c_uint64
    lib.orc_highest_set_bit.argtypes = [C.c_uint64]
    lib.orc_quat_mul.argtypes = [dp, dp, dp]
    lib.orc_quat_rotmat.argtypes = [dp, dp]
    lib.orc_quat_rotate.argtypes = [dp, dp, dp]
    lib.orc_quat_from_vector.argtypes = [dp, dp]
    lib.orc_axis_angle_quat.argtypes = [d, dp, dp]
    lib.orc_axis_angle_rotmat.argtypes = [d, dp, dp]
    lib.orc_cholesky_solve.argtypes = [dp, dp, C.c_int, d]
    lib.orc_cholesky_decompose.argtypes = [dp, dp, C.c_int, d]
    lib.orc_rk4_ivp.argtypes = [C.c_int, dp, C.c_int, d, d, d, dp]
    lib.orc_scene_create.restype = C.c_void_p
    lib.orc_scene_create.argtypes = [C.POINTER(T.KteOp), C.c_int, C.POINTER(T.ChainBase), C.POINTER(T.Shape), C.c_int]
    lib.orc_scene_destroy.argtypes = [C.c_void_p]
    lib.orc_scene_num_frames.argtypes = [C.c_void_p]
    lib.orc_scene_num_finders.argtypes = [C.c_void_p]
    lib.orc_state_derivative.argtypes = [C.c_void_p, dp, dp, C.c_int, dp, dp, dp]
    lib.orc_fk.argtypes = [C.c_void_p, dp, C.c_int, dp]
    lib.orc_min_distance.argtypes = [C.c_void_p, dp, C.c_int, dp]
    lib.orc_pair_distance.restype = d
    lib.orc_pair_distance.argtypes = [C.POINTER(T.Shape), C.POINTER(T.Shape)]
    lib.orc_rk4_step.argtypes = [C.c_void_p, C.POINTER(T.DynSpace), dp, dp, C.c_int, d, dp]
    lib.orc_steer.argtypes = [C.c_void_p, C.POINTER(T.DynSpace), dp, dp, C.c_int, d, dp, u32p, dp]
    lib.orc_rrt_dyn.argtypes = [C.c_void_p, C.POINTER(T.DynSpace), C.POINTER(T.RrtParams), C.c_int64, C.POINTER(RrtOut)]
    lib.orc_rrt_dyn_warm.argtypes = [C.c_void_p, C.POINTER(T.DynSpace), C.POINTER(T.RrtParams), C.POINTER(C.c_double),
                                     C.c_uint64, C.c_int64, C.POINTER(RrtOut)]
    lib.orc_rrt_qs.argtypes = [C.c_void_p, C.c_int, dp, dp, d, C.POINTER(T.RrtParams), C.c_int64, C.POINTER(RrtOut)]
    lib.orc_qs_move.argtypes = [C.c_void_p, C.c_int, dp, dp, d, dp, dp, C.c_int, d, dp, u32p]
    lib.orc_rrtstar_qs.argtypes = [C.c_void_p, C.c_int, dp, dp, d, C.POINTER(T.RrtParams), C.c_int64, C.POINTER(RrtStarOut)]
    lib.orc_bnb_rrtstar_qs.argtypes = [C.c_void_p, C.c_int, dp, dp, d, C.POINTER(T.RrtParams), C.c_int64, C.POINTER(RrtStarOut),
                                       C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.orc_bnb_removed_copy.argtypes = [C.POINTER(C.c_uint8)]
    lib.orc_birrtstar_qs.argtypes = [C.c_void_p, C.c_int, dp, dp, d, C.POINTER(T.RrtParams), C.c_int64, C.POINTER(BiRrtStarOut)]
    lib.orc_birrtstar_copy.argtypes = [dp, u32p, dp, u32p, dp, u32p, u32p]
    lib.orc_prm_dyn.argtypes = [C.c_void_p, C.POINTER(T.DynSpace), C.POINTER(T.PrmParams), C.c_int64, C.POINTER(PrmOut)]
    lib.orc_rrtstar_dyn.argtypes = [C.c_void_p, C.POINTER(T.DynSpace), C.POINTER(T.RrtParams), C.c_int64, C.POINTER(RrtStarOut)]
    lib.orc_rrtstar_copy.argtypes = [dp, u32p, dp, u32p]
    lib.orc_prm_qs.argtypes = [C.c_void_p, C.c_int, dp, dp, d, C.POINTER(T.PrmParams), C.c_int64, C.POINTER(PrmOut)]
    lib.orc_prm_copy.argtypes = [dp, u32p, u32p, dp, dp, u32p, C.POINTER(C.c_uint8), u32p]
    lib.orc_birrt_qs.argtypes = [C.c_void_p, C.c_int, dp, dp, d, C.POINTER(T.RrtParams), C.c_int64, C.POINTER(BiRrtOut)]
    lib.orc_birrt_copy.argtypes = [dp, u32p, dp, u32p, u32p, C.POINTER(C.c_uint8)]
    lib.orc_vptree_nn1.argtypes = [dp, C.c_uint32, dp, C.c_uint64, C.c_int, u32p, dp, dp, dp]
    lib.orc_rrt_copy.argtypes = [dp, u32p, u32p, C.POINTER(C.c_uint8), dp]
    _libs[name] = lib
    return lib


class OracleScene:
    """A KTE chain + proxy environment held by the oracle."""

    def __init__(self, scn, fast=False):
        self.lib = load(fast)
        self.scn = scn
        self._ops = scn.ops_array()
        self._shapes = scn.shapes_array() if scn.shapes else (T.Shape * 1)()
        verts = getattr(scn, "mesh_vertices", None)
        if verts is not None and len(verts):
            self._verts = np.ascontiguousarray(verts, dtype=np.float64).reshape(-1, 3)
            self.lib.orc_scene_create_with_meshes.restype = C.c_void_p
            self.lib.orc_scene_create_with_meshes.argtypes = [C.POINTER(T.KteOp), C.c_int, C.POINTER(T.ChainBase),
                                                              C.POINTER(T.Shape), C.c_int, C.POINTER(C.c_double), C.c_int]
            self.h = self.lib.orc_scene_create_with_meshes(self._ops, len(scn.ops), C.byref(scn.base), self._shapes,
                                                           len(scn.shapes), T.dptr(self._verts), len(self._verts))
        else:
            self.h = self.lib.orc_scene_create(self._ops, len(scn.ops), C.byref(scn.base), self._shapes, len(scn.shapes))
        self.n = scn.n_dof
        self.D = 2 * scn.n_dof

    def __del__(self):
        try:
            self.lib.orc_scene_destroy(self.h)
        except Exception:
            pass

    def state_derivative(self, x, u):
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1, self.D)
        u = np.ascontiguousarray(u, dtype=np.float64).reshape(-1, self.n)
        B = x.shape[0]
        pd = np.zeros((B, self.D))
        M = np.zeros((B, self.n, self.n))
        f = np.zeros((B, self.n))
        rc = self.lib.orc_state_derivative(self.h, T.dptr(x), T.dptr(u), B, T.dptr(pd), T.dptr(M), T.dptr(f))
        return rc, pd, M, f

    def fk(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1, self.D)
        nf = self.lib.orc_scene_num_frames(self.h)
        out = np.zeros((x.shape[0], nf, 7))
        self.lib.orc_fk(self.h, T.dptr(x), x.shape[0], T.dptr(out))
        return out

    def min_distance(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1, self.D)
        d = np.zeros(x.shape[0])
        self.lib.orc_min_distance(self.h, T.dptr(x), x.shape[0], T.dptr(d))
        return d

    def rk4_step(self, x, u, t=0.0):
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1, self.D)
        u = np.ascontiguousarray(u, dtype=np.float64).reshape(-1, self.n)
        xn = np.zeros_like(x)
        rc = self.lib.orc_rk4_step(self.h, C.byref(self.scn.dyn), T.dptr(x), T.dptr(u), x.shape[0], float(t), T.dptr(xn))
        return rc, xn

    def steer(self, a, b, fraction=1.0, record=False):
        a = np.ascontiguousarray(a, dtype=np.float64).reshape(-1, self.D)
        b = np.ascontiguousarray(b, dtype=np.float64).reshape(-1, self.D)
        B = a.shape[0]
        out = np.zeros_like(a)
        steps = np.zeros(B, dtype=np.uint32)
        rec = np.zeros((B, self.scn.dyn.steps_per_edge + 1, self.D)) if record else None
        rc = self.lib.orc_steer(self.h, C.byref(self.scn.dyn), T.dptr(a), T.dptr(b), B, float(fraction), T.dptr(out),
                                T.u32ptr(steps), T.dptr(rec) if record else None)
        return rc, out, steps, rec

    def rrt_dyn(self, prm, max_iterations=-1):
        out = RrtOut()
        rc = self.lib.orc_rrt_dyn(self.h, C.byref(self.scn.dyn), C.byref(prm), int(max_iterations), C.byref(out))
        return rc, out, self._copy_rrt(out, self.D)

    def rrt_dyn_warm(self, prm, warm_pos, iterations):
        """Timing only (bench.py cpu_baseline): `iterations` loop iterations of the sequential RRT started on a tree that
        already holds the vertices warm_pos [n][D] (besides the root): the CPU's cost per iteration AT that tree size."""
        warm_pos = np.ascontiguousarray(warm_pos, dtype=np.float64)
        out = RrtOut()
        rc = self.lib.orc_rrt_dyn_warm(self.h, C.byref(self.scn.dyn), C.byref(prm), T.dptr(warm_pos), warm_pos.shape[0],
                                       int(iterations), C.byref(out))
        return rc, out

    def rrt_qs(self, lower, upper, min_interval, prm, max_iterations=-1):
        lower = np.ascontiguousarray(lower, dtype=np.float64)
        upper = np.ascontiguousarray(upper, dtype=np.float64)
        out = RrtOut()
        rc = self.lib.orc_rrt_qs(self.h, len(lower), T.dptr(lower), T.dptr(upper), float(min_interval), C.byref(prm),
                                 int(max_iterations), C.byref(out))
        return rc, out, self._copy_rrt(out, len(lower))

    def rrtstar_qs(self, lower, upper, min_interval, prm, max_loop_iterations=-1):
        lower = np.ascontiguousarray(lower, dtype=np.float64)
        upper = np.ascontiguousarray(upper, dtype=np.float64)
        out = RrtStarOut()
        rc = self.lib.orc_rrtstar_qs(self.h, len(lower), T.dptr(lower), T.dptr(upper), float(min_interval), C.byref(prm),
                                     int(max_loop_iterations), C.byref(out))
        nv, D = int(out.num_vertices), len(lower)
        pos = np.zeros((nv, D)); pred = np.zeros(nv, dtype=np.uint32); dist = np.zeros(nv)
        near = np.zeros(max(int(out.loop_iterations), 1), dtype=np.uint32)
        self.lib.orc_rrtstar_copy(T.dptr(pos), T.u32ptr(pred), T.dptr(dist), T.u32ptr(near))
        return rc, out, {"pos": pos, "pred": pred, "dist": dist, "near_seq": near[: int(out.loop_iterations)]}

    def rrtstar_dyn(self, prm, max_loop_iterations=-1):
        """RRT* over the scenario's steerable dynamic space (vertices = states)."""
        out = RrtStarOut()
        rc = self.lib.orc_rrtstar_dyn(self.h, C.byref(self.scn.dyn), C.byref(prm), int(max_loop_iterations), C.byref(out))
        nv, D = int(out.num_vertices), 2 * self.scn.n_dof
        pos = np.zeros((nv, D)); pred = np.zeros(nv, dtype=np.uint32); dist = np.zeros(nv)
        near = np.zeros(max(int(out.loop_iterations), 1), dtype=np.uint32)
        self.lib.orc_rrtstar_copy(T.dptr(pos), T.u32ptr(pred), T.dptr(dist), T.u32ptr(near))
        return rc, out, {"pos": pos, "pred": pred, "dist": dist, "near_seq": near[: int(out.loop_iterations)]}

    def prm_qs(self, lower, upper, min_interval, prm, max_loop_iterations=-1):
        lower = np.ascontiguousarray(lower, dtype=np.float64)
        upper = np.ascontiguousarray(upper, dtype=np.float64)
        out = PrmOut()
        rc = self.lib.orc_prm_qs(self.h, len(lower), T.dptr(lower), T.dptr(upper), float(min_interval), C.byref(prm),
                                 int(max_loop_iterations), C.byref(out))
        nv, ne, it, D = int(out.num_vertices), int(out.num_edges), int(out.loop_iterations), len(lower)
        pos = np.zeros((nv, D)); eu = np.zeros(max(ne, 1), dtype=np.uint32); ev = np.zeros(max(ne, 1), dtype=np.uint32)
        ew = np.zeros(max(ne, 1)); dens = np.zeros(nv); cc = np.zeros(nv, dtype=np.uint32)
        kind = np.zeros(max(it, 1), dtype=np.uint8); exp = np.zeros(max(it, 1), dtype=np.uint32)
        self.lib.orc_prm_copy(T.dptr(pos), T.u32ptr(eu), T.u32ptr(ev), T.dptr(ew), T.dptr(dens), T.u32ptr(cc),
                              kind.ctypes.data_as(C.POINTER(C.c_uint8)), T.u32ptr(exp))
        return rc, out, {"pos": pos, "edge_u": eu[:ne], "edge_v": ev[:ne], "edge_w": ew[:ne], "density": dens,
                         "cc_root": cc, "kind": kind[:it], "expanded": exp[:it]}

    def bnb_rrtstar_qs(self, lower, upper, min_interval, prm, max_loop_iterations=-1):
        """RRT* with branch-and-bound pruning: (rc, out, graph incl. "removed", pruned, skipped)."""
        lower = np.ascontiguousarray(lower, dtype=np.float64)
        upper = np.ascontiguousarray(upper, dtype=np.float64)
        out = RrtStarOut()
        pruned, skipped = C.c_uint64(), C.c_uint64()
        rc = self.lib.orc_bnb_rrtstar_qs(self.h, len(lower), T.dptr(lower), T.dptr(upper), float(min_interval), C.byref(prm),
                                         int(max_loop_iterations), C.byref(out), C.byref(pruned), C.byref(skipped))
        nv, D = int(out.num_vertices), len(lower)
        pos = np.zeros((nv, D)); pred = np.zeros(nv, dtype=np.uint32); dist = np.zeros(nv)
        near = np.zeros(max(int(out.loop_iterations), 1), dtype=np.uint32)
        removed = np.zeros(nv, dtype=np.uint8)
        self.lib.orc_rrtstar_copy(T.dptr(pos), T.u32ptr(pred), T.dptr(dist), T.u32ptr(near))
        self.lib.orc_bnb_removed_copy(removed.ctypes.data_as(C.POINTER(C.c_uint8)))
        return rc, out, {"pos": pos, "pred": pred, "dist": dist, "near_seq": near[: int(out.loop_iterations)],
                         "removed": removed}, pruned.value, skipped.value

    def birrtstar_qs(self, lower, upper, min_interval, prm, max_loop_iterations=-1):
        """Bidirectional RRT* over the quasi-static space."""
        lower = np.ascontiguousarray(lower, dtype=np.float64)
        upper = np.ascontiguousarray(upper, dtype=np.float64)
        out = BiRrtStarOut()
        rc = self.lib.orc_birrtstar_qs(self.h, len(lower), T.dptr(lower), T.dptr(upper), float(min_interval), C.byref(prm),
                                       int(max_loop_iterations), C.byref(out))
        nv, D, it = int(out.num_vertices), len(lower), max(int(out.loop_iterations), 1)
        pos = np.zeros((nv, D)); pred = np.zeros(nv, dtype=np.uint32); dist = np.zeros(nv)
        succ = np.zeros(nv, dtype=np.uint32); fwd = np.zeros(nv)
        npred = np.zeros(it, dtype=np.uint32); nsucc = np.zeros(it, dtype=np.uint32)
        self.lib.orc_birrtstar_copy(T.dptr(pos), T.u32ptr(pred), T.dptr(dist), T.u32ptr(succ), T.dptr(fwd), T.u32ptr(npred),
                                    T.u32ptr(nsucc))
        n = int(out.loop_iterations)
        return rc, out, {"pos": pos, "pred": pred, "dist": dist, "succ": succ, "fwd_dist": fwd, "near_pred": npred[:n],
                         "near_succ": nsucc[:n]}

    def prm_dyn(self, prm, max_loop_iterations=-1):
        """PRM over the scenario's steerable dynamic space (vertices = states)."""
        out = PrmOut()
        rc = self.lib.orc_prm_dyn(self.h, C.byref(self.scn.dyn), C.byref(prm), int(max_loop_iterations), C.byref(out))
        nv, ne, it, D = int(out.num_vertices), int(out.num_edges), int(out.loop_iterations), 2 * self.scn.n_dof
        pos = np.zeros((nv, D)); eu = np.zeros(max(ne, 1), dtype=np.uint32); ev = np.zeros(max(ne, 1), dtype=np.uint32)
        ew = np.zeros(max(ne, 1)); dens = np.zeros(nv); cc = np.zeros(nv, dtype=np.uint32)
        kind = np.zeros(max(it, 1), dtype=np.uint8); exp = np.zeros(max(it, 1), dtype=np.uint32)
        self.lib.orc_prm_copy(T.dptr(pos), T.u32ptr(eu), T.u32ptr(ev), T.dptr(ew), T.dptr(dens), T.u32ptr(cc),
                              kind.ctypes.data_as(C.POINTER(C.c_uint8)), T.u32ptr(exp))
        return rc, out, {"pos": pos, "edge_u": eu[:ne], "edge_v": ev[:ne], "edge_w": ew[:ne], "density": dens,
                         "cc_root": cc, "kind": kind[:it], "expanded": exp[:it]}

    def birrt_qs(self, lower, upper, min_interval, prm, max_loop_iterations=-1):
        lower = np.ascontiguousarray(lower, dtype=np.float64)
        upper = np.ascontiguousarray(upper, dtype=np.float64)
        out = BiRrtOut()
        rc = self.lib.orc_birrt_qs(self.h, len(lower), T.dptr(lower), T.dptr(upper), float(min_interval), C.byref(prm),
                                   int(max_loop_iterations), C.byref(out))
        n1, n2, it, D = int(out.n1), int(out.n2), int(out.loop_iterations), len(lower)
        p1 = np.zeros((n1, D)); q1 = np.zeros(n1, dtype=np.uint32); p2 = np.zeros((n2, D)); q2 = np.zeros(n2, dtype=np.uint32)
        nn = np.zeros(max(2 * it, 1), dtype=np.uint32); acc = np.zeros(max(2 * it, 1), dtype=np.uint8)
        self.lib.orc_birrt_copy(T.dptr(p1), T.u32ptr(q1), T.dptr(p2), T.u32ptr(q2), T.u32ptr(nn),
                                acc.ctypes.data_as(C.POINTER(C.c_uint8)))
        return rc, out, {"pos1": p1, "parent1": q1, "pos2": p2, "parent2": q2, "nn_seq": nn[: 2 * it], "accept": acc[: 2 * it]}

    def qs_move(self, lower, upper, min_interval, a, b, fraction=1.0):
        lower = np.ascontiguousarray(lower, dtype=np.float64)
        upper = np.ascontiguousarray(upper, dtype=np.float64)
        D = len(lower)
        a = np.ascontiguousarray(a, dtype=np.float64).reshape(-1, D)
        b = np.ascontiguousarray(b, dtype=np.float64).reshape(-1, D)
        out = np.zeros_like(a)
        nchk = np.zeros(a.shape[0], dtype=np.uint32)
        self.lib.orc_qs_move(self.h, D, T.dptr(lower), T.dptr(upper), float(min_interval), T.dptr(a), T.dptr(b),
                             a.shape[0], float(fraction), T.dptr(out), T.u32ptr(nchk))
        return out, nchk

    def _copy_rrt(self, out, D):
        nv, it = int(out.num_vertices), int(out.iterations)
        pos = np.zeros((nv, D))
        parent = np.zeros(nv, dtype=np.uint32)
        nn_seq = np.zeros(max(it, 1), dtype=np.uint32)
        accept = np.zeros(max(it, 1), dtype=np.uint8)
        goal_dist = np.zeros(max(nv - 1, 1))
        self.lib.orc_rrt_copy(T.dptr(pos), T.u32ptr(parent), T.u32ptr(nn_seq),
                              accept.ctypes.data_as(C.POINTER(C.c_uint8)), T.dptr(goal_dist))
        return {"pos": pos, "parent": parent, "nn_seq": nn_seq[:it], "accept": accept[:it], "goal_dist": goal_dist[: nv - 1]}


def gjk_distance(a, b, mesh_vertices=None):
    """The oracle's GJK on world-anchored shape pairs (a[i], b[i])."""
    lib = load()
    n = len(a)
    aa, bb = T.as_array(list(a), T.Shape), T.as_array(list(b), T.Shape)
    verts = np.zeros((1, 3)) if mesh_vertices is None else np.ascontiguousarray(mesh_vertices, dtype=np.float64).reshape(-1, 3)
    out = np.zeros(n)
    lib.orc_gjk_distance.argtypes = [C.POINTER(T.Shape), C.POINTER(T.Shape), C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.orc_gjk_distance(aa, bb, n, T.dptr(verts), T.dptr(out))
    return out


def pair_distance(a, b):
    """The restated reference's closed form for one world-anchored pair (NaN: no finder)."""
    lib = load()
    lib.orc_pair_distance.restype = C.c_double
    return lib.orc_pair_distance(C.byref(a), C.byref(b))


def set_qs_speed_limits(speed, fast=False):
    """Speed limits of the rate-limited joint space for the quasi-static spaces created from now on (None = ordinary)."""
    lib = load(fast)
    lib.orc_set_qs_speed_limits.argtypes = [C.POINTER(C.c_double), C.c_int]
    if speed is None:
        lib.orc_set_qs_speed_limits(None, 0)
    else:
        sp = np.ascontiguousarray(speed, dtype=np.float64)
        lib.orc_set_qs_speed_limits(T.dptr(sp), len(sp))


def nn1(q, pts, fast=False):
    lib = load(fast)
    q = np.ascontiguousarray(q, dtype=np.float64)
    pts = np.ascontiguousarray(pts, dtype=np.float64)
    B, D = q.shape
    idx = np.zeros(B, dtype=np.uint32)
    dist = np.zeros(B)
    lib.orc_nn1(T.dptr(q), B, T.dptr(pts), pts.shape[0], D, T.u32ptr(idx), T.dptr(dist))
    return idx, dist


def vptree_nn1(q, pts, fast=False):
    """Exact 1-NN through the static vantage-point tree: (idx, dist, build seconds, query seconds)."""
    lib = load(fast)
    q = np.ascontiguousarray(q, dtype=np.float64)
    pts = np.ascontiguousarray(pts, dtype=np.float64)
    B, D = q.shape
    idx = np.zeros(B, dtype=np.uint32)
    dist = np.zeros(B)
    tb, tq = C.c_double(), C.c_double()
    lib.orc_vptree_nn1(T.dptr(q), B, T.dptr(pts), pts.shape[0], D, T.u32ptr(idx), T.dptr(dist), C.byref(tb), C.byref(tq))
    return idx, dist, tb.value, tq.value


def dvptree(q, pts, arity=2, incremental=False, seed=1, k=1, radius=np.inf, fast=False):
    """The restated DVP-tree of the reference (oracle/dvp_tree.hpp): k = 1 -> (idx, dist, info); k > 1 -> (idx, dist, cnt,
    info); info = {build_s, query_s, dist_evals (distance evaluations of the queries)}."""
    lib = load(fast)
    q = np.ascontiguousarray(q, dtype=np.float64)
    pts = np.ascontiguousarray(pts, dtype=np.float64)
    B, D = q.shape
    idx = np.zeros((B, k) if k > 1 else B, dtype=np.uint32)
    dist = np.zeros((B, k) if k > 1 else B)
    cnt = np.zeros(B, dtype=np.uint32)
    tb, tq, ev = C.c_double(), C.c_double(), C.c_uint64()
    lib.orc_dvptree.argtypes = [C.POINTER(C.c_double), C.c_uint32, C.POINTER(C.c_double), C.c_uint64, C.c_int, C.c_int, C.c_int,
                                C.c_uint32, C.c_uint32, C.c_double, C.POINTER(C.c_uint32), C.POINTER(C.c_double),
                                C.POINTER(C.c_uint32), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    rc = lib.orc_dvptree(T.dptr(q), B, T.dptr(pts), pts.shape[0], D, arity, 1 if incremental else 0, seed, k, float(radius),
                         T.u32ptr(idx), T.dptr(dist), T.u32ptr(cnt), C.byref(tb), C.byref(tq), C.byref(ev))
    assert rc == 0
    info = {"build_s": tb.value, "query_s": tq.value, "dist_evals": int(ev.value)}
    return (idx, dist, info) if k <= 1 else (idx, dist, cnt, info)


def knn(q, pts, k, radius=np.inf, fast=False):
    lib = load(fast)
    q = np.ascontiguousarray(q, dtype=np.float64)
    pts = np.ascontiguousarray(pts, dtype=np.float64)
    B, D = q.shape
    idx = np.zeros((B, k), dtype=np.uint32)
    dist = np.zeros((B, k))
    cnt = np.zeros(B, dtype=np.uint32)
    lib.orc_knn(T.dptr(q), B, T.dptr(pts), pts.shape[0], D, k, float(radius), T.u32ptr(idx), T.dptr(dist), T.u32ptr(cnt))
    return idx, dist, cnt
