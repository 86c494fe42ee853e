"""ctypes binding of the C-ABI in include/rkh.h (reak_amd/librkh.so = hand-written HIP kernels, gfx950).

There is no CPU fallback: if the shared library is missing or the GPU is absent the calls fail loudly.
The classes mirror the reference interfaces the kernels replace (paths relative to /root/reference/src/ReaK/):
  HipNeighborSearch  ~ linear_neighbor_search + any_knn_synchro  (ctrl/path_planning/topological_search.hpp:529-690)
  Scene              ~ kte_map_chain + mass_matrix_calc + proxy_query_pair_3D
  RrtPlanner         ~ rrt_planner::solve_planning_query          (ctrl/path_planning/rrt_path_planner.tpp:66-145)
"""
import ctypes as C
import os
import subprocess

import numpy as np

from . import types as T

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "librkh.so")

RKH_OK = 0
STATUS_NAMES = {0: "RKH_OK", -1: "RKH_ERR_BAD_ARG", -2: "RKH_ERR_OOM", -3: "RKH_ERR_SINGULAR", -4: "RKH_ERR_DEVICE",
                -5: "RKH_ERR_UNSUPPORTED", -6: "RKH_ERR_CAPACITY"}


class RkhError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"{STATUS_NAMES.get(status, status)}: {msg}")
        self.status = status


class SingularityError(RkhError):
    """singularity_error of the reference (core/lin_alg/mat_num_exceptions.hpp), status RKH_ERR_SINGULAR."""


class RrtStarStats(C.Structure):
    _fields_ = [("num_vertices", C.c_uint64), ("samples", C.c_uint64), ("loop_iterations", C.c_uint64),
                ("num_solutions", C.c_uint64), ("rewires", C.c_uint64), ("edges_checked", C.c_uint64),
                ("best_cost", C.c_double), ("pruned", C.c_uint64), ("skipped", C.c_uint64)]


class PrmStats(C.Structure):
    _fields_ = [("num_vertices", C.c_uint64), ("num_edges", C.c_uint64), ("samples", C.c_uint64),
                ("rejected", C.c_uint64), ("loop_iterations", C.c_uint64), ("num_components", C.c_uint64),
                ("publish_calls", C.c_uint64), ("merged_at_vertex", C.c_int64), ("edges_checked", C.c_uint64),
                ("device_steps", C.c_uint64)]


class BiRrtStats(C.Structure):
    _fields_ = [("num_vertices_1", C.c_uint64), ("num_vertices_2", C.c_uint64), ("loop_iterations", C.c_uint64),
                ("samples", C.c_uint64), ("num_solutions", C.c_uint64), ("joins", C.c_uint64),
                ("edges_checked", C.c_uint64), ("best_cost", C.c_double)]


class PlannerStats(C.Structure):
    _fields_ = [
        ("num_vertices", C.c_uint64),
        ("iterations", C.c_uint64),
        ("edges_checked", C.c_uint64),
        ("edges_speculated", C.c_uint64),
        ("rounds", C.c_uint64),
        ("num_solutions", C.c_uint64),
        ("best_cost", C.c_double),
        ("done", C.c_uint32),
    ]


ABI_VERSION = 3  # RKH_ABI_VERSION of include/rkh.h this binding mirrors


def abi_sizes():
    """sizeof of the mirrored PODs, in the argument order of rkh_abi_check."""
    return [C.sizeof(T.DynSpace), C.sizeof(T.QsSpace), C.sizeof(T.RrtParams), C.sizeof(T.PrmParams), C.sizeof(PlannerStats),
            C.sizeof(RrtStarStats), C.sizeof(PrmStats), C.sizeof(BiRrtStats), C.sizeof(T.Shape), C.sizeof(T.KteOp)]


EXPORTS = [
    "rkh_last_error", "rkh_version", "rkh_abi_version", "rkh_abi_check", "rkh_ctx_create", "rkh_ctx_destroy", "rkh_ctx_synchronize", "rkh_ctx_stream",
    "rkh_nn_create", "rkh_nn_destroy", "rkh_nn_clear", "rkh_nn_size", "rkh_nn_remove", "rkh_nn_live_size", "rkh_nn_append", "rkh_nn_query1",
    "rkh_nn_queryk", "rkh_nn_query1_async", "rkh_nn_queryk_async", "rkh_nn_fill_uniform", "rkh_nn_kernel_name",
    "rkh_nn_set_coord_bound",
    "rkh_scene_create", "rkh_scene_create_with_meshes", "rkh_diag_gjk_distance", "rkh_scene_destroy", "rkh_scene_num_dof", "rkh_scene_num_pairs", "rkh_state_derivative",
    "rkh_min_distance", "rkh_propagate", "rkh_edge_check", "rkh_planner_create", "rkh_planner_destroy",
    "rkh_planner_enqueue", "rkh_planner_sync", "rkh_planner_solve", "rkh_planner_get_tree", "rkh_planner_stream",
    "rkh_planner_nn_profile", "rkh_planner_nn_pairs", "rkh_planner_steer_profile", "rkh_planner_steer_steps", "rkh_diag_nn_mirror_query", "rkh_diag_feval_cycles", "rkh_diag_proximity_counts", "rkh_planner_create_batch", "rkh_planner_num_problems", "rkh_nn_set_events", "rkh_planner_create_qs_batch", "rkh_rrtstar_create_qs_batch", "rkh_rrtstar_create_batch", "rkh_birrtstar_create_qs_batch", "rkh_birrtstar_solve",
    "rkh_birrtstar_get_graph", "rkh_rrtstar_set_branch_and_bound", "rkh_rrtstar_get_removed", "rkh_rrtstar_destroy", "rkh_rrtstar_solve",
    "rkh_rrtstar_get_graph", "rkh_prm_create_qs_batch", "rkh_prm_create_batch", "rkh_prm_destroy", "rkh_prm_solve", "rkh_prm_get_graph", "rkh_birrt_create_qs_batch", "rkh_birrt_destroy", "rkh_birrt_solve", "rkh_birrt_get_trees", "rkh_planner_get_solution", "rkh_rrtstar_get_solution", "rkh_birrt_get_solution",
]


def build(verbose=False):
    """Compile every HIP source for gfx950 into reak_amd/librkh.so (hipcc cross-compiles without a GPU)."""
    subprocess.run(["make", "-s", "-C", os.path.join(_HERE, "csrc")], check=True,
                   stdout=None if verbose else subprocess.DEVNULL, stderr=None if verbose else subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise RuntimeError(f"{_LIB_PATH} is missing: run reak_amd.lib.build() (or __graft_entry__.build()); "
                           "there is no CPU fallback for the HIP path")
    lib = C.CDLL(_LIB_PATH)
    vp, d, dp, u32, u32p, u64 = C.c_void_p, C.c_double, C.POINTER(C.c_double), C.c_uint32, C.POINTER(C.c_uint32), C.c_uint64
    lib.rkh_last_error.restype = C.c_char_p
    lib.rkh_version.restype = C.c_char_p
    lib.rkh_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
    lib.rkh_ctx_destroy.argtypes = [vp]
    lib.rkh_ctx_synchronize.argtypes = [vp]
    lib.rkh_ctx_stream.restype = vp
    lib.rkh_ctx_stream.argtypes = [vp]
    lib.rkh_nn_create.argtypes = [vp, C.c_int, u64, C.POINTER(vp)]
    lib.rkh_nn_destroy.argtypes = [vp]
    lib.rkh_nn_clear.argtypes = [vp]
    lib.rkh_nn_size.restype = u64
    lib.rkh_nn_size.argtypes = [vp]
    lib.rkh_nn_remove.argtypes = [vp, u64]
    lib.rkh_nn_live_size.restype = u64
    lib.rkh_nn_live_size.argtypes = [vp]
    lib.rkh_nn_append.argtypes = [vp, dp, u64]
    lib.rkh_nn_query1.argtypes = [vp, dp, u32, u32p, dp]
    lib.rkh_nn_queryk.argtypes = [vp, dp, u32, u32, d, u32p, dp, u32p]
    lib.rkh_nn_query1_async.argtypes = [vp, vp, u32, vp, vp]
    lib.rkh_nn_queryk_async.argtypes = [vp, vp, u32, u32, d, vp, vp, vp]
    lib.rkh_nn_fill_uniform.argtypes = [vp, u64, u64]
    lib.rkh_nn_set_coord_bound.argtypes = [vp, C.c_double]
    lib.rkh_nn_set_events.argtypes = [vp, vp, vp]
    lib.rkh_nn_kernel_name.restype = C.c_char_p
    lib.rkh_scene_create.argtypes = [vp, C.POINTER(T.KteOp), C.c_int, C.POINTER(T.ChainBase), C.POINTER(T.Shape), C.c_int,
                                     C.POINTER(vp)]
    lib.rkh_scene_create_with_meshes.argtypes = [vp, C.POINTER(T.KteOp), C.c_int, C.POINTER(T.ChainBase), C.POINTER(T.Shape),
                                                 C.c_int, dp, u32, C.POINTER(vp)]
    lib.rkh_diag_gjk_distance.argtypes = [vp, C.POINTER(T.Shape), C.POINTER(T.Shape), u32, dp, u32, dp]
    lib.rkh_scene_destroy.argtypes = [vp]
    lib.rkh_scene_num_dof.argtypes = [vp]
    lib.rkh_scene_num_pairs.argtypes = [vp]
    lib.rkh_state_derivative.argtypes = [vp, dp, dp, u32, dp, dp, dp]
    lib.rkh_min_distance.argtypes = [vp, dp, u32, dp]
    lib.rkh_propagate.argtypes = [vp, C.POINTER(T.DynSpace), dp, dp, u32, d, dp, u32p, dp]
    lib.rkh_diag_feval_cycles.argtypes = [vp, dp, dp, u32, C.c_int, C.POINTER(C.c_uint64)]
    lib.rkh_diag_proximity_counts.argtypes = [vp, dp, u32, C.POINTER(C.c_uint64)]
    lib.rkh_edge_check.argtypes = [vp, dp, dp, d, dp, dp, u32, d, dp, u32p]
    lib.rkh_planner_create.argtypes = [vp, C.POINTER(T.DynSpace), C.POINTER(T.RrtParams), C.POINTER(vp)]
    lib.rkh_planner_create_batch.argtypes = [vp, C.POINTER(T.DynSpace), C.POINTER(T.RrtParams), u32, C.POINTER(vp)]
    lib.rkh_planner_create_qs_batch.argtypes = [vp, C.POINTER(T.QsSpace), C.POINTER(T.RrtParams), u32, C.POINTER(vp)]
    lib.rkh_rrtstar_create_qs_batch.argtypes = [vp, C.POINTER(T.QsSpace), C.POINTER(T.RrtParams), u32, C.POINTER(vp)]
    lib.rkh_rrtstar_destroy.argtypes = [vp]
    lib.rkh_rrtstar_solve.argtypes = [vp, C.c_int64, C.POINTER(RrtStarStats)]
    lib.rkh_rrtstar_get_graph.argtypes = [vp, u32, dp, u32p, dp, u32p]
    lib.rkh_rrtstar_set_branch_and_bound.argtypes = [vp, C.c_int]
    lib.rkh_rrtstar_get_removed.argtypes = [vp, u32, C.POINTER(C.c_uint8)]
    lib.rkh_birrtstar_solve.argtypes = [vp, C.c_int64, C.POINTER(BiRrtStarStats)]
    lib.rkh_birrtstar_get_graph.argtypes = [vp, u32, dp, u32p, dp, u32p, dp, u32p, u32p]
    lib.rkh_prm_create_qs_batch.argtypes = [vp, C.POINTER(T.QsSpace), C.POINTER(T.PrmParams), u32, C.POINTER(vp)]
    lib.rkh_prm_destroy.argtypes = [vp]
    lib.rkh_prm_solve.argtypes = [vp, C.c_int64, C.POINTER(PrmStats)]
    lib.rkh_planner_get_solution.argtypes = [vp, u32, u32p, u32, u32p, dp]
    lib.rkh_rrtstar_get_solution.argtypes = [vp, u32, u32p, u32, u32p, dp]
    lib.rkh_birrt_get_solution.argtypes = [vp, u32, u32p, u32p, u32p, u32p, u32, dp]
    lib.rkh_birrt_create_qs_batch.argtypes = [vp, C.POINTER(T.QsSpace), C.POINTER(T.RrtParams), u32, C.POINTER(vp)]
    lib.rkh_birrt_destroy.argtypes = [vp]
    lib.rkh_birrt_solve.argtypes = [vp, C.c_int64, C.POINTER(BiRrtStats)]
    lib.rkh_birrt_get_trees.argtypes = [vp, u32, dp, u32p, dp, u32p, u32p, C.POINTER(C.c_uint8)]
    lib.rkh_prm_get_graph.argtypes = [vp, u32, dp, u32p, u32p, dp, dp, u32p, C.POINTER(C.c_uint8), u32p]
    lib.rkh_planner_num_problems.restype = u32
    lib.rkh_planner_num_problems.argtypes = [vp]
    lib.rkh_planner_destroy.argtypes = [vp]
    lib.rkh_planner_enqueue.argtypes = [vp, u32]
    lib.rkh_planner_sync.argtypes = [vp, C.POINTER(PlannerStats)]
    lib.rkh_planner_solve.argtypes = [vp, C.POINTER(PlannerStats)]
    lib.rkh_planner_get_tree.argtypes = [vp, u32, dp, u32p, u32p, C.POINTER(C.c_uint8), dp]
    lib.rkh_planner_stream.restype = vp
    lib.rkh_planner_stream.argtypes = [vp]
    lib.rkh_planner_nn_profile.argtypes = [vp, dp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.rkh_planner_nn_pairs.argtypes = [vp, C.POINTER(C.c_uint64)]
    lib.rkh_planner_steer_profile.argtypes = [vp, dp, C.POINTER(C.c_uint64)]
    lib.rkh_planner_steer_steps.argtypes = [vp, C.POINTER(C.c_uint64)]
    lib.rkh_diag_nn_mirror_query.argtypes = [vp, dp, C.c_uint64, C.c_int, dp, C.c_uint32, C.c_double, C.POINTER(C.c_uint32), dp]
    lib.rkh_abi_version.restype = C.c_uint32
    lib.rkh_abi_check.argtypes = [C.c_uint32] + [C.c_size_t] * 10
    # the handshake of include/rkh.h: this binding's struct mirrors against the library's (RKH_ABI_CHECK)
    st = lib.rkh_abi_check(ABI_VERSION, *abi_sizes())
    if st != 0:
        raise RkhError(st, lib.rkh_last_error().decode())
    _lib = lib
    return lib


def _check(status):
    if status != RKH_OK:
        msg = load().rkh_last_error().decode()
        if status == -3:
            raise SingularityError(status, msg)
        raise RkhError(status, msg)


class Context:
    def __init__(self, device=0):
        self.lib = load()
        self.h = C.c_void_p()
        _check(self.lib.rkh_ctx_create(device, C.byref(self.h)))
        self.device = device

    def synchronize(self):
        _check(self.lib.rkh_ctx_synchronize(self.h))

    @property
    def stream(self):
        return self.lib.rkh_ctx_stream(self.h)

    def close(self):
        if self.h:
            self.lib.rkh_ctx_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HipNeighborSearch:
    """Exact brute-force NN over a growing vertex set kept in HBM (NNFinder + KNN synchro)."""

    def __init__(self, ctx, dims, capacity):
        self.ctx, self.lib, self.D = ctx, ctx.lib, dims
        self.h = C.c_void_p()
        _check(self.lib.rkh_nn_create(ctx.h, dims, capacity, C.byref(self.h)))

    def __len__(self):
        return int(self.lib.rkh_nn_size(self.h))

    def added_vertices(self, pts):
        pts = np.ascontiguousarray(pts, dtype=np.float64).reshape(-1, self.D)
        _check(self.lib.rkh_nn_append(self.h, T.dptr(pts), pts.shape[0]))

    def removed_vertex(self, index):
        _check(self.lib.rkh_nn_remove(self.h, int(index)))

    def live_size(self):
        return int(self.lib.rkh_nn_live_size(self.h))

    def clear(self):
        _check(self.lib.rkh_nn_clear(self.h))

    def set_coord_bound(self, bound):
        _check(self.lib.rkh_nn_set_coord_bound(self.h, float(bound)))

    def kernel_name(self):
        return self.lib.rkh_nn_kernel_name().decode()

    def fill_uniform(self, n, seed=1):
        _check(self.lib.rkh_nn_fill_uniform(self.h, n, seed))

    def nearest(self, q):
        q = np.ascontiguousarray(q, dtype=np.float64).reshape(-1, self.D)
        B = q.shape[0]
        idx = np.zeros(B, dtype=np.uint32)
        dist = np.zeros(B)
        _check(self.lib.rkh_nn_query1(self.h, T.dptr(q), B, T.u32ptr(idx), T.dptr(dist)))
        return idx, dist

    def nearest_async(self, d_q, B, d_idx, d_dist, events=None):
        if events is not None:
            _check(self.lib.rkh_nn_set_events(self.h, events[0], events[1]))
        _check(self.lib.rkh_nn_query1_async(self.h, d_q, B, d_idx, d_dist))

    def k_nearest(self, q, k, radius=np.inf):
        q = np.ascontiguousarray(q, dtype=np.float64).reshape(-1, self.D)
        B = q.shape[0]
        idx = np.zeros((B, k), dtype=np.uint32)
        dist = np.zeros((B, k))
        cnt = np.zeros(B, dtype=np.uint32)
        _check(self.lib.rkh_nn_queryk(self.h, T.dptr(q), B, k, float(radius), T.u32ptr(idx), T.dptr(dist), T.u32ptr(cnt)))
        return idx, dist, cnt

    def close(self):
        if self.h:
            self.lib.rkh_nn_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def gjk_distance(ctx, a, b, mesh_vertices=None):
    """GJK distance of the world-anchored shape pairs (a[i], b[i]) (rkh_diag_gjk_distance)."""
    n = len(a)
    aa, bb = T.as_array(list(a), T.Shape), T.as_array(list(b), T.Shape)
    verts = np.zeros((0, 3)) if mesh_vertices is None else np.ascontiguousarray(mesh_vertices, dtype=np.float64).reshape(-1, 3)
    out = np.zeros(n)
    _check(ctx.lib.rkh_diag_gjk_distance(ctx.h, aa, bb, n, T.dptr(verts) if len(verts) else None, len(verts), T.dptr(out)))
    return out


def nn_mirror_query(ctx, pts, q, coord_bound):
    """1-NN of every query through the planner-regime sweep (half-precision mirror + exact resolution,
    rkh_diag_nn_mirror_query): (index, distance) arrays."""
    pts = np.ascontiguousarray(pts, dtype=np.float64)
    q = np.ascontiguousarray(q, dtype=np.float64)
    n, D = pts.shape
    B = q.shape[0]
    idx = np.zeros(B, dtype=np.uint32)
    dist = np.zeros(B)
    _check(ctx.lib.rkh_diag_nn_mirror_query(ctx.h, T.dptr(pts), n, D, T.dptr(q), B, float(coord_bound), T.u32ptr(idx),
                                            T.dptr(dist)))
    return idx, dist


class Scene:
    def __init__(self, ctx, scn):
        self.ctx, self.lib, self.scn = ctx, ctx.lib, scn
        self.n, self.D = scn.n_dof, 2 * scn.n_dof
        self._ops = scn.ops_array()
        self._shapes = scn.shapes_array() if scn.shapes else (T.Shape * 1)()
        self.h = C.c_void_p()
        verts = getattr(scn, "mesh_vertices", None)
        if verts is not None and len(verts):  # convex vertex sets among the shapes (RKH_SHAPE_MESH)
            self._verts = np.ascontiguousarray(verts, dtype=np.float64).reshape(-1, 3)
            _check(self.lib.rkh_scene_create_with_meshes(ctx.h, self._ops, len(scn.ops), C.byref(scn.base), self._shapes,
                                                         len(scn.shapes), T.dptr(self._verts), len(self._verts),
                                                         C.byref(self.h)))
        else:
            _check(self.lib.rkh_scene_create(ctx.h, self._ops, len(scn.ops), C.byref(scn.base), self._shapes,
                                             len(scn.shapes), C.byref(self.h)))

    @property
    def num_pairs(self):
        return self.lib.rkh_scene_num_pairs(self.h)

    def state_derivative(self, x, u):
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1, self.D)
        u = np.ascontiguousarray(u, dtype=np.float64).reshape(-1, self.n)
        B = x.shape[0]
        pd, M, f = np.zeros((B, self.D)), np.zeros((B, self.n, self.n)), np.zeros((B, self.n))
        _check(self.lib.rkh_state_derivative(self.h, T.dptr(x), T.dptr(u), B, T.dptr(pd), T.dptr(M), T.dptr(f)))
        return pd, M, f

    def proximity_counts(self, x):
        """Stage counts of the two-lanes steer kernels' proximity test on the states x (rkh_diag_proximity_counts)."""
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1, self.D)
        c = (C.c_uint64 * 8)()
        _check(self.lib.rkh_diag_proximity_counts(self.h, T.dptr(x), x.shape[0], c))
        return {"states": int(c[0]), "pairs_past_cull": int(c[1]), "closed_forms": int(c[2]), "golden_section": int(c[3]),
                "states_in_collision": int(c[4]), "pairs_per_state": int(c[5]), "pairs_in_static_reach": int(c[6])}

    def diag_feval_cycles(self, x, u, iters=100):
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1, self.D)
        u = np.ascontiguousarray(u, dtype=np.float64).reshape(-1, self.n)
        out = np.zeros((x.shape[0], 8), dtype=np.uint64)
        _check(self.lib.rkh_diag_feval_cycles(self.h, T.dptr(x), T.dptr(u), x.shape[0], iters,
                                              out.ctypes.data_as(C.POINTER(C.c_uint64))))
        return out

    def min_distance(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1, self.D)
        d = np.zeros(x.shape[0])
        _check(self.lib.rkh_min_distance(self.h, T.dptr(x), x.shape[0], T.dptr(d)))
        return d

    def move_position_toward(self, lower, upper, min_interval, a, b, fraction=1.0):
        """manip_quasi_static_env::move_position_toward for B (a,b) joint-position pairs (rkh_edge_check)."""
        lower = np.ascontiguousarray(lower, dtype=np.float64)
        upper = np.ascontiguousarray(upper, dtype=np.float64)
        a = np.ascontiguousarray(a, dtype=np.float64).reshape(-1, self.n)
        b = np.ascontiguousarray(b, dtype=np.float64).reshape(-1, self.n)
        out = np.zeros_like(a)
        nchk = np.zeros(a.shape[0], dtype=np.uint32)
        _check(self.lib.rkh_edge_check(self.h, T.dptr(lower), T.dptr(upper), float(min_interval), T.dptr(a), T.dptr(b),
                                       a.shape[0], float(fraction), T.dptr(out), T.u32ptr(nchk)))
        return out, nchk

    def steer_position_toward(self, a, b, fraction=1.0, record=False, dyn=None):
        a = np.ascontiguousarray(a, dtype=np.float64).reshape(-1, self.D)
        b = np.ascontiguousarray(b, dtype=np.float64).reshape(-1, self.D)
        dyn = dyn if dyn is not None else self.scn.dyn
        B = a.shape[0]
        out = np.zeros_like(a)
        steps = np.zeros(B, dtype=np.uint32)
        rec = np.zeros((B, dyn.steps_per_edge + 1, self.D)) if record else None
        _check(self.lib.rkh_propagate(self.h, C.byref(dyn), T.dptr(a), T.dptr(b), B, float(fraction), T.dptr(out),
                                      T.u32ptr(steps), T.dptr(rec) if record else None))
        return out, steps, rec

    def close(self):
        if self.h:
            self.lib.rkh_scene_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def make_qs_space(n_dof, lower, upper, min_interval, speed_limits=None):
    """rkh_qs_space; speed_limits makes it the rate-limited joint space (points = joint / speed limit)."""
    q = T.QsSpace()
    q.n_dof = n_dof
    q.min_interval = float(min_interval)
    for i in range(n_dof):
        q.lower[i] = float(lower[i])
        q.upper[i] = float(upper[i])
        q.speed_limits[i] = 0.0 if speed_limits is None else float(speed_limits[i])
    return q


class RrtPlanner:
    """rrt_planner over the steerable dynamic space.  `prm` is one rkh_rrt_params or a list of them (a batch of
    independent problems sharing every kernel launch); `stats` is the first problem's, `all_stats` the array."""

    def __init__(self, scene, prm, dyn=None, qs=None):
        self.scene, self.lib = scene, scene.lib
        self.dyn = dyn if dyn is not None else scene.scn.dyn
        self.prms = list(prm) if isinstance(prm, (list, tuple)) else [prm]
        self.P = len(self.prms)
        self._prm_arr = T.as_array(self.prms, T.RrtParams)
        self.h = C.c_void_p()
        self.qs = qs
        if qs is not None:  # quasi-static free space (points = joint positions)
            self.D = qs.n_dof
            _check(self.lib.rkh_planner_create_qs_batch(scene.h, C.byref(qs), self._prm_arr, self.P, C.byref(self.h)))
        else:
            self.D = scene.D
            _check(self.lib.rkh_planner_create_batch(scene.h, C.byref(self.dyn), self._prm_arr, self.P, C.byref(self.h)))
        self.all_stats = (PlannerStats * self.P)()

    @property
    def stats(self):
        return self.all_stats[0]

    @property
    def done(self):
        return all(s.done for s in self.all_stats)

    @property
    def stream(self):
        return self.lib.rkh_planner_stream(self.h)

    def enqueue(self, rounds):
        _check(self.lib.rkh_planner_enqueue(self.h, rounds))

    def sync(self):
        _check(self.lib.rkh_planner_sync(self.h, self.all_stats))
        return self.all_stats[0]

    def solve_planning_query(self):
        _check(self.lib.rkh_planner_solve(self.h, self.all_stats))
        return self.all_stats[0]

    def nn_profile(self):
        ms, by, ln = C.c_double(), C.c_uint64(), C.c_uint64()
        _check(self.lib.rkh_planner_nn_profile(self.h, C.byref(ms), C.byref(by), C.byref(ln)))
        return ms.value, by.value, ln.value

    def nn_pairs(self):
        pairs = C.c_uint64()
        _check(self.lib.rkh_planner_nn_pairs(self.h, C.byref(pairs)))
        return pairs.value

    def solution(self, problem=0):
        n, cost = C.c_uint32(), C.c_double()
        _check(self.lib.rkh_planner_get_solution(self.h, problem, None, 0, C.byref(n), C.byref(cost)))
        path = np.zeros(max(n.value, 1), dtype=np.uint32)
        _check(self.lib.rkh_planner_get_solution(self.h, problem, T.u32ptr(path), len(path), C.byref(n), C.byref(cost)))
        return path[: n.value], cost.value

    def steer_profile(self):
        ms, ln = C.c_double(), C.c_uint64()
        _check(self.lib.rkh_planner_steer_profile(self.h, C.byref(ms), C.byref(ln)))
        return ms.value, ln.value

    def steer_steps(self):
        """RK4 steps the steer kernels integrated so far (executed work, not n_steps per launched edge)."""
        n = C.c_uint64()
        _check(self.lib.rkh_planner_steer_steps(self.h, C.byref(n)))
        return n.value

    def tree(self, problem=0):
        st = self.all_stats[problem]
        nv, it, D = int(st.num_vertices), int(st.iterations), self.D
        pos = np.zeros((nv, D))
        parent = np.zeros(nv, dtype=np.uint32)
        nn_seq = np.zeros(max(it, 1), dtype=np.uint32)
        accept = np.zeros(max(it, 1), dtype=np.uint8)
        gd = np.zeros(max(nv - 1, 1))
        _check(self.lib.rkh_planner_get_tree(self.h, problem, T.dptr(pos), T.u32ptr(parent), T.u32ptr(nn_seq),
                                             accept.ctypes.data_as(C.POINTER(C.c_uint8)), T.dptr(gd)))
        return {"pos": pos, "parent": parent, "nn_seq": nn_seq[:it], "accept": accept[:it], "goal_dist": gd[: nv - 1]}

    def close(self):
        if self.h:
            self.lib.rkh_planner_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class RrtPlannerPool:
    """A batch of RRT problems split over `groups` rkh_planner handles, each with its own HIP stream, driven round-robin
    from one host thread: enqueue is asynchronous, so while one group's steer kernel drains its last waves the other
    group's NN sweep and bookkeeping kernels fill the machine (measured +13-15 % with two groups of 128 problems over one
    group of 256; four groups: no further gain).  Same interface as RrtPlanner; problem indices are global."""

    def __init__(self, scene, prms, groups=2, dyn=None, qs=None):
        prms = list(prms)
        groups = max(1, min(groups, len(prms)))
        cut = [len(prms) * g // groups for g in range(groups + 1)]
        self.planners = [RrtPlanner(scene, prms[cut[g]:cut[g + 1]], dyn=dyn, qs=qs) for g in range(groups)]
        self.offsets = cut
        self.P, self.D = len(prms), self.planners[0].D

    @property
    def all_stats(self):
        return [s for pl in self.planners for s in pl.all_stats]

    @property
    def stats(self):
        return self.planners[0].all_stats[0]

    @property
    def done(self):
        return all(pl.done for pl in self.planners)

    def enqueue(self, rounds):
        for pl in self.planners:
            if rounds == 0 or not pl.done:
                pl.enqueue(rounds)

    def sync(self):
        for pl in self.planners:
            pl.sync()
        return self.stats

    def solve_planning_query(self, rounds_per_sync=16):
        self.enqueue(0)
        while True:
            self.enqueue(rounds_per_sync)
            self.sync()
            if self.done:
                return self.stats

    def _local(self, problem):
        for g, pl in enumerate(self.planners):
            if problem < self.offsets[g + 1]:
                return pl, problem - self.offsets[g]
        raise IndexError(problem)

    def tree(self, problem=0):
        pl, i = self._local(problem)
        return pl.tree(i)

    def solution(self, problem=0):
        pl, i = self._local(problem)
        return pl.solution(i)

    def nn_profile(self):
        prof = [pl.nn_profile() for pl in self.planners]
        return sum(p[0] for p in prof), sum(p[1] for p in prof), sum(p[2] for p in prof)

    def nn_pairs(self):
        return sum(pl.nn_pairs() for pl in self.planners)

    def steer_profile(self):
        prof = [pl.steer_profile() for pl in self.planners]
        return sum(p[0] for p in prof), sum(p[1] for p in prof)

    def steer_steps(self):
        return sum(pl.steer_steps() for pl in self.planners)

    def close(self):
        for pl in self.planners:
            pl.close()


class RrtStarPlanner:
    """rrtstar_planner (unidirectional, linear-search k-NN), batch of problems; `space` is a quasi-static space
    (make_qs_space: vertices = joint positions) or a steerable dynamic space (T.DynSpace: vertices = states (q, qd))."""

    def __init__(self, scene, prm, space):
        self.scene, self.lib, self.qs = scene, scene.lib, space
        self.prms = list(prm) if isinstance(prm, (list, tuple)) else [prm]
        dynamic = isinstance(space, T.DynSpace)
        self.P, self.D = len(self.prms), (2 * space.n_dof if dynamic else space.n_dof)
        self._prm_arr = T.as_array(self.prms, T.RrtParams)
        self.h = C.c_void_p()
        create = self.lib.rkh_rrtstar_create_batch if dynamic else self.lib.rkh_rrtstar_create_qs_batch
        _check(create(scene.h, C.byref(space), self._prm_arr, self.P, C.byref(self.h)))
        self.all_stats = (RrtStarStats * self.P)()

    @property
    def stats(self):
        return self.all_stats[0]

    def solve_planning_query(self, max_loop_iterations=-1):
        _check(self.lib.rkh_rrtstar_solve(self.h, int(max_loop_iterations), self.all_stats))
        return self.all_stats[0]

    def set_branch_and_bound(self, enabled=True):
        """USE_BRANCH_AND_BOUND_PRUNING_FLAG: branch_and_bound_connector instead of lazy_node_connector."""
        _check(self.lib.rkh_rrtstar_set_branch_and_bound(self.h, 1 if enabled else 0))

    def removed(self, problem=0):
        out = np.zeros(int(self.all_stats[problem].num_vertices), dtype=np.uint8)
        _check(self.lib.rkh_rrtstar_get_removed(self.h, problem, out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out

    def solution(self, problem=0):
        n, cost = C.c_uint32(), C.c_double()
        _check(self.lib.rkh_rrtstar_get_solution(self.h, problem, None, 0, C.byref(n), C.byref(cost)))
        path = np.zeros(max(n.value, 1), dtype=np.uint32)
        _check(self.lib.rkh_rrtstar_get_solution(self.h, problem, T.u32ptr(path), len(path), C.byref(n), C.byref(cost)))
        return path[: n.value], cost.value

    def graph(self, problem=0):
        st = self.all_stats[problem]
        nv, it = int(st.num_vertices), int(st.loop_iterations)
        pos = np.zeros((nv, self.D))
        pred = np.zeros(nv, dtype=np.uint32)
        dist = np.zeros(nv)
        near = np.zeros(max(it, 1), dtype=np.uint32)
        _check(self.lib.rkh_rrtstar_get_graph(self.h, problem, T.dptr(pos), T.u32ptr(pred), T.dptr(dist), T.u32ptr(near)))
        return {"pos": pos, "pred": pred, "dist": dist, "near_seq": near[:it]}

    def close(self):
        if self.h:
            self.lib.rkh_rrtstar_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BiRrtStarStats(C.Structure):
    _fields_ = [("num_vertices", C.c_uint64), ("samples", C.c_uint64), ("loop_iterations", C.c_uint64),
                ("rewires", C.c_uint64), ("fwd_rewires", C.c_uint64), ("joins", C.c_uint64), ("edges_checked", C.c_uint64),
                ("best_join_cost", C.c_double)]


class BiRrtStarPlanner:
    """rrtstar_planner with BIDIRECTIONAL_PLANNING over the quasi-static free space, batch of problems."""

    def __init__(self, scene, prm, qs):
        self.scene, self.lib, self.qs = scene, scene.lib, qs
        self.prms = list(prm) if isinstance(prm, (list, tuple)) else [prm]
        self.P, self.D = len(self.prms), qs.n_dof
        self._prm_arr = T.as_array(self.prms, T.RrtParams)
        self.h = C.c_void_p()
        _check(self.lib.rkh_birrtstar_create_qs_batch(scene.h, C.byref(qs), self._prm_arr, self.P, C.byref(self.h)))
        self.all_stats = (BiRrtStarStats * self.P)()

    @property
    def stats(self):
        return self.all_stats[0]

    def solve_planning_query(self, max_loop_iterations=-1):
        _check(self.lib.rkh_birrtstar_solve(self.h, int(max_loop_iterations), self.all_stats))
        return self.all_stats[0]

    def graph(self, problem=0):
        st = self.all_stats[problem]
        nv, it = int(st.num_vertices), max(int(st.loop_iterations), 1)
        pos = np.zeros((nv, self.D))
        pred = np.zeros(nv, dtype=np.uint32); succ = np.zeros(nv, dtype=np.uint32)
        dist = np.zeros(nv); fwd = np.zeros(nv)
        npred = np.zeros(it, dtype=np.uint32); nsucc = np.zeros(it, dtype=np.uint32)
        _check(self.lib.rkh_birrtstar_get_graph(self.h, problem, T.dptr(pos), T.u32ptr(pred), T.dptr(dist), T.u32ptr(succ),
                                                T.dptr(fwd), T.u32ptr(npred), T.u32ptr(nsucc)))
        n = int(st.loop_iterations)
        return {"pos": pos, "pred": pred, "dist": dist, "succ": succ, "fwd_dist": fwd, "near_pred": npred[:n],
                "near_succ": nsucc[:n]}

    def close(self):
        if self.h:
            self.lib.rkh_rrtstar_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PrmPlanner:
    """prm_planner (linear-search k-NN, adjacency-list motion graph), batch of problems; `space` is a quasi-static space
    (make_qs_space) or a steerable dynamic space (T.DynSpace: vertices = states (q, qd))."""

    def __init__(self, scene, prm, space):
        self.scene, self.lib, self.qs = scene, scene.lib, space
        self.prms = list(prm) if isinstance(prm, (list, tuple)) else [prm]
        dynamic = isinstance(space, T.DynSpace)
        self.P, self.D = len(self.prms), (2 * space.n_dof if dynamic else space.n_dof)
        self._prm_arr = T.as_array(self.prms, T.PrmParams)
        self.h = C.c_void_p()
        create = self.lib.rkh_prm_create_batch if dynamic else self.lib.rkh_prm_create_qs_batch
        _check(create(scene.h, C.byref(space), self._prm_arr, self.P, C.byref(self.h)))
        self.all_stats = (PrmStats * self.P)()

    @property
    def stats(self):
        return self.all_stats[0]

    def solve_planning_query(self, max_loop_iterations=-1):
        _check(self.lib.rkh_prm_solve(self.h, int(max_loop_iterations), self.all_stats))
        return self.all_stats[0]

    def graph(self, problem=0):
        st = self.all_stats[problem]
        nv, ne, it = int(st.num_vertices), int(st.num_edges), int(st.loop_iterations)
        pos = np.zeros((nv, self.D))
        eu = np.zeros(max(ne, 1), dtype=np.uint32)
        ev = np.zeros(max(ne, 1), dtype=np.uint32)
        ew = np.zeros(max(ne, 1))
        dens = np.zeros(nv)
        cc = np.zeros(nv, dtype=np.uint32)
        kind = np.zeros(max(it, 1), dtype=np.uint8)
        exp = np.zeros(max(it, 1), dtype=np.uint32)
        _check(self.lib.rkh_prm_get_graph(self.h, problem, T.dptr(pos), T.u32ptr(eu), T.u32ptr(ev), T.dptr(ew), T.dptr(dens),
                                          T.u32ptr(cc), kind.ctypes.data_as(C.POINTER(C.c_uint8)), T.u32ptr(exp)))
        return {"pos": pos, "edge_u": eu[:ne], "edge_v": ev[:ne], "edge_w": ew[:ne], "density": dens, "cc_root": cc,
                "kind": kind[:it], "expanded": exp[:it]}

    def close(self):
        if self.h:
            self.lib.rkh_prm_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BiRrtPlanner:
    """rrt_planner with BIDIRECTIONAL_PLANNING over the quasi-static free space, batch of problems."""

    def __init__(self, scene, prm, qs):
        self.scene, self.lib, self.qs = scene, scene.lib, qs
        self.prms = list(prm) if isinstance(prm, (list, tuple)) else [prm]
        self.P, self.D = len(self.prms), qs.n_dof
        self._prm_arr = T.as_array(self.prms, T.RrtParams)
        self.h = C.c_void_p()
        _check(self.lib.rkh_birrt_create_qs_batch(scene.h, C.byref(qs), self._prm_arr, self.P, C.byref(self.h)))
        self.all_stats = (BiRrtStats * self.P)()

    @property
    def stats(self):
        return self.all_stats[0]

    def solve_planning_query(self, max_loop_iterations=-1):
        _check(self.lib.rkh_birrt_solve(self.h, int(max_loop_iterations), self.all_stats))
        return self.all_stats[0]

    def solution(self, problem=0):
        n1, n2, cost = C.c_uint32(), C.c_uint32(), C.c_double()
        _check(self.lib.rkh_birrt_get_solution(self.h, problem, None, C.byref(n1), None, C.byref(n2), 0, C.byref(cost)))
        cap = max(n1.value, n2.value, 1)
        p1 = np.zeros(cap, dtype=np.uint32); p2 = np.zeros(cap, dtype=np.uint32)
        _check(self.lib.rkh_birrt_get_solution(self.h, problem, T.u32ptr(p1), C.byref(n1), T.u32ptr(p2), C.byref(n2), cap,
                                               C.byref(cost)))
        return p1[: n1.value], p2[: n2.value], cost.value

    def trees(self, problem=0):
        st = self.all_stats[problem]
        n1, n2, it = int(st.num_vertices_1), int(st.num_vertices_2), int(st.loop_iterations)
        p1 = np.zeros((n1, self.D)); q1 = np.zeros(n1, dtype=np.uint32)
        p2 = np.zeros((n2, self.D)); q2 = np.zeros(n2, dtype=np.uint32)
        nn = np.zeros(max(2 * it, 1), dtype=np.uint32); acc = np.zeros(max(2 * it, 1), dtype=np.uint8)
        _check(self.lib.rkh_birrt_get_trees(self.h, problem, T.dptr(p1), T.u32ptr(q1), T.dptr(p2), T.u32ptr(q2), T.u32ptr(nn),
                                            acc.ctypes.data_as(C.POINTER(C.c_uint8))))
        return {"pos1": p1, "parent1": q1, "pos2": p2, "parent2": q2, "nn_seq": nn[: 2 * it], "accept": acc[: 2 * it]}

    def close(self):
        if self.h:
            self.lib.rkh_birrt_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
