"""Committed golden vectors (tests/golden/c2_golden.npz, written by tests/make_golden.py from the oracle).
CPU: the oracle still reproduces them.  GPU: the HIP path reproduces them (same bars as test_gpu_parity.py)."""
import os

import numpy as np
import pytest

from reak_amd import scenarios
from reak_amd import types as T

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "c2_golden.npz")
GOLD_C1 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "c1_golden.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD, allow_pickle=False)


@pytest.fixture(scope="module")
def c2():
    return scenarios.make_c2(world_seed=1)


def test_oracle_reproduces_golden(oracle, gold, c2):
    osc = oracle.OracleScene(c2)
    lib = oracle.load()
    lo = np.array([c2.dyn.lower[i] for i in range(12)])
    hi = np.array([c2.dyn.upper[i] for i in range(12)])
    for seed in (1, 2, 3):
        s = np.zeros((64, 12))
        lib.orc_sample_hyperbox(seed, T.dptr(lo), T.dptr(hi), 12, 64, T.dptr(s))
        assert np.array_equal(s, gold[f"samples_seed{seed}"])
    rc, pd, M, f = osc.state_derivative(gold["fe_x"], gold["fe_u"])
    assert rc == 0 and np.allclose(pd, gold["fe_pd"], rtol=1e-12, atol=1e-12) and np.allclose(M, gold["fe_M"], rtol=1e-13)
    assert np.allclose(osc.min_distance(gold["prox_x"]), gold["prox_d"], atol=1e-13)
    rc, xo, steps, _ = osc.steer(gold["steer_a"], gold["steer_b"])
    assert np.array_equal(steps, gold["steer_steps"]) and np.allclose(xo, gold["steer_x"], rtol=1e-11, atol=1e-12)
    idx, dist = oracle.nn1(gold["nn_q"], gold["nn_pts"])
    assert np.array_equal(idx, gold["nn_idx"]) and np.array_equal(dist, gold["nn_dist"])
    kidx, kdist, kcnt = oracle.knn(gold["nn_q"], gold["nn_pts"], 52, radius=4.0)
    assert np.array_equal(kidx, gold["knn_idx"]) and np.array_equal(kdist, gold["knn_dist"]) and np.array_equal(kcnt, gold["knn_cnt"])
    prm = c2.rrt_params(seed=2, max_vertices=1500)
    rc, o, tree = osc.rrt_dyn(prm)
    assert list(gold["rrt2_counts"]) == [o.num_vertices, o.iterations, o.edges_checked, o.num_solutions]
    assert np.array_equal(tree["parent"], gold["rrt2_parent"]) and np.array_equal(tree["accept"], gold["rrt2_accept"])


@pytest.mark.gpu
def test_hip_reproduces_golden(gold, c2):
    from reak_amd import lib as L

    ctx = L.Context(0)
    sc = L.Scene(ctx, c2)
    pd, M, f = sc.state_derivative(gold["fe_x"], gold["fe_u"])
    assert np.allclose(pd, gold["fe_pd"], rtol=1e-10, atol=1e-10)
    assert np.max(np.abs(M - gold["fe_M"])) <= 1e-13 * np.abs(gold["fe_M"]).max()
    assert np.allclose(f, gold["fe_f"], rtol=1e-11, atol=1e-11)
    d = sc.min_distance(gold["prox_x"])
    assert np.allclose(d, gold["prox_d"], atol=1e-12)
    far = np.abs(gold["prox_d"]) > 1e-12
    assert np.array_equal((d < 0)[far], (gold["prox_d"] < 0)[far])
    xo, steps, _ = sc.steer_position_toward(gold["steer_a"], gold["steer_b"])
    assert np.array_equal(steps, gold["steer_steps"])
    assert np.allclose(xo, gold["steer_x"], rtol=1e-10, atol=1e-12)
    nn = L.HipNeighborSearch(ctx, 12, 3000)
    nn.added_vertices(gold["nn_pts"])
    idx, dist = nn.nearest(gold["nn_q"])
    assert np.array_equal(idx, gold["nn_idx"]) and np.array_equal(dist, gold["nn_dist"])
    # k-NN + radius (k = 52 = 4 (floor(log2 5000) + 1), the C1 neighbourhood): counts and distances bit for bit; the
    # indices too wherever the distances are distinct (among exact ties the reference's order is std::heap-defined)
    kidx, kdist, kcnt = nn.k_nearest(gold["nn_q"], 52, radius=4.0)
    assert np.array_equal(kcnt, gold["knn_cnt"]) and np.array_equal(kdist, gold["knn_dist"])
    for b in range(len(kcnt)):
        c = int(kcnt[b])
        if c == 0:
            continue
        d = kdist[b, :c]
        distinct = np.r_[True, d[1:] != d[:-1]] & np.r_[d[:-1] != d[1:], True]
        assert np.array_equal(kidx[b, :c][distinct], gold["knn_idx"][b, :c][distinct])
        assert set(kidx[b, :c]) == set(gold["knn_idx"][b, :c])
    # planner level: three seeds in ONE batch (one launch per kernel per round for all of them)
    pl = L.RrtPlanner(sc, [c2.rrt_params(seed=s, max_vertices=1500) for s in (1, 2, 3)])
    pl.solve_planning_query()
    for i, seed in enumerate((1, 2, 3)):
        st = pl.all_stats[i]
        tree = pl.tree(i)
        assert [st.num_vertices, st.iterations, st.edges_checked, st.num_solutions] == list(gold[f"rrt{seed}_counts"])
        assert np.array_equal(tree["nn_seq"], gold[f"rrt{seed}_nn_seq"])
        assert np.array_equal(tree["accept"], gold[f"rrt{seed}_accept"])
        assert np.array_equal(tree["parent"], gold[f"rrt{seed}_parent"])
        assert np.allclose(tree["pos"], gold[f"rrt{seed}_pos"], rtol=1e-10, atol=1e-12)


# ------------------------------------------------------------------ C1: quasi-static RRT / RRT* / PRM
# two realisations of C1: the 3D KTEs / shapes, and the reference's own 2D classes (c1_planar_golden.npz)
@pytest.fixture(scope="module", params=["3d", "planar"])
def gold1(request):
    name = "c1_golden.npz" if request.param == "3d" else "c1_planar_golden.npz"
    g = dict(np.load(os.path.join(os.path.dirname(GOLD_C1), name), allow_pickle=False))
    g["_variant"] = request.param
    return g


def _c1(gold1):
    c1 = scenarios.make_c1(world_seed=1) if gold1["_variant"] == "3d" else scenarios.make_c1_planar(world_seed=1)
    return c1, c1.meta["lower"], c1.meta["upper"], c1.meta["min_interval"]


def _check_c1(gold1, walk, rrt, star, prm):
    mv, nchk = walk
    assert np.array_equal(nchk, gold1["walk_nchk"]) and np.array_equal(mv, gold1["walk_out"])
    counts, tree = rrt
    assert list(gold1["rrt_counts"]) == counts
    assert np.array_equal(tree["parent"], gold1["rrt_parent"]) and np.array_equal(tree["pos"], gold1["rrt_pos"])
    assert np.array_equal(tree["accept"], gold1["rrt_accept"])
    counts, best, g = star
    assert list(gold1["star_counts"]) == counts and best == gold1["star_best"][0]
    assert np.array_equal(g["pred"], gold1["star_pred"]) and np.array_equal(g["dist"], gold1["star_dist"])
    assert np.array_equal(g["pos"], gold1["star_pos"]) and np.array_equal(g["near_seq"], gold1["star_near"])
    counts, g = prm
    assert list(gold1["prm_counts"]) == counts
    for k in ("pos", "edge_u", "edge_v", "edge_w", "density", "cc_root", "kind", "expanded"):
        assert np.array_equal(g[k], gold1["prm_" + k]), k


def test_oracle_reproduces_c1_golden(oracle, gold1):
    c1, lo, hi, mi = _c1(gold1)
    osc = oracle.OracleScene(c1)
    walk = osc.qs_move(lo, hi, mi, gold1["walk_a"], gold1["walk_b"], fraction=1.0)
    rc, o, tree = osc.rrt_qs(lo, hi, mi, c1.rrt_params(seed=1, max_vertices=800))
    rrt = ([o.num_vertices, o.iterations, o.edges_checked, o.num_solutions], tree)
    rc, o, g = osc.rrtstar_qs(lo, hi, mi, c1.rrt_params(seed=1, max_vertices=600))
    star = ([o.num_vertices, o.samples, o.loop_iterations, o.num_solutions, o.rewires, o.edges_checked], o.best_cost, g)
    rc, o, g = osc.prm_qs(lo, hi, mi, c1.prm_params(seed=1, max_vertices=500, sampling_radius=1.0))
    prm = ([o.num_vertices, o.num_edges, o.samples, o.rejected, o.loop_iterations, o.num_components, o.publish_calls,
            o.merged_at_vertex, o.edges_checked], g)
    _check_c1(gold1, walk, rrt, star, prm)


@pytest.mark.gpu
def test_hip_reproduces_c1_golden(gold1):
    from reak_amd import lib as L

    c1, lo, hi, mi = _c1(gold1)
    ctx = L.Context(0)
    sc = L.Scene(ctx, c1)
    qs = L.make_qs_space(3, lo, hi, mi)
    walk = sc.move_position_toward(lo, hi, mi, gold1["walk_a"], gold1["walk_b"], fraction=1.0)
    pl = L.RrtPlanner(sc, c1.rrt_params(seed=1, max_vertices=800), qs=qs)
    st = pl.solve_planning_query()
    rrt = ([st.num_vertices, st.iterations, st.edges_checked, st.num_solutions], pl.tree())
    ps = L.RrtStarPlanner(sc, c1.rrt_params(seed=1, max_vertices=600), qs)
    st = ps.solve_planning_query()
    star = ([st.num_vertices, st.samples, st.loop_iterations, st.num_solutions, st.rewires, st.edges_checked], st.best_cost,
            ps.graph())
    pp = L.PrmPlanner(sc, c1.prm_params(seed=1, max_vertices=500, sampling_radius=1.0), qs)
    st = pp.solve_planning_query()
    prm = ([st.num_vertices, st.num_edges, st.samples, st.rejected, st.loop_iterations, st.num_components, st.publish_calls,
            st.merged_at_vertex, st.edges_checked], pp.graph())
    _check_c1(gold1, walk, rrt, star, prm)


# ------------------------------------------------------------------ round-2 planners
STATE_KEYS = ("dynstar_pos", "dynstar_dist", "dynprm_pos")   # propagated states: fp64 to 1e-10 (device sincos), the rest exact


def _check_round2(got):
    gold = dict(np.load(os.path.join(os.path.dirname(GOLD_C1), "round2_planners_golden.npz"), allow_pickle=False))
    assert set(got) == set(gold)
    for k, v in gold.items():
        if k in STATE_KEYS:
            fin = np.isfinite(v)
            assert np.array_equal(np.isfinite(got[k]), fin) and np.allclose(got[k][fin], v[fin], rtol=1e-9, atol=1e-12), k
        else:
            assert np.array_equal(got[k], v), k


def test_oracle_reproduces_round2_golden(oracle):
    import make_golden

    _check_round2(make_golden.round2_cases(make_golden.OracleSurface))


@pytest.mark.gpu
def test_hip_reproduces_round2_golden():
    import make_golden
    from reak_amd import lib as L

    ctx = L.Context(0)

    class HipSurface:
        def __init__(self, scn):
            self.scn, self.sc = scn, L.Scene(ctx, scn)

        def birrtstar(self, lo, hi, mi, prm):
            pl = L.BiRrtStarPlanner(self.sc, prm, L.make_qs_space(self.scn.n_dof, lo, hi, mi))
            o = pl.solve_planning_query()
            return [o.num_vertices, o.samples, o.loop_iterations, o.rewires, o.fwd_rewires, o.joins, o.edges_checked], pl.graph()

        def bnb(self, lo, hi, mi, prm, iters):
            pl = L.RrtStarPlanner(self.sc, prm, L.make_qs_space(self.scn.n_dof, lo, hi, mi))
            pl.set_branch_and_bound(True)
            o = pl.solve_planning_query(max_loop_iterations=iters)
            g = pl.graph()
            g["removed"] = pl.removed()
            return [o.num_vertices, o.samples, o.loop_iterations, o.num_solutions, o.rewires, o.edges_checked, o.pruned,
                    o.skipped], g

        def rrtstar_rl(self, lo, hi, mi, speed, prm):
            pl = L.RrtStarPlanner(self.sc, prm, L.make_qs_space(self.scn.n_dof, lo, hi, mi, speed_limits=speed))
            o = pl.solve_planning_query()
            return [o.num_vertices, o.samples, o.loop_iterations, o.num_solutions, o.rewires, o.edges_checked], pl.graph()

        def rrtstar_dyn(self, prm):
            pl = L.RrtStarPlanner(self.sc, prm, self.scn.dyn)
            o = pl.solve_planning_query()
            return [o.num_vertices, o.samples, o.loop_iterations, o.num_solutions, o.rewires, o.edges_checked], pl.graph()

        def prm_dyn(self, pp):
            pl = L.PrmPlanner(self.sc, pp, self.scn.dyn)
            o = pl.solve_planning_query()
            return [o.num_vertices, o.num_edges, o.samples, o.rejected, o.loop_iterations, o.num_components,
                    o.edges_checked], pl.graph()

    _check_round2(make_golden.round2_cases(HipSurface))
