"""Generates tests/golden/c2_golden.npz from the CPU oracle (the restatement of the reference; the reference itself
cannot be built here).  Fixtures are data only: seeded inputs and the oracle's outputs.
Run:  python tests/make_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import oracle_lib  # noqa: E402
from reak_amd import scenarios  # noqa: E402
from reak_amd import types as T  # noqa: E402


def main():
    oracle_lib.build()
    scn = scenarios.make_c2(world_seed=1)
    osc = oracle_lib.OracleScene(scn)
    lo = np.array([scn.dyn.lower[i] for i in range(12)])
    hi = np.array([scn.dyn.upper[i] for i in range(12)])
    out = {}
    # a1: sample stream (first 64 samples of seeds 1..3)
    lib = oracle_lib.load()
    for seed in (1, 2, 3):
        s = np.zeros((64, 12))
        lib.orc_sample_hyperbox(seed, T.dptr(lo), T.dptr(hi), 12, 64, T.dptr(s))
        out[f"samples_seed{seed}"] = s
    # a10-a19: x' = f(x,u)
    rng = np.random.default_rng(20260101)
    x = rng.uniform(lo, hi, size=(256, 12))
    u = rng.uniform(-50, 50, size=(256, 6))
    rc, pd, M, f = osc.state_derivative(x, u)
    assert rc == 0
    out.update(fe_x=x, fe_u=u, fe_pd=pd, fe_M=M, fe_f=f)
    # a20-a23: proximity
    xq = np.zeros((256, 12))
    xq[:, 0::2] = rng.uniform(-np.pi, np.pi, size=(256, 6))
    out.update(prox_x=xq, prox_d=osc.min_distance(xq))
    # steer (a8-a23): 64 edges
    a = rng.uniform(lo, hi, size=(200, 12)) * 0.6
    a[:, 0::2] = rng.uniform(-2.5, 2.5, size=(200, 6))
    a = a[osc.min_distance(a) > 0.01][:64]
    b = rng.uniform(lo, hi, size=(a.shape[0], 12))
    rc, xo, steps, _ = osc.steer(a, b)
    assert rc == 0
    out.update(steer_a=a, steer_b=b, steer_x=xo, steer_steps=steps)
    # a7: planner level
    for seed in (1, 2, 3):
        prm = scn.rrt_params(seed=seed, max_vertices=1500)
        rc, o, tree = osc.rrt_dyn(prm)
        assert rc == 0
        out[f"rrt{seed}_counts"] = np.array([o.num_vertices, o.iterations, o.edges_checked, o.num_solutions], dtype=np.int64)
        out[f"rrt{seed}_nn_seq"] = tree["nn_seq"]
        out[f"rrt{seed}_accept"] = tree["accept"]
        out[f"rrt{seed}_parent"] = tree["parent"]
        out[f"rrt{seed}_pos"] = tree["pos"]
    # a3/a4: NN on a seeded cloud
    pts = rng.uniform(-3, 3, size=(3000, 12))
    q = rng.uniform(-3, 3, size=(40, 12))
    idx, dist = oracle_lib.nn1(q, pts)
    kidx, kdist, kcnt = oracle_lib.knn(q, pts, 52, radius=4.0)
    out.update(nn_pts=pts, nn_q=q, nn_idx=idx, nn_dist=dist, knn_idx=kidx, knn_dist=kdist, knn_cnt=kcnt)
    os.makedirs(os.path.join(ROOT, "tests", "golden"), exist_ok=True)
    path = os.path.join(ROOT, "tests", "golden", "c2_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")
    make_c1_golden(scenarios.make_c1(world_seed=1), "c1_golden.npz")
    make_c1_golden(scenarios.make_c1_planar(world_seed=1), "c1_planar_golden.npz")
    make_round2_golden()


def make_c1_golden(c1, file_name):
    """C1 (3-DOF planar arm, quasi-static space; once with the 3D KTEs / shapes, once with the reference's 2D classes):
    RRT, RRT* and PRM graphs of the oracle, small sizes."""
    osc = oracle_lib.OracleScene(c1)
    lo, hi, mi = c1.meta["lower"], c1.meta["upper"], c1.meta["min_interval"]
    out = {}
    rng = np.random.default_rng(20260102)
    a = rng.uniform(lo, hi, size=(400, 3))
    x = np.zeros((400, 6)); x[:, 0::2] = a
    a = a[osc.min_distance(x) > 0.0][:128]
    b = rng.uniform(lo, hi, size=(a.shape[0], 3))
    mv, nchk = osc.qs_move(lo, hi, mi, a, b, fraction=1.0)
    out.update(walk_a=a, walk_b=b, walk_out=mv, walk_nchk=nchk)
    rc, o, tree = osc.rrt_qs(lo, hi, mi, c1.rrt_params(seed=1, max_vertices=800))
    out.update(rrt_counts=np.array([o.num_vertices, o.iterations, o.edges_checked, o.num_solutions], dtype=np.int64),
               rrt_parent=tree["parent"], rrt_pos=tree["pos"], rrt_accept=tree["accept"])
    rc, o, g = osc.rrtstar_qs(lo, hi, mi, c1.rrt_params(seed=1, max_vertices=600))
    out.update(star_counts=np.array([o.num_vertices, o.samples, o.loop_iterations, o.num_solutions, o.rewires,
                                     o.edges_checked], dtype=np.int64),
               star_best=np.array([o.best_cost]), star_pred=g["pred"], star_dist=g["dist"], star_pos=g["pos"],
               star_near=g["near_seq"])
    rc, o, g = osc.prm_qs(lo, hi, mi, c1.prm_params(seed=1, max_vertices=500, sampling_radius=1.0))
    out.update(prm_counts=np.array([o.num_vertices, o.num_edges, o.samples, o.rejected, o.loop_iterations,
                                    o.num_components, o.publish_calls, o.merged_at_vertex, o.edges_checked], dtype=np.int64),
               prm_pos=g["pos"], prm_edge_u=g["edge_u"], prm_edge_v=g["edge_v"], prm_edge_w=g["edge_w"],
               prm_density=g["density"], prm_cc_root=g["cc_root"], prm_kind=g["kind"], prm_expanded=g["expanded"])
    path = os.path.join(ROOT, "tests", "golden", file_name)
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


def round2_cases(osc_of):
    """The planners added in round 2, small sizes: bidirectional RRT*, RRT* with branch-and-bound pruning, RRT* in the
    rate-limited joint space (all C1), RRT* and PRM over the dynamic state space of the planar arm.  `osc_of(scn)` gives
    the object with the oracle's call surface (the oracle itself, or the HIP path behind the same names)."""
    out = {}
    c1 = scenarios.make_c1(world_seed=1)
    lo, hi, mi = c1.meta["lower"], c1.meta["upper"], c1.meta["min_interval"]
    o = osc_of(c1)
    c, g = o.birrtstar(lo, hi, mi, c1.rrt_params(seed=5, max_vertices=400))
    out.update(bistar_counts=np.array(c, dtype=np.int64), bistar_pred=g["pred"], bistar_succ=g["succ"],
               bistar_dist=g["dist"], bistar_fwd=g["fwd_dist"], bistar_pos=g["pos"])
    c, g = o.bnb(lo, hi, mi, c1.rrt_params(seed=5, max_vertices=400), 1500)
    out.update(bnb_counts=np.array(c, dtype=np.int64), bnb_pred=g["pred"], bnb_dist=g["dist"], bnb_pos=g["pos"],
               bnb_removed=g["removed"])
    speed = np.array([2.0, 0.5, 1.25])
    prm = c1.rrt_params(seed=6, max_vertices=300)
    for d in range(3):
        prm.start[d] /= speed[d]
        prm.goal[d] /= speed[d]
    c, g = o.rrtstar_rl(np.asarray(lo) / speed, np.asarray(hi) / speed, mi, speed, prm)
    out.update(rl_counts=np.array(c, dtype=np.int64), rl_pred=g["pred"], rl_dist=g["dist"], rl_pos=g["pos"])
    pd = scenarios.make_c1_planar(world_seed=1, dynamics=True)
    o = osc_of(pd)
    prm = pd.rrt_params(seed=7, max_vertices=250)
    prm.conn_tol = 3.0
    c, g = o.rrtstar_dyn(prm)
    out.update(dynstar_counts=np.array(c, dtype=np.int64), dynstar_pred=g["pred"], dynstar_pos=g["pos"],
               dynstar_dist=g["dist"])
    pp = pd.prm_params(seed=8, max_vertices=200, sampling_radius=1.0)
    pp.base.conn_tol = 3.0
    c, g = o.prm_dyn(pp)
    out.update(dynprm_counts=np.array(c, dtype=np.int64), dynprm_pos=g["pos"], dynprm_edge_u=g["edge_u"],
               dynprm_edge_v=g["edge_v"], dynprm_kind=g["kind"])
    return out


class OracleSurface:
    """oracle_lib behind the call surface of round2_cases."""

    def __init__(self, scn):
        self.osc = oracle_lib.OracleScene(scn)

    def birrtstar(self, lo, hi, mi, prm):
        rc, o, g = self.osc.birrtstar_qs(lo, hi, mi, prm)
        return [o.num_vertices, o.samples, o.loop_iterations, o.rewires, o.fwd_rewires, o.joins, o.edges_checked], g

    def bnb(self, lo, hi, mi, prm, iters):
        rc, o, g, pruned, skipped = self.osc.bnb_rrtstar_qs(lo, hi, mi, prm, max_loop_iterations=iters)
        return [o.num_vertices, o.samples, o.loop_iterations, o.num_solutions, o.rewires, o.edges_checked, pruned, skipped], g

    def rrtstar_rl(self, lo, hi, mi, speed, prm):
        oracle_lib.set_qs_speed_limits(speed)
        try:
            rc, o, g = self.osc.rrtstar_qs(lo, hi, mi, prm)
        finally:
            oracle_lib.set_qs_speed_limits(None)
        return [o.num_vertices, o.samples, o.loop_iterations, o.num_solutions, o.rewires, o.edges_checked], g

    def rrtstar_dyn(self, prm):
        rc, o, g = self.osc.rrtstar_dyn(prm)
        return [o.num_vertices, o.samples, o.loop_iterations, o.num_solutions, o.rewires, o.edges_checked], g

    def prm_dyn(self, pp):
        rc, o, g = self.osc.prm_dyn(pp)
        return [o.num_vertices, o.num_edges, o.samples, o.rejected, o.loop_iterations, o.num_components, o.edges_checked], g


def make_round2_golden():
    out = round2_cases(OracleSurface)
    path = os.path.join(ROOT, "tests", "golden", "round2_planners_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
