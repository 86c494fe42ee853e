"""GPU parity tests: the HIP path (through the C-ABI) against the CPU oracle on the same seeded inputs.

Bars: bit-exact for integer / index work (NN indices and distances, accept bits, parents, node counts,
free-step counts); propagated fp64 states within 1e-10 relative (sin/cos are OCML on the device and glibc
in the oracle, everything else rounds identically because the kernels are built with -ffp-contract=off)."""
import os

import numpy as np
import pytest

from reak_amd import scenarios
from reak_amd import types as T

pytestmark = pytest.mark.gpu

STATE_RTOL = 1e-10


@pytest.fixture(scope="module")
def L():
    from reak_amd import lib

    return lib


@pytest.fixture(scope="module")
def ctx(L):
    return L.Context(0)


@pytest.fixture(scope="module")
def c2():
    return scenarios.make_c2(world_seed=1)


# ------------------------------------------------------------------ nearest neighbour
@pytest.mark.parametrize("D,n,B", [(12, 5000, 37), (12, 1, 5), (3, 777, 300), (6, 100000, 8), (12, 65536, 512), (2, 300, 1)])
def test_nn1_bit_exact(L, ctx, oracle, D, n, B):
    rng = np.random.default_rng(100 + D + n)
    pts = rng.uniform(-3, 3, size=(n, D))
    q = rng.uniform(-3, 3, size=(B, D))
    nn = L.HipNeighborSearch(ctx, D, n + 10)
    nn.added_vertices(pts)
    idx, dist = nn.nearest(q)
    ridx, rdist = oracle.nn1(q, pts)
    assert np.array_equal(idx, ridx)
    assert np.array_equal(dist, rdist)  # bit-exact incl. the correctly rounded sqrt


@pytest.mark.parametrize("D,n,B,expect", [(12, 30000, 300, "nn1_sweep_bf16_kernel"), (12, 777, 129, "nn1_sweep_bf16_kernel"),
                                           (6, 20000, 200, "nn1_sweep_bf16_kernel"), (3, 5000, 1000, "nn1_sweep_mfma_kernel"),
                                           (16, 9000, 130, "nn1_sweep_bf16_kernel"), (12, 30000, 48, "nn1_sweep_bf16_kernel"), (24, 4000, 48, "nn1_sweep_f32_kernel"), (7, 5000, 64, "nn1_sweep_bf16_kernel"), (8, 40000, 500, "nn1_sweep_bf16_kernel"),
                                           (12, 600000, 200, "nn1_sweep_bf16_kernel"), (6, 8192, 1000, "nn1_sweep_bf16_kernel"),
                                           (24, 4000, 200, "nn1_sweep_f32_kernel"),
                                           # few queries over a large tree: one wave per 32-row slab, no LDS tile
                                           (12, 70001, 8, "nn1_few_mfma_kernel"), (12, 65536, 1, "nn1_stream_kernel"),
                                           (6, 100003, 32, "nn1_few_mfma_kernel"), (3, 131072, 17, "nn1_few_mfma_kernel"),
                                           (16, 66000, 5, "nn1_few_mfma_kernel"), (7, 80000, 31, "nn1_few_mfma_kernel"),
                                           (2, 70000, 8, "nn1_few_mfma_kernel"), (12, 8192, 8, "nn1_few_mfma_kernel"),
                                           (7, 70000, 3, "nn1_few_mfma_kernel"), (12, 8191, 8, "nn1_stream_kernel"),
                                           (12, 1048577, 32, "nn1_few_mfma_kernel")])
def test_nn1_single_precision_prefilters_are_bit_exact(L, ctx, oracle, D, n, B, expect):
    """Sweeps with a coordinate bound run a float pre-filter (fp32 matrix cores above 64 queries and, over trees of at
    least 8192 rows, for 5 to 32 queries; packed fp32 VALU between) in front of the exact fp64 test: indices and distances must not change.  The cloud contains exact duplicates (ties
    resolve to the lower index), near-duplicates one ulp apart and queries sitting on vertices (distance 0)."""
    rng = np.random.default_rng(7 * D + n + B)
    pts = rng.uniform(-np.pi, np.pi, size=(n, D))
    dup = rng.integers(0, n // 2, size=n // 50)
    pts[n // 2 + np.arange(len(dup))] = pts[dup]                       # exact duplicates at higher indices
    near = rng.integers(0, n // 2, size=n // 50)
    pts[n - 1 - np.arange(len(near))] = np.nextafter(pts[near], np.inf)  # one ulp away
    q = rng.uniform(-np.pi, np.pi, size=(B, D))
    q[::7] = pts[rng.integers(0, n, size=len(q[::7]))]                 # queries on vertices
    q[1::7] = pts[dup[rng.integers(0, len(dup), size=len(q[1::7]))]] + rng.normal(0, 1e-9, size=(len(q[1::7]), D))
    if n > 400:  # more coincident vertices than a lane's candidate list holds, inside one 32-row slab and across slabs
        pts[100:124] = pts[100]
        pts[300:400:7] = pts[100]
        q[2 % B] = pts[100] + 1e-9
        q[3 % B] = pts[100]
    nn = L.HipNeighborSearch(ctx, D, n + 10)
    nn.added_vertices(pts)
    nn.set_coord_bound(np.pi + 1e-6)
    idx, dist = nn.nearest(q)
    assert nn.kernel_name() == expect
    ridx, rdist = oracle.nn1(q, pts)
    assert np.array_equal(idx, ridx)
    assert np.array_equal(dist, rdist)


@pytest.mark.parametrize("D,n,B,bound", [(12, 30000, 300, np.pi), (12, 777, 129, np.pi), (12, 1, 5, np.pi), (12, 33, 1, 3.0),
                                         (6, 100000, 500, np.pi), (3, 5000, 1000, 1.0), (2, 300, 37, 0.5), (7, 5000, 64, 2.0),
                                         (12, 65536, 385, 30.0), (12, 200000, 777, np.pi), (8, 40000, 32, 0.01)])
def test_nn1_mirror_sweep_is_bit_exact(L, ctx, oracle, D, n, B, bound):
    """The planner-regime sweep (nn_mirror.hip): one half-precision matrix instruction per 32 x 32 (vertex, query) pairs
    over the stored mirror of the rows selects the rows whose fp64 distance is evaluated -- index and distance must be
    those of the linear search.  The cloud contains exact duplicates (ties resolve to the lower index), near-duplicates
    one ulp apart, queries sitting on vertices, queries 1e-9 away from duplicated vertices, and more coincident vertices
    than a query's candidate list holds (the resolve kernel's exact scan); sizes cover one row, partial slabs, several
    query blocks (B > 384) and coordinate bounds from 0.01 to 30."""
    rng = np.random.default_rng(11 * D + n + B)
    pts = rng.uniform(-bound, bound, size=(n, D))
    q = rng.uniform(-bound, bound, size=(B, D))
    if n >= 100:
        dup = rng.integers(0, n // 2, size=n // 50)
        pts[n // 2 + np.arange(len(dup))] = pts[dup]                       # exact duplicates at higher indices
        near = rng.integers(0, n // 2, size=n // 50)
        pts[n - 1 - np.arange(len(near))] = np.clip(np.nextafter(pts[near], np.inf), -bound, bound)  # one ulp away
        q[::7] = pts[rng.integers(0, n, size=len(q[::7]))]                 # queries on vertices
        q[1::7] = np.clip(pts[dup[rng.integers(0, len(dup), size=len(q[1::7]))]] + rng.normal(0, 1e-9 * bound, size=(len(q[1::7]), D)),
                          -bound, bound)
    if n > 400:  # 24 + 15 coincident vertices: beyond the 16 candidate rows a query keeps
        pts[100:124] = pts[100]
        pts[300:400:7] = pts[100]
        q[2 % B] = np.clip(pts[100] + 1e-9 * bound, -bound, bound)
        q[3 % B] = pts[100]
    idx, dist = L.nn_mirror_query(ctx, pts, q, bound)
    ridx, rdist = oracle.nn1(q, pts)
    assert np.array_equal(idx, ridx)
    assert np.array_equal(dist, rdist)


@pytest.mark.parametrize("scale", [1e-9, 1e9])
def test_prefilters_step_aside_for_extreme_scales(L, ctx, oracle, scale):
    """The pre-filters' error analysis assumes float products and bf16 pieces in the normal range: clouds with a coordinate
    bound outside [1e-6, 1e6] are swept by the exact fp64 kernels."""
    rng = np.random.default_rng(77)
    pts = rng.uniform(-1, 1, size=(20000, 12)) * scale
    q = rng.uniform(-1, 1, size=(300, 12)) * scale
    nn = L.HipNeighborSearch(ctx, 12, 20000)
    nn.added_vertices(pts)
    nn.set_coord_bound(scale)
    idx, dist = nn.nearest(q)
    assert nn.kernel_name() == "nn1_sweep_kernel"
    ridx, rdist = oracle.nn1(q, pts)
    assert np.array_equal(idx, ridx) and np.array_equal(dist, rdist)


def test_coordinate_bound_is_checked_where_the_library_holds_the_data(L, ctx):
    """rkh_nn_set_coord_bound is a promise the pre-filters' exactness rests on: rows and host queries beyond it are refused."""
    rng = np.random.default_rng(5)
    pts = rng.uniform(-1, 1, size=(1000, 6))
    nn = L.HipNeighborSearch(ctx, 6, 2000)
    nn.added_vertices(pts)
    with pytest.raises(L.RkhError):
        nn.set_coord_bound(0.5)          # rows already stored exceed it
    nn.set_coord_bound(1.0)
    bad = pts[:10].copy()
    bad[3, 2] = 1.5
    with pytest.raises(L.RkhError):
        nn.added_vertices(bad)
    assert len(nn) == 1000
    q = rng.uniform(-1, 1, size=(100, 6))
    q[50, 0] = -1.0001
    with pytest.raises(L.RkhError):
        nn.nearest(q)
    q[50, 0] = -1.0
    idx, _ = nn.nearest(q)
    assert np.all(idx < 1000)


@pytest.mark.parametrize("D,n,bound", [(12, 40000, 0.0), (12, 40000, 3.0), (6, 3000, 3.0), (3, 900, 0.0), (12, 70000, 3.0)])
def test_removed_vertices_are_never_returned(L, ctx, oracle, D, n, bound):
    """any_knn_synchro::removed_vertex: tombstoned rows keep their index and vanish from every sweep (register-direct,
    tiled fp64, packed-fp32 and matrix-core 1-NN; k-NN with radius).  Checked against the oracle's linear search over the
    remaining vertices, indices mapped back."""
    rng = np.random.default_rng(31 + D + n)
    pts = rng.uniform(-3, 3, size=(n, D))
    nn = L.HipNeighborSearch(ctx, D, n)
    nn.added_vertices(pts)
    if bound > 0.0:
        nn.set_coord_bound(bound)
    queries = {B: rng.uniform(-3, 3, size=(B, D)) for B in (5, 20, 48, 200)}
    dead = np.zeros(n, dtype=bool)
    for B, q in queries.items():     # remove the current nearest neighbour of every query, and a random tenth of the rest
        idx, _ = nn.nearest(q)
        dead[idx] = True
    dead[rng.choice(n, size=n // 10, replace=False)] = True
    dead[n - 1] = True               # the last row
    dead[:300] = True                # a whole tile and more
    for i in np.flatnonzero(dead):
        nn.removed_vertex(i)
    nn.removed_vertex(int(np.flatnonzero(dead)[0]))   # twice: allowed
    assert len(nn) == n and nn.live_size() == n - dead.sum()
    live = np.flatnonzero(~dead)
    for B, q in queries.items():
        idx, dist = nn.nearest(q)
        ridx, rdist = oracle.nn1(q, pts[live])
        assert np.array_equal(idx, live[ridx]) and np.array_equal(dist, rdist)
    q = queries[20]
    kidx, kdist, kcnt = nn.k_nearest(q, 24, radius=2.5 if D > 3 else 0.4)
    ridx, rdist, rcnt = oracle.knn(q, pts[live], 24, radius=2.5 if D > 3 else 0.4)
    assert np.array_equal(kcnt, rcnt)
    for b in range(len(q)):
        c = int(kcnt[b])
        assert np.array_equal(np.sort(kidx[b, :c]), np.sort(live[ridx[b, :c]])) and np.array_equal(kdist[b, :c], rdist[b, :c])
    with pytest.raises(L.RkhError):
        nn.removed_vertex(n)         # no such vertex
    # rows appended after removals take new indices; a cleared store forgets its tombstones
    nn.clear()
    nn.added_vertices(pts[:10])
    idx, _ = nn.nearest(pts[:10])
    assert np.array_equal(idx, np.arange(10))


def test_nn1_ties_first_minimum_wins(L, ctx, oracle):
    rng = np.random.default_rng(7)
    base = rng.uniform(-1, 1, size=(500, 12))
    pts = np.concatenate([base, base, base[::-1]])  # every point three times, at scattered indices
    q = base[rng.integers(0, 500, size=64)] + 1e-3
    nn = L.HipNeighborSearch(ctx, 12, pts.shape[0])
    nn.added_vertices(pts[:700])
    nn.added_vertices(pts[700:])  # appended in two goes (added_vertex synchro)
    idx, dist = nn.nearest(q)
    ridx, rdist = oracle.nn1(q, pts)
    assert np.array_equal(idx, ridx) and np.array_equal(dist, rdist)
    assert np.all(idx < 500)


def test_nn1_empty_and_growth(L, ctx, oracle):
    nn = L.HipNeighborSearch(ctx, 12, 1000)
    idx, dist = nn.nearest(np.zeros((3, 12)))
    assert np.all(idx == 0xFFFFFFFF) and np.all(np.isinf(dist))
    rng = np.random.default_rng(3)
    pts = rng.normal(size=(1000, 12))
    for lo, hi in [(0, 1), (1, 255), (255, 257), (257, 1000)]:
        nn.added_vertices(pts[lo:hi])
        q = rng.normal(size=(9, 12))
        idx, dist = nn.nearest(q)
        ridx, rdist = oracle.nn1(q, pts[:hi])
        assert np.array_equal(idx, ridx) and np.array_equal(dist, rdist)
    with pytest.raises(L.RkhError):
        nn.added_vertices(pts[:100])  # capacity exceeded -> error, not silent truncation


@pytest.mark.parametrize("D,n,B,k,radius", [(12, 5000, 33, 52, np.inf), (12, 40, 7, 52, np.inf), (12, 70000, 64, 68, 2.5),
                                               (3, 3000, 5, 8, 0.4), (6, 1, 3, 4, np.inf), (12, 300000, 16, 76, np.inf)])
def test_knn_matches_linear_search(L, ctx, oracle, D, n, B, k, radius):
    rng = np.random.default_rng(500 + n + k)
    pts = rng.uniform(-3, 3, size=(n, D))
    q = rng.uniform(-3, 3, size=(B, D))
    nn = L.HipNeighborSearch(ctx, D, n)
    nn.added_vertices(pts)
    idx, dist, cnt = nn.k_nearest(q, k, radius)
    ridx, rdist, rcnt = oracle.knn(q, pts, k, radius)
    assert np.array_equal(cnt, rcnt)
    assert np.array_equal(dist, rdist)   # bit-exact distances, ascending
    assert np.array_equal(idx, ridx)     # random data: no exact ties, so the order is fully determined


def test_knn_ties_and_star_neighborhood(L, ctx, oracle):
    base = np.random.default_rng(1).uniform(-1, 1, size=(400, 12))
    pts = np.concatenate([base, base])  # every vertex twice: ties at every rank
    q = base[:9] + 1e-3
    nn = L.HipNeighborSearch(ctx, 12, pts.shape[0])
    nn.added_vertices(pts)
    idx, dist, cnt = nn.k_nearest(q, 10)
    ridx, rdist, rcnt = oracle.knn(q, pts, 10)
    assert np.array_equal(cnt, rcnt) and np.array_equal(dist, rdist)  # same distance multiset
    assert np.array_equal(np.sort(idx % 400, axis=1), np.sort(ridx % 400, axis=1))  # same vertices up to the duplicate


def test_sqrt_and_divide_are_correctly_rounded(L, ctx, oracle):
    """The NN distance is sqrt(sum of squares): device sqrt must equal IEEE sqrt bit for bit."""
    rng = np.random.default_rng(11)
    pts = np.zeros((1, 2))
    q = np.stack([np.exp(rng.uniform(-40, 40, size=20000)), np.zeros(20000)], axis=1)
    nn = L.HipNeighborSearch(ctx, 2, 1)
    nn.added_vertices(pts)
    idx, dist = nn.nearest(q)
    assert np.array_equal(dist, np.sqrt(q[:, 0] * q[:, 0]))


# ------------------------------------------------------------------ dynamics
def test_state_derivative_pendulum(L, ctx, oracle):
    scn = scenarios.make_pendulum()
    sc, osc = L.Scene(ctx, scn), oracle.OracleScene(scn)
    rng = np.random.default_rng(5)
    x = rng.uniform(-3, 3, size=(64, 2))
    u = rng.uniform(-2, 2, size=(64, 1))
    pd, M, f = sc.state_derivative(x, u)
    rc, rpd, rM, rf = osc.state_derivative(x, u)
    assert rc == 0
    assert np.allclose(M, rM, rtol=1e-13) and np.allclose(f, rf, rtol=1e-12, atol=1e-13)
    assert np.allclose(pd, rpd, rtol=1e-12, atol=1e-12)


def test_state_derivative_c2(L, ctx, oracle, c2):
    sc, osc = L.Scene(ctx, c2), oracle.OracleScene(c2)
    rng = np.random.default_rng(6)
    lo = np.array([c2.dyn.lower[i] for i in range(12)])
    hi = np.array([c2.dyn.upper[i] for i in range(12)])
    x = rng.uniform(lo, hi, size=(1024, 12))
    u = rng.uniform(-50, 50, size=(1024, 6))
    pd, M, f = sc.state_derivative(x, u)
    rc, rpd, rM, rf = osc.state_derivative(x, u)
    assert rc == 0
    scale = np.abs(rM).max()
    assert np.max(np.abs(M - rM)) <= 1e-13 * scale
    assert np.allclose(f, rf, rtol=1e-11, atol=1e-11)
    assert np.allclose(pd, rpd, rtol=1e-10, atol=1e-10)
    assert np.array_equal(pd[:, 0::2], x[:, 1::2])  # q_dot rows are copies


def test_planar_chain_dynamics(L, ctx, oracle):
    """The 2D halves of the KTE rows: revolute_joint_2D / rigid_link_2D / inertia_2D doForce, the 2D rows of the mass
    matrix, the steer loop and the planners over the planar arm's dynamic state space (one lane per edge,
    propagate_planar.hip) against the oracle's 2D classes."""
    scn = scenarios.make_c1_planar(world_seed=1, dynamics=True)
    sc, osc = L.Scene(ctx, scn), oracle.OracleScene(scn)
    rng = np.random.default_rng(21)
    lo = np.array([scn.dyn.lower[i] for i in range(6)])
    hi = np.array([scn.dyn.upper[i] for i in range(6)])
    x = rng.uniform(lo, hi, size=(1000, 6))
    u = rng.uniform(-50, 50, size=(1000, 3))
    pd, M, f = sc.state_derivative(x, u)
    rc, rpd, rM, rf = osc.state_derivative(x, u)
    assert rc == 0
    assert np.max(np.abs(M - rM)) <= 1e-13 * np.abs(rM).max()
    assert np.allclose(f, rf, rtol=1e-11, atol=1e-11) and np.allclose(pd, rpd, rtol=1e-10, atol=1e-10)
    assert np.array_equal(pd[:, 0::2], x[:, 1::2])
    # steer: free starts, random targets; full edges and edges cut by obstacles / bounds
    a = rng.uniform(lo, hi, size=(600, 6)) * 0.6
    a[:, 0::2] = rng.uniform(-3.0, 3.0, size=(600, 3))
    a = a[osc.min_distance(a) > 0.01][:256]
    b = rng.uniform(lo, hi, size=(len(a), 6))
    out, steps, rec = sc.steer_position_toward(a, b, record=True)
    rc, rout, rsteps, rrec = osc.steer(a, b, record=True)
    assert rc == 0 and np.array_equal(steps, rsteps)
    assert np.allclose(out, rout, rtol=STATE_RTOL, atol=1e-12) and np.allclose(rec, rrec, rtol=STATE_RTOL, atol=1e-12)
    assert steps.min() < 20 <= steps.max()
    for fr in (0.0, 0.5):
        o2, s2, _ = sc.steer_position_toward(a[:16], b[:16], fraction=fr)
        rc, ro2, rs2, _ = osc.steer(a[:16], b[:16], fraction=fr)
        assert np.array_equal(s2, rs2) and np.allclose(o2, ro2, rtol=STATE_RTOL, atol=1e-13)
    # RRT over the dynamic space (batch planner), then RRT* over it (graph batch)
    prm = scn.rrt_params(seed=2, max_vertices=800)
    rc, rout, rtree = osc.rrt_dyn(prm)
    pl = L.RrtPlanner(sc, prm)
    st = pl.solve_planning_query()
    tree = pl.tree()
    assert rc == 0 and st.num_vertices == rout.num_vertices == 801 and st.iterations == rout.iterations
    assert np.array_equal(tree["nn_seq"], rtree["nn_seq"]) and np.array_equal(tree["accept"], rtree["accept"])
    assert np.array_equal(tree["parent"], rtree["parent"])
    assert np.allclose(tree["pos"], rtree["pos"], rtol=STATE_RTOL, atol=1e-12)
    pl.close()
    prm = scn.rrt_params(seed=3, max_vertices=300)
    prm.conn_tol = 3.0
    rc, sout, rg = osc.rrtstar_dyn(prm)
    ps = L.RrtStarPlanner(sc, prm, scn.dyn)
    ss = ps.solve_planning_query()
    g = ps.graph()
    assert (ss.num_vertices, ss.samples, ss.rewires, ss.edges_checked) == (sout.num_vertices, sout.samples, sout.rewires,
                                                                            sout.edges_checked)
    assert np.array_equal(g["pred"], rg["pred"]) and np.array_equal(g["near_seq"], rg["near_seq"])
    assert np.allclose(g["pos"], rg["pos"], rtol=STATE_RTOL, atol=1e-12)
    ps.close()
    # a planar chain given at position level still refuses the dynamics entry points
    sq = L.Scene(ctx, scenarios.make_c1_planar(world_seed=1))
    with pytest.raises(L.RkhError):
        sq.state_derivative(x[:1], u[:1])


def test_singular_mass_matrix_is_reported(L, ctx):
    scn = scenarios.make_pendulum(length=0.0, mass=0.0)  # M = 0 -> pivot < 1e-8 -> singularity_error
    sc = L.Scene(ctx, scn)
    with pytest.raises(L.SingularityError):
        sc.state_derivative(np.zeros((1, 2)), np.zeros((1, 1)))


# ------------------------------------------------------------------ proximity
def test_min_distance_c2(L, ctx, oracle, c2):
    sc, osc = L.Scene(ctx, c2), oracle.OracleScene(c2)
    assert sc.num_pairs == osc.lib.orc_scene_num_finders(osc.h) == 300
    rng = np.random.default_rng(8)
    x = np.zeros((2048, 12))
    x[:, 0::2] = rng.uniform(-np.pi, np.pi, size=(2048, 6))
    d = sc.min_distance(x)
    rd = osc.min_distance(x)
    assert np.allclose(d, rd, rtol=0, atol=1e-12)
    close = np.abs(rd) < 1e-12
    assert np.array_equal((d < 0)[~close], (rd < 0)[~close])  # identical collision verdicts
    assert 0.02 < np.mean(rd < 0) < 0.98  # the world really has both verdicts


# ------------------------------------------------------------------ propagate (steer)
def test_propagate_c2_matches_oracle(L, ctx, oracle, c2):
    sc, osc = L.Scene(ctx, c2), oracle.OracleScene(c2)
    rng = np.random.default_rng(9)
    lo = np.array([c2.dyn.lower[i] for i in range(12)])
    hi = np.array([c2.dyn.upper[i] for i in range(12)])
    B = 256
    a = rng.uniform(lo, hi, size=(B, 12)) * 0.6
    a[:, 0::2] = rng.uniform(-2.5, 2.5, size=(B, 6))
    free0 = osc.min_distance(a) > 0.01
    a, B = a[free0], int(free0.sum())
    b = rng.uniform(lo, hi, size=(B, 12))
    out, steps, rec = sc.steer_position_toward(a, b, record=True)
    rc, rout, rsteps, rrec = osc.steer(a, b, record=True)
    assert rc == 0
    assert np.array_equal(steps, rsteps)  # identical step-collision verdict sequence
    assert np.allclose(out, rout, rtol=STATE_RTOL, atol=1e-12)
    assert np.allclose(rec, rrec, rtol=STATE_RTOL, atol=1e-12)
    assert steps.min() < 20 <= steps.max()  # both truncated and full edges occur


def test_propagate_fraction_and_goal_tolerance(L, ctx, oracle, c2):
    sc, osc = L.Scene(ctx, c2), oracle.OracleScene(c2)
    a = np.tile(c2.start, (4, 1))
    b = np.tile(c2.goal, (4, 1))
    for fr in (0.0, 0.25, 0.5):
        out, steps, _ = sc.steer_position_toward(a, b, fraction=fr)
        rc, rout, rsteps, _ = osc.steer(a, b, fraction=fr)
        assert np.array_equal(steps, rsteps) and np.allclose(out, rout, rtol=STATE_RTOL, atol=1e-13)
    out, steps, _ = sc.steer_position_toward(a, a)  # already at the target: no step
    assert np.all(steps == 0) and np.array_equal(out, a)


# ------------------------------------------------------------------ quasi-static edge walk (a9)
def test_edge_check_matches_interp_topo_walk(L, ctx, oracle):
    c1 = scenarios.make_c1(world_seed=1)
    sc, osc = L.Scene(ctx, c1), oracle.OracleScene(c1)
    lo, hi, mi = c1.meta["lower"], c1.meta["upper"], c1.meta["min_interval"]
    rng = np.random.default_rng(12)
    a = rng.uniform(lo, hi, size=(600, 3))
    x = np.zeros((600, 6)); x[:, 0::2] = a
    a = a[osc.min_distance(x) > 0.0][:256]
    b = rng.uniform(lo, hi, size=(a.shape[0], 3))
    b[:8] = a[:8] + 0.01          # shorter than min_interval: no predicate call at all
    for fr in (0.0, 0.5, 1.0):
        out, nchk = sc.move_position_toward(lo, hi, mi, a, b, fraction=fr)
        rout, rnchk = osc.qs_move(lo, hi, mi, a, b, fraction=fr)
        assert np.array_equal(nchk, rnchk)       # same number of is_free calls (integer loop count)
        assert np.array_equal(out, rout)         # interpolation is pure IEEE arithmetic: bit-exact
    assert (nchk == 0).sum() >= 8 and nchk.max() > 50
    full = np.all(rout == b, axis=1)
    assert 0.1 < full.mean() < 0.95              # both blocked and complete edges occur


def test_edge_walks_in_a_scene_whose_bounding_spheres_nearly_all_overlap(L, ctx, oracle):
    """The lane-per-point edge walk (edge_points_kernel) queues the (point, pair) combinations that survive the
    bounding-sphere cull and runs their closed forms one per thread; when more survive than the queue holds it redoes
    the pair list in slices.  Forty thin rods of length 1.6 around the C3 arm have bounding spheres that reach across the
    whole workspace while the rods themselves are rarely touched: ~64 points x several hundred surviving pairs per pass,
    far beyond the queue's 1024 entries.  Number of predicate calls and returned points against the oracle's walk,
    for 64 and (a launch of thousands of edges) 32 points per pass."""
    import copy

    c3 = scenarios.make_c3(1)
    rng = np.random.default_rng(77)
    scn = copy.copy(c3)
    rods = []
    for _ in range(40):
        s = T.Shape(kind=T.SHAPE_CCYLINDER, anchor=-1)
        q = rng.normal(size=4)
        s.pose = T.make_pose(rng.uniform(-0.9, 0.9, size=3), q / np.linalg.norm(q))
        s.dims[:] = (1.6, 0.01, 0.0)
        rods.append(s)
    scn.shapes = list(c3.shapes) + rods
    sc, osc = L.Scene(ctx, scn), oracle.OracleScene(scn)
    lo, hi, mi = c3.meta["lower"], c3.meta["upper"], c3.meta["min_interval"]
    n = sc.n
    a = rng.uniform(-1.2, 1.2, size=(4000, n))
    x = np.zeros((len(a), 2 * n)); x[:, 0::2] = a
    a = a[osc.min_distance(x) > 0.0]
    assert len(a) >= 300
    b = rng.uniform(lo, hi, size=(len(a), n))
    for count in (96, len(a)):       # a launch of few edges: 64 points per pass; of >= 2048: 32 (pad by repetition)
        aa, bb = a[:count], b[:count]
        if count == len(a):
            reps = (2100 + count - 1) // count
            aa, bb = np.tile(aa, (reps, 1)), np.tile(bb, (reps, 1))
        out, nchk = sc.move_position_toward(lo, hi, mi, aa, bb)
        rout, rnchk = osc.qs_move(lo, hi, mi, aa[:count], bb[:count])
        assert np.array_equal(nchk[:count], rnchk) and np.array_equal(out[:count], rout)
        assert np.array_equal(nchk, np.tile(rnchk, len(aa) // count)[: len(aa)])
    assert nchk.max() > 64 and (np.all(rout == b[: len(rout)], axis=1)).mean() > 0.05


# ------------------------------------------------------------------ planner
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_rrt_tree_identical_to_sequential_planner(L, ctx, oracle, c2, seed):
    sc, osc = L.Scene(ctx, c2), oracle.OracleScene(c2)
    prm = c2.rrt_params(seed=seed, max_vertices=1500)
    rc, rout, rtree = osc.rrt_dyn(prm)
    assert rc == 0
    pl = L.RrtPlanner(sc, prm)
    st = pl.solve_planning_query()
    tree = pl.tree()
    assert st.num_vertices == rout.num_vertices == 1501
    assert st.iterations == rout.iterations
    assert st.edges_checked == rout.edges_checked
    assert np.array_equal(tree["nn_seq"], rtree["nn_seq"])    # same nearest neighbour for every sample
    assert np.array_equal(tree["accept"], rtree["accept"])    # same accept / reject bitstring
    assert np.array_equal(tree["parent"], rtree["parent"])    # same topology
    assert np.allclose(tree["pos"], rtree["pos"], rtol=STATE_RTOL, atol=1e-12)
    assert np.array_equal(np.isinf(tree["goal_dist"]), np.isinf(rtree["goal_dist"]))


def test_rrt_stops_on_max_results(L, ctx, oracle, c2):
    """Goal next to the start: the first goal probe that connects ends the run (max_num_results = 1)."""
    import copy

    scn = copy.copy(c2)
    scn.goal = c2.start.copy()
    scn.goal[1] = 0.5  # reachable within one edge from states near the start
    sc, osc = L.Scene(ctx, scn), oracle.OracleScene(scn)
    prm = scn.rrt_params(seed=4, max_vertices=400, max_results=1)
    rc, rout, rtree = osc.rrt_dyn(prm)
    pl = L.RrtPlanner(sc, prm)
    st = pl.solve_planning_query()
    tree = pl.tree()
    assert (st.num_vertices, st.iterations, st.num_solutions) == (rout.num_vertices, rout.iterations, rout.num_solutions)
    assert np.array_equal(tree["parent"], rtree["parent"])
    if rout.num_solutions:
        assert st.best_cost == pytest.approx(rout.best_cost, rel=1e-10)


def test_c1_quasi_static_rrt_identical_to_sequential_planner(L, ctx, oracle):
    """BASELINE config C1: 3-DOF planar arm, quasi-static RRT, 10 box obstacles, 5k nodes, single seed."""
    c1 = scenarios.make_c1(world_seed=1)
    sc, osc = L.Scene(ctx, c1), oracle.OracleScene(c1)
    lo, hi, mi = c1.meta["lower"], c1.meta["upper"], c1.meta["min_interval"]
    prm = c1.rrt_params(seed=1, max_vertices=5000)
    rc, rout, rtree = osc.rrt_qs(lo, hi, mi, prm)
    pl = L.RrtPlanner(sc, prm, qs=L.make_qs_space(3, lo, hi, mi))
    st = pl.solve_planning_query()
    tree = pl.tree()
    assert (st.num_vertices, st.iterations, st.edges_checked) == (rout.num_vertices, rout.iterations, rout.edges_checked)
    assert st.num_vertices == 5001
    assert np.array_equal(tree["nn_seq"], rtree["nn_seq"])
    assert np.array_equal(tree["accept"], rtree["accept"])
    assert np.array_equal(tree["parent"], rtree["parent"])
    assert np.array_equal(tree["pos"], rtree["pos"])              # straight-line edges: pure IEEE arithmetic, bit-exact
    assert np.array_equal(tree["goal_dist"], rtree["goal_dist"])  # goal probes, incl. which vertices see the goal
    assert st.num_solutions == rout.num_solutions and rout.num_solutions > 0
    assert st.best_cost == rout.best_cost


def test_c1_stops_at_first_solution(L, ctx, oracle):
    c1 = scenarios.make_c1(world_seed=1)
    sc, osc = L.Scene(ctx, c1), oracle.OracleScene(c1)
    lo, hi, mi = c1.meta["lower"], c1.meta["upper"], c1.meta["min_interval"]
    prm = c1.rrt_params(seed=3, max_vertices=5000, max_results=1)
    rc, rout, rtree = osc.rrt_qs(lo, hi, mi, prm)
    pl = L.RrtPlanner(sc, prm, qs=L.make_qs_space(3, lo, hi, mi))
    st = pl.solve_planning_query()
    assert rout.num_solutions == 1 and rout.num_vertices < 5001
    assert (st.num_vertices, st.iterations, st.num_solutions) == (rout.num_vertices, rout.iterations, 1)
    assert st.best_cost == rout.best_cost
    assert np.array_equal(pl.tree()["parent"], rtree["parent"])


def test_sample_stream_grows_past_its_first_capacity(L, ctx, oracle):
    """generate_rrt has no iteration cap (rr_tree.hpp:192-196): a cluttered scene with a demanding steer tolerance accepts
    ~9 % of its samples, so 2000 vertices take far more iterations than the first allocation of the device-resident
    sample stream (16384); the planner must grow it and stay the sequential planner (the pool / solve loops used to
    spin on empty rounds or return RKH_ERR_CAPACITY there)."""
    c1 = scenarios.make_c1_planar(world_seed=1, n_obstacles=25)
    sc, osc = L.Scene(ctx, c1), oracle.OracleScene(c1)
    lo, hi, mi = c1.meta["lower"], c1.meta["upper"], c1.meta["min_interval"]
    prm = c1.rrt_params(seed=4, max_vertices=2000, steer_tol=0.9)
    rc, rout, rtree = osc.rrt_qs(lo, hi, mi, prm)
    assert rout.iterations > 17000
    for runner in ("solve", "pool"):
        if runner == "solve":
            pl = L.RrtPlanner(sc, prm, qs=L.make_qs_space(3, lo, hi, mi))
        else:
            pl = L.RrtPlannerPool(sc, [prm], groups=1, qs=L.make_qs_space(3, lo, hi, mi))
        st = pl.solve_planning_query()
        tree = pl.tree()
        assert (st.num_vertices, st.iterations, st.edges_checked) == (rout.num_vertices, rout.iterations, rout.edges_checked)
        assert np.array_equal(tree["nn_seq"], rtree["nn_seq"]) and np.array_equal(tree["accept"], rtree["accept"])
        assert np.array_equal(tree["parent"], rtree["parent"]) and np.array_equal(tree["pos"], rtree["pos"])
        pl.close()


# ------------------------------------------------------------------ RRT* (a24)
@pytest.mark.parametrize("seed", [1, 2])
def test_rrtstar_graph_identical_to_sequential_planner(L, ctx, oracle, seed):
    """RRT* with k-NN rewiring over the quasi-static space: same vertices, predecessors, accumulated costs, rewires."""
    c1 = scenarios.make_c1(world_seed=1)
    sc, osc = L.Scene(ctx, c1), oracle.OracleScene(c1)
    lo, hi, mi = c1.meta["lower"], c1.meta["upper"], c1.meta["min_interval"]
    prm = c1.rrt_params(seed=seed, max_vertices=1200)
    rc, rout, rg = osc.rrtstar_qs(lo, hi, mi, prm)
    pl = L.RrtStarPlanner(sc, prm, L.make_qs_space(3, lo, hi, mi))
    st = pl.solve_planning_query()
    g = pl.graph()
    assert (st.num_vertices, st.samples, st.loop_iterations, st.num_solutions, st.rewires, st.edges_checked) == (
        rout.num_vertices, rout.samples, rout.loop_iterations, rout.num_solutions, rout.rewires, rout.edges_checked)
    assert st.rewires > 100 and st.num_solutions > 0
    assert np.array_equal(g["near_seq"], rg["near_seq"])
    assert np.array_equal(g["pred"], rg["pred"])
    assert np.array_equal(g["pos"], rg["pos"])    # straight-line edges: bit-exact
    assert np.array_equal(g["dist"], rg["dist"])  # accumulated costs after rewiring and cost propagation
    assert st.best_cost == rout.best_cost


@pytest.mark.parametrize("seed,conn_tol,max_vertices", [(3, 3.0, 1000), (4, 0.05, 250)])
def test_rrtstar_over_the_dynamic_space_identical_to_sequential_planner(L, ctx, oracle, c2, seed, conn_tol, max_vertices):
    """RRT* whose vertices are states (q, qd) of the C2 chain and whose candidate edges -- expand_to_nearest from every
    neighbour, can_be_connected in each direction -- are RK4 propagations with collision checks (the directed branch of
    the connector; the space's metric is symmetric, so predecessor and successor neighbourhoods coincide).  A loose
    connection tolerance makes rewiring happen; the reference's default (5 %) practically never connects two states."""
    sc, osc = L.Scene(ctx, c2), oracle.OracleScene(c2)
    prm = c2.rrt_params(seed=seed, max_vertices=max_vertices)
    prm.conn_tol = conn_tol
    rc, rout, rg = osc.rrtstar_dyn(prm)
    assert rc == 0
    pl = L.RrtStarPlanner(sc, prm, c2.dyn)
    st = pl.solve_planning_query()
    g = pl.graph()
    assert (st.num_vertices, st.samples, st.loop_iterations, st.num_solutions, st.rewires, st.edges_checked) == (
        rout.num_vertices, rout.samples, rout.loop_iterations, rout.num_solutions, rout.rewires, rout.edges_checked)
    assert st.num_vertices == max_vertices + 2
    if conn_tol > 1.0:
        assert st.rewires > 20
    assert np.array_equal(g["near_seq"], rg["near_seq"])
    assert np.array_equal(g["pred"], rg["pred"])
    # end states of the accepted propagations: fp64 to the tolerance of the dynamics tests (the device's sincos differs
    # from libm's in the last place); the graph itself -- neighbours, predecessors, rewires, counts -- is identical
    assert np.allclose(g["pos"], rg["pos"], rtol=STATE_RTOL, atol=1e-12)
    fin = np.isfinite(rg["dist"])
    assert np.array_equal(np.isfinite(g["dist"]), fin)
    assert np.allclose(g["dist"][fin], rg["dist"][fin], rtol=10 * STATE_RTOL, atol=1e-12)
    assert np.isclose(st.best_cost, rout.best_cost, rtol=10 * STATE_RTOL) or (np.isinf(st.best_cost) and np.isinf(rout.best_cost))
    pl.close()


@pytest.mark.parametrize("seed", [1, 2])
def test_bidirectional_rrtstar_graph_identical_to_sequential_planner(L, ctx, oracle, seed):
    """RRT* with BIDIRECTIONAL_PLANNING: a forward tree (predecessor / distance_accum) and a backward tree (successor /
    fwd_distance_accum) grown and rewired together; the generator pulls one point from each side per sample."""
    c1 = scenarios.make_c1(world_seed=1)
    sc, osc = L.Scene(ctx, c1), oracle.OracleScene(c1)
    lo, hi, mi = c1.meta["lower"], c1.meta["upper"], c1.meta["min_interval"]
    prm = c1.rrt_params(seed=seed, max_vertices=1200)
    rc, rout, rg = osc.birrtstar_qs(lo, hi, mi, prm)
    pl = L.BiRrtStarPlanner(sc, prm, L.make_qs_space(3, lo, hi, mi))
    st = pl.solve_planning_query()
    g = pl.graph()
    assert (st.num_vertices, st.samples, st.loop_iterations, st.rewires, st.fwd_rewires, st.joins, st.edges_checked) == (
        rout.num_vertices, rout.samples, rout.loop_iterations, rout.rewires, rout.fwd_rewires, rout.joins, rout.edges_checked)
    assert st.num_vertices >= 1202 and st.joins > 100
    assert (g["near_succ"] != 0xFFFFFFFF).sum() > 50 and (g["near_pred"] != 0xFFFFFFFF).sum() > 50   # both pulls happen
    # a joining vertex pulls the same point into both trees: coincident vertices, i.e. exactly equal k-NN distances whose
    # order (and membership at the k-th place) follows the reference's heap, not (distance, index)
    assert len(np.unique(g["pos"], axis=0)) < len(g["pos"])
    for key in ("near_pred", "near_succ", "pred", "succ", "pos", "dist", "fwd_dist"):
        assert np.array_equal(g[key], rg[key]), key
    assert st.best_join_cost == rout.best_join_cost
    # a shorter run resumed gives the same graph as the run in one go
    pl2 = L.BiRrtStarPlanner(sc, prm, L.make_qs_space(3, lo, hi, mi))
    pl2.solve_planning_query(max_loop_iterations=300)
    st2 = pl2.solve_planning_query()
    assert st2.num_vertices == st.num_vertices and np.array_equal(pl2.graph()["fwd_dist"], g["fwd_dist"])
    pl.close(); pl2.close()


@pytest.mark.parametrize("scene,seed,iterations", [("c4", 2, 1500), ("c4", 1, 1500), ("c1", 1, 4000)])
def test_rrtstar_with_branch_and_bound_pruning_identical_to_sequential_planner(L, ctx, oracle, scene, seed, iterations):
    """USE_BRANCH_AND_BOUND_PRUNING_FLAG: points that cannot improve on the best solution are dropped, vertices whose
    cost + distance to the goal exceeds it are removed from the graph (tombstones in the device NN store).  With
    uniform sampling almost every point is dropped once the goal is connected (reference behaviour), so the runs are
    ended by their iteration budget."""
    scn = scenarios.make_c4(world_seed=1) if scene == "c4" else scenarios.make_c1(world_seed=1)
    sc, osc = L.Scene(ctx, scn), oracle.OracleScene(scn)
    lo, hi, mi = scn.meta["lower"], scn.meta["upper"], scn.meta["min_interval"]
    prm = scn.rrt_params(seed=seed, max_vertices=1500)
    rc, rout, rg, rpruned, rskipped = osc.bnb_rrtstar_qs(lo, hi, mi, prm, max_loop_iterations=iterations)
    pl = L.RrtStarPlanner(sc, prm, L.make_qs_space(scn.n_dof, lo, hi, mi))
    pl.set_branch_and_bound(True)
    st = pl.solve_planning_query(max_loop_iterations=iterations)
    g = pl.graph()
    assert (st.num_vertices, st.samples, st.loop_iterations, st.num_solutions, st.rewires, st.edges_checked, st.pruned,
            st.skipped) == (rout.num_vertices, rout.samples, rout.loop_iterations, rout.num_solutions, rout.rewires,
                            rout.edges_checked, rpruned, rskipped)
    assert st.pruned >= 4 and st.skipped > 1000 and st.num_solutions >= 2
    for key in ("near_seq", "pred", "pos", "dist"):
        assert np.array_equal(g[key], rg[key]), key
    assert np.array_equal(pl.removed(), rg["removed"]) and rg["removed"].sum() == st.pruned
    assert st.best_cost == rout.best_cost
    with pytest.raises(L.RkhError):
        pl.set_branch_and_bound(False)   # only before the first solve
    pl.close()


def test_rate_limited_joint_space(L, ctx, oracle):
    """The reference's manipulator environments plan in the rate-limited joint space (Ndof_rl_space): a point holds
    reach times q_i / speed_limit_i, the hyperbox, the metric and the interpolation live in those coordinates and the
    model is posed at point_i * speed_limit_i.  RRT* and PRM graphs against the oracle, and the space's is_free against
    the ordinary distance query at the mapped configuration."""
    c1 = scenarios.make_c1(world_seed=1)
    sc, osc = L.Scene(ctx, c1), oracle.OracleScene(c1)
    speed = np.array([2.0, 0.5, 1.25])
    lo, hi, mi = np.asarray(c1.meta["lower"]) / speed, np.asarray(c1.meta["upper"]) / speed, c1.meta["min_interval"]
    qs = L.make_qs_space(3, lo, hi, mi, speed_limits=speed)
    prm = c1.rrt_params(seed=3, max_vertices=600)
    for d in range(3):
        prm.start[d] /= speed[d]
        prm.goal[d] /= speed[d]
    oracle.set_qs_speed_limits(speed)
    try:
        rc, rout, rg = osc.rrtstar_qs(lo, hi, mi, prm)
        pprm = c1.prm_params(seed=4, max_vertices=300, sampling_radius=0.8)
        for d in range(3):
            pprm.base.start[d] /= speed[d]
            pprm.base.goal[d] /= speed[d]
        rc2, pout, pg = osc.prm_qs(lo, hi, mi, pprm)
    finally:
        oracle.set_qs_speed_limits(None)
    pl = L.RrtStarPlanner(sc, prm, qs)
    st = pl.solve_planning_query()
    g = pl.graph()
    assert (st.num_vertices, st.samples, st.rewires, st.edges_checked) == (rout.num_vertices, rout.samples, rout.rewires,
                                                                            rout.edges_checked)
    for key in ("near_seq", "pred", "pos", "dist"):
        assert np.array_equal(g[key], rg[key]), key
    pp = L.PrmPlanner(sc, pprm, qs)
    _prm_same(pp.solve_planning_query(), pp.graph(), pout, pg)
    # the vertices are free where the model is posed at point * speed (all but the odd end point: a completed walk
    # returns its target without testing it, interpolated_topologies.hpp:146-160), and the plain joint-space planner on
    # the same seed builds a different tree (the metric is another one)
    x = np.zeros((len(g["pos"]), 6)); x[:, 0::2] = g["pos"] * speed
    d = sc.min_distance(x)
    assert (d >= 0.0).mean() > 0.98 and d.min() > -0.05
    x[:, 0::2] = g["pos"]   # posed at the raw reach times instead, many of the same points collide
    assert (sc.min_distance(x) < 0.0).sum() > 2 * (d < 0.0).sum()
    plain = L.RrtStarPlanner(sc, c1.rrt_params(seed=3, max_vertices=600), L.make_qs_space(3, c1.meta["lower"], c1.meta["upper"], mi))
    plain.solve_planning_query()
    assert not np.allclose(plain.graph()["pos"][2:50] , g["pos"][2:50] * speed)
    pl.close(); pp.close(); plain.close()


def test_rrtstar_batch_of_seeds(L, ctx, oracle):
    c1 = scenarios.make_c1(world_seed=1)
    sc, osc = L.Scene(ctx, c1), oracle.OracleScene(c1)
    lo, hi, mi = c1.meta["lower"], c1.meta["upper"], c1.meta["min_interval"]
    prms = [c1.rrt_params(seed=s, max_vertices=300) for s in (5, 6, 7, 8)]
    pl = L.RrtStarPlanner(sc, prms, L.make_qs_space(3, lo, hi, mi))
    pl.solve_planning_query()
    for i, prm in enumerate(prms):
        rc, rout, rg = osc.rrtstar_qs(lo, hi, mi, prm)
        assert pl.all_stats[i].num_vertices == rout.num_vertices and pl.all_stats[i].rewires == rout.rewires
        assert np.array_equal(pl.graph(i)["pred"], rg["pred"]) and np.array_equal(pl.graph(i)["dist"], rg["dist"])


# ------------------------------------------------------------------ PRM (a25)
def _prm_same(st, g, rout, rg):
    assert (st.num_vertices, st.num_edges, st.samples, st.rejected, st.loop_iterations, st.num_components,
            st.publish_calls, st.merged_at_vertex, st.edges_checked) == (
        rout.num_vertices, rout.num_edges, rout.samples, rout.rejected, rout.loop_iterations, rout.num_components,
        rout.publish_calls, rout.merged_at_vertex, rout.edges_checked)
    assert np.array_equal(g["kind"], rg["kind"]) and np.array_equal(g["expanded"], rg["expanded"])
    assert np.array_equal(g["pos"], rg["pos"])
    assert np.array_equal(g["edge_u"], rg["edge_u"]) and np.array_equal(g["edge_v"], rg["edge_v"])
    assert np.array_equal(g["edge_w"], rg["edge_w"])
    assert np.array_equal(g["density"], rg["density"]) and np.array_equal(g["cc_root"], rg["cc_root"])


@pytest.mark.parametrize("seed,conn_tol,sampling_radius,max_vertices", [(3, 3.0, 1.0, 1000), (5, 0.05, 0.5, 200)])
def test_prm_over_the_dynamic_space_identical_to_sequential_planner(L, ctx, oracle, c2, seed, conn_tol, sampling_radius,
                                                                    max_vertices):
    """PRM whose vertices are states (q, qd) of the C2 chain: is_free(state) rejection sampling, random walks that are
    RK4 propagations over a fraction of the edge time, connections by full propagations.  Same loop decisions, vertices,
    edges, components and expansion queue as the sequential planner; states to the tolerance of the dynamics tests."""
    sc, osc = L.Scene(ctx, c2), oracle.OracleScene(c2)
    prm = c2.prm_params(seed=seed, max_vertices=max_vertices, sampling_radius=sampling_radius)
    prm.base.conn_tol = conn_tol
    rc, rout, rg = osc.prm_dyn(prm)
    assert rc == 0
    pl = L.PrmPlanner(sc, prm, c2.dyn)
    st = pl.solve_planning_query()
    g = pl.graph()
    assert (st.num_vertices, st.num_edges, st.samples, st.rejected, st.loop_iterations, st.num_components,
            st.publish_calls, st.merged_at_vertex, st.edges_checked) == (
        rout.num_vertices, rout.num_edges, rout.samples, rout.rejected, rout.loop_iterations, rout.num_components,
        rout.publish_calls, rout.merged_at_vertex, rout.edges_checked)
    assert st.num_vertices == max_vertices + 2 and np.bincount(g["kind"], minlength=3)[1] > 10   # walks did expand
    if conn_tol > 1.0:
        assert st.num_edges > 2 * max_vertices
    assert np.array_equal(g["kind"], rg["kind"]) and np.array_equal(g["expanded"], rg["expanded"])
    assert np.array_equal(g["edge_u"], rg["edge_u"]) and np.array_equal(g["edge_v"], rg["edge_v"])
    assert np.array_equal(g["cc_root"], rg["cc_root"])
    assert np.allclose(g["pos"], rg["pos"], rtol=STATE_RTOL, atol=1e-12)
    assert np.allclose(g["edge_w"], rg["edge_w"], rtol=10 * STATE_RTOL, atol=1e-12)
    assert np.allclose(g["density"], rg["density"], rtol=1e-8, atol=1e-12)
    pl.close()


@pytest.mark.parametrize("seed", [1, 2])
def test_prm_roadmap_identical_to_sequential_planner(L, ctx, oracle, seed):
    """PRM (construct / expand with the density heap, random walks, connected components): the roadmap, the order of
    its edges, the densities and the random-stream consumption equal the sequential planner's."""
    c1 = scenarios.make_c1(world_seed=1)
    sc, osc = L.Scene(ctx, c1), oracle.OracleScene(c1)
    lo, hi, mi = c1.meta["lower"], c1.meta["upper"], c1.meta["min_interval"]
    prm = c1.prm_params(seed=seed, max_vertices=800, sampling_radius=1.0)
    rc, rout, rg = osc.prm_qs(lo, hi, mi, prm)
    pl = L.PrmPlanner(sc, prm, L.make_qs_space(3, lo, hi, mi))
    st = pl.solve_planning_query()
    assert st.rejected > 50 and (rg["kind"] == 1).sum() > 50  # both branches and the rejection loop were exercised
    _prm_same(st, pl.graph(), rout, rg)


def test_prm_batch_and_resume(L, ctx, oracle):
    c1 = scenarios.make_c1(world_seed=1)
    sc, osc = L.Scene(ctx, c1), oracle.OracleScene(c1)
    lo, hi, mi = c1.meta["lower"], c1.meta["upper"], c1.meta["min_interval"]
    prms = [c1.prm_params(seed=s, max_vertices=250, sampling_radius=0.5 + 0.25 * i) for i, s in enumerate((5, 6, 7))]
    pl = L.PrmPlanner(sc, prms, L.make_qs_space(3, lo, hi, mi))
    pl.solve_planning_query(max_loop_iterations=100)   # stop half way, then resume
    pl.solve_planning_query()
    for i, prm in enumerate(prms):
        rc, rout, rg = osc.prm_qs(lo, hi, mi, prm)
        _prm_same(pl.all_stats[i], pl.graph(i), rout, rg)


# ------------------------------------------------------------------ kernel mappings of the steer kernel
def test_propagate_mappings_are_bit_identical(L, ctx, oracle, c2, monkeypatch):
    """Two waves per edge (state_derivative_duo), one wave per edge, 16 lanes per edge and the two-lanes mappings follow
    the same operation order: identical bits."""
    sc, osc = L.Scene(ctx, c2), oracle.OracleScene(c2)
    rng = np.random.default_rng(21)
    lo = np.array([c2.dyn.lower[i] for i in range(12)])
    hi = np.array([c2.dyn.upper[i] for i in range(12)])
    B = 300
    a = rng.uniform(lo, hi, size=(B, 12)) * 0.6
    a[:, 0::2] = rng.uniform(-2.5, 2.5, size=(B, 6))
    a = a[osc.min_distance(a) > 0.01][:200]   # not a multiple of 64: the last wave is ragged
    b = rng.uniform(lo, hi, size=(a.shape[0], 12))
    res = {}
    for lanes in ("64", "128", "16", "1", "2"):
        monkeypatch.setenv("RKH_LANES_PER_EDGE", lanes)
        res[lanes] = sc.steer_position_toward(a, b, record=True)
    for lanes in ("128", "16", "1", "2"):
        assert np.array_equal(res[lanes][1], res["64"][1])
        assert np.array_equal(res[lanes][0], res["64"][0])
        assert np.array_equal(res[lanes][2], res["64"][2], equal_nan=True)
    rc, rout, rsteps, _ = osc.steer(a, b)
    assert np.array_equal(res["1"][1], rsteps) and np.allclose(res["1"][0], rout, rtol=STATE_RTOL, atol=1e-12)
    assert res["1"][1].min() < 20 <= res["1"][1].max()


def test_every_robot_shape_is_tested_by_every_mapping(L, ctx, oracle, c2, monkeypatch):
    """An obstacle sitting ON link k must stop the edge at once, whichever lane of the edge's pair tests that link's
    shape (the two-lanes kernels split the robot shapes between the edge's lanes; round 1 lost the second lane's
    verdict in a short-circuited cross-lane exchange).  Obstacle kinds: sphere, box, capped cylinder."""
    import copy
    osc = oracle.OracleScene(c2)
    rng = np.random.default_rng(3)
    lo = np.array([c2.dyn.lower[i] for i in range(12)])
    hi = np.array([c2.dyn.upper[i] for i in range(12)])
    x = rng.uniform(lo, hi, size=(1, 12)) * 0.5
    t = rng.uniform(lo, hi, size=(1, 12))
    robot = [s for s in c2.shapes if s.anchor >= 0]
    fr = osc.fk(x)[0]
    for k in (2, 3, 4, 5):
        p = fr[2 * k + 1][:3]  # joint k's end frame = the base of link k's capsule
        for kind, dims in ((T.SHAPE_SPHERE, [0.08, 0, 0]), (T.SHAPE_BOX, [0.15, 0.2, 0.1]), (T.SHAPE_CCYLINDER, [0.2, 0.05, 0])):
            ob = T.Shape(kind=kind, anchor=-1)
            ob.pose = T.make_pose(tuple(p), (1.0, 0.0, 0.0, 0.0))
            ob.dims[:] = dims
            for order in ([0, k], [k, 0], [0, 1, 2, 3, 4, 5]):
                s2 = copy.copy(c2)
                s2.shapes = [robot[i] for i in order] + [ob]
                sc, o2 = L.Scene(ctx, s2), oracle.OracleScene(s2)
                rc, rout, rsteps, _ = o2.steer(x, t)
                assert rsteps[0] == 0
                for lanes in ("64", "128", "16", "1", "2"):
                    monkeypatch.setenv("RKH_LANES_PER_EDGE", lanes)
                    assert sc.steer_position_toward(x, t)[1][0] == 0, (k, kind, order, lanes)


def test_c2_with_floor_plane_cylinders_and_a_spherical_tool(L, ctx, oracle, monkeypatch):
    """The plane / cylinder finders (prox_plane_{sphere,ccylinder,box,..}, prox_sphere_cylinder): a C2 world with a floor
    plane, flat-ended cylinders and a spherical tool on the end effector.  Distances to 1e-12, verdicts, free-step
    counts of every steer mapping and a planner run against the oracle."""
    scn = scenarios.make_c2(world_seed=2, floor=-0.12, n_cylinders=6, tool_sphere=0.06)
    sc, osc = L.Scene(ctx, scn), oracle.OracleScene(scn)
    rng = np.random.default_rng(17)
    lo = np.array([scn.dyn.lower[i] for i in range(12)])
    hi = np.array([scn.dyn.upper[i] for i in range(12)])
    x = rng.uniform(lo, hi, size=(2048, 12))
    rd, d = osc.min_distance(x), sc.min_distance(x)
    assert np.allclose(d, rd, rtol=0, atol=1e-12)
    sure = np.abs(rd) > 1e-12
    assert np.array_equal((d < 0)[sure], (rd < 0)[sure])
    # the floor is what stops many of them: without it the same configurations are mostly free
    plain = oracle.OracleScene(scenarios.make_c2(world_seed=2, n_cylinders=6, tool_sphere=0.06))
    assert ((rd < 0) & (plain.min_distance(x) > 0)).sum() > 200
    a = x[:1024] * 0.6
    b = rng.uniform(lo, hi, size=(a.shape[0], 12))
    rc, rout, rsteps, _ = osc.steer(a, b)
    assert (rsteps == 0).sum() > 20 and (rsteps == 20).sum() > 200
    for lanes in ("64", "128", "16", "2"):
        monkeypatch.setenv("RKH_LANES_PER_EDGE", lanes)
        out, steps, _ = sc.steer_position_toward(a, b)
        assert np.array_equal(steps, rsteps), lanes
        assert np.allclose(out, rout, rtol=STATE_RTOL, atol=1e-12), lanes
    monkeypatch.delenv("RKH_LANES_PER_EDGE")
    prm = scn.rrt_params(seed=3, max_vertices=700)
    rc, ro, rtree = osc.rrt_dyn(prm)
    pl = L.RrtPlanner(sc, prm)
    st = pl.solve_planning_query()
    tree = pl.tree()
    assert (st.num_vertices, st.iterations, st.edges_checked) == (ro.num_vertices, ro.iterations, ro.edges_checked)
    assert np.array_equal(tree["parent"], rtree["parent"]) and np.array_equal(tree["accept"], rtree["accept"])


def test_small_plane_is_refused(L, ctx):
    """A plane whose bounding sphere (plane.cpp:31) does not always overlap the robot's makes the reference's result
    depend on the finder order (proxy_query_model.cpp:384-389): refused instead of silently different."""
    scn = scenarios.make_c2(world_seed=2, floor=-0.12)
    scn.shapes[-1].dims[:] = [0.5, 0.5, 0.0]
    with pytest.raises(Exception):
        L.Scene(ctx, scn)


def test_steer_from_colliding_and_free_starts_matches_oracle(L, ctx, oracle, c2, monkeypatch):
    """Unfiltered random start states (some already inside an obstacle): free-step counts of every mapping against the
    oracle, states within tolerance."""
    sc, osc = L.Scene(ctx, c2), oracle.OracleScene(c2)
    rng = np.random.default_rng(0)
    lo = np.array([c2.dyn.lower[i] for i in range(12)])
    hi = np.array([c2.dyn.upper[i] for i in range(12)])
    B = 1536
    a = rng.uniform(lo, hi, size=(B, 12)) * 0.6
    b = rng.uniform(lo, hi, size=(B, 12))
    rc, rout, rsteps, _ = osc.steer(a, b)
    assert (rsteps == 0).sum() > 20 and (rsteps == 20).sum() > 500
    for lanes in ("64", "1", "2"):
        monkeypatch.setenv("RKH_LANES_PER_EDGE", lanes)
        out, steps, _ = sc.steer_position_toward(a, b)
        assert np.array_equal(steps, rsteps), lanes
        assert np.allclose(out, rout, rtol=STATE_RTOL, atol=1e-12), lanes


@pytest.mark.parametrize("lanes", ["1", "2"])
def test_rrt_tree_with_one_lane_per_edge(L, ctx, oracle, c2, monkeypatch, lanes):
    """Every round through one of the two-lanes-per-edge kernels (1: LDS-resident, 2: registers + DPP, two waves / SIMD)."""
    monkeypatch.setenv("RKH_LANES_PER_EDGE", lanes)
    sc, osc = L.Scene(ctx, c2), oracle.OracleScene(c2)
    prm = c2.rrt_params(seed=2, max_vertices=1500)
    rc, rout, rtree = osc.rrt_dyn(prm)
    pl = L.RrtPlanner(sc, prm)
    st = pl.solve_planning_query()
    tree = pl.tree()
    assert (st.num_vertices, st.iterations, st.edges_checked) == (rout.num_vertices, rout.iterations, rout.edges_checked)
    assert np.array_equal(tree["nn_seq"], rtree["nn_seq"]) and np.array_equal(tree["accept"], rtree["accept"])
    assert np.array_equal(tree["parent"], rtree["parent"])
    assert np.allclose(tree["pos"], rtree["pos"], rtol=STATE_RTOL, atol=1e-12)
    assert np.array_equal(np.isinf(tree["goal_dist"]), np.isinf(rtree["goal_dist"]))


# ------------------------------------------------------------------ flexible_beam_3D (a17)
def test_flexible_beam_dynamics_and_planner(L, ctx, oracle, monkeypatch):
    """C2 chain with a flexible_beam_3D (k = 1e4 N/m, k_theta = 1e2, the C4 parameters) from the end effector to a world
    anchor: x' = f(x,u), the steer kernels (all mappings) and a planner run against the oracle."""
    scn = scenarios.make_c2(world_seed=1, tether=(0.4, 1e4, 1e2))
    sc, osc = L.Scene(ctx, scn), oracle.OracleScene(scn)
    plain = oracle.OracleScene(scenarios.make_c2(world_seed=1))
    rng = np.random.default_rng(31)
    lo = np.array([scn.dyn.lower[i] for i in range(12)])
    hi = np.array([scn.dyn.upper[i] for i in range(12)])
    x = rng.uniform(lo, hi, size=(128, 12))
    u = rng.uniform(-50, 50, size=(128, 6))
    rc, rpd, rM, rf = osc.state_derivative(x, u)
    pd, M, f = sc.state_derivative(x, u)
    assert rc == 0
    assert np.max(np.abs(M - rM)) <= 1e-13 * np.abs(rM).max()      # the beam adds no inertia
    assert np.allclose(f, rf, rtol=1e-10, atol=1e-8)               # bias force with the beam (acos: OCML vs glibc)
    assert np.allclose(pd, rpd, rtol=1e-9, atol=1e-8)
    assert np.abs(rf - plain.state_derivative(x, u)[3]).max() > 10.0  # and the beam does pull
    a = x[:96] * 0.5
    a = a[osc.min_distance(a) > 0.01][:60]
    b = rng.uniform(lo, hi, size=(a.shape[0], 12))
    res = {}
    for lanes in ("64", "128", "1", "2"):
        monkeypatch.setenv("RKH_LANES_PER_EDGE", lanes)
        res[lanes] = sc.steer_position_toward(a, b)
    assert np.array_equal(res["1"][0], res["64"][0]) and np.array_equal(res["1"][1], res["64"][1])
    assert np.array_equal(res["2"][0], res["64"][0]) and np.array_equal(res["2"][1], res["64"][1])
    assert np.array_equal(res["128"][0], res["64"][0]) and np.array_equal(res["128"][1], res["64"][1])
    rc, rout, rsteps, _ = osc.steer(a, b)
    assert np.array_equal(res["64"][1], rsteps) and np.allclose(res["64"][0], rout, rtol=1e-9, atol=1e-10)
    monkeypatch.delenv("RKH_LANES_PER_EDGE")
    prm = scn.rrt_params(seed=3, max_vertices=400)
    rc, ro, rtree = osc.rrt_dyn(prm)
    pl = L.RrtPlanner(sc, prm)
    st = pl.solve_planning_query()
    tree = pl.tree()
    assert (st.num_vertices, st.iterations, st.edges_checked) == (ro.num_vertices, ro.iterations, ro.edges_checked)
    assert np.array_equal(tree["parent"], rtree["parent"]) and np.array_equal(tree["accept"], rtree["accept"])


# ------------------------------------------------------------------ ragged / degenerate sizes
def test_ragged_planner_batch(L, ctx, oracle, c2):
    """Problems of very different sizes in one batch (they finish in different rounds; the small ones idle while the
    large ones go on), each still the sequential planner on its seed."""
    sc, osc = L.Scene(ctx, c2), oracle.OracleScene(c2)
    sizes = (1, 37, 400, 1100)
    prms = [c2.rrt_params(seed=10 + i, max_vertices=mv) for i, mv in enumerate(sizes)]
    pl = L.RrtPlanner(sc, prms)
    pl.solve_planning_query()
    for i, prm in enumerate(prms):
        rc, ro, rtree = osc.rrt_dyn(prm)
        st, tree = pl.all_stats[i], pl.tree(i)
        assert (st.num_vertices, st.iterations, st.edges_checked) == (ro.num_vertices, ro.iterations, ro.edges_checked)
        assert np.array_equal(tree["parent"], rtree["parent"]) and np.array_equal(tree["nn_seq"], rtree["nn_seq"])


def test_steer_kernels_ragged_wave_sizes(L, ctx, oracle, c2, monkeypatch):
    """Batch sizes around the wave granularities of the two steer mappings (1, 27, 28, 29, 57 edges)."""
    sc, osc = L.Scene(ctx, c2), oracle.OracleScene(c2)
    rng = np.random.default_rng(77)
    lo = np.array([c2.dyn.lower[i] for i in range(12)])
    hi = np.array([c2.dyn.upper[i] for i in range(12)])
    a = rng.uniform(lo, hi, size=(300, 12)) * 0.5
    a = a[osc.min_distance(a) > 0.01]
    b = rng.uniform(lo, hi, size=(a.shape[0], 12))
    for B in (1, 27, 28, 29, 57):
        rc, rout, rsteps, _ = osc.steer(a[:B], b[:B])
        for lanes in ("64", "1", "2"):
            monkeypatch.setenv("RKH_LANES_PER_EDGE", lanes)
            out, steps, _ = sc.steer_position_toward(a[:B], b[:B])
            assert np.array_equal(steps, rsteps) and np.allclose(out, rout, rtol=STATE_RTOL, atol=1e-12)


def test_knn_degenerate_requests(L, ctx, oracle):
    """k larger than the tree, a radius that excludes everything, a single-vertex tree."""
    rng = np.random.default_rng(5)
    pts = rng.uniform(-1, 1, size=(5, 3))
    q = rng.uniform(-1, 1, size=(4, 3))
    nn = L.HipNeighborSearch(ctx, 3, 64)
    nn.added_vertices(pts[:1])
    idx, dist, cnt = nn.k_nearest(q, 8, radius=np.inf)
    assert np.all(cnt == 1) and np.all(idx[:, 0] == 0)
    nn.added_vertices(pts[1:])
    idx, dist, cnt = nn.k_nearest(q, 8, radius=np.inf)       # k > n: everything, nearest first
    ridx, rdist, rcnt = oracle.knn(q, pts, 8, radius=np.inf)
    assert np.array_equal(cnt, rcnt) and np.all(cnt == 5)
    assert np.array_equal(idx[:, :5], ridx[:, :5]) and np.array_equal(dist[:, :5], rdist[:, :5])
    idx, dist, cnt = nn.k_nearest(q, 3, radius=1e-9)          # nothing inside the radius
    assert np.all(cnt == 0)


# ------------------------------------------------------------------ C4: 12-DOF dual arm (branching chain), 200 obstacles
def test_c4_dual_arm_quasi_static(L, ctx, oracle):
    """Two 6-R arms on fixed mounts of one base (rigid links from the chain base), 200 obstacles, 12-D quasi-static space:
    distance queries, edge walks, RRT and PRM against the oracle (whose KTE interpreter handles the branching natively)."""
    c4 = scenarios.make_c4(world_seed=1)
    sc, osc = L.Scene(ctx, c4), oracle.OracleScene(c4)
    lo, hi, mi = c4.meta["lower"], c4.meta["upper"], c4.meta["min_interval"]
    rng = np.random.default_rng(44)
    q = rng.uniform(lo, hi, size=(200, 12))
    x = np.zeros((200, 24)); x[:, 0::2] = q
    d, rd = sc.min_distance(x), osc.min_distance(x)
    assert np.allclose(d, rd, atol=1e-12)
    far = np.abs(rd) > 1e-12
    assert np.array_equal((d < 0)[far], (rd < 0)[far]) and 0.05 < (rd < 0).mean() < 0.95
    a = q[rd > 0.0][:96]
    b = rng.uniform(lo, hi, size=(a.shape[0], 12))
    out, nchk = sc.move_position_toward(lo, hi, mi, a, b, fraction=1.0)
    rout, rnchk = osc.qs_move(lo, hi, mi, a, b, fraction=1.0)
    assert np.array_equal(nchk, rnchk) and np.array_equal(out, rout)
    qs = L.make_qs_space(12, lo, hi, mi)
    prm = c4.rrt_params(seed=1, max_vertices=300)
    rc, ro, rtree = osc.rrt_qs(lo, hi, mi, prm)
    pl = L.RrtPlanner(sc, prm, qs=qs)
    st = pl.solve_planning_query()
    assert (st.num_vertices, st.iterations, st.edges_checked) == (ro.num_vertices, ro.iterations, ro.edges_checked)
    assert np.array_equal(pl.tree()["parent"], rtree["parent"]) and np.array_equal(pl.tree()["pos"], rtree["pos"])
    pp = c4.prm_params(seed=2, max_vertices=200, sampling_radius=1.5)
    rc, ro, rg = osc.prm_qs(lo, hi, mi, pp)
    pm = L.PrmPlanner(sc, pp, qs)
    st = pm.solve_planning_query()
    _prm_same(st, pm.graph(), ro, rg)


# ------------------------------------------------------------------ GJK / convex meshes (north_star N1, config C4)
def test_gjk_on_the_device_matches_the_oracle_and_the_closed_forms(L, ctx, oracle):
    """10 000 random poses per primitive pair type: device GJK against the oracle's twin (1e-12) and, while the cores are
    apart, against the restated closed forms (1e-10; capped cylinder / box: within the reference's own golden-section
    tolerance).  Then boxes given as eight-corner meshes, and mesh / mesh pairs."""
    from test_oracle_kat import _rand_quat, _shape, gjk_pair_sets

    rng = np.random.default_rng(123)
    for (ka, kb), (A, B) in gjk_pair_sets(rng, 10000).items():
        g = L.gjk_distance(ctx, A, B)
        og = oracle.gjk_distance(A, B)
        assert np.max(np.abs(g - og)) <= 1e-12, (ka, kb)
        c = np.array([oracle.pair_distance(a, b) for a, b in zip(A[:2000], B[:2000])])
        rad = lambda s: s.dims[0] if s.kind == T.SHAPE_SPHERE else (s.dims[1] if s.kind == T.SHAPE_CCYLINDER else 0.0)
        rsum = np.array([rad(a) + rad(b) for a, b in zip(A[:2000], B[:2000])])
        apart = g[:2000] > -rsum - 1e-10
        assert np.all(c[~apart] < 0.0)
        if (ka, kb) == ("ccyl", "box"):
            tol = 1e-3 * 0.5 * np.array([a.dims[0] for a in A[:2000]])
            assert np.all(g[:2000][apart] <= c[apart] + 1e-10) and np.all(c[apart] - g[:2000][apart] <= tol[apart] + 1e-10)
        else:
            assert np.max(np.abs(g[:2000][apart] - c[apart])) <= 1e-10, (ka, kb)
    rng = np.random.default_rng(9)
    pool, m1, m2 = [], [], []
    for i in range(2000):
        for lst in (m1, m2):
            v = scenarios.random_convex_mesh(rng, int(rng.integers(12, 33)), rng.uniform(0.05, 0.25))
            lst.append(_shape(T.SHAPE_MESH, rng.uniform(-0.6, 0.6, 3), (float(sum(len(x) for x in pool)), float(len(v)), 0.0),
                              _rand_quat(rng)))
            pool.append(v)
    pool = np.concatenate(pool)
    g, og = L.gjk_distance(ctx, m1, m2, pool), oracle.gjk_distance(m1, m2, pool)
    assert np.max(np.abs(g - og)) <= 1e-12 and (g > 0).sum() > 1000 and (g < 0).sum() > 20
    caps = [_shape(T.SHAPE_CCYLINDER, rng.uniform(-0.6, 0.6, 3), (rng.uniform(0.1, 0.5), 0.05, 0), _rand_quat(rng)) for _ in m1]
    g, og = L.gjk_distance(ctx, caps, m2, pool), oracle.gjk_distance(caps, m2, pool)
    assert np.max(np.abs(g - og)) <= 1e-12


def test_nlp_proximity_poses_through_gjk(L, ctx, oracle):
    """Shapes and poses of test_nlp_proximity.cpp:40-58 (six cylinders, six boxes, four poses) and the twenty pairs it
    queues: the device GJK against the oracle's (1e-12).  The reference asserts no value for these pairs -- parity
    unpinned; test_oracle_kat.py checks the oracle against a surface sampling."""
    sh = scenarios.nlp_proximity_shapes()
    pairs = scenarios.NLP_PROXIMITY_PAIRS
    A, B = [sh[a] for a, _ in pairs], [sh[b] for _, b in pairs]
    g, og = L.gjk_distance(ctx, A, B), oracle.gjk_distance(A, B)
    assert np.max(np.abs(g - og)) <= 1e-12 and np.all(g > 1.0)
    g2 = L.gjk_distance(ctx, B, A)
    assert np.max(np.abs(g2 - oracle.gjk_distance(B, A))) <= 1e-12 and np.max(np.abs(g - g2)) <= 1e-10


def test_c4_with_200_convex_mesh_obstacles(L, ctx, oracle):
    """BASELINE config C4 as written: the 12-DOF dual arm among 200 convex MESH obstacles (12-32 vertices each),
    proximity through batched GJK.  Distance queries, edge walks, RRT, bidirectional RRT and the PRM roadmap against
    the oracle."""
    c4 = scenarios.make_c4(world_seed=1, meshes=True)
    assert sum(1 for s in c4.shapes if s.kind == T.SHAPE_MESH) == 200 and 12 * 200 <= len(c4.mesh_vertices) <= 32 * 200
    sc, osc = L.Scene(ctx, c4), oracle.OracleScene(c4)
    lo, hi, mi = c4.meta["lower"], c4.meta["upper"], c4.meta["min_interval"]
    rng = np.random.default_rng(45)
    q = rng.uniform(lo, hi, size=(400, 12))
    x = np.zeros((400, 24)); x[:, 0::2] = q
    d, rd = sc.min_distance(x), osc.min_distance(x)
    assert np.allclose(d, rd, rtol=0, atol=1e-10)
    far = np.abs(rd) > 1e-10
    assert np.array_equal((d < 0)[far], (rd < 0)[far]) and 0.05 < (rd < 0).mean() < 0.95
    a = q[rd > 0.0][:96]
    b = rng.uniform(lo, hi, size=(a.shape[0], 12))
    out, nchk = sc.move_position_toward(lo, hi, mi, a, b, fraction=1.0)
    rout, rnchk = osc.qs_move(lo, hi, mi, a, b, fraction=1.0)
    assert np.array_equal(nchk, rnchk) and np.array_equal(out, rout)
    qs = L.make_qs_space(12, lo, hi, mi)
    prm = c4.rrt_params(seed=1, max_vertices=300)
    rc, ro, rtree = osc.rrt_qs(lo, hi, mi, prm)
    pl = L.RrtPlanner(sc, prm, qs=qs)
    st = pl.solve_planning_query()
    assert (st.num_vertices, st.iterations, st.edges_checked) == (ro.num_vertices, ro.iterations, ro.edges_checked)
    assert np.array_equal(pl.tree()["parent"], rtree["parent"]) and np.array_equal(pl.tree()["pos"], rtree["pos"])
    pp = c4.prm_params(seed=2, max_vertices=200, sampling_radius=1.5)
    rc, ro, rg = osc.prm_qs(lo, hi, mi, pp)
    pm = L.PrmPlanner(sc, pp, qs)
    st = pm.solve_planning_query()
    _prm_same(st, pm.graph(), ro, rg)
    bp = c4.rrt_params(seed=3, max_vertices=150, max_results=2)
    rc, ro, rt = osc.birrt_qs(lo, hi, mi, bp)
    bl = L.BiRrtPlanner(sc, bp, qs)
    st = bl.solve_planning_query()
    assert (st.num_vertices_1, st.num_vertices_2, st.loop_iterations, st.num_solutions) == (ro.n1, ro.n2, ro.loop_iterations,
                                                                                              ro.num_solutions)
    assert np.array_equal(bl.trees()["parent1"], rt["parent1"]) and np.array_equal(bl.trees()["parent2"], rt["parent2"])


# ------------------------------------------------------------------ bidirectional RRT (a7, rr_tree.hpp:256-317)
@pytest.mark.parametrize("seed,max_results", [(1, 3), (2, 1 << 30)])
def test_bidirectional_rrt_identical_to_sequential_planner(L, ctx, oracle, seed, max_results):
    c1 = scenarios.make_c1(world_seed=1)
    sc, osc = L.Scene(ctx, c1), oracle.OracleScene(c1)
    lo, hi, mi = c1.meta["lower"], c1.meta["upper"], c1.meta["min_interval"]
    prm = c1.rrt_params(seed=seed, max_vertices=1500, max_results=max_results)
    rc, ro, rt = osc.birrt_qs(lo, hi, mi, prm)
    pl = L.BiRrtPlanner(sc, prm, L.make_qs_space(3, lo, hi, mi))
    st = pl.solve_planning_query()
    t = pl.trees()
    assert (st.num_vertices_1, st.num_vertices_2, st.loop_iterations, st.samples, st.num_solutions, st.joins,
            st.edges_checked) == (ro.n1, ro.n2, ro.loop_iterations, ro.samples, ro.num_solutions, ro.joins, ro.edges_checked)
    assert st.best_cost == ro.best_cost and st.joins > 0
    for k in ("nn_seq", "accept", "parent1", "parent2", "pos1", "pos2"):
        assert np.array_equal(t[k], rt[k]), k


def test_bidirectional_rrt_batch_c4(L, ctx, oracle):
    """Three seeds of the 12-DOF dual arm in one batch, stopped half way and resumed."""
    c4 = scenarios.make_c4(world_seed=1)
    sc, osc = L.Scene(ctx, c4), oracle.OracleScene(c4)
    lo, hi, mi = c4.meta["lower"], c4.meta["upper"], c4.meta["min_interval"]
    prms = [c4.rrt_params(seed=s, max_vertices=150, max_results=2) for s in (3, 4, 5)]
    pl = L.BiRrtPlanner(sc, prms, L.make_qs_space(12, lo, hi, mi))
    pl.solve_planning_query(max_loop_iterations=40)
    pl.solve_planning_query()
    for i, prm in enumerate(prms):
        rc, ro, rt = osc.birrt_qs(lo, hi, mi, prm)
        st, t = pl.all_stats[i], pl.trees(i)
        assert (st.num_vertices_1, st.num_vertices_2, st.loop_iterations, st.num_solutions) == (
            ro.n1, ro.n2, ro.loop_iterations, ro.num_solutions)
        assert np.array_equal(t["parent1"], rt["parent1"]) and np.array_equal(t["pos2"], rt["pos2"])


# ------------------------------------------------------------------ solution paths
def _path_len(pos, path):
    return sum(float(np.sqrt(((pos[a] - pos[b]) ** 2).sum())) for a, b in zip(path[:-1], path[1:]))


def test_solution_paths_are_consistent_with_the_registered_costs(L, ctx, oracle):
    """The vertex paths of the best solutions follow the parent / predecessor arrays from the start and re-add to the
    registered costs (solution_path_factories.hpp walks)."""
    c1 = scenarios.make_c1(world_seed=1)
    sc = L.Scene(ctx, c1)
    lo, hi, mi = c1.meta["lower"], c1.meta["upper"], c1.meta["min_interval"]
    qs = L.make_qs_space(3, lo, hi, mi)
    # RRT: path to the vertex whose goal probe connected + that probe's distance
    pl = L.RrtPlanner(sc, c1.rrt_params(seed=1, max_vertices=1500), qs=qs)
    st = pl.solve_planning_query()
    tree = pl.tree()
    path, cost = pl.solution()
    assert st.num_solutions > 0 and cost == st.best_cost and path[0] == 0
    assert all(tree["parent"][b] == a for a, b in zip(path[:-1], path[1:]))
    assert np.isclose(_path_len(tree["pos"], path) + tree["goal_dist"][path[-1] - 1], cost, rtol=1e-12)
    # RRT*: predecessor chain of the goal vertex
    ps = L.RrtStarPlanner(sc, c1.rrt_params(seed=1, max_vertices=1200), qs)
    st = ps.solve_planning_query()
    g = ps.graph()
    path, cost = ps.solution()
    assert path[0] == 0 and path[-1] == 1 and cost == g["dist"][1] == st.best_cost
    assert all(g["pred"][b] == a for a, b in zip(path[:-1], path[1:]))
    # bidirectional RRT: tree-1 path to the joining vertex, joining gap, tree-2 path to the goal
    pb = L.BiRrtPlanner(sc, c1.rrt_params(seed=2, max_vertices=800), qs)
    st = pb.solve_planning_query()
    t = pb.trees()
    p1, p2, cost = pb.solution()
    assert st.num_solutions > 0 and cost == st.best_cost and p1[0] == 0 and p2[-1] == 0
    total = _path_len(t["pos1"], p1) + _path_len(t["pos2"], p2) + float(np.sqrt(((t["pos1"][p1[-1]] - t["pos2"][p2[0]]) ** 2).sum()))
    assert np.isclose(total, cost, rtol=1e-12)


# ------------------------------------------------------------------ other chain sizes (template instantiations)
@pytest.mark.parametrize("n", [2, 3, 4, 7])
def test_random_chains_of_other_sizes(L, ctx, oracle, n, monkeypatch):
    """Oblique joint axes, skew offsets, full inertia tensors, rotated base: x' = f(x,u), both steer mappings and a short
    planner run for every joint count the dynamics kernels are instantiated for (6 is BASELINE C2, 1 the pendulum)."""
    scn = scenarios.make_random_chain(n, seed=n)
    sc, osc = L.Scene(ctx, scn), oracle.OracleScene(scn)
    rng = np.random.default_rng(100 + n)
    lo = np.array([scn.dyn.lower[i] for i in range(2 * n)])
    hi = np.array([scn.dyn.upper[i] for i in range(2 * n)])
    x = rng.uniform(lo, hi, size=(96, 2 * n))
    u = rng.uniform(-30, 30, size=(96, n))
    rc, rpd, rM, rf = osc.state_derivative(x, u)
    pd, M, f = sc.state_derivative(x, u)
    assert rc == 0 and np.max(np.abs(M - rM)) <= 1e-12 * np.abs(rM).max()
    assert np.allclose(f, rf, rtol=1e-10, atol=1e-9) and np.allclose(pd, rpd, rtol=1e-9, atol=1e-8)
    a = x[osc.min_distance(x) > 0.01][:40]
    b = rng.uniform(lo, hi, size=(a.shape[0], 2 * n))
    res = {}
    for lanes in ("64", "1", "2"):
        monkeypatch.setenv("RKH_LANES_PER_EDGE", lanes)
        res[lanes] = sc.steer_position_toward(a, b)
    assert np.array_equal(res["1"][0], res["64"][0]) and np.array_equal(res["1"][1], res["64"][1])
    assert np.array_equal(res["2"][0], res["64"][0]) and np.array_equal(res["2"][1], res["64"][1])
    rc, rout, rsteps, _ = osc.steer(a, b)
    assert np.array_equal(res["64"][1], rsteps) and np.allclose(res["64"][0], rout, rtol=1e-9, atol=1e-10)
    monkeypatch.delenv("RKH_LANES_PER_EDGE")
    prm = scn.rrt_params(seed=1, max_vertices=250)
    rc, ro, rtree = osc.rrt_dyn(prm)
    pl = L.RrtPlanner(sc, prm)
    st = pl.solve_planning_query()
    assert (st.num_vertices, st.iterations) == (ro.num_vertices, ro.iterations)
    assert np.array_equal(pl.tree()["parent"], rtree["parent"])


def test_large_batch_in_auto_mode_matches_oracle(L, ctx, oracle, c2):
    """A batch big enough for the planner's own choices to matter: rounds above the lane threshold run the two-lanes-per-
    edge kernel, small ones the wave-per-edge kernel, the NN sweep runs its fp32 pre-filter, launch sizes come from the
    host-side bounds.  Spot-checked problems must still be the sequential planner on their seeds."""
    sc, osc = L.Scene(ctx, c2), oracle.OracleScene(c2, fast=True)
    P, mv = 48, 3000
    prms = [c2.rrt_params(seed=100 + i, max_vertices=mv) for i in range(P)]
    pl = L.RrtPlanner(sc, prms)
    pl.solve_planning_query()
    for i in (0, 17, 47):
        rc, ro, rtree = osc.rrt_dyn(prms[i])
        st, tree = pl.all_stats[i], pl.tree(i)
        assert (st.num_vertices, st.iterations, st.edges_checked) == (ro.num_vertices, ro.iterations, ro.edges_checked)
        assert np.array_equal(tree["nn_seq"], rtree["nn_seq"]) and np.array_equal(tree["accept"], rtree["accept"])
        assert np.array_equal(tree["parent"], rtree["parent"])
        assert np.allclose(tree["pos"], rtree["pos"], rtol=STATE_RTOL, atol=1e-12)


# ------------------------------------------------------------------ C4 dynamics: branching chain + inter-arm beam
def test_c4_dual_arm_dynamics_with_flexible_beam(L, ctx, oracle):
    """12-DOF dual arm with the flexible_beam_3D between the two end effectors (k = 1e4 N/m, k_theta = 1e2): block-
    diagonal mass matrix, bias forces with the beam on both tips, steer kernel and a short dynamic RRT in the 24-D state
    space against the oracle."""
    import copy

    scn = scenarios.make_c4(world_seed=1, beam=(0.7, 1e4, 1e2))
    sc, osc = L.Scene(ctx, scn), oracle.OracleScene(scn)
    rng = np.random.default_rng(12)
    lo = np.array([scn.dyn.lower[i] for i in range(24)])
    hi = np.array([scn.dyn.upper[i] for i in range(24)])
    x = rng.uniform(lo, hi, size=(64, 24))
    u = rng.uniform(-40, 40, size=(64, 12))
    rc, rpd, rM, rf = osc.state_derivative(x, u)
    pd, M, f = sc.state_derivative(x, u)
    assert rc == 0
    assert np.max(np.abs(M - rM)) <= 1e-12 * np.abs(rM).max() and np.abs(rM[:, :6, 6:]).max() == 0.0
    assert np.allclose(f, rf, rtol=1e-10, atol=1e-7) and np.allclose(pd, rpd, rtol=1e-9, atol=1e-7)
    nobeam = oracle.OracleScene(scenarios.make_c4(world_seed=1))
    assert np.abs(nobeam.state_derivative(x, u)[3] - rf).max() > 100.0     # the beam loads both arms
    a = x * 0.4
    a = a[osc.min_distance(a) > 0.01][:24]
    b = rng.uniform(lo, hi, size=(a.shape[0], 24))
    out, steps, _ = sc.steer_position_toward(a, b)
    rc, rout, rsteps, _ = osc.steer(a, b)
    assert np.array_equal(steps, rsteps) and np.allclose(out, rout, rtol=1e-9, atol=1e-9)
    # two waves per edge on the branching chain with the beam: the same bits as one wave per edge
    os.environ["RKH_LANES_PER_EDGE"] = "128"
    try:
        out2, steps2, _ = sc.steer_position_toward(a, b)
    finally:
        del os.environ["RKH_LANES_PER_EDGE"]
    assert np.array_equal(steps2, steps) and np.array_equal(out2, out)
    dyn_scn = copy.copy(scn)
    dyn_scn.start = np.zeros(24)
    dyn_scn.goal = np.zeros(24)
    dyn_scn.goal[0::2] = scn.goal
    prm = dyn_scn.rrt_params(seed=2, max_vertices=120)
    rc, ro, rtree = osc.rrt_dyn(prm)
    pl = L.RrtPlanner(sc, prm)
    st = pl.solve_planning_query()
    assert (st.num_vertices, st.iterations) == (ro.num_vertices, ro.iterations)
    assert np.array_equal(pl.tree()["parent"], rtree["parent"])
    assert np.allclose(pl.tree()["pos"], rtree["pos"], rtol=1e-9, atol=1e-9)


# ------------------------------------------------------------------ planar chains: the reference's 2D classes (true C1)
def _planar_checks(L, ctx, oracle, scn, n_cfg, rrt_vertices, seed):
    n = scn.n_dof
    sc, osc = L.Scene(ctx, scn), oracle.OracleScene(scn)
    lo, hi, mi = scn.meta["lower"], scn.meta["upper"], scn.meta["min_interval"]
    rng = np.random.default_rng(100 + seed)
    q = rng.uniform(lo, hi, size=(n_cfg, n))
    x = np.zeros((n_cfg, 2 * n)); x[:, 0::2] = q
    d, rd = sc.min_distance(x), osc.min_distance(x)
    assert np.allclose(d, rd, atol=1e-12)
    far = np.abs(rd) > 1e-12
    assert np.array_equal((d < 0)[far], (rd < 0)[far]) and 0.02 < (rd < 0).mean() < 0.98
    a = q[rd > 0.0][:128]
    b = rng.uniform(lo, hi, size=(a.shape[0], n))
    out, nchk = sc.move_position_toward(lo, hi, mi, a, b, fraction=1.0)
    rout, rnchk = osc.qs_move(lo, hi, mi, a, b, fraction=1.0)
    assert np.array_equal(nchk, rnchk) and np.array_equal(out, rout)
    qs = L.make_qs_space(n, lo, hi, mi)
    prm = scn.rrt_params(seed=seed, max_vertices=rrt_vertices)
    rc, ro, rtree = osc.rrt_qs(lo, hi, mi, prm)
    pl = L.RrtPlanner(sc, prm, qs=qs)
    st = pl.solve_planning_query()
    assert (st.num_vertices, st.iterations, st.edges_checked, st.num_solutions) == (
        ro.num_vertices, ro.iterations, ro.edges_checked, ro.num_solutions)
    t = pl.tree()
    assert np.array_equal(t["parent"], rtree["parent"]) and np.array_equal(t["pos"], rtree["pos"])
    assert np.array_equal(t["nn_seq"], rtree["nn_seq"]) and np.array_equal(t["accept"], rtree["accept"])
    return sc, osc, qs


def test_c1_planar_arm_with_the_reference_2d_classes(L, ctx, oracle):
    """BASELINE C1 as the reference builds it: revolute_joint_2D / rigid_link_2D chain, capped_rectangle links,
    rectangle obstacles, proxy_query_pair_2D (with its order-dependent cull) -- distance queries, edge walks, RRT to
    5000 vertices, RRT*, PRM and bidirectional RRT against the oracle."""
    c1 = scenarios.make_c1_planar(world_seed=1)
    sc, osc, qs = _planar_checks(L, ctx, oracle, c1, 600, 5000, 1)
    lo, hi, mi = c1.meta["lower"], c1.meta["upper"], c1.meta["min_interval"]
    prm = c1.rrt_params(seed=2, max_vertices=600)
    rc, ro, rg = osc.rrtstar_qs(lo, hi, mi, prm)
    ps = L.RrtStarPlanner(sc, prm, qs)
    st = ps.solve_planning_query()
    g = ps.graph()
    assert (st.num_vertices, st.loop_iterations, st.rewires, st.num_solutions) == (ro.num_vertices, ro.loop_iterations,
                                                                                 ro.rewires, ro.num_solutions)
    assert np.array_equal(g["pred"], rg["pred"]) and np.array_equal(g["pos"], rg["pos"]) and np.array_equal(g["dist"], rg["dist"])
    pp = c1.prm_params(seed=3, max_vertices=400, sampling_radius=1.0)
    rc, ro, rgp = osc.prm_qs(lo, hi, mi, pp)
    pm = L.PrmPlanner(sc, pp, qs)
    stp = pm.solve_planning_query()
    _prm_same(stp, pm.graph(), ro, rgp)
    bp = c1.rrt_params(seed=4, max_vertices=1000, max_results=2)
    rc, ro, rt = osc.birrt_qs(lo, hi, mi, bp)
    pb = L.BiRrtPlanner(sc, bp, qs)
    stb = pb.solve_planning_query()
    assert (stb.num_vertices_1, stb.num_vertices_2, stb.loop_iterations, stb.num_solutions, stb.joins) == (
        ro.n1, ro.n2, ro.loop_iterations, ro.num_solutions, ro.joins)
    tb = pb.trees()
    assert np.array_equal(tb["parent1"], rt["parent1"]) and np.array_equal(tb["pos2"], rt["pos2"]) and stb.best_cost == ro.best_cost


@pytest.mark.parametrize("n,seed", [(4, 1), (2, 4), (7, 3)])
def test_planar_chains_with_every_2d_pair_routine(L, ctx, oracle, n, seed):
    """Circles, capped rectangles and rectangles on both sides (all six finders of proxy_query_pair_2D), oblique link
    offsets and a turned base."""
    scn = scenarios.make_planar_mixed(seed=seed, n=n)
    _planar_checks(L, ctx, oracle, scn, 400, 800, seed)


def test_planar_cull_order_is_the_references(L, ctx, oracle):
    """The order-dependent verdict of proxy_query_pair_2D::findMinimumDistance (tests/test_oracle_kat.py works the
    numbers): the same three shapes give 'free' or 'colliding' depending on the environment's order."""
    ops = [T.KteOp(kind=T.KTE_REVOLUTE_JOINT_2D, coord=0, base_frame=0, end_frame=1, joint_op=-1),
           T.KteOp(kind=T.KTE_RIGID_LINK_2D, coord=-1, base_frame=1, end_frame=2, joint_op=-1)]
    ops[1].offset = T.make_pose_2d((0.2, 0.0))
    link = T.Shape(kind=T.SHAPE_CRECT, anchor=1)
    link.pose = T.make_pose_2d((0.1, 0.0))
    link.dims[:] = [0.2, 6.0, 0.0]

    def circle(pos, r):
        s = T.Shape(kind=T.SHAPE_CIRCLE, anchor=-1)
        s.pose = T.make_pose_2d(pos)
        s.dims[:] = [r, 0.0, 0.0]
        return s
    far, near = circle((0.1, 3.11), 0.1), circle((3.2, 0.0), 0.05)
    base = T.ChainBase()
    base.pose = T.make_pose_2d()
    for shapes, expect in (([link, far, near], 0.01), ([link, near, far], -0.05)):
        scn = scenarios.Scenario(name="cull", ops=ops, base=base, shapes=shapes, dyn=T.DynSpace(), n_dof=1, n_frames=3,
                                 start=np.zeros(1), goal=np.zeros(1), meta={})
        d = L.Scene(ctx, scn).min_distance(np.zeros((1, 2)))[0]
        assert d == oracle.OracleScene(scn).min_distance(np.zeros((1, 2)))[0]
        assert d == pytest.approx(expect, rel=1e-9)


def test_planar_scene_refuses_dynamics(L, ctx):
    c1 = scenarios.make_c1_planar()
    sc = L.Scene(ctx, c1)
    with pytest.raises(L.RkhError):
        sc.state_derivative(np.zeros((1, 6)), np.zeros((1, 3)))


def test_batch_scale_fit_and_wave_packing_do_not_change_results(L, ctx, oracle, c2, monkeypatch):
    """The per-round batch scale (fitted on the device to whole passes of steer waves) and the packed wave mapping only
    move work around: the trees of a batch are the same with the fit switched off, and equal to the oracle's."""
    prms = [c2.rrt_params(seed=70 + i, max_vertices=2500) for i in range(40)]
    trees = {}
    for fit in ("1", "0"):
        monkeypatch.setenv("RKH_WAVE_FIT", fit)
        pl = L.RrtPlanner(L.Scene(ctx, c2), prms)
        pl.solve_planning_query()
        trees[fit] = [(int(pl.all_stats[i].num_vertices), int(pl.all_stats[i].iterations), pl.tree(i)) for i in (0, 17, 39)]
        pl.close()
    osc = oracle.OracleScene(c2, fast=True)
    for k, i in enumerate((0, 17, 39)):
        a, b = trees["1"][k], trees["0"][k]
        assert a[:2] == b[:2]
        for key in ("parent", "nn_seq", "accept", "pos"):
            assert np.array_equal(a[2][key], b[2][key]), key
    rc, ro, rt = osc.rrt_dyn(prms[17])
    t = trees["1"][1]
    assert (t[0], t[1]) == (ro.num_vertices, ro.iterations)
    assert np.array_equal(t[2]["parent"], rt["parent"]) and np.array_equal(t[2]["accept"], rt["accept"])


def test_planner_pool_splits_a_batch_without_changing_results(L, ctx, oracle, c2):
    """RrtPlannerPool (problems split over planner handles = HIP streams, enqueued round-robin) returns, per global
    problem index, what one handle with all problems returns."""
    prms = [c2.rrt_params(seed=300 + i, max_vertices=600) for i in range(7)]
    sc = L.Scene(ctx, c2)
    one = L.RrtPlanner(sc, prms)
    one.solve_planning_query()
    pool = L.RrtPlannerPool(sc, prms, groups=3)
    pool.solve_planning_query()
    assert pool.done and len(pool.all_stats) == 7
    for i in range(7):
        a, b = one.tree(i), pool.tree(i)
        assert int(one.all_stats[i].num_vertices) == int(pool.all_stats[i].num_vertices)
        for k in ("parent", "nn_seq", "accept", "pos"):
            assert np.array_equal(a[k], b[k]), (i, k)
        assert one.solution(i)[1] == pool.solution(i)[1]
    pool.close()
    one.close()


def test_stepwise_and_two_phase_steer_launches_do_not_change_results(L, ctx, oracle, c2, monkeypatch):
    """The steer launch of a large round in its forms -- one launch for the whole edge, two phases with a compaction of
    the survivors between them, one launch per RK4 step over the live edges of all problems, and the default: a resident
    set of waves whose lane pairs take a new edge from the round's pool whenever theirs ends, followed by list launches
    for the edges the waves hand over at the end -- runs the same arithmetic per edge.  The round sizes at which the planner switches between them are far above what a test can
    afford (32 k / 65 k edges), so RKH_STEER_SPLIT_MIN_EDGES = 0 sends every round of the two-lanes mapping (>= 1024 edges)
    through the form under test.  Trees, NN sequences, accept bits, goal probes and counters must be identical across
    the forms, equal to the oracle's, and the executed-step counter must agree with the free-step counts."""
    prms = [c2.rrt_params(seed=70 + i, max_vertices=2500) for i in range(40)]
    picks = (0, 17, 39)
    runs = {}
    for name, env in (("whole", {"RKH_STEER_SPLIT": "0", "RKH_STEER_STEPWISE": "0"}),
                      ("two_phase", {"RKH_STEER_STEPWISE": "0", "RKH_STEER_SPLIT": "5", "RKH_STEER_SPLIT_MIN_EDGES": "0"}),
                      ("stepwise", {"RKH_STEER_STEPWISE": "1", "RKH_STEER_POOL": "0", "RKH_STEER_SPLIT_MIN_EDGES": "0"}),
                      ("pool", {"RKH_STEER_STEPWISE": "1", "RKH_STEER_POOL": "1", "RKH_STEER_SPLIT_MIN_EDGES": "0"}),
                      # few resident waves: every wave refills its lanes many times and hands orphans over at the end
                      ("pool_few_waves", {"RKH_STEER_STEPWISE": "1", "RKH_STEER_POOL": "1", "RKH_STEER_SPLIT_MIN_EDGES": "0",
                                          "RKH_STEER_POOL_WAVES": "24"})):
        for k in ("RKH_STEER_SPLIT", "RKH_STEER_STEPWISE", "RKH_STEER_SPLIT_MIN_EDGES", "RKH_STEER_POOL", "RKH_STEER_POOL_WAVES"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        pl = L.RrtPlanner(L.Scene(ctx, c2), prms)
        pl.solve_planning_query()
        runs[name] = {"stats": [(int(s.num_vertices), int(s.iterations), int(s.edges_checked), int(s.num_solutions),
                                 float(s.best_cost)) for s in pl.all_stats],
                      "trees": [pl.tree(i) for i in picks], "steps": pl.steer_steps(),
                      "spec": sum(int(s.edges_speculated) for s in pl.all_stats)}
        pl.close()
    for name in ("two_phase", "stepwise", "pool", "pool_few_waves"):
        assert runs[name]["stats"] == runs["whole"]["stats"], name
        for a, b in zip(runs[name]["trees"], runs["whole"]["trees"]):
            for key in ("parent", "nn_seq", "accept", "pos", "goal_dist"):
                assert np.array_equal(a[key], b[key]), (name, key)
    # executed work: every form integrates the same steps (a step counts when it starts from a live edge), fewer than
    # the 20 per propagated edge the launches are sized for
    assert runs["stepwise"]["steps"] == runs["whole"]["steps"] == runs["two_phase"]["steps"] == runs["pool"]["steps"]
    assert runs["pool_few_waves"]["steps"] == runs["whole"]["steps"]
    assert 0 < runs["stepwise"]["steps"] < 20 * 2 * runs["stepwise"]["spec"]
    osc = oracle.OracleScene(c2, fast=True)
    rc, ro, rt = osc.rrt_dyn(prms[17])
    t = runs["stepwise"]["trees"][1]
    assert runs["stepwise"]["stats"][17][:3] == (ro.num_vertices, ro.iterations, ro.edges_checked)
    for key in ("parent", "nn_seq", "accept"):
        assert np.array_equal(t[key], rt[key]), key
    assert np.allclose(t["pos"], rt["pos"], rtol=STATE_RTOL, atol=1e-12)


@pytest.mark.parametrize("D", [3, 6, 12])
def test_reference_hidim_scenario_without_obstacles(L, ctx, oracle, D):
    """The reference's own planner scenario (test_hidim_planners.cpp:151,195-212: unit hypercube, no obstacles, start
    0.05 * 1 -> goal 0.95 * 1; see scenarios.make_hidim for what is and is not modelled).  The scene has ZERO proximity
    pairs: rkh_scene_create must accept it, rkh_min_distance reports +inf (findMinimumDistance over an empty finder list
    leaves the initial infinity, proxy_query_model.cpp:376-402), every edge walk reaches its target, and RRT, RRT*, PRM and
    the bidirectional RRT build the oracle's graphs.  The reference holds no expected output for the scenario: the pin is
    the oracle (parity unpinned beyond it)."""
    hd = scenarios.make_hidim(D)
    sc, osc = L.Scene(ctx, hd), oracle.OracleScene(hd)
    assert sc.num_pairs == 0
    lo, hi, mi = hd.meta["lower"], hd.meta["upper"], hd.meta["min_interval"]
    rng = np.random.default_rng(D)
    x = np.zeros((64, 2 * D))
    x[:, 0::2] = rng.uniform(0, 1, size=(64, D))
    d = sc.min_distance(x)
    assert np.all(np.isinf(d)) and np.all(d > 0) and np.array_equal(d, osc.min_distance(x))
    a, b = rng.uniform(0, 1, size=(32, D)), rng.uniform(0, 1, size=(32, D))
    out, steps = sc.move_position_toward(lo, hi, mi, a, b)
    assert np.array_equal(out, b)  # nothing in the way: the walk ends on its target (the exact-end shortcut of fraction 1)
    qs = L.make_qs_space(D, lo, hi, mi)
    prm = hd.rrt_params(seed=2, max_vertices=1500)
    rc, ro, rt = osc.rrt_qs(lo, hi, mi, prm)
    pl = L.RrtPlanner(sc, prm, qs=qs)
    st, tree = pl.solve_planning_query(), pl.tree()
    assert (st.num_vertices, st.iterations, st.edges_checked, st.num_solutions) == (ro.num_vertices, ro.iterations,
                                                                                  ro.edges_checked, ro.num_solutions)
    for k in ("nn_seq", "accept", "parent", "pos", "goal_dist"):
        assert np.array_equal(tree[k], rt[k]), k
    assert st.num_solutions > 0 and st.best_cost == ro.best_cost
    assert np.all(tree["accept"] == 1)  # every sample is reached: an expansion is rejected by obstacles only
    pl.close()
    prm = hd.rrt_params(seed=3, max_vertices=600)
    rc, ro, rg = osc.rrtstar_qs(lo, hi, mi, prm)
    ps = L.RrtStarPlanner(sc, prm, qs)
    st, g = ps.solve_planning_query(), ps.graph()
    assert (st.num_vertices, st.loop_iterations, st.rewires, st.edges_checked) == (ro.num_vertices, ro.loop_iterations,
                                                                                   ro.rewires, ro.edges_checked)
    for k in ("pred", "pos", "dist"):
        assert np.array_equal(g[k], rg[k]), k
    # in free space the optimal cost to every vertex is its straight-line distance from the start, and RRT* may only
    # approach it from above
    conn = np.flatnonzero(g["pred"] != 0xFFFFFFFF)
    assert np.all(g["dist"][conn] >= np.sqrt(((g["pos"][conn] - g["pos"][0]) ** 2).sum(axis=1)) * (1 - 1e-12))
    ps.close()
    pp = hd.prm_params(seed=4, max_vertices=400, sampling_radius=0.2 * np.sqrt(D))
    rc, ro, rg = osc.prm_qs(lo, hi, mi, pp)
    pr = L.PrmPlanner(sc, pp, qs)
    st, g = pr.solve_planning_query(), pr.graph()
    assert (st.num_vertices, st.num_edges, st.num_components, st.loop_iterations) == (ro.num_vertices, ro.num_edges,
                                                                                     ro.num_components, ro.loop_iterations)
    for k in ("pos", "edge_u", "edge_v", "edge_w", "density"):
        assert np.array_equal(g[k], rg[k]), k
    pr.close()


def test_planning_in_an_obstacle_course_read_from_an_rkx_archive(L, ctx, oracle, tmp_path):
    """f4: the `window_crossing` course of R/examples/misc/build_X8_obstacle_courses.cpp:69-152 (floor plane + eight wall
    boxes) goes through a ReaK XML archive (reak_amd/rkx.py), is read back, and becomes the environment of the 6-R arm
    standing in front of the window wall; the whole scene then round-trips through write_scene / read_scene and
    the COPY is what the device and the oracle plan in: distances, edge walks and the quasi-static RRT must agree.  (The
    reference flies a vehicle in SE(3) through the course; the hot path's spaces are joint spaces, so a manipulator
    stands in.  The reference holds no expected outputs for the course: parity unpinned beyond the oracle.)"""
    from reak_amd import rkx
    from reak_amd import types as T

    path = tmp_path / "window_crossing.rkx"
    path.write_text(rkx.write_obstacle_course("window_crossing"))
    shapes, names, start_pos, end_pos = rkx.read_obstacle_course(path.read_text())
    assert names[0] == "floor" and len(shapes) == 9
    c3 = scenarios.make_c3(world_seed=1)
    arm = [s for s in c3.shapes if s.anchor >= 0]
    c3.shapes = arm + shapes
    c3.base.pose = T.make_pose((2.55, 4.5, -2.5))  # 0.35 m in front of wall2 (x = 2.9 .. 3.1), inside the course's volume (z < 0)
    scene_file = tmp_path / "arm_in_course.rkx"
    scene_file.write_text(rkx.write_scene(c3))
    scn = rkx.read_scene(scene_file.read_text(), c3)
    sc, osc = L.Scene(ctx, scn), oracle.OracleScene(scn)
    lo, hi, mi = scn.meta["lower"], scn.meta["upper"], scn.meta["min_interval"]
    rng = np.random.default_rng(5)
    x = np.zeros((512, 12))
    x[:, 0::2] = rng.uniform(-np.pi, np.pi, size=(512, 6))
    d, rd = sc.min_distance(x), osc.min_distance(x)
    assert np.allclose(d, rd, rtol=0, atol=1e-12) and np.array_equal(d < 0, rd < 0)
    assert 0.05 < np.mean(d < 0) < 0.95  # the walls matter: some configurations collide, some are free
    prm = scn.rrt_params(seed=3, max_vertices=1200)
    rc, ro, rt = osc.rrt_qs(lo, hi, mi, prm)
    pl = L.RrtPlanner(sc, prm, qs=L.make_qs_space(6, lo, hi, mi))
    st, tree = pl.solve_planning_query(), pl.tree()
    assert (st.num_vertices, st.iterations, st.edges_checked, st.num_solutions) == (ro.num_vertices, ro.iterations,
                                                                                  ro.edges_checked, ro.num_solutions)
    for k in ("nn_seq", "accept", "parent", "pos"):
        assert np.array_equal(tree[k], rt[k]), k
    assert 0 < np.sum(tree["accept"] == 0)  # walls stop some expansions
    pl.close()


def test_prm_random_walks_over_the_planar_dynamic_space(L, ctx, oracle):
    """PRM -- rejection sampling on is_free(state), random_walk as an RK4 propagation over a fraction of the edge time
    (EDGE_WALK_ACCEPT, planning_visitors.hpp:403-432), can_be_connected by full propagations -- over the state space of the
    PLANAR 3R arm (revolute_joint_2D / rigid_link_2D / inertia_2D; one lane per edge, propagate_planar.hip): the entry
    point existed since round 2 without a test.  Same loop decisions, roadmap and densities as the sequential planner.
    (The bidirectional planners need a reversible space -- rrtstar_path_planner.tpp:70 -- and are not offered over any
    dynamic space, planar or not.)"""
    scn = scenarios.make_c1_planar(world_seed=1, dynamics=True)
    sc, osc = L.Scene(ctx, scn), oracle.OracleScene(scn)
    prm = scn.prm_params(seed=4, max_vertices=400, sampling_radius=1.0)
    prm.base.conn_tol = 3.0
    rc, rout, rg = osc.prm_dyn(prm)
    assert rc == 0
    pl = L.PrmPlanner(sc, prm, scn.dyn)
    st = pl.solve_planning_query()
    g = pl.graph()
    assert (st.num_vertices, st.num_edges, st.samples, st.rejected, st.loop_iterations, st.num_components,
            st.edges_checked) == (rout.num_vertices, rout.num_edges, rout.samples, rout.rejected, rout.loop_iterations,
                                  rout.num_components, rout.edges_checked)
    k = np.bincount(g["kind"], minlength=3)
    assert k[1] > 10 and st.num_edges > 100  # random walks expanded the roadmap, connections were made
    assert np.array_equal(g["kind"], rg["kind"]) and np.array_equal(g["expanded"], rg["expanded"])
    assert np.array_equal(g["edge_u"], rg["edge_u"]) and np.array_equal(g["edge_v"], rg["edge_v"])
    assert np.array_equal(g["cc_root"], rg["cc_root"])
    assert np.allclose(g["pos"], rg["pos"], rtol=STATE_RTOL, atol=1e-12)
    assert np.allclose(g["density"], rg["density"], rtol=1e-8, atol=1e-12)
    pl.close()


def test_proximity_stage_counts_are_consistent(L, ctx, oracle, c2):
    """rkh_diag_proximity_counts (the `collide` object of bench.py): the counted stages are nested, the number of states
    found in collision equals the oracle's verdicts on the same states, and a scene's pairs within static reach are a
    subset of its proxy pairs."""
    sc, osc = L.Scene(ctx, c2), oracle.OracleScene(c2)
    rng = np.random.default_rng(3)
    lo = np.array([c2.dyn.lower[i] for i in range(12)]); hi = np.array([c2.dyn.upper[i] for i in range(12)])
    x = rng.uniform(lo, hi, size=(5000, 12))
    c = sc.proximity_counts(x)
    assert c["states"] == 5000
    assert 0 < c["pairs_in_static_reach"] <= c["pairs_per_state"] == sc.num_pairs
    assert c["closed_forms"] + c["golden_section"] <= c["pairs_past_cull"] <= c["states"] * c["pairs_in_static_reach"]
    assert c["states_in_collision"] == int((osc.min_distance(x) < 0.0).sum())
