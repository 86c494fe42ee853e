"""Size-independent properties at BASELINE's full sizes (where the oracle would take minutes to hours): the HIP path must
satisfy what the domain guarantees.  GPU only; everything goes through the C-ABI."""
import numpy as np
import pytest

from reak_amd import scenarios

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    from reak_amd import lib

    return lib


@pytest.fixture(scope="module")
def ctx(L):
    return L.Context(0)


def test_nn_sweep_properties_at_one_million_vertices(L, ctx):
    """BASELINE C3 tree size (1 M vertices, 6-D): a vertex is its own nearest neighbour at distance 0; the answer does not
    depend on how the queries are batched; the first of the k nearest is the nearest; k-NN distances ascend and respect
    the radius; appending vertices can only shrink a query's NN distance."""
    n, D = 1_000_000, 6
    nn = L.HipNeighborSearch(ctx, D, n + 64)
    nn.fill_uniform(n, seed=11)
    rng = np.random.default_rng(3)
    q = rng.random((96, D))
    idx, dist = nn.nearest(q)
    assert idx.max() < n and np.all(dist > 0)
    for lo, hi in ((0, 1), (1, 9), (9, 96)):       # any batching gives the same answers
        i2, d2 = nn.nearest(q[lo:hi])
        assert np.array_equal(i2, idx[lo:hi]) and np.array_equal(d2, dist[lo:hi])
    kidx, kdist, kcnt = nn.k_nearest(q[:16], 80, radius=np.inf)
    assert np.all(kcnt == 80)
    assert np.array_equal(kidx[:, 0], idx[:16]) and np.array_equal(kdist[:, 0], dist[:16])
    assert np.all(np.diff(kdist, axis=1) >= 0)
    r = float(np.median(kdist[:, 40]))
    kidx2, kdist2, kcnt2 = nn.k_nearest(q[:16], 80, radius=r)
    for b in range(16):
        m = int((kdist[b] < r).sum())
        assert kcnt2[b] == m and np.array_equal(kidx2[b, :m], kidx[b, :m])
    nn.added_vertices(q[:64])                        # the queries themselves become vertices n .. n+63
    i3, d3 = nn.nearest(q)
    assert np.array_equal(i3[:64], np.arange(n, n + 64)) and np.all(d3[:64] == 0.0)
    assert np.all(d3[64:] <= dist[64:])


def test_full_size_rrt_run_invariants(L, ctx):
    """BASELINE C2 at its full tree size (100 000 vertices), four seeds in one batch: structural invariants of
    generate_rrt, the vertex budget, bounds and collision-freedom of every vertex, and run-to-run determinism."""
    c2 = scenarios.make_c2(world_seed=1)
    sc = L.Scene(ctx, c2)
    mv = 100_000
    prms = [c2.rrt_params(seed=900 + i, max_vertices=mv) for i in range(4)]
    pl = L.RrtPlanner(sc, prms)
    pl.solve_planning_query()
    lo = np.array([c2.dyn.lower[i] for i in range(12)])
    hi = np.array([c2.dyn.upper[i] for i in range(12)])
    first = None
    for i in range(4):
        st, t = pl.all_stats[i], pl.tree(i)
        n, it = int(st.num_vertices), int(st.iterations)
        assert n == mv + 1 and it >= mv                      # keep_going(): the root is not counted
        assert st.edges_checked == it + (n - 1)              # one steer per iteration + one goal probe per vertex
        assert int(t["accept"].sum()) == n - 1               # every accepted steer added exactly one vertex
        par = t["parent"]
        assert par[0] == 0xFFFFFFFF and np.all(par[1:] < np.arange(1, n))   # a parent precedes its child
        # the k-th accepted sample created vertex k+1; its parent is that sample's nearest neighbour
        acc_iter = np.flatnonzero(t["accept"])
        assert np.array_equal(par[1:], t["nn_seq"][acc_iter])
        # the nearest neighbour of sample s existed when s was drawn
        n_before = 1 + np.concatenate(([0], np.cumsum(t["accept"])[:-1]))
        assert np.all(t["nn_seq"] < n_before)
        assert np.all(t["pos"] >= lo - 1e-12) and np.all(t["pos"] <= hi + 1e-12)
        sub = t["pos"][:: max(1, n // 4000)]
        assert np.all(sc.min_distance(sub) >= 0.0)           # vertices are collision-free states
        if i == 0:
            first = (par.copy(), t["pos"].copy())
    pl.close()
    pl2 = L.RrtPlanner(sc, prms[:1])                          # same seed alone: identical tree (batching is invisible)
    pl2.solve_planning_query()
    t2 = pl2.tree(0)
    assert np.array_equal(t2["parent"], first[0]) and np.array_equal(t2["pos"], first[1])
