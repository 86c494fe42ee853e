// abi_smoke.cpp -- the C++ adaptors of include/rkh_adaptors.hpp compiled by g++ and linked against librkh.so.
//
// A minimal vector graph stands in for ReaK's BGL motion graph; the program
//   1. drives generate_rrt (R/ctrl/graph_alg/rr_tree.hpp:179-199) with the visitor predicates of
//      planning_visitor_base::steer_towards_position (R/ctrl/path_planning/planning_visitors.hpp:349-360) ONE QUERY AT
//      A TIME through the sockets: NNFinder functor, KNN synchro, steerable C_free topology, proximity pair;
//   2. runs the batched planner entry (hip_rrt_planner) on the same query and seed;
//   3. checks that both grew the same tree (they are the same sequential algorithm), and prints one JSON line the
//      calling test compares with the oracle.
// usage: abi_smoke <scene.bin>   (written by tests/test_cpp_adaptors.py: ops, base, shapes, rkh_dyn_space, rkh_rrt_params)
#include <cmath>
#include <cstdio>
#include <cstring>
#include <iterator>
#include <random>
#include <vector>

#include "rkh_adaptors.hpp"

typedef std::vector<double> Point;

struct VertexData {
  Point position;
  std::size_t parent;
};
struct VectorGraph {  // stands in for boost::adjacency_list_BC<vecBC, ...>: vertex descriptors are indices
  std::vector<VertexData> v;
};
inline std::size_t vertex(std::size_t i, VectorGraph&) { return i; }
inline std::size_t num_vertices(const VectorGraph& g) { return g.v.size(); }
struct PositionMap {
  VectorGraph* g;
};
inline const Point& get(PositionMap m, std::size_t u) { return m.g->v[u].position; }

template <typename T>
static bool read_pod(FILE* f, T* out, std::size_t n = 1) {
  return std::fread(out, sizeof(T), n, f) == n;
}

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  FILE* f = std::fopen(argv[1], "rb");
  if (!f) return 2;
  int32_t n_ops = 0, n_shapes = 0;
  rkh_chain_base base;
  rkh_dyn_space sp;
  rkh_rrt_params prm;
  std::vector<rkh_kte_op> ops;
  std::vector<rkh_shape> shapes;
  bool ok = read_pod(f, &n_ops);
  ops.resize(n_ops);
  ok = ok && read_pod(f, ops.data(), n_ops) && read_pod(f, &base) && read_pod(f, &n_shapes);
  shapes.resize(n_shapes);
  ok = ok && read_pod(f, shapes.data(), n_shapes) && read_pod(f, &sp) && read_pod(f, &prm);
  std::fclose(f);
  if (!ok) return 2;
  try {
    const std::size_t D = 2 * std::size_t(sp.n_dof);
    auto ctx = rkh::make_context(0);
    auto scene = rkh::make_scene(ctx, ops.data(), n_ops, base, shapes.data(), n_shapes);
    typedef rkh::kte_dynamic_free_space<Point> FreeSpace;
    auto space = std::make_shared<FreeSpace>(scene, sp);
    const Point start(prm.start, prm.start + D), goal(prm.goal, prm.goal + D);

    // ---- 1. the stock algorithm through the sockets, one query at a time
    VectorGraph g;
    PositionMap pm{&g};
    auto nn = rkh::make_nn(ctx, int(D), uint64_t(prm.max_vertices) + 2);
    rkh::hip_linear_neighbor_search<VectorGraph> find_nearest(nn);
    rkh::hip_knn_synchro<VectorGraph, PositionMap> synchro(nn, pm);
    rkh::hip_proxy_query_pair proxy(scene);
    std::mt19937 eng(prm.seed);  // get_global_rng().seed(seed)
    g.v.push_back(VertexData{start, std::size_t(-1)});  // create_root (rrt_path_planner.tpp:131-133)
    synchro.added_vertex(std::size_t(0), g);
    unsigned long added = 0, iterations = 0, proxy_disagree = 0;
    while (added < prm.max_vertices) {  // keep_going (planning_visitors.hpp:203-205)
      ++iterations;
      const Point p_rnd = space->random_point(eng);                                     // rr_tree.hpp:194
      const std::size_t u = find_nearest(p_rnd, g, space->get_super_space(), pm);       // rr_tree.hpp:195
      const Point pu = get(pm, u);
      const std::pair<Point, FreeSpace::steer_record_type> st = space->steer_position_toward(pu, 1.0, p_rnd);
      const double traveled = space->get_super_space().distance(pu, st.first);
      const double best_case = space->get_super_space().distance(pu, p_rnd);
      const bool reached_new = (!std::isinf(traveled)) && (traveled < 2.0 * best_case) && (traveled > prm.steer_tol * best_case);
      if (!reached_new) continue;
      // the proximity socket must agree with the topology's is_free on the accepted state (it is collision-free)
      proxy.apply_to_model(st.first);
      auto finder = proxy.findMinimumDistance();
      if (finder && (finder->getLastResult().mDistance < 0.0)) ++proxy_disagree;
      if (!space->is_free(st.first)) ++proxy_disagree;
      g.v.push_back(VertexData{st.first, u});                                             // add_child_vertex
      synchro.added_vertex(num_vertices(g) - 1, g);                                       // vis.vertex_added
      ++added;
    }
    // k-NN through the same functor: nearest first, every distance below the radius
    std::vector<std::size_t> nb;
    find_nearest(goal, std::back_inserter(nb), g, space->get_super_space(), pm, 8, 1e9);
    bool knn_sorted = nb.size() == std::min<std::size_t>(8, num_vertices(g));
    for (std::size_t i = 1; i < nb.size(); ++i)
      knn_sorted = knn_sorted && space->get_super_space().distance(goal, get(pm, nb[i - 1])) <=
                                     space->get_super_space().distance(goal, get(pm, nb[i]));

    // ---- 2. the batched planner entry on the same query
    rkh::hip_rrt_planner<FreeSpace> planner(space, prm.max_vertices, prm.steer_tol, prm.conn_tol);
    const auto r = planner.solve_planning_query(start, goal, prm.seed, prm.max_results);

    // ---- 3. same tree?
    bool same = (r.positions.size() == num_vertices(g)) && (r.stats.iterations == iterations);
    double max_diff = 0.0;
    for (std::size_t v = 0; same && v < num_vertices(g); ++v) {
      same = same && (v == 0 ? r.parent[v] == 0xFFFFFFFFu : std::size_t(r.parent[v]) == g.v[v].parent);
      for (std::size_t i = 0; i < D; ++i) max_diff = std::fmax(max_diff, std::fabs(r.positions[v][i] - g.v[v].position[i]));
    }
    same = same && (max_diff == 0.0);  // the same kernels computed both
    // errors arrive as exceptions: a state of the wrong size is a std::range_error like in kte_nl_system.hpp:181-188
    bool threw = false;
    try {
      rkh::check(rkh_min_distance(scene.get(), nullptr, 1, nullptr));
    } catch (const std::range_error&) {
      threw = true;
    }
    std::printf("{\"vertices\": %zu, \"iterations\": %lu, \"planner_vertices\": %llu, \"planner_iterations\": %llu, "
                "\"same_tree\": %s, \"max_abs_diff\": %.3g, \"knn_sorted\": %s, \"proxy_disagree\": %lu, \"bad_arg_throws\": %s, "
                "\"parents\": [",
                num_vertices(g), iterations, (unsigned long long)r.stats.num_vertices, (unsigned long long)r.stats.iterations,
                same ? "true" : "false", max_diff, knn_sorted ? "true" : "false", proxy_disagree, threw ? "true" : "false");
    for (std::size_t v = 1; v < num_vertices(g); ++v) std::printf("%s%zu", v > 1 ? ", " : "", g.v[v].parent);
    std::printf("]}\n");
    return (same && knn_sorted && proxy_disagree == 0 && threw) ? 0 : 1;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "abi_smoke: %s\n", e.what());
    return 3;
  }
}
