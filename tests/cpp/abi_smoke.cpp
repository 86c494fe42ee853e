// abi_smoke.cpp -- the C++ adaptors of include/rkh_adaptors.hpp compiled by g++ and linked against librkh.so.
//
// A minimal vector graph stands in for ReaK's BGL motion graph; the program
//   1. drives generate_rrt (R/ctrl/graph_alg/rr_tree.hpp:179-199) with the visitor predicates of
//      planning_visitor_base::steer_towards_position (R/ctrl/path_planning/planning_visitors.hpp:349-360) ONE QUERY AT
//      A TIME through the sockets: NNFinder functor, KNN synchro, steerable C_free topology, proximity pair;
//   2. runs the planner entry -- hip_rrt_planner::solve_planning_query(Query&), the signature of
//      sample_based_planner::solve_planning_query(planning_query&) (motion_planner_base.hpp:102) -- on the same query and
//      seed, with a point-to-point query object (p2p_planning_query.hpp:74-229) collecting register_solution calls and a
//      progress hook counting report_progress calls;
//   3. checks that both grew the same tree (they are the same sequential algorithm);
//   4. on a quasi-static scene drives RRT, RRT*, PRM and the bidirectional RRT through their adaptors and query objects
//      and prints what the calling test compares with the oracle, one JSON line in all.
// usage: abi_smoke <dyn_scene.bin> <qs_scene.bin>   (written by tests/test_cpp_adaptors.py)
#include <algorithm>
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
inline std::size_t null_vertex(const VectorGraph&) { return std::size_t(-1); }
struct PositionMap {
  VectorGraph* g;
};
inline const Point& get(PositionMap m, std::size_t u) { return m.g->v[u].position; }

template <typename T>
static bool read_pod(FILE* f, T* out, std::size_t n = 1) {
  return std::fread(out, sizeof(T), n, f) == n;
}

// path_planning_p2p_query (p2p_planning_query.hpp:74-229) without Boost: start / goal, max_num_results, keep_going, and
// register_solution walking the predecessors like register_basic_solution_path_impl (solution_path_factories.hpp:58-110)
struct P2PQuery {
  Point start_pos, goal_pos;
  std::size_t max_num_results;
  std::vector<std::pair<double, std::vector<std::size_t> > > solutions;  // (cost, vertex path start .. last)
  std::size_t register_calls = 0;
  const Point& get_start_position() const { return start_pos; }
  const Point& get_goal_position() const { return goal_pos; }
  bool keep_going() const { return max_num_results > solutions.size(); }
  double get_best_solution_distance() const {
    double b = std::numeric_limits<double>::infinity();
    for (const auto& s : solutions) b = std::min(b, s.first);
    return b;
  }
  void reset_solution_records() { solutions.clear(); }
  template <typename Vertex, typename Graph>
  bool register_solution(Vertex start_node, Vertex goal_node, double goal_distance, Graph& g) {
    ++register_calls;
    std::vector<std::size_t> path;
    Vertex v = goal_node;
    while (v != Graph::null_vertex() && v != start_node) {
      path.push_back(v);
      v = g[v].predecessor;
    }
    if (v != start_node) return false;
    path.push_back(start_node);
    std::reverse(path.begin(), path.end());
    const double cost = g[goal_node].distance_accum + goal_distance;
    if (!(cost < get_best_solution_distance())) return false;  // only improvements are recorded (:91-97)
    solutions.push_back(std::make_pair(cost, path));
    return true;
  }
  template <typename Vertex, typename Graph>
  bool register_joining_point(Vertex, Vertex, Vertex join1, Vertex join2, double joining_distance, Graph& g1, Graph& g2) {
    ++register_calls;
    const double cost = g1[join1].distance_accum + joining_distance + g2[join2].distance_accum;
    if (!(cost < get_best_solution_distance())) return false;
    solutions.push_back(std::make_pair(cost, std::vector<std::size_t>{join1, join2}));
    return true;
  }
};

static void print_u32_list(const char* key, const std::vector<std::size_t>& v, std::size_t cap = 4000) {
  std::printf("\"%s\": [", key);
  for (std::size_t i = 0; i < v.size() && i < cap; ++i)
    std::printf("%s%lld", i ? ", " : "", v[i] == std::size_t(-1) ? -1ll : (long long)v[i]);
  std::printf("]");
}

int main(int argc, char** argv) {
  if (argc < 3) return 2;
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
    rkh::check(RKH_ABI_CHECK());  // this translation unit's rkh.h against the library's
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

    // ---- 2. the planner entry on the same query: solve_planning_query(Query&)
    rkh::hip_rrt_planner<FreeSpace> planner(space, prm.max_vertices, /*progress interval*/ 25, prm.steer_tol, prm.conn_tol);
    unsigned long progress_reports = 0, solution_reports = 0;
    planner.progress_reporter = [&](const FreeSpace&, const rkh::hip_motion_graph<Point>&) { ++progress_reports; };
    planner.solution_reporter = [&](const FreeSpace&, const rkh::hip_motion_graph<Point>&, double) { ++solution_reports; };
    P2PQuery query{start, goal, std::size_t(prm.max_results), {}, 0};
    bool stale_rng_throws = false;
    try {  // draws were made from the global generator since it was seeded (the loop above used its own engine: seed now)
      rkh::global_rng_seed(prm.seed);
      (void)space->random_point();
      planner.solve_planning_query(query);
    } catch (const rkh::unsupported_error&) {
      stale_rng_throws = true;
    }
    rkh::global_rng_seed(prm.seed);
    planner.solve_planning_query(query);
    const rkh::hip_motion_graph<Point>& mg = planner.motion_graph();
    // afterwards the global generator stands where the sequential planner would have left it
    std::mt19937 expect_eng(prm.seed);
    expect_eng.discard(planner.last_stats.iterations * D);
    const bool rng_advanced = (rkh::get_global_rng() == expect_eng) && (eng == expect_eng);

    // ---- 3. same tree?
    bool same = (num_vertices(mg) == num_vertices(g)) && (planner.last_stats.iterations == iterations);
    double max_diff = 0.0;
    for (std::size_t v = 0; same && v < num_vertices(g); ++v) {
      same = same && (v == 0 ? mg[v].predecessor == std::size_t(-1) : mg[v].predecessor == g.v[v].parent);
      for (std::size_t i = 0; i < D; ++i) max_diff = std::fmax(max_diff, std::fabs(mg[v].position[i] - g.v[v].position[i]));
    }
    same = same && (max_diff == 0.0);  // the same kernels computed both
    const bool reports_ok = progress_reports == (num_vertices(mg) - 1) / 25 && solution_reports == query.solutions.size() &&
                            planner.last_stats.num_solutions == query.solutions.size() &&
                            (query.solutions.empty() || query.get_best_solution_distance() == planner.last_stats.best_cost);
    // the Topology / MetricSpace concept surface (metric_space_concept.hpp:86-223)
    const Point mid = space->get_super_space().origin();
    const Point dp = space->difference(goal, start);
    const bool topo_ok = get(rkh::distance_metric, space->get_super_space())(start, goal, space->get_super_space()) ==
                             space->get_super_space().distance(start, goal) &&
                         space->norm(dp) == space->get_super_space().distance(goal, start) && mid.size() == D &&
                         space->get_super_space().adjust(start, dp) == goal;
    // errors arrive as exceptions: a state of the wrong size is a std::range_error like in kte_nl_system.hpp:181-188
    bool threw = false;
    try {
      rkh::check(rkh_min_distance(scene.get(), nullptr, 1, nullptr));
    } catch (const std::range_error&) {
      threw = true;
    }
    std::printf("{\"vertices\": %zu, \"iterations\": %lu, \"planner_vertices\": %llu, \"planner_iterations\": %llu, "
                "\"same_tree\": %s, \"max_abs_diff\": %.3g, \"knn_sorted\": %s, \"proxy_disagree\": %lu, \"bad_arg_throws\": %s, "
                "\"reports_ok\": %s, \"topo_ok\": %s, \"stale_rng_throws\": %s, \"rng_advanced\": %s, \"rrt_solutions\": %zu, "
                "\"parents\": [",
                num_vertices(g), iterations, (unsigned long long)planner.last_stats.num_vertices,
                (unsigned long long)planner.last_stats.iterations, same ? "true" : "false", max_diff,
                knn_sorted ? "true" : "false", proxy_disagree, threw ? "true" : "false", reports_ok ? "true" : "false",
                topo_ok ? "true" : "false", stale_rng_throws ? "true" : "false", rng_advanced ? "true" : "false",
                query.solutions.size());
    for (std::size_t v = 1; v < num_vertices(g); ++v) std::printf("%s%zu", v > 1 ? ", " : "", g.v[v].parent);
    std::printf("]");

    // ---- 4. the graph planners over a quasi-static scene, each through its adaptor and a query object
    {
      FILE* f2 = std::fopen(argv[2], "rb");
      if (!f2) return 2;
      int32_t n_ops2 = 0, n_shapes2 = 0;
      rkh_chain_base base2;
      rkh_qs_space qs;
      rkh_rrt_params qp;
      double sampling_radius = 1.0;
      std::vector<rkh_kte_op> ops2;
      std::vector<rkh_shape> shapes2;
      bool ok2 = read_pod(f2, &n_ops2);
      ops2.resize(n_ops2);
      ok2 = ok2 && read_pod(f2, ops2.data(), n_ops2) && read_pod(f2, &base2) && read_pod(f2, &n_shapes2);
      shapes2.resize(n_shapes2);
      ok2 = ok2 && read_pod(f2, shapes2.data(), n_shapes2) && read_pod(f2, &qs) && read_pod(f2, &qp) &&
            read_pod(f2, &sampling_radius);
      std::fclose(f2);
      if (!ok2) return 2;
      auto scene2 = rkh::make_scene(ctx, ops2.data(), n_ops2, base2, shapes2.data(), n_shapes2);
      typedef rkh::manip_quasi_static_free_space<Point> QsSpace;
      auto qspace = std::make_shared<QsSpace>(scene2, qs);
      const std::size_t n = std::size_t(qs.n_dof);
      const Point qstart(qp.start, qp.start + n), qgoal(qp.goal, qp.goal + n);
      // the topology itself: free start / goal, a walk that ends early is "unreachable"
      const bool qs_topo = qspace->is_free(qstart) && qspace->is_free(qgoal) &&
                           (qspace->distance(qstart, qgoal) == qspace->get_super_space().distance(qstart, qgoal) ||
                            std::isinf(qspace->distance(qstart, qgoal)));
      std::printf(", \"qs_topo_ok\": %s", qs_topo ? "true" : "false");
      {  // RRT (unidirectional) over the quasi-static space
        rkh::hip_rrt_planner<QsSpace> pl(qspace, qp.max_vertices, 0, qp.steer_tol, qp.conn_tol);
        P2PQuery q{qstart, qgoal, std::size_t(qp.max_results), {}, 0};
        rkh::global_rng_seed(qp.seed);
        pl.solve_planning_query(q);
        std::vector<std::size_t> par;
        for (std::size_t v = 0; v < num_vertices(pl.motion_graph()); ++v) par.push_back(pl.motion_graph()[v].predecessor);
        std::printf(", \"qs_rrt\": {\"vertices\": %zu, \"iterations\": %llu, \"solutions\": %zu, \"best\": %.17g, ",
                    par.size(), (unsigned long long)pl.last_stats.iterations, q.solutions.size(),
                    q.solutions.empty() ? -1.0 : q.get_best_solution_distance());
        print_u32_list("parent", par);
        std::printf("}");
      }
      {  // RRT*
        rkh::hip_rrtstar_planner<QsSpace> pl(qspace, qp.max_vertices, 100, qp.steer_tol, qp.conn_tol);
        unsigned long reports = 0;
        pl.progress_reporter = [&](const QsSpace&, const rkh::hip_motion_graph<Point>&) { ++reports; };
        P2PQuery q{qstart, qgoal, std::size_t(qp.max_results), {}, 0};
        rkh::global_rng_seed(qp.seed);
        pl.solve_planning_query(q);
        std::vector<std::size_t> pred;
        for (std::size_t v = 0; v < num_vertices(pl.motion_graph()); ++v) pred.push_back(pl.motion_graph()[v].predecessor);
        std::printf(", \"qs_rrtstar\": {\"vertices\": %zu, \"rewires\": %llu, \"solutions\": %zu, \"best\": %.17g, "
                    "\"reports\": %lu, \"path_len\": %zu, ",
                    pred.size(), (unsigned long long)pl.last_stats.rewires, q.solutions.size(),
                    q.solutions.empty() ? -1.0 : q.get_best_solution_distance(), reports,
                    q.solutions.empty() ? std::size_t(0) : q.solutions.back().second.size());
        print_u32_list("pred", pred);
        std::printf("}");
      }
      {  // PRM: the roadmap; no solution is registered (as in the reference)
        rkh::hip_prm_planner<QsSpace> pl(qspace, qp.max_vertices, 0, qp.steer_tol, qp.conn_tol, sampling_radius);
        P2PQuery q{qstart, qgoal, std::size_t(qp.max_results), {}, 0};
        rkh::global_rng_seed(qp.seed);
        pl.solve_planning_query(q);
        double wsum = 0.0;
        for (double w : pl.motion_graph().edge_weight) wsum += w;
        std::printf(", \"qs_prm\": {\"vertices\": %zu, \"edges\": %zu, \"components\": %llu, \"weight_sum\": %.17g, "
                    "\"register_calls\": %zu}",
                    num_vertices(pl.motion_graph()), pl.motion_graph().edges.size(),
                    (unsigned long long)pl.last_stats.num_components, wsum, q.register_calls);
      }
      {  // bidirectional RRT
        rkh::hip_birrt_planner<QsSpace> pl(qspace, qp.max_vertices, 0, qp.steer_tol, qp.conn_tol);
        P2PQuery q{qstart, qgoal, std::size_t(qp.max_results), {}, 0};
        rkh::global_rng_seed(qp.seed);
        pl.solve_planning_query(q);
        std::printf(", \"qs_birrt\": {\"vertices_1\": %zu, \"vertices_2\": %zu, \"solutions\": %llu, \"best\": %.17g, "
                    "\"query_best\": %.17g}",
                    num_vertices(pl.graph1()), num_vertices(pl.graph2()), (unsigned long long)pl.last_stats.num_solutions,
                    pl.last_stats.num_solutions ? pl.last_stats.best_cost : -1.0,
                    q.solutions.empty() ? -1.0 : q.get_best_solution_distance());
      }
    }
    std::printf("}\n");
    return (same && knn_sorted && proxy_disagree == 0 && threw && reports_ok && topo_ok && stale_rng_throws && rng_advanced) ? 0
                                                                                                                        : 1;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "abi_smoke: %s\n", e.what());
    return 3;
  }
}
