// rkh_adaptors.hpp -- C++ adaptors that model ReaK's planning concepts on top of the C-ABI (rkh.h).
//
// ReaK wires its sampling-based planners through C++ template concepts, not through an ABI.  The
// classes below are the sockets' drop-ins: same call signatures, same argument meaning, errors re-thrown as C++
// exceptions (no status codes above this header).  They are Boost-free on purpose -- the only things they ask of the
// surrounding code are the free functions BGL and ReaK already provide and that any graph / point type can supply:
//
//     vertex(i, g)                 i-th vertex descriptor in vertices(g) order        (boost/graph/graph_traits.hpp)
//     get(position_map, v)         the vertex's point; p[i], p.size()                 (boost/property_map)
//     num_vertices(g)
//
// so the same header serves the ReaK tree (Graph = boost::adjacency_list_BC<...>, Point = vect_n<double>) and the
// self-contained check tests/cpp/abi_smoke.cpp (Graph = a vector of vertices).  Paths `R/...` = src/ReaK/... of the
// reference tree.
//
//   socket                                      reference                                               adaptor
//   NNFinder (1-NN, k-NN, pred/succ k-NN)       R/ctrl/path_planning/topological_search.hpp:585-634,773  hip_linear_neighbor_search
//   KNN synchro                                 R/ctrl/path_planning/any_knn_synchro.hpp:69-84            hip_knn_synchro
//   steerable C_free topology                   R/ctrl/path_planning/steerable_space_concept.hpp:78-93    kte_dynamic_free_space
//   proximity (findMinimumDistance)             R/geometry/proximity/proximity_finder_3D.hpp:49-82        hip_proxy_query_pair
//   planner entry (solve_planning_query)        R/ctrl/path_planning/motion_planner_base.hpp:102          hip_rrt_planner
#ifndef RKH_ADAPTORS_HPP
#define RKH_ADAPTORS_HPP

#include <cmath>
#include <cstddef>
#include <cstdint>
#include <iterator>
#include <limits>
#include <memory>
#include <random>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "rkh.h"

namespace rkh {

// ---- errors: no status code crosses this header -------------------------------------------------------------------
struct singularity_error : std::runtime_error {  // R/core/lin_alg/mat_num_exceptions.hpp (thrown by linsolve_Cholesky)
  explicit singularity_error(const std::string& m) : std::runtime_error(m) {}
};
struct unsupported_error : std::runtime_error {  // the scene is outside what the HIP kernels cover: keep the CPU path
  explicit unsupported_error(const std::string& m) : std::runtime_error(m) {}
};
inline void check(rkh_status s) {
  switch (s) {
    case RKH_OK: return;
    case RKH_ERR_SINGULAR: throw singularity_error(rkh_last_error());
    case RKH_ERR_BAD_ARG: throw std::range_error(rkh_last_error());  // kte_nl_system.hpp:181-188
    case RKH_ERR_OOM: throw std::bad_alloc();
    case RKH_ERR_UNSUPPORTED: throw unsupported_error(rkh_last_error());
    default: throw std::runtime_error(rkh_last_error());
  }
}

// ---- handles (ReaK::shared_ptr ownership: copies of a functor share the device object) ---------------------------
inline std::shared_ptr<rkh_ctx> make_context(int device = 0) {
  rkh_ctx* c = nullptr;
  check(rkh_ctx_create(device, &c));
  return std::shared_ptr<rkh_ctx>(c, [](rkh_ctx* p) { (void)rkh_ctx_destroy(p); });
}
inline std::shared_ptr<rkh_nn> make_nn(const std::shared_ptr<rkh_ctx>& ctx, int dims, uint64_t capacity) {
  rkh_nn* n = nullptr;
  check(rkh_nn_create(ctx.get(), dims, capacity, &n));
  return std::shared_ptr<rkh_nn>(n, [ctx](rkh_nn* p) { (void)rkh_nn_destroy(p); });
}
inline std::shared_ptr<rkh_scene> make_scene(const std::shared_ptr<rkh_ctx>& ctx, const rkh_kte_op* prog, int n_ops,
                                             const rkh_chain_base& base, const rkh_shape* shapes, int n_shapes) {
  rkh_scene* s = nullptr;
  check(rkh_scene_create(ctx.get(), prog, n_ops, &base, shapes, n_shapes, &s));
  return std::shared_ptr<rkh_scene>(s, [ctx](rkh_scene* p) { (void)rkh_scene_destroy(p); });
}

// ---- NNFinder: linear_neighbor_search<Graph> -----------------------------------------------------------------------
// Copyable functor (the planners pass it by value).  Vertex ids on the device are insertion indices = positions in
// vertices(g) order, which is what vecBC / poolBC storage gives while no vertex is removed.
template <typename Graph>
struct hip_linear_neighbor_search {
  std::shared_ptr<rkh_nn> nn;
  hip_linear_neighbor_search() {}
  explicit hip_linear_neighbor_search(const std::shared_ptr<rkh_nn>& h) : nn(h) {}

  // Vertex operator()(const Point& p, Graph& g, const Topology& space, PositionMap position) const
  // (topological_search.hpp:585-598 -> min_dist_linear_search :95-118; "first minimum wins")
  template <typename Point, typename Topology, typename PositionMap>
  auto operator()(const Point& p, Graph& g, const Topology&, PositionMap) const -> decltype(vertex(std::size_t(0), g)) {
    uint32_t idx = 0;
    double d = 0.0;
    check(rkh_nn_query1(nn.get(), &p[0], 1, &idx, &d));
    return vertex(std::size_t(idx), g);
  }
  // OutIt operator()(p, out, g, space, position, max_neighbors = 1, radius = inf) const
  // (topological_search.hpp:619-634 -> :244-274; candidates need d < radius strictly; nearest first)
  template <typename Point, typename OutputIterator, typename Topology, typename PositionMap>
  OutputIterator operator()(const Point& p, OutputIterator out, Graph& g, const Topology&, PositionMap,
                            std::size_t max_neighbors = 1,
                            double radius = std::numeric_limits<double>::infinity()) const {
    std::vector<uint32_t> idx(max_neighbors);
    std::vector<double> d(max_neighbors);
    uint32_t cnt = 0;
    check(rkh_nn_queryk(nn.get(), &p[0], 1, uint32_t(max_neighbors), radius, idx.data(), d.data(), &cnt));
    for (uint32_t i = 0; i < cnt; ++i) *(out++) = vertex(std::size_t(idx[i]), g);
    return out;
  }
  // directed graphs: (p, pred_out, succ_out, g, space, position, max_neighbors, radius)
  // (topological_search.hpp:773-, :337-378).  For a symmetric metric both lists are the same neighbours.
  template <typename Point, typename OutputIterator, typename Topology, typename PositionMap>
  std::pair<OutputIterator, OutputIterator> operator()(const Point& p, OutputIterator pred_out, OutputIterator succ_out,
                                                       Graph& g, const Topology& s, PositionMap pm,
                                                       std::size_t max_neighbors = 1,
                                                       double radius = std::numeric_limits<double>::infinity()) const {
    std::vector<decltype(vertex(std::size_t(0), g))> nb;
    (*this)(p, std::back_inserter(nb), g, s, pm, max_neighbors, radius);
    for (const auto& v : nb) {
      *(pred_out++) = v;
      *(succ_out++) = v;
    }
    return std::make_pair(pred_out, succ_out);
  }
};

// ---- KNN synchro: any_knn_synchro::{added_vertex, removed_vertex} (any_knn_synchro.hpp:69-84) --------------------
// Called from planning_visitor_base::vertex_added / vertex_to_be_removed (planning_visitors.hpp:166-168,213-216).
template <typename Graph, typename PositionMap>
struct hip_knn_synchro {
  std::shared_ptr<rkh_nn> nn;
  PositionMap position;
  hip_knn_synchro(const std::shared_ptr<rkh_nn>& h, PositionMap pm) : nn(h), position(pm) {}
  template <typename Vertex>
  void added_vertex(Vertex u, Graph& g) const {
    (void)g;
    const auto& p = get(position, u);
    check(rkh_nn_append(nn.get(), &p[0], 1));
  }
  // The device store is append-only with tombstones: row numbers = order of the added_vertex calls, so the graph's
  // vertex descriptor must convert to that number (vecS / pooled vertex lists: the descriptor is the index; a graph
  // that re-uses the holes of removed vertices has to map descriptors to insertion numbers in `index_of`).
  template <typename Vertex>
  void removed_vertex(Vertex u, Graph& g) const {
    (void)g;
    check(rkh_nn_remove(nn.get(), static_cast<uint64_t>(u)));
  }
};

// ---- super-space: hyperbox_topology< vect_n<double> > with the euclidean metric ---------------------------------
// (R/ctrl/topologies/hyperbox_topology.hpp:97-103,178-189; vect_distance_metrics.hpp:113-137).  random_point draws
// from the engine it is given -- ReaK uses the global mt19937 (global_rng.hpp:44-54), D draws of
// uniform_01<mt19937&, double> = eng() * 2^-32 per point.
template <typename Point>
struct hyperbox_super_space {
  Point lower, upper;
  typedef Point point_type;
  template <typename Engine>
  Point random_point(Engine& eng) const {
    Point p(lower);
    for (std::size_t i = 0; i < lower.size(); ++i) {
      double u;
      do {
        u = double(eng()) * (1.0 / 4294967296.0);
      } while (!(u < 1.0));
      p[i] = lower[i] + u * (upper[i] - lower[i]);
    }
    return p;
  }
  double distance(const Point& a, const Point& b) const {  // euclidean_distance_metric: left-to-right sum, then sqrt
    double s = 0.0;
    for (std::size_t i = 0; i < a.size(); ++i) {
      const double d = a[i] - b[i];
      s += d * d;
    }
    return std::sqrt(s);
  }
  bool is_in_bounds(const Point& p) const {
    for (std::size_t i = 0; i < p.size(); ++i) {
      if (lower[i] < upper[i]) {
        if ((p[i] < lower[i]) || (p[i] > upper[i])) return false;
      } else {
        if ((p[i] > lower[i]) || (p[i] < upper[i])) return false;
      }
    }
    return true;
  }
};

// ---- steerable C_free topology ---------------------------------------------------------------------------------------
// Models SubSpaceConcept (get_super_space), MetricSpace (distance, move_position_toward), is_free, and
// SteerableSpaceConcept: std::pair<point_type, steer_record_type> steer_position_toward(a, fraction, b)
// (steerable_space_concept.hpp:78-93; the loop shape of examples/misc/MEAQR_topology.hpp:503-565 with the PD law of
// rkh_dyn_space).  Points are value types holding 2 n_dof doubles (q, qd interleaved, kte_nl_system.hpp:190-193).
template <typename Point = std::vector<double> >
class kte_dynamic_free_space {
 public:
  typedef Point point_type;
  typedef Point point_difference_type;
  typedef std::vector<Point> steer_record_type;
  typedef hyperbox_super_space<Point> super_space_type;

  kte_dynamic_free_space(const std::shared_ptr<rkh_scene>& scene, const rkh_dyn_space& sp) : m_scene(scene), m_sp(sp) {
    const std::size_t D = 2 * std::size_t(sp.n_dof);
    m_super.lower = Point(D);
    m_super.upper = Point(D);
    for (std::size_t i = 0; i < D; ++i) {
      m_super.lower[i] = sp.lower[i];
      m_super.upper[i] = sp.upper[i];
    }
  }
  const super_space_type& get_super_space() const { return m_super; }
  const rkh_dyn_space& dyn_space() const { return m_sp; }
  const std::shared_ptr<rkh_scene>& scene() const { return m_scene; }

  // manip_free_workspace.hpp:79-99,154-156: bounds, then "any proxy pair closer than 0 -> colliding"
  bool is_free(const Point& p) const {
    if (!m_super.is_in_bounds(p)) return false;
    double d = 0.0;
    check(rkh_min_distance(m_scene.get(), &p[0], 1, &d));
    return !(d < 0.0);
  }
  std::pair<Point, steer_record_type> steer_position_toward(const Point& a, double fraction, const Point& b) const {
    const std::size_t D = a.size();
    Point out(a);
    uint32_t n_free = 0;
    std::vector<double> rec(std::size_t(m_sp.steps_per_edge + 1) * D);
    check(rkh_propagate(m_scene.get(), &m_sp, &a[0], &b[0], 1, fraction, &out[0], &n_free, rec.data()));
    steer_record_type r;
    for (uint32_t k = 0; k <= n_free; ++k) {
      Point x(a);
      for (std::size_t i = 0; i < D; ++i) x[i] = rec[k * D + i];
      r.push_back(x);
    }
    return std::make_pair(out, r);
  }
  Point move_position_toward(const Point& a, double fraction, const Point& b) const {
    return steer_position_toward(a, fraction, b).first;
  }
  // MEAQR_topology_with_CD::distance (MEAQR_topology.hpp:995-1003): infinite unless the steer comes within 5 %
  double distance(const Point& a, const Point& b) const {
    const Point r = move_position_toward(a, 1.0, b);
    const double dab = m_super.distance(a, b);
    return (dab * 0.05 > m_super.distance(r, b)) ? dab : std::numeric_limits<double>::infinity();
  }
  template <typename Engine>
  Point random_point(Engine& eng) const {  // default_random_sampler on the super-space (default_random_sampler.hpp:64-66)
    return m_super.random_point(eng);
  }

 private:
  std::shared_ptr<rkh_scene> m_scene;
  rkh_dyn_space m_sp;
  super_space_type m_super;
};
// trait tags the planners dispatch on (is_steerable_space / is_metric_space / ... are boost::mpl bools in ReaK;
// a maintainer specialises them to true_ for this class, manip_free_workspace.hpp:307-318)
template <typename T>
struct is_steerable_space_tag {
  static const bool value = false;
};
template <typename P>
struct is_steerable_space_tag<kte_dynamic_free_space<P> > {
  static const bool value = true;
};

// ---- proximity socket: proxy_query_pair_3D::findMinimumDistance ----------------------------------------------------
// manip_dk_proxy_env_impl::is_free reads `findMinimumDistance()->getLastResult().mDistance`
// (manip_free_workspace.hpp:85-95; proximity_finder_3D.hpp:49-82).  The device applies the state to the chain
// (manip_direct_kin_map::apply_to_model) and evaluates every finder of the pair list in one call.
struct proximity_record {  // proximity_record_3D (proximity_record_3D.hpp:47-56); points are not produced
  double mDistance = std::numeric_limits<double>::infinity();
};
class hip_proximity_finder {
 public:
  explicit hip_proximity_finder(double d) { m_last.mDistance = d; }
  const proximity_record& getLastResult() const { return m_last; }

 private:
  proximity_record m_last;
};
class hip_proxy_query_pair {
 public:
  explicit hip_proxy_query_pair(const std::shared_ptr<rkh_scene>& scene) : m_scene(scene) {}
  // the joint state the models are at (apply_to_model writes it into the KTE chain in ReaK)
  template <typename Point>
  void apply_to_model(const Point& p) {
    m_state.assign(&p[0], &p[0] + p.size());
  }
  std::shared_ptr<hip_proximity_finder> findMinimumDistance() const {
    if (rkh_scene_num_pairs(m_scene.get()) == 0) return std::shared_ptr<hip_proximity_finder>();  // empty finder list
    double d = 0.0;
    check(rkh_min_distance(m_scene.get(), m_state.data(), 1, &d));
    return std::make_shared<hip_proximity_finder>(d);
  }

 private:
  std::shared_ptr<rkh_scene> m_scene;
  std::vector<double> m_state;
};

// ---- planner entry: sample_based_planner<FreeSpace>::solve_planning_query ------------------------------------------
// (motion_planner_base.hpp:102, options :400-422; rrt_planner with UNIDIRECTIONAL_PLANNING | LINEAR_SEARCH_KNN,
// rrt_path_planner.tpp:66-145).  The batched device driver grows exactly the tree of the sequential generate_rrt on
// the query's seed; the adaptor rebuilds the caller's motion graph from it and reports the solutions.
// Query concept used here: get_start_position(), get_goal_position(), max_num_results, register_solution(cost, path)
// -- what planning_query / path_planning_p2p_query offer (p2p_planning_query.hpp:74-229).
template <typename FreeSpace>
class hip_rrt_planner {
 public:
  typedef typename FreeSpace::point_type point_type;
  hip_rrt_planner(const std::shared_ptr<FreeSpace>& space, std::size_t max_vertex_count, double steer_progress_tol = 0.1,
                  double connection_tol = 0.05)
      : m_space(space), m_max_vertex_count(max_vertex_count), m_steer_tol(steer_progress_tol), m_conn_tol(connection_tol) {}

  struct result {
    rkh_planner_stats stats;
    std::vector<point_type> positions;  // vertices in insertion order (vertex 0 = the query's start)
    std::vector<uint32_t> parent;       // 0xFFFFFFFF for the root
    std::vector<uint32_t> solution;     // vertex ids of the best registered solution, start first (empty: none)
    double solution_cost = std::numeric_limits<double>::infinity();
  };

  // seed = what the caller passed to get_global_rng().seed() (global_rng.hpp:44-54)
  result solve_planning_query(const point_type& start, const point_type& goal, uint32_t seed,
                              uint32_t max_num_results = 1u << 30) const {
    rkh_rrt_params prm = rkh_rrt_params();
    prm.seed = seed;
    prm.max_vertices = uint32_t(m_max_vertex_count);
    prm.max_results = max_num_results;
    prm.steer_tol = m_steer_tol;
    prm.conn_tol = m_conn_tol;
    const std::size_t D = start.size();
    for (std::size_t i = 0; i < D; ++i) {
      prm.start[i] = start[i];
      prm.goal[i] = goal[i];
    }
    rkh_planner* raw = nullptr;
    check(rkh_planner_create(m_space->scene().get(), &m_space->dyn_space(), &prm, &raw));
    std::shared_ptr<rkh_planner> pl(raw, [](rkh_planner* p) { (void)rkh_planner_destroy(p); });
    result r;
    check(rkh_planner_solve(pl.get(), &r.stats));
    const std::size_t nv = std::size_t(r.stats.num_vertices);
    std::vector<double> pos(nv * D);
    r.parent.resize(nv);
    check(rkh_planner_get_tree(pl.get(), 0, pos.data(), r.parent.data(), nullptr, nullptr, nullptr));
    r.positions.assign(nv, start);
    for (std::size_t v = 0; v < nv; ++v)
      for (std::size_t i = 0; i < D; ++i) r.positions[v][i] = pos[v * D + i];
    uint32_t n_path = 0;
    check(rkh_planner_get_solution(pl.get(), 0, nullptr, 0, &n_path, &r.solution_cost));
    if (n_path) {
      r.solution.resize(n_path);
      check(rkh_planner_get_solution(pl.get(), 0, r.solution.data(), n_path, &n_path, &r.solution_cost));
    }
    return r;
  }

 private:
  std::shared_ptr<FreeSpace> m_space;
  std::size_t m_max_vertex_count;
  double m_steer_tol, m_conn_tol;
};

}  // namespace rkh
#endif
