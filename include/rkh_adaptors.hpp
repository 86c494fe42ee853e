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
//   planner entry (solve_planning_query)        R/ctrl/path_planning/motion_planner_base.hpp:102          hip_rrt_planner, hip_rrtstar_planner,
//                                                                                                           hip_prm_planner, hip_birrt_planner
//   quasi-static C_free topology                R/ctrl/topologies/manip_free_workspace.hpp:113-300        manip_quasi_static_free_space
//   the global generator                        R/core/base/global_rng.hpp:44-54                          get_global_rng / global_rng_seed
#ifndef RKH_ADAPTORS_HPP
#define RKH_ADAPTORS_HPP

#include <cmath>
#include <cstddef>
#include <algorithm>
#include <cstdint>
#include <functional>
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
    // an empty (or fully removed) store answers 0xFFFFFFFF: min_dist_linear_search then returns its initial
    // `result`, a default-constructed descriptor = boost::graph_traits<Graph>::null_vertex() (:98-100); the graph
    // supplies it through the free function null_vertex(g)
    if (idx == 0xFFFFFFFFu) return null_vertex(g);
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
  // vertex descriptor must convert to that number.  vecBC vertex lists: the descriptor IS that index while nothing is
  // removed.  poolBC storage (rrtstar_path_planner.tpp:102-103) re-uses the hole of a removed vertex for the next
  // add_vertex (lazy_connector.hpp:359-364 removes unconnectable vertices; branch and bound prunes): from the first
  // removal on, descriptor != insertion number.  The mapping a maintainer keeps beside this synchro is
  //     insertion_number[descriptor]  (set in added_vertex: = count of added_vertex calls so far; read in removed_vertex)
  //     descriptor_of[insertion_number]  (for the NNFinder: vertex(idx, g) above becomes descriptor_of[idx])
  // -- `index_of` below is that hook (identity by default).  BGL-Extra is not part of the reference tree, so the hole
  // order itself is "parity unpinned" (DESIGN.md section 2): results are identical up to this renaming of vertices.
  std::function<uint64_t(uint64_t)> index_of;
  template <typename Vertex>
  void removed_vertex(Vertex u, Graph& g) const {
    (void)g;
    const uint64_t d = static_cast<uint64_t>(u);
    check(rkh_nn_remove(nn.get(), index_of ? index_of(d) : d));
  }
};

// ---- the global generator (R/core/base/global_rng.hpp:44-54) ----------------------------------------------------------
// ReaK's samplers draw from one process-global boost::mt19937 (bit-identical to std::mt19937).  The device planners
// generate their sample stream from a SEED (rkh_rrt_params::seed), so the adaptors need to know the seed the engine was
// last given: seed it through global_rng_seed(s) (= get_global_rng().seed(s)); a planner entry checks that the engine
// still is mt19937(s) -- nothing was drawn since -- and, after planning, advances it by the draws the planner consumed,
// so that host code drawing afterwards continues the same stream the sequential planner would have left.
inline std::mt19937& get_global_rng() {
  static std::mt19937 instance;  // default seed 5489, like the reference's never-seeded engine
  return instance;
}
inline uint32_t& global_rng_last_seed() {
  static uint32_t s = std::mt19937::default_seed;
  return s;
}
inline void global_rng_seed(uint32_t s) {
  get_global_rng().seed(s);
  global_rng_last_seed() = s;
}
inline uint32_t global_rng_fresh_seed() {  // the seed the device planners take; throws if draws were made since it was set
  if (!(get_global_rng() == std::mt19937(global_rng_last_seed())))
    throw unsupported_error("the device planners start from a freshly seeded global generator: call rkh::global_rng_seed(s) "
                            "right before solve_planning_query");
  return global_rng_last_seed();
}

// ---- super-space: hyperbox_topology< vect_n<double> > with the euclidean metric ---------------------------------
// (R/ctrl/topologies/hyperbox_topology.hpp:97-103,178-189; vect_distance_metrics.hpp:113-137; the Topology / MetricSpace
// / PointDistribution concepts of metric_space_concept.hpp:86-223).  random_point() draws from the global mt19937,
// D draws of uniform_01<mt19937&, double> = eng() * 2^-32 per point.
struct distance_metric_t {};  // tag of get(distance_metric, space) (metric_space_concept.hpp:60-62)
static const distance_metric_t distance_metric = distance_metric_t();
template <typename Space>
struct bound_distance_metric {  // what get(distance_metric, space) returns: d(a, b, space)
  template <typename Point>
  double operator()(const Point& a, const Point& b, const Space& s) const { return s.distance(a, b); }
  template <typename Point>
  double operator()(const Point& dp, const Space& s) const { return s.norm(dp); }
};
template <typename Point>
struct hyperbox_super_space {
  Point lower, upper;
  typedef Point point_type;
  typedef Point point_difference_type;
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
  Point random_point() const { return random_point(get_global_rng()); }
  double norm(const Point& dp) const {  // euclidean_distance_metric: left-to-right sum, then sqrt
    double s = 0.0;
    for (std::size_t i = 0; i < dp.size(); ++i) s += dp[i] * dp[i];
    return std::sqrt(s);
  }
  double distance(const Point& a, const Point& b) const {
    double s = 0.0;
    for (std::size_t i = 0; i < a.size(); ++i) {
      const double d = a[i] - b[i];
      s += d * d;
    }
    return std::sqrt(s);
  }
  Point difference(const Point& a, const Point& b) const {  // a - b (vector_topology.hpp)
    Point r(a);
    for (std::size_t i = 0; i < a.size(); ++i) r[i] = a[i] - b[i];
    return r;
  }
  Point origin() const {  // hyperbox_topology::origin: the centre of the box
    Point r(lower);
    for (std::size_t i = 0; i < lower.size(); ++i) r[i] = (lower[i] + upper[i]) * 0.5;
    return r;
  }
  Point adjust(const Point& a, const Point& dp) const {
    Point r(a);
    for (std::size_t i = 0; i < a.size(); ++i) r[i] = a[i] + dp[i];
    return r;
  }
  Point move_position_toward(const Point& a, double fraction, const Point& b) const {  // a + (b - a) * fraction
    Point r(a);
    for (std::size_t i = 0; i < a.size(); ++i) r[i] = a[i] + (b[i] - a[i]) * fraction;
    return r;
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
template <typename Point>
bound_distance_metric<hyperbox_super_space<Point> > get(distance_metric_t, const hyperbox_super_space<Point>&) {
  return bound_distance_metric<hyperbox_super_space<Point> >();
}

// ---- steerable C_free topology ---------------------------------------------------------------------------------------
// Models SubSpaceConcept (get_super_space), Topology (difference / origin / adjust), MetricSpace (distance, get(distance_
// metric, .), move_position_toward), PointDistribution (random_point), is_free, and SteerableSpaceConcept:
// std::pair<point_type, steer_record_type> steer_position_toward(a, fraction, b) (steerable_space_concept.hpp:78-93; the
// loop shape of examples/misc/MEAQR_topology.hpp:503-565 with the PD law of rkh_dyn_space).  Points are value types
// holding 2 n_dof doubles (q, qd interleaved, kte_nl_system.hpp:190-193).
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
  std::size_t dimensions() const { return 2 * std::size_t(m_sp.n_dof); }

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
  double norm(const Point& dp) const { return m_super.norm(dp); }
  // Topology concept on the sub-space: forwarded to the super-space, adjust through the free-space motion like the
  // reference's sub-spaces do (no_obstacle_space.hpp:166-168: move_position_toward(p, 1.0, super.adjust(p, dp)))
  Point difference(const Point& a, const Point& b) const { return m_super.difference(a, b); }
  Point origin() const { return m_super.origin(); }
  Point adjust(const Point& a, const Point& dp) const { return move_position_toward(a, 1.0, m_super.adjust(a, dp)); }
  template <typename Engine>
  Point random_point(Engine& eng) const {  // default_random_sampler on the super-space (default_random_sampler.hpp:64-66)
    return m_super.random_point(eng);
  }
  Point random_point() const { return m_super.random_point(get_global_rng()); }

 private:
  std::shared_ptr<rkh_scene> m_scene;
  rkh_dyn_space m_sp;
  super_space_type m_super;
};
template <typename P>
bound_distance_metric<kte_dynamic_free_space<P> > get(distance_metric_t, const kte_dynamic_free_space<P>&) {
  return bound_distance_metric<kte_dynamic_free_space<P> >();
}

// ---- quasi-static C_free topology: manip_quasi_static_env (manip_free_workspace.hpp:113-300) -------------------------
// Points are joint positions (n_dof doubles); move_position_toward walks the straight line in min_interval steps and
// returns the last free point (interp_topo_move_position_toward_pred, interpolated_topologies.hpp:137-163), distance is
// the length if the walk arrives, +inf otherwise (:193-199).
template <typename Point = std::vector<double> >
class manip_quasi_static_free_space {
 public:
  typedef Point point_type;
  typedef Point point_difference_type;
  typedef hyperbox_super_space<Point> super_space_type;
  manip_quasi_static_free_space(const std::shared_ptr<rkh_scene>& scene, const rkh_qs_space& sp) : m_scene(scene), m_sp(sp) {
    const std::size_t n = std::size_t(sp.n_dof);
    m_super.lower = Point(n);
    m_super.upper = Point(n);
    for (std::size_t i = 0; i < n; ++i) {
      m_super.lower[i] = sp.lower[i];
      m_super.upper[i] = sp.upper[i];
    }
  }
  const super_space_type& get_super_space() const { return m_super; }
  const rkh_qs_space& qs_space() const { return m_sp; }
  const std::shared_ptr<rkh_scene>& scene() const { return m_scene; }
  std::size_t dimensions() const { return std::size_t(m_sp.n_dof); }
  bool is_free(const Point& p) const {
    if (!m_super.is_in_bounds(p)) return false;
    std::vector<double> x(2 * p.size(), 0.0);  // rkh_min_distance takes states (q, qd interleaved)
    for (std::size_t i = 0; i < p.size(); ++i) x[2 * i] = p[i];
    double d = 0.0;
    check(rkh_min_distance(m_scene.get(), x.data(), 1, &d));
    return !(d < 0.0);
  }
  Point move_position_toward(const Point& a, double fraction, const Point& b) const {
    Point out(a);
    uint32_t n_checked = 0;
    check(rkh_edge_check(m_scene.get(), m_sp.lower, m_sp.upper, m_sp.min_interval, &a[0], &b[0], 1, fraction, &out[0],
                         &n_checked));
    return out;
  }
  double distance(const Point& a, const Point& b) const {  // interp_topo_get_distance_pred
    const Point r = move_position_toward(a, 1.0, b);
    return (m_super.distance(r, b) < std::numeric_limits<double>::epsilon()) ? m_super.distance(a, b)
                                                                            : std::numeric_limits<double>::infinity();
  }
  double norm(const Point& dp) const { return m_super.norm(dp); }
  Point difference(const Point& a, const Point& b) const { return m_super.difference(a, b); }
  Point origin() const { return m_super.origin(); }
  Point adjust(const Point& a, const Point& dp) const { return move_position_toward(a, 1.0, m_super.adjust(a, dp)); }
  template <typename Engine>
  Point random_point(Engine& eng) const {
    return m_super.random_point(eng);
  }
  Point random_point() const { return m_super.random_point(get_global_rng()); }

 private:
  std::shared_ptr<rkh_scene> m_scene;
  rkh_qs_space m_sp;
  super_space_type m_super;
};
template <typename P>
bound_distance_metric<manip_quasi_static_free_space<P> > get(distance_metric_t, const manip_quasi_static_free_space<P>&) {
  return bound_distance_metric<manip_quasi_static_free_space<P> >();
}
// trait tags the planners dispatch on (is_steerable_space / is_metric_space / ... are boost::mpl bools in ReaK;
// a maintainer specialises them to true_ for these classes, manip_free_workspace.hpp:307-318)
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

// ---- planner entry: sample_based_planner<FreeSpace>::solve_planning_query(planning_query<FreeSpace>&) ---------------
// (motion_planner_base.hpp:102, options :400-422, report_progress / report_solution :343-351).  The batched device
// drivers build exactly the graphs of the sequential algorithms on the seed of the global generator; an adaptor hands
// the caller's QUERY OBJECT what the sequential planner's visitor would: every solution through
//     query.register_solution(start_node, goal_node, goal_distance, graph)            (planning_queries.hpp:168-212)
//     query.register_joining_point(start, goal, join1, join2, joining_distance, g1, g2)   (:214-273, bidirectional)
// in the order the sequential planner finds them, and stops reporting once query.keep_going() turns false.  The Query
// type is whatever offers the members planning_query / path_planning_p2p_query do (p2p_planning_query.hpp:74-229):
// get_start_position(), get_goal_position(), max_num_results, keep_going(), register_solution(...), reset_solution_records().
// The reporter chain (any_sbmp_reporter_chain) is two std::function hooks with the arguments of
// do_report_progress_impl(space, g, reporter) / reporter.draw_solution(space, srp).
//
// The motion graph handed to the query: vertices in insertion order (vecBC semantics), bundles named like ReaK's
// (any_motion_graphs.hpp:170-184: position, distance_accum, predecessor, weight).
template <typename Point>
struct hip_motion_graph {
  struct vertex_bundled {
    Point position;
    double distance_accum;    // cost from the start along the predecessors (optimal_mg_vertex)
    std::size_t predecessor;  // null_vertex() for a root / unconnected vertex
    double weight;            // length of the edge from the predecessor
    double density;           // PRM: prm_density_calculator
  };
  typedef std::size_t vertex_descriptor;
  std::vector<vertex_bundled> v;
  std::vector<std::pair<std::size_t, std::size_t> > edges;  // PRM roadmap edges in insertion order
  std::vector<double> edge_weight;
  static std::size_t null_vertex() { return std::size_t(-1); }
  vertex_bundled& operator[](std::size_t u) { return v[u]; }
  const vertex_bundled& operator[](std::size_t u) const { return v[u]; }
};
template <typename Point>
std::size_t num_vertices(const hip_motion_graph<Point>& g) {
  return g.v.size();
}
template <typename Point>
std::size_t vertex(std::size_t i, const hip_motion_graph<Point>&) {
  return i;
}
template <typename Point>
std::size_t null_vertex(const hip_motion_graph<Point>&) {
  return std::size_t(-1);
}

namespace detail {
template <typename Point>
inline void fill_params(rkh_rrt_params& prm, const Point& start, const Point& goal, uint32_t seed, std::size_t max_vertices,
                        std::size_t max_results, double steer_tol, double conn_tol) {
  prm = rkh_rrt_params();
  prm.seed = seed;
  prm.max_vertices = uint32_t(max_vertices);
  prm.max_results = uint32_t(std::min<std::size_t>(max_results, std::size_t(1) << 30));
  prm.steer_tol = steer_tol;
  prm.conn_tol = conn_tol;
  for (std::size_t i = 0; i < start.size(); ++i) {
    prm.start[i] = start[i];
    prm.goal[i] = goal[i];
  }
}
// the draws a finished planner consumed leave the global engine where the sequential planner would have left it
inline void consume_global_draws(uint64_t draws) { get_global_rng().discard(draws); }
inline const rkh_dyn_space* dyn_of(const void*) { return nullptr; }
}  // namespace detail

// common options of sample_based_planner (motion_planner_base.hpp:400-422)
template <typename FreeSpace>
class hip_planner_base {
 public:
  typedef typename FreeSpace::point_type point_type;
  typedef hip_motion_graph<point_type> graph_type;
  hip_planner_base(const std::shared_ptr<FreeSpace>& world, std::size_t max_vertex_count, std::size_t progress_interval,
                   double steer_progress_tol, double connection_tol, double sampling_radius)
      : m_space(world), m_max_vertex_count(max_vertex_count), m_progress_interval(progress_interval),
        m_steer_progress_tol(steer_progress_tol), m_connection_tol(connection_tol), m_sampling_radius(sampling_radius) {}
  virtual ~hip_planner_base() {}
  // set_reporter (motion_planner_base.hpp:112): the two calls a reporter chain receives
  std::function<void(const FreeSpace&, const graph_type&)> progress_reporter;
  std::function<void(const FreeSpace&, const graph_type&, double /*solution cost*/)> solution_reporter;
  std::size_t get_max_vertex_count() const { return m_max_vertex_count; }
  std::size_t get_progress_interval() const { return m_progress_interval; }
  double get_steer_progress_tolerance() const { return m_steer_progress_tol; }
  double get_connection_tolerance() const { return m_connection_tol; }
  double get_sampling_radius() const { return m_sampling_radius; }
  const graph_type& motion_graph() const { return m_graph; }  // the graph of the last query

 protected:
  // report_progress (motion_planner_base.hpp:343-347) counts vertex_added calls and reports every m_progress_interval-th.
  // The device grows the graph in batches, so the reports for the interval boundaries crossed since the last call are
  // issued together, each seeing the graph as it is NOW (at most one batch ahead of its boundary).
  void report_progress_upto(std::size_t vertices_added) {
    if (!m_progress_interval) return;
    while (m_reported + m_progress_interval <= vertices_added) {
      m_reported += m_progress_interval;
      if (progress_reporter) progress_reporter(*m_space, m_graph);
    }
  }
  std::shared_ptr<FreeSpace> m_space;
  std::size_t m_max_vertex_count, m_progress_interval;
  double m_steer_progress_tol, m_connection_tol, m_sampling_radius;
  graph_type m_graph;
  std::size_t m_reported = 0;
};

// ---- rrt_planner, UNIDIRECTIONAL_PLANNING | LINEAR_SEARCH_KNN (rrt_path_planner.tpp:66-145) over the steerable dynamic
// space or the quasi-static space
template <typename FreeSpace>
class hip_rrt_planner : public hip_planner_base<FreeSpace> {
  typedef hip_planner_base<FreeSpace> base;

 public:
  typedef typename base::point_type point_type;
  typedef typename base::graph_type graph_type;
  hip_rrt_planner(const std::shared_ptr<FreeSpace>& world, std::size_t max_vertex_count, std::size_t progress_interval = 0,
                  double steer_progress_tol = 0.1, double connection_tol = 0.05)
      : base(world, max_vertex_count, progress_interval, steer_progress_tol, connection_tol, 1.0) {}

  rkh_planner_stats last_stats = rkh_planner_stats();

  template <typename Query>
  void solve_planning_query(Query& aQuery) {
    const point_type& start = aQuery.get_start_position();
    const point_type& goal = aQuery.get_goal_position();
    const std::size_t D = start.size();
    rkh_rrt_params prm;
    detail::fill_params(prm, start, goal, global_rng_fresh_seed(), this->m_max_vertex_count, aQuery.max_num_results,
                        this->m_steer_progress_tol, this->m_connection_tol);
    std::shared_ptr<rkh_planner> pl = create(prm, static_cast<const FreeSpace*>(nullptr));
    this->m_graph = graph_type();
    this->m_reported = 0;
    rkh_planner_stats st = rkh_planner_stats();
    for (;;) {  // generate_rrt (rr_tree.hpp:179-199) in batches of rounds
      check(rkh_planner_enqueue(pl.get(), 16));
      check(rkh_planner_sync(pl.get(), &st));
      pull_tree(pl.get(), D, st, st.done != 0);
      this->report_progress_upto(std::size_t(st.num_vertices) - 1);
      if (st.done) break;
    }
    // The goal probe of a vertex rides in the steer launch of the round after the one that added it, so the probes are
    // complete once the planner is done (it stops by itself at max_num_results solutions): the solutions are handed to
    // the query then, in the order the sequential planner registers them.
    register_solutions(aQuery);
    last_stats = st;
    detail::consume_global_draws(uint64_t(st.iterations) * D);
  }

 private:
  std::shared_ptr<rkh_planner> create(const rkh_rrt_params& prm, const kte_dynamic_free_space<point_type>*) {
    rkh_planner* raw = nullptr;
    check(rkh_planner_create(this->m_space->scene().get(), &this->m_space->dyn_space(), &prm, &raw));
    return std::shared_ptr<rkh_planner>(raw, [](rkh_planner* p) { (void)rkh_planner_destroy(p); });
  }
  std::shared_ptr<rkh_planner> create(const rkh_rrt_params& prm, const manip_quasi_static_free_space<point_type>*) {
    rkh_planner* raw = nullptr;
    check(rkh_planner_create_qs_batch(this->m_space->scene().get(), &this->m_space->qs_space(), &prm, 1, &raw));
    return std::shared_ptr<rkh_planner>(raw, [](rkh_planner* p) { (void)rkh_planner_destroy(p); });
  }
  void pull_tree(rkh_planner* pl, std::size_t D, const rkh_planner_stats& st, bool with_goal_probes) {
    const std::size_t nv = std::size_t(st.num_vertices);
    std::vector<double> pos(nv * D), gd(nv > 1 ? nv - 1 : 1, std::numeric_limits<double>::infinity());
    std::vector<uint32_t> parent(nv);
    check(rkh_planner_get_tree(pl, 0, pos.data(), parent.data(), nullptr, nullptr, with_goal_probes ? gd.data() : nullptr));
    graph_type& g = this->m_graph;
    const std::size_t old = g.v.size();
    g.v.resize(nv);
    m_goal_dist.resize(nv, std::numeric_limits<double>::infinity());
    for (std::size_t v = old; v < nv; ++v) {
      typename graph_type::vertex_bundled& b = g.v[v];
      b.position = point_type(pos.begin() + v * D, pos.begin() + (v + 1) * D);
      b.predecessor = v == 0 ? graph_type::null_vertex() : std::size_t(parent[v]);
      b.weight = v == 0 ? 0.0 : this->m_space->get_super_space().distance(g.v[b.predecessor].position, b.position);
      b.distance_accum = v == 0 ? 0.0 : g.v[b.predecessor].distance_accum + b.weight;
      b.density = 0.0;
    }
    if (with_goal_probes)
      for (std::size_t v = 1; v < nv; ++v) m_goal_dist[v] = gd[v - 1];
  }
  // planning_visitor_base::edge_added (planning_visitors.hpp:186-201): a vertex whose goal probe is finite registers a
  // solution, in vertex order, while the query wants more
  template <typename Query>
  void register_solutions(Query& aQuery) {
    const graph_type& g = this->m_graph;
    for (std::size_t v = 1; v < g.v.size(); ++v) {
      if (!(m_goal_dist[v] < std::numeric_limits<double>::infinity())) continue;
      if (!aQuery.keep_going()) break;
      if (aQuery.register_solution(std::size_t(0), v, m_goal_dist[v], this->m_graph) && this->solution_reporter)
        this->solution_reporter(*this->m_space, g, g.v[v].distance_accum + m_goal_dist[v]);
    }
  }
  std::vector<double> m_goal_dist;
};

// ---- rrtstar_planner, UNIDIRECTIONAL_PLANNING | LINEAR_SEARCH_KNN (rrtstar_path_planner.tpp:298-; optionally
// USE_BRANCH_AND_BOUND_PRUNING_FLAG).  Vertex 0 = start, vertex 1 = goal (the goal node is part of the graph, so
// vertex_added -> dispatched_register_solution reports whenever the goal's distance_accum improved,
// planning_visitors.hpp:108-116,178-182); the adaptor registers the goal's final best cost.
template <typename FreeSpace>
class hip_rrtstar_planner : public hip_planner_base<FreeSpace> {
  typedef hip_planner_base<FreeSpace> base;

 public:
  typedef typename base::point_type point_type;
  typedef typename base::graph_type graph_type;
  hip_rrtstar_planner(const std::shared_ptr<FreeSpace>& world, std::size_t max_vertex_count, std::size_t progress_interval = 0,
                      double steer_progress_tol = 0.1, double connection_tol = 0.05, bool branch_and_bound = false)
      : base(world, max_vertex_count, progress_interval, steer_progress_tol, connection_tol, 1.0), m_bnb(branch_and_bound) {}
  rkh_rrtstar_stats last_stats = rkh_rrtstar_stats();

  template <typename Query>
  void solve_planning_query(Query& aQuery) {
    const point_type& start = aQuery.get_start_position();
    const point_type& goal = aQuery.get_goal_position();
    const std::size_t D = start.size();
    rkh_rrt_params prm;
    detail::fill_params(prm, start, goal, global_rng_fresh_seed(), this->m_max_vertex_count, aQuery.max_num_results,
                        this->m_steer_progress_tol, this->m_connection_tol);
    std::shared_ptr<rkh_rrtstar> pl = create(prm, static_cast<const FreeSpace*>(nullptr));
    if (m_bnb) check(rkh_rrtstar_set_branch_and_bound(pl.get(), 1));
    rkh_rrtstar_stats st = rkh_rrtstar_stats();
    check(rkh_rrtstar_solve(pl.get(), -1, &st));
    const std::size_t nv = std::size_t(st.num_vertices);
    std::vector<double> pos(nv * D), dist(nv);
    std::vector<uint32_t> pred(nv);
    check(rkh_rrtstar_get_graph(pl.get(), 0, pos.data(), pred.data(), dist.data(), nullptr));
    graph_type& g = this->m_graph;
    g = graph_type();
    g.v.resize(nv);
    for (std::size_t v = 0; v < nv; ++v) {
      typename graph_type::vertex_bundled& b = g.v[v];
      b.position = point_type(pos.begin() + v * D, pos.begin() + (v + 1) * D);
      b.predecessor = (pred[v] == 0xFFFFFFFFu || v == 0) ? graph_type::null_vertex() : std::size_t(pred[v]);
      b.distance_accum = dist[v];
      b.weight = b.predecessor == graph_type::null_vertex() ? 0.0 : dist[v] - dist[b.predecessor];
      b.density = 0.0;
    }
    this->m_reported = 0;
    this->report_progress_upto(nv);
    if (nv > 1 && g.v[1].predecessor != graph_type::null_vertex() && aQuery.keep_going()) {
      if (aQuery.register_solution(std::size_t(0), std::size_t(1), 0.0, g) && this->solution_reporter)
        this->solution_reporter(*this->m_space, g, g.v[1].distance_accum);
    }
    last_stats = st;
    detail::consume_global_draws(uint64_t(st.samples) * D);
  }

 private:
  std::shared_ptr<rkh_rrtstar> create(const rkh_rrt_params& prm, const kte_dynamic_free_space<point_type>*) {
    rkh_rrtstar* raw = nullptr;
    check(rkh_rrtstar_create_batch(this->m_space->scene().get(), &this->m_space->dyn_space(), &prm, 1, &raw));
    return std::shared_ptr<rkh_rrtstar>(raw, [](rkh_rrtstar* p) { (void)rkh_rrtstar_destroy(p); });
  }
  std::shared_ptr<rkh_rrtstar> create(const rkh_rrt_params& prm, const manip_quasi_static_free_space<point_type>*) {
    rkh_rrtstar* raw = nullptr;
    check(rkh_rrtstar_create_qs_batch(this->m_space->scene().get(), &this->m_space->qs_space(), &prm, 1, &raw));
    return std::shared_ptr<rkh_rrtstar>(raw, [](rkh_rrtstar* p) { (void)rkh_rrtstar_destroy(p); });
  }
  bool m_bnb;
};

// ---- prm_planner (prm_path_planner.tpp:131-365).  As in the reference (density_plan_visitor; rkh.h), the roadmap grows to
// max_vertex_count and no solution is registered: the query's records stay as they were; the roadmap is motion_graph().
template <typename FreeSpace>
class hip_prm_planner : public hip_planner_base<FreeSpace> {
  typedef hip_planner_base<FreeSpace> base;

 public:
  typedef typename base::point_type point_type;
  typedef typename base::graph_type graph_type;
  hip_prm_planner(const std::shared_ptr<FreeSpace>& world, std::size_t max_vertex_count, std::size_t progress_interval = 0,
                  double steer_progress_tol = 0.1, double connection_tol = 0.05, double sampling_radius = 1.0)
      : base(world, max_vertex_count, progress_interval, steer_progress_tol, connection_tol, sampling_radius) {}
  rkh_prm_stats last_stats = rkh_prm_stats();

  template <typename Query>
  void solve_planning_query(Query& aQuery) {
    const point_type& start = aQuery.get_start_position();
    const point_type& goal = aQuery.get_goal_position();
    const std::size_t D = start.size();
    rkh_prm_params prm = rkh_prm_params();
    detail::fill_params(prm.base, start, goal, global_rng_fresh_seed(), this->m_max_vertex_count, aQuery.max_num_results,
                        this->m_steer_progress_tol, this->m_connection_tol);
    prm.sampling_radius = this->m_sampling_radius;
    prm.expand_probability = 0.2;  // fixed by the reference's generate_prm call (prm_path_planner.tpp:250-253)
    std::shared_ptr<rkh_prm> pl = create(prm, static_cast<const FreeSpace*>(nullptr));
    rkh_prm_stats st = rkh_prm_stats();
    check(rkh_prm_solve(pl.get(), -1, &st));
    const std::size_t nv = std::size_t(st.num_vertices), ne = std::size_t(st.num_edges);
    std::vector<double> pos(nv * D), ew(ne ? ne : 1), dens(nv);
    std::vector<uint32_t> eu(ne ? ne : 1), ev(ne ? ne : 1);
    check(rkh_prm_get_graph(pl.get(), 0, pos.data(), eu.data(), ev.data(), ew.data(), dens.data(), nullptr, nullptr, nullptr));
    graph_type& g = this->m_graph;
    g = graph_type();
    g.v.resize(nv);
    for (std::size_t v = 0; v < nv; ++v) {
      typename graph_type::vertex_bundled& b = g.v[v];
      b.position = point_type(pos.begin() + v * D, pos.begin() + (v + 1) * D);
      b.predecessor = graph_type::null_vertex();
      b.distance_accum = std::numeric_limits<double>::infinity();
      b.weight = 0.0;
      b.density = dens[v];
    }
    for (std::size_t e = 0; e < ne; ++e) {
      g.edges.push_back(std::make_pair(std::size_t(eu[e]), std::size_t(ev[e])));
      g.edge_weight.push_back(ew[e]);
    }
    this->m_reported = 0;
    this->report_progress_upto(nv);
    last_stats = st;
    // PRM's draws depend on its control flow (rejection samples, expansion coin, random walks): the planner reports them
    detail::consume_global_draws(uint64_t(st.samples));
  }

 private:
  std::shared_ptr<rkh_prm> create(const rkh_prm_params& prm, const kte_dynamic_free_space<point_type>*) {
    rkh_prm* raw = nullptr;
    check(rkh_prm_create_batch(this->m_space->scene().get(), &this->m_space->dyn_space(), &prm, 1, &raw));
    return std::shared_ptr<rkh_prm>(raw, [](rkh_prm* p) { (void)rkh_prm_destroy(p); });
  }
  std::shared_ptr<rkh_prm> create(const rkh_prm_params& prm, const manip_quasi_static_free_space<point_type>*) {
    rkh_prm* raw = nullptr;
    check(rkh_prm_create_qs_batch(this->m_space->scene().get(), &this->m_space->qs_space(), &prm, 1, &raw));
    return std::shared_ptr<rkh_prm>(raw, [](rkh_prm* p) { (void)rkh_prm_destroy(p); });
  }
};

// ---- rrt_planner with BIDIRECTIONAL_PLANNING, the reference's default flag (rrt_path_planner.hpp:114 ->
// generate_bidirectional_rrt, rr_tree.hpp:256-317) over the quasi-static space (a reversible space).  Two trees; a
// joining vertex registers a solution through query.register_joining_point (planning_visitors.hpp:223-231).  The best
// registered solution is handed over after the run: graph1() / graph2() are the trees, the joining pair are the last
// vertices of the two solution paths.
template <typename FreeSpace>
class hip_birrt_planner : public hip_planner_base<FreeSpace> {
  typedef hip_planner_base<FreeSpace> base;

 public:
  typedef typename base::point_type point_type;
  typedef typename base::graph_type graph_type;
  hip_birrt_planner(const std::shared_ptr<FreeSpace>& world, std::size_t max_vertex_count, std::size_t progress_interval = 0,
                    double steer_progress_tol = 0.1, double connection_tol = 0.05)
      : base(world, max_vertex_count, progress_interval, steer_progress_tol, connection_tol, 1.0) {}
  rkh_birrt_stats last_stats = rkh_birrt_stats();
  const graph_type& graph1() const { return this->m_graph; }
  const graph_type& graph2() const { return m_graph2; }

  template <typename Query>
  void solve_planning_query(Query& aQuery) {
    const point_type& start = aQuery.get_start_position();
    const point_type& goal = aQuery.get_goal_position();
    const std::size_t D = start.size();
    rkh_rrt_params prm;
    detail::fill_params(prm, start, goal, global_rng_fresh_seed(), this->m_max_vertex_count, aQuery.max_num_results,
                        this->m_steer_progress_tol, this->m_connection_tol);
    rkh_birrt* raw = nullptr;
    check(rkh_birrt_create_qs_batch(this->m_space->scene().get(), &this->m_space->qs_space(), &prm, 1, &raw));
    std::shared_ptr<rkh_birrt> pl(raw, [](rkh_birrt* p) { (void)rkh_birrt_destroy(p); });
    rkh_birrt_stats st = rkh_birrt_stats();
    check(rkh_birrt_solve(pl.get(), -1, &st));
    const std::size_t n1 = std::size_t(st.num_vertices_1), n2 = std::size_t(st.num_vertices_2);
    std::vector<double> p1(n1 * D), p2(n2 * D);
    std::vector<uint32_t> par1(n1), par2(n2);
    check(rkh_birrt_get_trees(pl.get(), 0, p1.data(), par1.data(), p2.data(), par2.data(), nullptr, nullptr));
    fill_tree(this->m_graph, p1, par1, D);
    fill_tree(m_graph2, p2, par2, D);
    this->m_reported = 0;
    this->report_progress_upto(n1 + n2);
    std::vector<uint32_t> s1(n1), s2(n2);
    uint32_t m1 = 0, m2 = 0;
    double cost = std::numeric_limits<double>::infinity();
    check(rkh_birrt_get_solution(pl.get(), 0, s1.data(), &m1, s2.data(), &m2, uint32_t(std::max(n1, n2)), &cost));
    if (m1 && m2 && aQuery.keep_going()) {
      const std::size_t j1 = s1[m1 - 1], j2 = s2[m2 - 1];
      const double join = cost - this->m_graph.v[j1].distance_accum - m_graph2.v[j2].distance_accum;
      if (aQuery.register_joining_point(std::size_t(0), std::size_t(0), j1, j2, join, this->m_graph, m_graph2) &&
          this->solution_reporter)
        this->solution_reporter(*this->m_space, this->m_graph, cost);
    }
    last_stats = st;
    detail::consume_global_draws(uint64_t(st.samples) * D);
  }

 private:
  void fill_tree(graph_type& g, const std::vector<double>& pos, const std::vector<uint32_t>& parent, std::size_t D) {
    g = graph_type();
    g.v.resize(parent.size());
    for (std::size_t v = 0; v < parent.size(); ++v) {
      typename graph_type::vertex_bundled& b = g.v[v];
      b.position = point_type(pos.begin() + v * D, pos.begin() + (v + 1) * D);
      b.predecessor = parent[v] == 0xFFFFFFFFu ? graph_type::null_vertex() : std::size_t(parent[v]);
      b.weight = b.predecessor == graph_type::null_vertex()
                     ? 0.0
                     : this->m_space->get_super_space().distance(g.v[b.predecessor].position, b.position);
      b.distance_accum = b.predecessor == graph_type::null_vertex() ? 0.0 : g.v[b.predecessor].distance_accum + b.weight;
      b.density = 0.0;
    }
  }
  graph_type m_graph2;
};

}  // namespace rkh
#endif
