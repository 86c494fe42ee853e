// ORACLE -- TEST INFRASTRUCTURE ONLY (see reak_math.hpp header).
//
// CPU restatement of the sampling-based-planning layer: global RNG + hyperbox sampling, Euclidean
// metric, linear 1-NN / k-NN search, star_neighborhood, the quasi-static edge-stepping loop,
// RK4 (runge_kutta4_integrate_impl), the steerable dynamic free space, the planning-visitor
// predicates and generate_rrt.
//
// Third-party arithmetic restated from its published definition (source not in /root/reference):
//   Boost.Random mt19937 + uniform_01<Engine&,double> (Boost >= 1.46, R/CMakeLists.txt:73-75):
//   std::mt19937 is the same generator (10000th output of the default seed is 4123659995);
//   uniform_01 on a 32-bit engine is  u = eng() * 2^-32  (one draw per coordinate, redrawn if it
//   rounds to 1.0, which cannot happen in double).  "parity unpinned": no reference test fixes a
//   seed or an expected sample (global_rng.hpp:50-54 seeds from random_device).
// Planner-level results (node counts, trees) are pinned by nothing in the reference; the golden
// vectors under tests/golden/ generated from this restatement are the pin.
#ifndef REAK_ORACLE_PLANNING_HPP
#define REAK_ORACLE_PLANNING_HPP

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <random>
#include <set>
#include <utility>
#include <vector>

#include "../include/rkh_types.h"
#include "reak_kte.hpp"
#include "reak_proximity.hpp"

namespace oracle {

typedef std::vector<double> Point;

// get_global_rng(): core/base/global_rng.hpp:44-54 ; boost::uniform_01<global_rng_type&,double>
struct GlobalRng {
  std::mt19937 eng;
  explicit GlobalRng(uint32_t seed = 5489u) : eng(seed) {}
  double uniform_01() {
    for (;;) {
      double result = double(eng() - std::mt19937::min()) * (1.0 / 4294967296.0);
      if (result < 1.0) return result;
    }
  }
};

// hyperbox_topology::random_point: ctrl/topologies/hyperbox_topology.hpp:97-103
inline Point hyperbox_random_point(GlobalRng& rng, const double* lower, const double* upper, int D) {
  Point p(lower, lower + D);
  for (int i = 0; i < D; ++i) p[i] += rng.uniform_01() * (upper[i] - lower[i]);
  return p;
}
// hyperbox_topology::is_in_bounds: hyperbox_topology.hpp:178-189
inline bool hyperbox_is_in_bounds(const double* a, const double* lower, const double* upper, int D) {
  for (int i = 0; i < D; ++i) {
    if (lower[i] < upper[i]) {
      if ((a[i] < lower[i]) || (a[i] > upper[i])) return false;
    } else {
      if ((a[i] > lower[i]) || (a[i] < upper[i])) return false;
    }
  }
  return true;
}
// euclidean_distance_metric: ctrl/topologies/vect_distance_metrics.hpp:113-150
// (difference = a - b, vector_topology.hpp; result += d*d left to right; sqrt)
inline double euclid(const double* a, const double* b, int D) {
  double result = 0.0;
  for (int i = 0; i < D; ++i) {
    double d = a[i] - b[i];
    result += d * d;
  }
  return std::sqrt(result);
}

// min_dist_linear_search (1-NN): ctrl/path_planning/topological_search.hpp:95-118
// vertices are visited in insertion order; strict '<' => first minimum wins.  Returns n if empty.
inline std::size_t linear_nn(const double* q, const double* pts, std::size_t n, int D, double* d_out = nullptr) {
  if (n == 0) return n;
  double d_best = std::numeric_limits<double>::infinity();
  std::size_t result = n;
  for (std::size_t i = 0; i < n; ++i) {
    double d = euclid(q, pts + i * D, D);
    if (d < d_best) {
      d_best = d;
      result = i;
    }
  }
  if (d_out) *d_out = d_best;
  return result;
}

// min_dist_linear_search (k-NN + radius): topological_search.hpp:244-274
// compare_pair_first orders the heap by distance only (detail::compare_pair_first :57-66).
// `removed` (optional): vertices taken out of the graph (remove_vertex) -- they are not in vertices(g) any more
inline void linear_knn(const double* q, const double* pts, std::size_t n, int D, std::size_t max_neighbors,
                       double radius, std::vector<std::pair<double, std::size_t>>& out,
                       const std::vector<uint8_t>* removed = nullptr) {
  out.clear();
  if (n == 0) return;
  auto p_compare = [](const std::pair<double, std::size_t>& a, const std::pair<double, std::size_t>& b) {
    return a.first < b.first;
  };
  for (std::size_t i = 0; i < n; ++i) {
    if (removed && (*removed)[i]) continue;
    double d = euclid(q, pts + i * D, D);
    if (!(d < radius)) continue;
    out.push_back(std::make_pair(d, i));
    std::push_heap(out.begin(), out.end(), p_compare);
    if (out.size() > max_neighbors) {
      std::pop_heap(out.begin(), out.end(), p_compare);
      out.pop_back();
      radius = out.front().first;
    }
  }
  std::sort_heap(out.begin(), out.end(), p_compare);
}

// math::highest_set_bit: core/base/misc_math.hpp:50-59
inline std::size_t highest_set_bit(std::size_t N) {
  std::size_t temp = 0;
  for (std::size_t shift = sizeof(std::size_t) * 4; (shift && (N != 1)); shift >>= 1) {
    if (N >> shift) {
      temp |= shift;
      N >>= shift;
    }
  }
  return temp;
}
// star_neighborhood::operator(): ctrl/graph_alg/neighborhood_functors.hpp:95-102
inline void star_neighborhood(std::size_t N, double c_space_dimensions, double gamma_value, std::size_t* k,
                              double* radius) {
  std::size_t log_N = highest_set_bit(N) + 1;
  *k = 4 * log_N;
  *radius = gamma_value * std::pow(log_N / double(N), 1.0 / c_space_dimensions);
}

// --------------------------------------------------------------------------------------------
// Free spaces.  Common surface used by the visitor predicates / generate_rrt below.
struct SpaceCounters {
  long edges_checked = 0;   // steer_towards_position + can_be_connected + goal probes
  long states_checked = 0;  // is_free calls
  long f_evals = 0;
};

// Quasi-static manipulator free space over joint positions:
// manip_quasi_static_env (ctrl/topologies/manip_free_workspace.hpp:113-300) on a hyperbox joint space
// with linear interpolation (vector_topology::move_position_toward).
struct QuasiStaticSpace {
  int D = 0;
  std::vector<double> lower, upper;
  double min_interval = 0.1;
  // rate-limited joint space (Ndof_rl_space): the point holds q_i / speed_limit_i, the model is evaluated at
  // point_i * speed_limit_i (joint_space_limits_detail.hpp:1552-1561,1868-1875; manip_free_workspace.hpp: apply_to_model
  // maps the point to the normal joint space first); empty = an ordinary joint space
  std::vector<double> speed;
  KteChain chain;
  ProxyEnv env;
  SpaceCounters cnt;

  Point random_point(GlobalRng& rng) const { return hyperbox_random_point(rng, lower.data(), upper.data(), D); }
  double metric(const Point& a, const Point& b) const { return euclid(a.data(), b.data(), D); }
  // manip_quasi_static_env::is_free :154-156 ; manip_dk_proxy_env_impl::is_free :79-99
  bool is_free(const Point& p) {
    ++cnt.states_checked;
    if (!hyperbox_is_in_bounds(p.data(), lower.data(), upper.data(), D)) return false;
    std::vector<double> x(2 * D, 0.0);
    for (int i = 0; i < D; ++i) x[2 * i] = speed.empty() ? p[i] : p[i] * speed[i];
    chain.apply_kinematics(x.data());
    return env.is_free(chain);
  }
  // vector_topology::move_position_toward: a + (b - a) * fraction
  Point lin_move(const Point& a, double fraction, const Point& b) const {
    Point r(D);
    for (int i = 0; i < D; ++i) r[i] = a[i] + (b[i] - a[i]) * fraction;
    return r;
  }
  // interp_topo_move_position_toward_pred: ctrl/interpolation/interpolated_topologies.hpp:137-163
  Point move_position_toward(const Point& a, double fraction, const Point& b) {
    ++cnt.edges_checked;
    double dist_tot = metric(a, b);
    if (dist_tot == std::numeric_limits<double>::infinity()) return a;
    if (dist_tot < min_interval) return lin_move(a, fraction, b);
    double dist_inter = dist_tot * fraction;
    double dist_cur = min_interval;
    Point result = a;
    Point last_result = a;
    while (dist_cur < dist_inter) {
      result = lin_move(a, dist_cur / dist_tot, b);
      if (!is_free(result)) return last_result;
      dist_cur += min_interval;
      last_result = result;
    }
    if (fraction == 1.0) return b;
    else if (fraction == 0.0) return a;
    return lin_move(a, fraction, b);
  }
  // interp_topo_get_distance_pred: interpolated_topologies.hpp:193-199
  double distance(const Point& a, const Point& b) {
    Point b_tmp = move_position_toward(a, 1.0, b);
    if (metric(b_tmp, b) < std::numeric_limits<double>::epsilon()) return metric(a, b);
    return std::numeric_limits<double>::infinity();
  }
  // planning_visitor_base::dispatched_steer_towards_position case 4 (planning_visitors.hpp:288-296)
  double steer(const Point& src, const Point& dest, double fraction, Point& p_result) {
    p_result = move_position_toward(src, fraction, dest);
    return metric(src, p_result);
  }
  // vector_topology::move_position_back_to: b + (a - b) * fraction (vector_topology.hpp:115-118)
  Point lin_move_back(const Point& a, double fraction, const Point& b) const {
    Point r(D);
    for (int i = 0; i < D; ++i) r[i] = b[i] + (a[i] - b[i]) * fraction;
    return r;
  }
  // interp_topo_move_position_back_to_pred: interpolated_topologies.hpp:165-191 -- the walk starts at b and moves
  // towards a.  Kept as written: a walk that completes with fraction == 1.0 returns b, its own starting point (:185-186).
  Point move_position_back_to(const Point& a, double fraction, const Point& b) {
    ++cnt.edges_checked;
    double dist_tot = metric(a, b);
    if (dist_tot == std::numeric_limits<double>::infinity()) return b;
    if (dist_tot < min_interval) return lin_move_back(a, fraction, b);
    double dist_inter = dist_tot * fraction;
    double dist_cur = min_interval;
    Point result = b;
    Point last_result = b;
    while (dist_cur < dist_inter) {
      result = lin_move_back(a, dist_cur / dist_tot, b);
      if (!is_free(result)) return last_result;
      dist_cur += min_interval;
      last_result = result;
    }
    if (fraction == 1.0) return b;
    else if (fraction == 0.0) return a;
    return lin_move_back(a, fraction, b);
  }
  // planning_visitor_base::dispatched_steer_back_to_position case 2 (planning_visitors.hpp:311-320)
  double steer_back(const Point& src, const Point& dest, double fraction, Point& p_result) {
    p_result = move_position_back_to(src, fraction, dest);
    return metric(p_result, dest);
  }
};

// runge_kutta4_integrate_impl: ctrl/sys_integrators/runge_kutta4_integrator_sys.hpp:53-97 with a
// constant_trajectory input (ctrl/interpolation/constant_trajectory.hpp:138-150: the waypoint is
// the same u at every time).  Returns the number of loop iterations.
template <typename F>
int runge_kutta4_integrate(F get_state_derivative, int D, const double* start_point, double* end_point,
                           const double* u, double start_time, double end_time, double time_step) {
  std::vector<double> dp(D), w(D), k1(D), k2(D), k3(D);
  get_state_derivative(start_point, u, dp.data());
  double t = start_time;
  for (int i = 0; i < D; ++i) end_point[i] = start_point[i];
  int iters = 0;
  while (((time_step > 0.0) && (t < end_time)) || ((time_step < 0.0) && (t > end_time))) {
    ++iters;
    for (int i = 0; i < D; ++i) w[i] = end_point[i];
    for (int i = 0; i < D; ++i) k1[i] = time_step * dp[i];
    for (int i = 0; i < D; ++i) end_point[i] = end_point[i] + 0.5 * k1[i];
    t += time_step * 0.5;
    get_state_derivative(end_point, u, dp.data());
    for (int i = 0; i < D; ++i) k2[i] = time_step * dp[i];
    for (int i = 0; i < D; ++i) end_point[i] = w[i] + 0.5 * k2[i];
    get_state_derivative(end_point, u, dp.data());
    for (int i = 0; i < D; ++i) k3[i] = time_step * dp[i];
    for (int i = 0; i < D; ++i) end_point[i] = w[i] + k3[i];
    t += time_step * 0.5;
    get_state_derivative(end_point, u, dp.data());
    for (int i = 0; i < D; ++i)
      end_point[i] = end_point[i] + ((((1.0 / 6.0) * k1[i] + (2.0 / 6.0) * k2[i]) + (time_step / 6.0) * dp[i]) -
                                     (2.0 / 3.0) * k3[i]);
    get_state_derivative(end_point, u, dp.data());  // primes the next iteration (:95)
  }
  return iters;
}

// Steerable dynamic free space over a KTE chain ("kte_dynamic_free_space", build-defined; loop shape
// of examples/misc/MEAQR_topology.hpp:503-565 (steer_with_constant_control) and :995-1003 (distance)).
struct DynSpace {
  rkh_dyn_space P;
  int D = 0;
  std::vector<double> lower, upper;  // the state hyperbox (copies of P.lower / P.upper for the planners that read them)
  KteChain chain;
  ProxyEnv env;
  SpaceCounters cnt;

  Point random_point(GlobalRng& rng) const { return hyperbox_random_point(rng, P.lower, P.upper, D); }
  double metric(const Point& a, const Point& b) const { return euclid(a.data(), b.data(), D); }
  // MEAQR_topology_with_CD::is_free_impl :919-940 : in_bounds, then DK, then proximity
  bool is_free(const Point& x) {
    ++cnt.states_checked;
    if (!hyperbox_is_in_bounds(x.data(), P.lower, P.upper, D)) return false;
    chain.apply_kinematics(x.data());
    return env.is_free(chain);
  }
  void control(const Point& x, const Point& target, double* u) const {
    for (int i = 0; i < P.n_dof; ++i) {
      double v = P.kp * (target[2 * i] - x[2 * i]) + P.kd * (target[2 * i + 1] - x[2 * i + 1]);
      if (v > P.u_max) v = P.u_max;
      else if (v < -P.u_max) v = -P.u_max;
      u[i] = v;
    }
  }
  void rk4_step(const Point& x, const double* u, double t, Point& x_next) {
    x_next.resize(D);
    auto f = [&](const double* p, const double* uu, double* pd) {
      ++cnt.f_evals;
      chain.get_state_derivative(p, uu, pd);
    };
    runge_kutta4_integrate(f, D, x.data(), x_next.data(), u, t, t + P.dt, P.dt);
  }
  // steer_position_toward(a, fraction, b): returns the last free state; steps_free = number of
  // accepted RK4 steps; record (optional) = the steer_record (a, then every accepted state).
  Point steer_position_toward(const Point& a, double fraction, const Point& b, int* steps_free = nullptr,
                              std::vector<Point>* record = nullptr) {
    ++cnt.edges_checked;
    double T_goal = fraction * (P.steps_per_edge * P.dt);
    double current_time = 0.0;
    Point x_current = a;
    if (record) record->push_back(x_current);
    int n_free = 0;
    Point x_next;
    std::vector<double> u(P.n_dof);
    while ((current_time < T_goal) && (metric(x_current, b) > P.goal_tol)) {
      control(x_current, b, u.data());
      rk4_step(x_current, u.data(), current_time, x_next);
      if (is_free(x_next)) {
        x_current = x_next;
        current_time += P.dt;
        ++n_free;
        if (record) record->push_back(x_current);
      } else {
        break;
      }
    }
    if (steps_free) *steps_free = n_free;
    return x_current;
  }
  // C_free distance (goal probe): MEAQR_topology_with_CD::distance :995-1003
  double distance(const Point& a, const Point& b) {
    Point result = steer_position_toward(a, 1.0, b);
    if (metric(a, b) * 0.05 > metric(result, b)) return metric(a, b);
    return std::numeric_limits<double>::infinity();
  }
  // dispatched_steer_towards_position case 3 (planning_visitors.hpp:277-285)
  double steer(const Point& src, const Point& dest, double fraction, Point& p_result) {
    p_result = steer_position_toward(src, fraction, dest);
    return metric(src, p_result);
  }
};

// --------------------------------------------------------------------------------------------
// RRT (unidirectional, linear-search NN): rrt_planner::solve_planning_query
// (ctrl/path_planning/rrt_path_planner.tpp:66-145) -> generate_rrt (ctrl/graph_alg/rr_tree.hpp:179-199).
struct RrtResult {
  int D = 0;
  std::vector<double> pos;       // n x D, insertion order (vertices(g) order)
  std::vector<uint32_t> parent;  // parent index; root: 0xFFFFFFFF
  std::vector<uint32_t> nn_seq;  // per iteration: nearest-neighbour vertex
  std::vector<uint8_t> accept;   // per iteration: steer accepted (vertex added)
  std::vector<double> goal_dist; // per added vertex (non-root): goal-probe distance
  long iterations = 0;
  long num_solutions = 0;
  double best_cost = std::numeric_limits<double>::infinity();
  SpaceCounters cnt;
};

// warm_pos / warm_n (timing only, bench.py's cpu_baseline): the loop starts on a tree that already holds these vertices
// (rows 1 .. warm_n of a tree grown elsewhere on the same world, hung under the root), so that the cost of an iteration
// AT that tree size can be measured without growing the tree on the CPU first.
template <typename Space>
void generate_rrt(Space& space, const rkh_rrt_params& prm, long max_iterations, RrtResult& res,
                  const double* warm_pos = nullptr, std::size_t warm_n = 0) {
  const int D = space.D;
  GlobalRng rng(prm.seed);
  res = RrtResult();
  res.D = D;
  Point start(prm.start, prm.start + D), goal(prm.goal, prm.goal + D);
  // create_root(vp_start): rrt_path_planner.tpp:131-133 (no vertex_added call => not counted)
  res.pos.insert(res.pos.end(), start.begin(), start.end());
  res.parent.push_back(0xFFFFFFFFu);
  if (warm_pos && warm_n) {
    res.pos.insert(res.pos.end(), warm_pos, warm_pos + warm_n * std::size_t(D));
    res.parent.insert(res.parent.end(), warm_n, 0u);
    res.goal_dist.insert(res.goal_dist.end(), warm_n, std::numeric_limits<double>::infinity());
  }
  unsigned long m_iteration_count = (unsigned long)warm_n;  // sample_based_planner::m_iteration_count
  // keep_going: planning_visitors.hpp:203-205 -> motion_planner_base.hpp:374 && p2p_planning_query.hpp:121-123
  auto keep_going = [&]() {
    return (m_iteration_count < prm.max_vertices) && (prm.max_results > (unsigned long)res.num_solutions);
  };
  while (keep_going() && (max_iterations < 0 || res.iterations < max_iterations)) {
    ++res.iterations;
    Point p_rnd = space.random_point(rng);                                   // rr_tree.hpp:194
    std::size_t n = res.parent.size();
    std::size_t u = linear_nn(p_rnd.data(), res.pos.data(), n, D);           // rr_tree.hpp:195
    res.nn_seq.push_back(uint32_t(u));
    // steer_towards_position: planning_visitors.hpp:349-360
    Point pu(res.pos.begin() + u * D, res.pos.begin() + (u + 1) * D);
    Point p_v;
    double traveled_dist = space.steer(pu, p_rnd, 1.0, p_v);
    double best_case_dist = space.metric(pu, p_rnd);
    bool reached_new = (!std::isinf(traveled_dist)) && (traveled_dist < 2.0 * best_case_dist) &&
                       (traveled_dist > prm.steer_tol * best_case_dist);
    res.accept.push_back(reached_new ? 1 : 0);
    if (!reached_new) continue;
    // add_child_vertex; vis.vertex_added (iteration_count++); vis.edge_added (goal probe)
    res.pos.insert(res.pos.end(), p_v.begin(), p_v.end());
    res.parent.push_back(uint32_t(u));
    ++m_iteration_count;  // report_progress: motion_planner_base.hpp:343-347
    // edge_added: planning_visitors.hpp:186-201
    double goal_dist = space.distance(p_v, goal);
    res.goal_dist.push_back(goal_dist);
    if (goal_dist < std::numeric_limits<double>::infinity()) {
      // register_basic_solution_path_impl: solution_path_factories.hpp:132-152
      double total = goal_dist;
      std::size_t v = res.parent.size() - 1;
      while (res.parent[v] != 0xFFFFFFFFu) {
        std::size_t pv = res.parent[v];
        total += euclid(&res.pos[pv * D], &res.pos[v * D], D);
        v = pv;
      }
      if (res.num_solutions == 0 || total < res.best_cost) {
        res.best_cost = total;
        ++res.num_solutions;
      }
    }
  }
  res.cnt = space.cnt;
}

}  // namespace oracle
#endif

// --------------------------------------------------------------------------------------------
// RRT* (unidirectional, linear-search k-NN, undirected motion graph = symmetric metric space):
// rrtstar_planner::solve_planning_query_impl (ctrl/path_planning/rrtstar_path_planner.tpp:298-)
//  -> generate_rrt_star (ctrl/graph_alg/rrt_star.hpp:530-570) -> generate_rrt_star_loop (:169-190)
// with rrg_node_generator (node_generators.hpp:137-172), star_neighborhood
// (neighborhood_functors.hpp:95-102) and lazy_node_connector (lazy_connector.hpp:332-372).
namespace oracle {

struct RrtStarResult {
  int D = 0;
  std::vector<double> pos;        // vertex 0 = start, vertex 1 = goal (init_motion_graph, rrtstar_path_planner.tpp:188-203)
  std::vector<uint32_t> pred;     // predecessor (start: itself; unconnected: 0xFFFFFFFF)
  std::vector<double> dist;       // distance_accum
  std::vector<double> weight;     // weight of the edge (pred -> v)
  std::vector<uint32_t> near_seq; // per loop iteration: x_near returned by the node generator (0xFFFFFFFF = none)
  long samples = 0;               // samples drawn (node generator retries included)
  long loop_iterations = 0;
  long num_solutions = 0;
  long rewires = 0;
  double best_cost = std::numeric_limits<double>::infinity();
  SpaceCounters cnt;
};

template <typename Space>
void generate_rrt_star(Space& space, const rkh_rrt_params& prm, long max_loop_iterations, RrtStarResult& res) {
  const int D = space.D;
  const uint32_t NIL = 0xFFFFFFFFu;
  GlobalRng rng(prm.seed);
  res = RrtStarResult();
  res.D = D;
  Point start(prm.start, prm.start + D), goal(prm.goal, prm.goal + D);
  auto add_vertex = [&](const Point& p, double d, uint32_t pr) {
    res.pos.insert(res.pos.end(), p.begin(), p.end());
    res.dist.push_back(d);
    res.pred.push_back(pr);
    res.weight.push_back(0.0);
    return uint32_t(res.pred.size() - 1);
  };
  auto P = [&](uint32_t v) { return Point(res.pos.begin() + std::size_t(v) * D, res.pos.begin() + std::size_t(v + 1) * D); };
  const double inf = std::numeric_limits<double>::infinity();
  add_vertex(start, 0.0, 0);   // put(distance, start, 0.0); put(predecessor, start, start)  (rrt_star.hpp:563-564)
  add_vertex(goal, inf, NIL);  // initialize_vertex (planning_visitors.hpp:136-140)
  std::vector<std::vector<uint32_t>> children(2);
  const double space_dim = double(D);                 // get_space_dimensionality()
  const double gamma = 3.0 * space.metric(start, goal);  // 3 * heuristic(start -> goal), rrtstar_path_planner.tpp:303,322
  unsigned long m_iteration_count = 0;
  auto keep_going = [&]() {
    return (m_iteration_count < prm.max_vertices) && (prm.max_results > (unsigned long)res.num_solutions);
  };
  std::vector<std::pair<double, std::size_t>> nc;
  auto select_neighborhood = [&](const Point& p, std::vector<uint32_t>& out) {  // star_neighborhood::operator()
    std::size_t k;
    double radius;
    star_neighborhood(res.pred.size(), space_dim, gamma, &k, &radius);
    linear_knn(p.data(), res.pos.data(), res.pred.size(), D, k, radius, nc);
    out.clear();
    for (auto& e : nc) out.push_back(uint32_t(e.second));
  };
  // planning_visitor_base::can_be_connected (planning_visitors.hpp:385-395)
  auto can_be_connected = [&](uint32_t u, uint32_t v, double* w) {
    Point p_result;
    Point pu = P(u), pv = P(v);
    double traveled = space.steer(pu, pv, 1.0, p_result);
    double remaining = space.metric(p_result, pv);
    *w = traveled;
    return (!std::isinf(traveled)) && (remaining < prm.conn_tol * traveled);
  };
  std::vector<uint32_t> Nc;
  while (keep_going() && (max_loop_iterations < 0 || res.loop_iterations < max_loop_iterations)) {
    ++res.loop_iterations;
    // ---- rrg_node_generator (node_generators.hpp:137-172)
    Point p_new;
    uint32_t x_near = NIL;
    double eweight = 0.0;
    for (std::size_t i = 0;; ++i) {
      p_new = space.random_point(rng);
      ++res.samples;
      select_neighborhood(p_new, Nc);
      bool was_expanded = false;
      for (uint32_t u : Nc) {  // rrg_node_puller::expand_to_nearest (:61-77)
        Point pu = P(u), p_tmp;
        double traveled = space.steer(pu, p_new, 1.0, p_tmp);
        double best_case = space.metric(pu, p_new);
        bool ok = (!std::isinf(traveled)) && (traveled < 2.0 * best_case) && (traveled > prm.steer_tol * best_case);
        if (ok) {
          p_new = p_tmp;
          x_near = u;
          eweight = traveled;
          was_expanded = true;
          break;
        }
      }
      if (was_expanded) break;
      if (i >= 10) { x_near = NIL; break; }
    }
    res.near_seq.push_back(x_near);
    if (x_near == NIL || res.dist[x_near] == inf) continue;  // rrt_star.hpp:181-182
    // ---- lazy_node_connector::operator() (lazy_connector.hpp:332-372)
    select_neighborhood(p_new, Nc);
    uint32_t v = add_vertex(p_new, inf, NIL);  // rrt_conn_visitor::create_vertex (rrt_star.hpp:112-128)
    children.emplace_back();
    ++m_iteration_count;  // vis.vertex_added -> report_progress
    // vertex_added: dispatched_register_solution for optimal graphs (planning_visitors.hpp:108-116,176-182)
    if (res.pred[1] != NIL && res.dist[1] < res.best_cost) {
      res.best_cost = res.dist[1];  // register_optimal_solution_path_impl (solution_path_factories.hpp:226-270)
      ++res.num_solutions;
    }
    // connect_best_predecessor (:79-123)
    {
      const uint32_t x_near_original = x_near;
      double d_near = res.dist[x_near] + eweight;
      for (uint32_t u : Nc) {
        if (u == x_near_original || res.pred[u] == NIL) continue;
        double tentative_weight = space.metric(P(u), P(v));
        double d_out = tentative_weight + res.dist[u];
        if (d_out < d_near) {
          double w;
          if (can_be_connected(u, v, &w)) {
            x_near = u;
            d_near = d_out;
            eweight = w;
          }
        }
      }
    }
    // pruned_node_connector::create_pred_edge (pruned_connector.hpp:366-382); edge_added returns early (goal node exists)
    res.dist[v] = eweight + res.dist[x_near];
    res.pred[v] = x_near;
    res.weight[v] = eweight;
    children[x_near].push_back(v);
    // connect_successors (:230-275)
    for (uint32_t u : Nc) {
      if (u == x_near) continue;
      double tentative_weight = space.metric(P(v), P(u));
      double d_in = tentative_weight + res.dist[v];
      if (d_in < res.dist[u]) {
        double w;
        if (can_be_connected(v, u, &w)) {
          res.dist[u] = d_in;
          uint32_t old_pred = res.pred[u];
          res.pred[u] = v;
          res.weight[u] = w;
          children[v].push_back(u);
          if (old_pred != u && old_pred != NIL) {  // remove_edge(old_pred, u)
            auto& ch = children[old_pred];
            ch.erase(std::find(ch.begin(), ch.end(), u));
          }
          ++res.rewires;
        }
      }
    }
    // pruned_node_connector::update_successors (pruned_connector.hpp:310-332)
    {
      std::vector<uint32_t> incons(1, v);
      while (!incons.empty()) {
        uint32_t s = incons.back();
        incons.pop_back();
        for (uint32_t t : children[s]) {
          if (res.pred[t] != s) continue;
          res.dist[t] = res.dist[s] + res.weight[t];
          incons.push_back(t);
        }
      }
    }
  }
  res.cnt = space.cnt;
}

}  // namespace oracle

// --------------------------------------------------------------------------------------------
// Bidirectional RRT* (undirected motion graph): generate_rrt_star_bidir (ctrl/graph_alg/rrt_star.hpp:612-659) ->
// generate_rrt_star_bidir_loop (:197-236) with rrg_bidir_generator (node_generators.hpp:215-277: expand_to_nearest
// over the neighbours that have a predecessor, retract_from_nearest over those that have a successor, :84-118) and the
// bidirectional lazy_node_connector::operator() (lazy_connector.hpp:465-518: connect_best_predecessor :79-123,
// connect_best_successor :125-168, create_pred_edge / create_succ_edge pruned_connector.hpp:366-404,
// connect_successors :230-275, update_successors, connect_predecessors :170-227, update_predecessors :338-360).
//
// Reference behaviour kept: (1) the goal vertex has a successor (itself), so connect_successors never gives it a
// predecessor, and vertex_added's solution test (goal.predecessor != null, planning_visitors.hpp:108-116) never fires:
// no solution is registered, the graph grows to max_vertex_count; a vertex with both a predecessor and a successor
// ("joining vertex") is reported here as `joins` / best_join_cost = distance + fwd_distance.  (2) a walk back
// (move_position_back_to) that completes returns its own starting point, so retract_from_nearest only succeeds on walks
// cut short by an obstacle.  Defined here where the reference reads indeterminate memory: the start vertex's successor /
// fwd_distance_accum (bidir_optimal_mg_vertex has no constructor, any_motion_graphs.hpp:255-261) are null / infinity.
// Not modelled: two parallel edges between the same pair of vertices (a predecessor edge and a successor edge), which
// BGL's remove_edge(u, v) / incident-edge loops would treat as one pair.
namespace oracle {

struct BiRrtStarResult {
  int D = 0;
  std::vector<double> pos;
  std::vector<uint32_t> pred, succ;       // 0xFFFFFFFF = none
  std::vector<double> dist, fwd_dist;     // distance_accum, fwd_distance_accum
  std::vector<double> weight, fwd_weight; // weight of the edge (pred -> v) / (v -> succ)
  std::vector<uint32_t> near_pred, near_succ;  // per loop iteration: x_pred / x_succ of the node generator
  long samples = 0, loop_iterations = 0, rewires = 0, fwd_rewires = 0, joins = 0;
  double best_join_cost = std::numeric_limits<double>::infinity();
  SpaceCounters cnt;
};

template <typename Space>
void generate_rrt_star_bidir(Space& space, const rkh_rrt_params& prm, long max_loop_iterations, BiRrtStarResult& res) {
  const int D = space.D;
  const uint32_t NIL = 0xFFFFFFFFu;
  const double inf = std::numeric_limits<double>::infinity();
  GlobalRng rng(prm.seed);
  res = BiRrtStarResult();
  res.D = D;
  Point start(prm.start, prm.start + D), goal(prm.goal, prm.goal + D);
  std::vector<std::vector<uint32_t>> children, parents;  // incident edges by role: v's predecessor-children / successor-parents
  auto add_vertex = [&](const Point& p) {
    res.pos.insert(res.pos.end(), p.begin(), p.end());
    res.dist.push_back(inf);
    res.pred.push_back(NIL);
    res.fwd_dist.push_back(inf);
    res.succ.push_back(NIL);
    res.weight.push_back(0.0);
    res.fwd_weight.push_back(0.0);
    children.emplace_back();
    parents.emplace_back();
    return uint32_t(res.pred.size() - 1);
  };
  auto P = [&](uint32_t v) { return Point(res.pos.begin() + std::size_t(v) * D, res.pos.begin() + std::size_t(v + 1) * D); };
  add_vertex(start);
  add_vertex(goal);
  res.dist[0] = 0.0;      // rrt_star.hpp:647-652
  res.pred[0] = 0;
  res.fwd_dist[1] = 0.0;
  res.succ[1] = 1;
  const double space_dim = double(D);
  const double gamma = 3.0 * space.metric(start, goal);
  unsigned long m_iteration_count = 0;
  auto keep_going = [&]() { return (m_iteration_count < prm.max_vertices) && (prm.max_results > 0ul); };
  std::vector<std::pair<double, std::size_t>> nc;
  auto select_neighborhood = [&](const Point& p, std::vector<uint32_t>& out) {
    std::size_t k;
    double radius;
    star_neighborhood(res.pred.size(), space_dim, gamma, &k, &radius);
    linear_knn(p.data(), res.pos.data(), res.pred.size(), D, k, radius, nc);
    out.clear();
    for (auto& e : nc) out.push_back(uint32_t(e.second));
  };
  auto can_be_connected = [&](uint32_t u, uint32_t v, double* w) {  // planning_visitors.hpp:385-395
    Point p_result;
    Point pu = P(u), pv = P(v);
    double traveled = space.steer(pu, pv, 1.0, p_result);
    double remaining = space.metric(p_result, pv);
    *w = traveled;
    return (!std::isinf(traveled)) && (remaining < prm.conn_tol * traveled);
  };
  std::vector<uint32_t> Nc;
  // lazy_node_connector::operator() (lazy_connector.hpp:465-518)
  auto connect_vertex = [&](const Point& p, uint32_t x_pred, double ep_pred, uint32_t x_succ, double ep_succ) {
    select_neighborhood(p, Nc);
    uint32_t v = add_vertex(p);  // rrt_conn_visitor::create_vertex (rrt_star.hpp:112-128)
    ++m_iteration_count;         // vis.vertex_added -> report_progress
    {  // connect_best_predecessor (:79-123)
      const uint32_t orig = x_pred;
      double d_near = inf;
      if (x_pred != NIL) d_near = res.dist[x_pred] + ep_pred;
      for (uint32_t u : Nc) {
        if (u == orig || res.pred[u] == NIL) continue;
        double d_out = space.metric(P(u), P(v)) + res.dist[u];
        if (d_out < d_near) {
          double w;
          if (can_be_connected(u, v, &w)) {
            x_pred = u;
            d_near = d_out;
            ep_pred = w;
          }
        }
      }
    }
    {  // connect_best_successor (:125-168)
      const uint32_t orig = x_succ;
      double d_near = inf;
      if (x_succ != NIL) d_near = res.fwd_dist[x_succ] + ep_succ;
      for (uint32_t u : Nc) {
        if (u == orig || res.succ[u] == NIL) continue;
        double d_in = space.metric(P(v), P(u)) + res.fwd_dist[u];
        if (d_in < d_near) {
          double w;
          if (can_be_connected(v, u, &w)) {
            x_succ = u;
            d_near = d_in;
            ep_succ = w;
          }
        }
      }
    }
    if (x_pred == NIL && x_succ == NIL) return;  // (the vertex would be removed; unreachable from the loop below)
    if (x_pred != NIL) {  // create_pred_edge
      res.dist[v] = ep_pred + res.dist[x_pred];
      res.pred[v] = x_pred;
      res.weight[v] = ep_pred;
      children[x_pred].push_back(v);
    }
    if (x_succ != NIL) {  // create_succ_edge
      res.fwd_dist[v] = ep_succ + res.fwd_dist[x_succ];
      res.succ[v] = x_succ;
      res.fwd_weight[v] = ep_succ;
      parents[x_succ].push_back(v);
    }
    if (res.pred[v] != NIL && res.succ[v] != NIL) {
      ++res.joins;
      if (res.dist[v] + res.fwd_dist[v] < res.best_join_cost) res.best_join_cost = res.dist[v] + res.fwd_dist[v];
    }
    // connect_successors (:230-275, with the successor map: vertices of the backward tree are left alone)
    for (uint32_t u : Nc) {
      if (u == x_pred || res.succ[u] != NIL) continue;
      double d_in = space.metric(P(v), P(u)) + res.dist[v];
      if (d_in < res.dist[u]) {
        double w;
        if (can_be_connected(v, u, &w)) {
          res.dist[u] = d_in;
          uint32_t old_pred = res.pred[u];
          res.pred[u] = v;
          res.weight[u] = w;
          children[v].push_back(u);
          if (old_pred != u && old_pred != NIL) {
            auto& ch = children[old_pred];
            ch.erase(std::find(ch.begin(), ch.end(), u));
          }
          ++res.rewires;
        }
      }
    }
    {  // update_successors (pruned_connector.hpp:310-332)
      std::vector<uint32_t> incons(1, v);
      while (!incons.empty()) {
        uint32_t s = incons.back();
        incons.pop_back();
        for (uint32_t t : children[s]) {
          if (res.pred[t] != s) continue;
          res.dist[t] = res.dist[s] + res.weight[t];
          incons.push_back(t);
        }
      }
    }
    // connect_predecessors (:170-227, with the predecessor map: vertices of the forward tree are left alone)
    for (uint32_t u : Nc) {
      if (u == x_succ || res.pred[u] != NIL) continue;
      double d_in = space.metric(P(u), P(v)) + res.fwd_dist[v];
      if (d_in < res.fwd_dist[u]) {
        double w;
        if (can_be_connected(u, v, &w)) {
          res.fwd_dist[u] = d_in;
          uint32_t old_succ = res.succ[u];
          res.succ[u] = v;
          res.fwd_weight[u] = w;
          parents[v].push_back(u);
          if (old_succ != u && old_succ != NIL) {
            auto& pa = parents[old_succ];
            pa.erase(std::find(pa.begin(), pa.end(), u));
          }
          ++res.fwd_rewires;
        }
      }
    }
    {  // update_predecessors (pruned_connector.hpp:338-360)
      std::vector<uint32_t> incons(1, v);
      while (!incons.empty()) {
        uint32_t t = incons.back();
        incons.pop_back();
        for (uint32_t s : parents[t]) {
          if (res.succ[s] != t) continue;
          res.fwd_dist[s] = res.fwd_dist[t] + res.fwd_weight[s];
          incons.push_back(s);
        }
      }
    }
  };
  while (keep_going() && (max_loop_iterations < 0 || res.loop_iterations < max_loop_iterations)) {
    ++res.loop_iterations;
    // ---- rrg_bidir_generator (node_generators.hpp:244-277)
    Point p_pred, p_succ;
    uint32_t x_pred = NIL, x_succ = NIL;
    double ep_pred = 0.0, ep_succ = 0.0;
    for (std::size_t i = 0;; ++i) {
      p_pred = space.random_point(rng);
      ++res.samples;
      p_succ = p_pred;
      select_neighborhood(p_pred, Nc);
      bool was_expanded = false, was_retracted = false;
      x_pred = NIL;
      x_succ = NIL;
      for (uint32_t u : Nc) {  // expand_to_nearest (:84-100)
        if (res.pred[u] == NIL) continue;
        Point pu = P(u), p_tmp;
        double traveled = space.steer(pu, p_pred, 1.0, p_tmp);
        double best_case = space.metric(pu, p_pred);
        if ((!std::isinf(traveled)) && (traveled < 2.0 * best_case) && (traveled > prm.steer_tol * best_case)) {
          p_pred = p_tmp;
          x_pred = u;
          ep_pred = traveled;
          was_expanded = true;
          break;
        }
      }
      for (uint32_t u : Nc) {  // retract_from_nearest (:102-118); steer_back_to_position planning_visitors.hpp:367-378
        if (res.succ[u] == NIL) continue;
        Point pu = P(u), p_tmp;
        double traveled = space.steer_back(p_succ, pu, 1.0, p_tmp);
        double best_case = space.metric(p_succ, pu);
        if ((!std::isinf(traveled)) && (traveled < 2.0 * best_case) && (traveled > prm.steer_tol * best_case)) {
          p_succ = p_tmp;
          x_succ = u;
          ep_succ = traveled;
          was_retracted = true;
          break;
        }
      }
      if (was_expanded || was_retracted) break;
      if (i >= 10) { x_pred = NIL; x_succ = NIL; break; }
    }
    res.near_pred.push_back(x_pred);
    res.near_succ.push_back(x_succ);
    if (x_pred != NIL) connect_vertex(p_pred, x_pred, ep_pred, NIL, 0.0);
    if (x_succ != NIL) connect_vertex(p_succ, NIL, 0.0, x_succ, ep_succ);
  }
  res.cnt = space.cnt;
}

}  // namespace oracle

// --------------------------------------------------------------------------------------------
// PRM (LINEAR_SEARCH_KNN, ADJ_LIST_MOTION_GRAPH, undirected motion graph):
// prm_planner::solve_planning_query (ctrl/path_planning/prm_path_planner.tpp:131-365)
//  -> generate_prm (ctrl/graph_alg/probabilistic_roadmap.hpp:309-404) -> generate_prm_impl (:211-249)
// with prm_node_connector (prm_connector.hpp:68-182), prm_conn_visitor (probabilistic_roadmap.hpp:75-196),
// density_plan_visitor<prm_density_calculator> (density_plan_visitors.hpp:50-224, density_calculators.hpp:45-73),
// random_walk (planning_visitors.hpp:403-432) and star_neighborhood.
//
// Reference behaviour kept: solve_planning_query instantiates density_plan_visitor, not prm_planner_visitor, so
// publish_path is planning_visitor_base's (planning_visitors.hpp:119-126): it tests the goal vertex's
// distance_accum, which nothing on this path ever updates from infinity -- no solution is registered and the
// roadmap grows until max_vertex_count.  (We report the start/goal component merge separately.)
//
// Third-party pieces restated from their published definitions ("parity unpinned", sources not in
// /root/reference): boost::d_ary_heap_indirect<V,4,...,std::less<double>> (boost/graph/detail/d_ary_heap.hpp:
// push, push_or_update = insert-or-sift-UP-only, pop, top); BGL-Extra adjacency_list_BC out_edges order, taken
// here as the insertion order of a vertex's incident edges.
namespace oracle {

// boost::d_ary_heap_indirect, Arity 4, keys read through the density map at comparison time
struct DAryHeap4 {
  std::vector<uint32_t> data;
  std::vector<std::size_t>* index_in_heap = nullptr;
  const std::vector<double>* key = nullptr;
  bool greater = false;  // std::greater<double> as the compare type: the top is the LARGEST key (branch_and_bound_connector)
  bool before(double a, double b) const { return greater ? a > b : a < b; }
  static std::size_t parent(std::size_t i) { return (i - 1) / 4; }
  std::size_t& idx(uint32_t v) {
    if (index_in_heap->size() <= v) index_in_heap->resize(v + 1, 0);  // vector_property_map grows with value 0
    return (*index_in_heap)[v];
  }
  bool empty() const { return data.empty(); }
  uint32_t top() const { return data[0]; }
  void preserve_heap_property_up(std::size_t index) {
    std::size_t orig_index = index, num_levels_moved = 0;
    if (index == 0) return;
    uint32_t moving = data[index];
    double moving_dist = (*key)[moving];
    for (;;) {
      if (index == 0) break;
      std::size_t parent_index = parent(index);
      uint32_t parent_value = data[parent_index];
      if (before(moving_dist, (*key)[parent_value])) {
        ++num_levels_moved;
        index = parent_index;
        continue;
      } else {
        break;
      }
    }
    index = orig_index;
    for (std::size_t i = 0; i < num_levels_moved; ++i) {
      std::size_t parent_index = parent(index);
      uint32_t parent_value = data[parent_index];
      idx(parent_value) = index;
      data[index] = parent_value;
      index = parent_index;
    }
    data[index] = moving;
    idx(moving) = index;
  }
  void preserve_heap_property_down() {
    if (data.empty()) return;
    std::size_t index = 0;
    uint32_t moving = data[0];
    double moving_dist = (*key)[moving];
    std::size_t heap_size = data.size();
    for (;;) {
      std::size_t first_child = index * 4 + 1;
      if (first_child >= heap_size) break;
      std::size_t smallest_child = 0;
      double smallest_dist = (*key)[data[first_child]];
      std::size_t n_children = (first_child + 4 <= heap_size) ? 4 : heap_size - first_child;
      for (std::size_t i = 1; i < n_children; ++i) {
        double i_dist = (*key)[data[first_child + i]];
        if (before(i_dist, smallest_dist)) {
          smallest_child = i;
          smallest_dist = i_dist;
        }
      }
      if (before(smallest_dist, moving_dist)) {
        std::size_t c = first_child + smallest_child;  // swap_heap_elements(c, index)
        uint32_t va = data[c], vb = data[index];
        data[c] = vb;
        data[index] = va;
        idx(va) = index;
        idx(vb) = c;
        index = c;
        continue;
      } else {
        break;
      }
    }
  }
  void push(uint32_t v) {
    std::size_t index = data.size();
    data.push_back(v);
    idx(v) = index;
    preserve_heap_property_up(index);
  }
  void push_or_update(uint32_t v) {  // insert if absent; in both cases only sift up
    std::size_t index = idx(v);
    if (index == std::size_t(-1)) {
      index = data.size();
      data.push_back(v);
      idx(v) = index;
    }
    preserve_heap_property_up(index);
  }
  void pop() {
    idx(data[0]) = std::size_t(-1);
    if (data.size() != 1) {
      data[0] = data.back();
      idx(data[0]) = 0;
      data.pop_back();
      preserve_heap_property_down();
    } else {
      data.pop_back();
    }
  }
};

struct PrmResult {
  int D = 0;
  std::vector<double> pos;           // vertex 0 = start, 1 = goal
  std::vector<uint32_t> edge_u, edge_v;
  std::vector<double> edge_w;
  std::vector<double> density;
  std::vector<uint32_t> cc_root;     // raw union-find parents at the end
  std::vector<uint8_t> kind;         // per loop iteration: 0 construct, 1 expand (vertex added), 2 expand failed (Q.pop)
  std::vector<uint32_t> expanded;    // per loop iteration: Q.top() of expansion iterations, else 0xFFFFFFFF
  long samples = 0, rejected = 0, loop_iterations = 0, num_components = 0;
  long publish_calls = 0;            // times cc_set.size() < 2 triggered publish_path
  long merged_at_vertex = -1;        // vertex count when start and goal first shared a component
  SpaceCounters cnt;
};

template <typename Space>
void generate_prm(Space& space, const rkh_prm_params& pp, long max_loop_iterations, PrmResult& res) {
  const rkh_rrt_params& prm = pp.base;
  const int D = space.D;
  const uint32_t NIL = 0xFFFFFFFFu;
  GlobalRng rng(prm.seed);
  res = PrmResult();
  res.D = D;
  Point start(prm.start, prm.start + D), goal(prm.goal, prm.goal + D);
  std::vector<std::vector<uint32_t>> incident;  // per vertex: incident edge ids, insertion order
  std::vector<std::size_t> index_in_heap;
  DAryHeap4 Q;
  Q.index_in_heap = &index_in_heap;
  Q.key = &res.density;
  std::vector<uint32_t>& cc_root = res.cc_root;
  std::set<uint32_t> cc_set;
  const double space_dim = double(D);
  const double gamma = 3.0 * space.metric(start, goal);
  const double sampling_radius = pp.sampling_radius;
  unsigned long m_iteration_count = 0;
  auto P = [&](uint32_t v) { return Point(res.pos.begin() + std::size_t(v) * D, res.pos.begin() + std::size_t(v + 1) * D); };
  auto keep_going = [&]() { return (m_iteration_count < prm.max_vertices) && (prm.max_results > 0ul); };
  // prm_density_calculator::update_density (density_calculators.hpp:55-72)
  auto update_density = [&](uint32_t u) {
    std::size_t deg_u = incident[u].size();
    if (deg_u == 0) {
      res.density[u] = 0.0;
      return;
    }
    std::size_t max_node_degree = std::size_t(D) + 1;
    double sum = 0.0;
    for (uint32_t e : incident[u]) sum += res.edge_w[e] / sampling_radius;
    sum /= double(deg_u) * double(deg_u) / double(max_node_degree);
    res.density[u] = std::exp(-sum * sum);
  };
  auto raw_add_vertex = [&](const Point& p) {
    res.pos.insert(res.pos.end(), p.begin(), p.end());
    res.density.push_back(0.0);
    incident.emplace_back();
    cc_root.push_back(0);
    return uint32_t(res.density.size() - 1);
  };
  // prm_conn_visitor::requeue_vertex / affected_vertex (probabilistic_roadmap.hpp:160-166)
  auto requeue = [&](uint32_t u) {
    update_density(u);  // density_plan_visitor::affected_vertex -> init_nonrecursive_density
    Q.push_or_update(u);
  };
  auto shortcut_cc_root = [&](uint32_t u) {  // :109-121
    std::vector<uint32_t> trace(1, u);
    while (cc_root[u] != u) {
      u = cc_root[u];
      trace.push_back(u);
    }
    while (!trace.empty()) {
      cc_root[trace.back()] = u;
      trace.pop_back();
    }
  };
  auto edge_added = [&](uint32_t u, uint32_t v) {  // :123-143 (m_vis.edge_added returns early: the goal node exists)
    shortcut_cc_root(u);
    shortcut_cc_root(v);
    if (cc_root[v] != cc_root[u]) {
      uint32_t r1 = cc_root[u], r2 = cc_root[v];
      cc_root[r2] = r1;
      cc_root[v] = r1;
      cc_set.erase(r2);
      if (cc_set.size() < 2) ++res.publish_calls;  // publish_path: registers nothing (see header)
    }
    if (res.merged_at_vertex < 0) {
      uint32_t a = 0, b = 1;
      while (cc_root[a] != a) a = cc_root[a];
      while (cc_root[b] != b) b = cc_root[b];
      if (a == b) res.merged_at_vertex = long(res.density.size());
    }
  };
  auto add_edge = [&](uint32_t u, uint32_t v, double w) {
    uint32_t e = uint32_t(res.edge_w.size());
    res.edge_u.push_back(u);
    res.edge_v.push_back(v);
    res.edge_w.push_back(w);
    incident[u].push_back(e);
    incident[v].push_back(e);
    edge_added(u, v);
  };
  // ---- solve_planning_query: start and goal vertices (RK_PRM_PLANNER_INITIALIZE_START_AND_GOAL)
  raw_add_vertex(start);
  raw_add_vertex(goal);
  // ---- generate_prm with a non-empty graph (:368-397)
  for (uint32_t u = 0; u < 2; ++u) {
    update_density(u);  // vis.affected_vertex
    Q.push(u);
    cc_root[u] = u;
  }
  for (uint32_t u = 0; u < 2; ++u) cc_set.insert(u);  // no edges yet: every vertex is its own component
  std::vector<std::pair<double, std::size_t>> nc;
  // planning_visitor_base::can_be_connected (planning_visitors.hpp:385-395)
  auto can_be_connected = [&](uint32_t u, uint32_t v, double* w) {
    Point p_result;
    Point pu = P(u), pv = P(v);
    double traveled = space.steer(pu, pv, 1.0, p_result);
    double remaining = space.metric(p_result, pv);
    *w = traveled;
    return (!std::isinf(traveled)) && (remaining < prm.conn_tol * traveled);
  };
  // prm_node_connector::operator(), undirected (prm_connector.hpp:136-182)
  auto connect_vertex = [&](const Point& p, uint32_t x_near, double eweight) {
    std::size_t k;
    double radius;
    star_neighborhood(res.density.size(), space_dim, gamma, &k, &radius);
    linear_knn(p.data(), res.pos.data(), res.density.size(), D, k, radius, nc);
    std::vector<uint32_t> Nc;
    for (auto& e : nc) Nc.push_back(uint32_t(e.second));
    // prm_conn_visitor::create_vertex (:92-107)
    uint32_t v = raw_add_vertex(p);
    cc_root[v] = v;
    cc_set.insert(v);
    update_density(v);   // vertex_added -> initialize_vertex -> init_nonrecursive_density
    ++m_iteration_count;  // report_progress; dispatched_register_solution: goal distance_accum is infinite
    if (index_in_heap.size() <= v) index_in_heap.resize(v + 1, 0);
    index_in_heap[v] = std::size_t(-1);
    if (x_near != NIL) {  // connect_to_first_pred (:71-91)
      add_edge(x_near, v, eweight);
      requeue(x_near);
    }
    requeue(v);
    for (uint32_t u : Nc) {
      if (u == x_near) continue;
      double w;
      bool can_connect = can_be_connected(u, v, &w);
      if (can_connect) add_edge(u, v, w);
      requeue(u);  // affected by travel attempts
    }
    requeue(v);
  };
  while (keep_going() && (max_loop_iterations < 0 || res.loop_iterations < max_loop_iterations)) {
    ++res.loop_iterations;
    double rand_value = rng.uniform_01();
    if (rand_value > pp.expand_probability) {
      // construction node (:229-235)
      Point p_rnd = space.random_point(rng);
      ++res.samples;
      while (!space.is_free(p_rnd)) {
        ++res.rejected;
        p_rnd = space.random_point(rng);
        ++res.samples;
      }
      connect_vertex(p_rnd, NIL, 0.0);
      res.kind.push_back(0);
      res.expanded.push_back(NIL);
    } else {
      // expansion node (:237-246); random_walk (planning_visitors.hpp:403-432)
      uint32_t v = Q.top();
      res.expanded.push_back(v);
      Point pv = P(v);
      Point origin(D);  // hyperbox_topology::origin (hyperbox_topology.hpp:194-196)
      for (int i = 0; i < D; ++i) origin[i] = space.lower[i] + 0.5 * (space.upper[i] - space.lower[i]);
      unsigned int i = 0;
      Point p_rnd = space.random_point(rng);
      ++res.samples;
      Point dp_rnd(D);
      for (int d = 0; d < D; ++d) dp_rnd[d] = p_rnd[d] - origin[d];
      Point p_result;
      bool worked = false;
      double w = 0.0;
      do {
        for (int d = 0; d < D; ++d) p_rnd[d] = pv[d] + dp_rnd[d];
        double dist = space.metric(pv, p_rnd);
        double target_dist = rng.uniform_01() * sampling_radius;
        double traveled = space.steer(pv, p_rnd, target_dist / dist, p_result);
        if ((!std::isinf(traveled)) && (traveled > prm.steer_tol * target_dist)) {
          worked = true;
          w = traveled;
          break;
        } else {
          p_rnd = space.random_point(rng);
          ++res.samples;
          for (int d = 0; d < D; ++d) dp_rnd[d] = p_rnd[d] - origin[d];
        }
      } while (++i <= 10);
      if (worked) {
        connect_vertex(p_result, v, w);
        res.kind.push_back(1);
      } else {
        Q.pop();
        res.kind.push_back(2);
      }
    }
  }
  res.num_components = long(cc_set.size());
  res.cnt = space.cnt;
}

}  // namespace oracle

// --------------------------------------------------------------------------------------------
// Bidirectional RRT: generate_bidirectional_rrt (ctrl/graph_alg/rr_tree.hpp:256-317) with expand_rrt_vertex
// (:86-112), planning_visitor_base::steer_towards_position / joining_vertex_found (planning_visitors.hpp:349-360,
// 223-231) and register_basic_solution_path_impl for two graphs (solution_path_factories.hpp:359-408).
// Tree 1 grows from the start, tree 2 from the goal; both roots exist before the loop (rrt_path_planner.tpp), so they
// are not counted by m_iteration_count.  vertex_added / edge_added do not probe the goal in a bidirectional planner
// (planning_visitors.hpp:174-176,189-192).
namespace oracle {

struct BiRrtResult {
  int D = 0;
  std::vector<double> pos[2];        // tree 1 (root = start), tree 2 (root = goal)
  std::vector<uint32_t> parent[2];   // root: 0xFFFFFFFF
  std::vector<uint32_t> nn_seq;      // per expansion (two per loop iteration): nearest vertex u
  std::vector<uint8_t> accept;       // per expansion: reached_new
  long loop_iterations = 0, samples = 0, num_solutions = 0, joins = 0;
  double best_cost = std::numeric_limits<double>::infinity();
  SpaceCounters cnt;
};

template <typename Space>
void generate_bidirectional_rrt(Space& space, const rkh_rrt_params& prm, long max_loop_iterations, BiRrtResult& res) {
  const int D = space.D;
  const uint32_t NIL = 0xFFFFFFFFu;
  GlobalRng rng(prm.seed);
  res = BiRrtResult();
  res.D = D;
  Point start(prm.start, prm.start + D), goal(prm.goal, prm.goal + D);
  res.pos[0].assign(start.begin(), start.end());
  res.parent[0].push_back(NIL);
  res.pos[1].assign(goal.begin(), goal.end());
  res.parent[1].push_back(NIL);
  unsigned long m_iteration_count = 0;
  auto keep_going = [&]() {
    return (m_iteration_count < prm.max_vertices) && (prm.max_results > (unsigned long)res.num_solutions);
  };
  auto P = [&](int t, uint32_t v) {
    return Point(res.pos[t].begin() + std::size_t(v) * D, res.pos[t].begin() + std::size_t(v + 1) * D);
  };
  // detail::expand_rrt_vertex on tree t towards p_target: returns (vertex, reached_new)
  auto expand = [&](int t, const Point& p_target) {
    std::size_t n = res.parent[t].size();
    uint32_t u = uint32_t(linear_nn(p_target.data(), res.pos[t].data(), n, D));
    res.nn_seq.push_back(u);
    Point pu = P(t, u), p_v;
    double traveled = space.steer(pu, p_target, 1.0, p_v);
    double best_case = space.metric(pu, p_target);
    bool reached_new = (!std::isinf(traveled)) && (traveled < 2.0 * best_case) && (traveled > prm.steer_tol * best_case);
    res.accept.push_back(reached_new ? 1 : 0);
    if (!reached_new) return std::make_pair(u, false);
    res.pos[t].insert(res.pos[t].end(), p_v.begin(), p_v.end());
    res.parent[t].push_back(u);
    ++m_iteration_count;  // vis.vertex_added -> report_progress
    return std::make_pair(uint32_t(res.parent[t].size() - 1), true);
  };
  // joining_vertex_found -> register_joining_point -> register_basic_solution_path_impl (two graphs)
  auto joining_vertex_found = [&](uint32_t u1, uint32_t u2) {
    ++res.joins;
    double total = space.metric(P(0, u1), P(1, u2));
    uint32_t j1 = u1;
    while (res.parent[0][j1] != NIL) {
      uint32_t v = res.parent[0][j1];
      total += space.metric(P(0, v), P(0, j1));
      j1 = v;
    }
    uint32_t j2 = u2;
    while (res.parent[1][j2] != NIL) {
      uint32_t v = res.parent[1][j2];
      total += space.metric(P(1, j2), P(1, v));
      j2 = v;
    }
    if (res.num_solutions == 0 || total < res.best_cost) {
      res.best_cost = total;
      ++res.num_solutions;
    }
  };
  std::pair<uint32_t, bool> v_target2(0u, true);
  Point p_target2 = P(0, 0);
  std::pair<uint32_t, bool> v_target1(0u, true);
  Point p_target1 = P(1, 0);
  while (keep_going() && (max_loop_iterations < 0 || res.loop_iterations < max_loop_iterations)) {
    ++res.loop_iterations;
    // first, expand the first graph towards its target (:282-299)
    {
      std::size_t n_before = res.parent[0].size();
      std::pair<uint32_t, bool> v1 = expand(0, p_target1);
      (void)n_before;
      if (v1.second && v_target1.second) {
        joining_vertex_found(v1.first, v_target1.first);
        p_target2 = space.random_point(rng);
        ++res.samples;
        v_target2.second = false;
      } else if (!v1.second) {  // v1.first == u1: unsuccessful expansion
        p_target2 = space.random_point(rng);
        ++res.samples;
        v_target2.second = false;
      } else {
        p_target2 = P(0, v1.first);
        v_target2 = std::make_pair(v1.first, true);
      }
    }
    // then, expand the second graph towards its target (:301-318)
    {
      std::pair<uint32_t, bool> v2 = expand(1, p_target2);
      if (v2.second && v_target2.second) {
        joining_vertex_found(v_target2.first, v2.first);
        p_target1 = space.random_point(rng);
        ++res.samples;
        v_target1.second = false;
      } else if (!v2.second) {
        p_target1 = space.random_point(rng);
        ++res.samples;
        v_target1.second = false;
      } else {
        p_target1 = P(1, v2.first);
        v_target1 = std::make_pair(v2.first, true);
      }
    }
  }
  res.cnt = space.cnt;
}

}  // namespace oracle

// --------------------------------------------------------------------------------------------
// RRT* with branch-and-bound pruning (USE_BRANCH_AND_BOUND_PRUNING_FLAG): generate_bnb_rrt_star
// (ctrl/graph_alg/rrt_star.hpp:690-730) = generate_rrt_star_loop (:169-190) with branch_and_bound_connector
// (ctrl/graph_alg/branch_and_bound_connector.hpp:105-330) in the place of lazy_node_connector: a new point that cannot
// improve on the best solution is dropped before (:287-293) or after (:311-317) its predecessor is chosen, every kept
// vertex sits in a 4-ary max-heap keyed distance_accum + distance to the goal, update_successors re-keys the vertices
// whose cost changed (push_or_update: sift-up only, also when a key went down) and then removes every vertex whose key
// exceeds the goal's cost (:174-185).
//
// Defined here where the reference leaves things open ("parity unpinned"): (1) vertex ids are append-only -- a removed
// vertex becomes a tombstone; BGL-Extra's pooled vertex container would hand its slot to the next vertex;
// (2) the pruning loop reads Q.top() again after the last pop; it stops here when the heap is empty; (3) clear_vertex
// drops the edges of a removed vertex but its children keep their predecessor field and their cost: they stay candidate
// parents exactly as in the reference, where the dangling descriptor is only ever compared, not followed.
namespace oracle {

struct BnbRrtStarResult {
  RrtStarResult g;
  std::vector<uint8_t> removed;  // per vertex
  long pruned = 0;               // vertices removed by the pruning loop or right after their creation
  long skipped = 0;              // points dropped before a vertex was created
};

template <typename Space>
void generate_bnb_rrt_star(Space& space, const rkh_rrt_params& prm, long max_loop_iterations, BnbRrtStarResult& out) {
  const int D = space.D;
  const uint32_t NIL = 0xFFFFFFFFu;
  GlobalRng rng(prm.seed);
  out = BnbRrtStarResult();
  RrtStarResult& res = out.g;
  res.D = D;
  Point start(prm.start, prm.start + D), goal(prm.goal, prm.goal + D);
  std::vector<std::vector<uint32_t>> children;
  std::vector<double> key;
  std::vector<std::size_t> index_in_heap;
  DAryHeap4 Q;
  Q.index_in_heap = &index_in_heap;
  Q.key = &key;
  Q.greater = true;
  auto add_vertex = [&](const Point& p, double d, uint32_t pr) {
    res.pos.insert(res.pos.end(), p.begin(), p.end());
    res.dist.push_back(d);
    res.pred.push_back(pr);
    res.weight.push_back(0.0);
    children.emplace_back();
    out.removed.push_back(0);
    key.push_back(0.0);
    uint32_t v = uint32_t(res.pred.size() - 1);
    Q.idx(v) = std::size_t(-1);
    return v;
  };
  auto P = [&](uint32_t v) { return Point(res.pos.begin() + std::size_t(v) * D, res.pos.begin() + std::size_t(v + 1) * D); };
  const double inf = std::numeric_limits<double>::infinity();
  add_vertex(start, 0.0, 0);
  add_vertex(goal, inf, NIL);
  const double space_dim = double(D);
  const double gamma = 3.0 * space.metric(start, goal);
  unsigned long m_iteration_count = 0;
  auto keep_going = [&]() {
    return (m_iteration_count < prm.max_vertices) && (prm.max_results > (unsigned long)res.num_solutions);
  };
  std::vector<std::pair<double, std::size_t>> nc;
  auto select_neighborhood = [&](const Point& p, std::vector<uint32_t>& o) {
    std::size_t k;
    double radius;
    std::size_t live = res.pred.size() - std::size_t(out.pruned);  // num_vertices(g)
    star_neighborhood(live, space_dim, gamma, &k, &radius);
    linear_knn(p.data(), res.pos.data(), res.pred.size(), D, k, radius, nc, &out.removed);
    o.clear();
    for (auto& e : nc) o.push_back(uint32_t(e.second));
  };
  auto can_be_connected = [&](uint32_t u, uint32_t v, double* w) {
    Point p_result;
    Point pu = P(u), pv = P(v);
    double traveled = space.steer(pu, pv, 1.0, p_result);
    double remaining = space.metric(p_result, pv);
    *w = traveled;
    return (!std::isinf(traveled)) && (remaining < prm.conn_tol * traveled);
  };
  auto remove_vertex = [&](uint32_t v) {  // vertex_to_be_removed; clear_vertex; remove_vertex
    out.removed[v] = 1;
    ++out.pruned;
    uint32_t pv = res.pred[v];
    if (pv != NIL && pv != v && !out.removed[pv]) {
      auto& ch = children[pv];
      auto it = std::find(ch.begin(), ch.end(), v);
      if (it != ch.end()) ch.erase(it);
    }
    children[v].clear();
  };
  std::vector<uint32_t> Nc;
  while (keep_going() && (max_loop_iterations < 0 || res.loop_iterations < max_loop_iterations)) {
    ++res.loop_iterations;
    // ---- rrg_node_generator (node_generators.hpp:137-172)
    Point p_new;
    uint32_t x_near = NIL;
    double eweight = 0.0;
    for (std::size_t i = 0;; ++i) {
      p_new = space.random_point(rng);
      ++res.samples;
      select_neighborhood(p_new, Nc);
      bool was_expanded = false;
      for (uint32_t u : Nc) {
        Point pu = P(u), p_tmp;
        double traveled = space.steer(pu, p_new, 1.0, p_tmp);
        double best_case = space.metric(pu, p_new);
        bool ok = (!std::isinf(traveled)) && (traveled < 2.0 * best_case) && (traveled > prm.steer_tol * best_case);
        if (ok) {
          p_new = p_tmp;
          x_near = u;
          eweight = traveled;
          was_expanded = true;
          break;
        }
      }
      if (was_expanded) break;
      if (i >= 10) { x_near = NIL; break; }
    }
    res.near_seq.push_back(x_near);
    if (x_near == NIL || res.dist[x_near] == inf) continue;
    // ---- branch_and_bound_connector::operator() (:277-330)
    const double dist_from_start = space.metric(P(0), p_new);
    const double dist_to_goal = space.metric(p_new, P(1));
    if (res.pred[1] != NIL && dist_from_start + dist_to_goal > res.dist[1]) {
      ++out.skipped;
      continue;
    }
    select_neighborhood(p_new, Nc);
    uint32_t v = add_vertex(p_new, inf, NIL);
    ++m_iteration_count;
    if (res.pred[1] != NIL && res.dist[1] < res.best_cost) {
      res.best_cost = res.dist[1];
      ++res.num_solutions;
    }
    {  // connect_best_predecessor
      const uint32_t x_near_original = x_near;
      double d_near = res.dist[x_near] + eweight;
      for (uint32_t u : Nc) {
        if (u == x_near_original || res.pred[u] == NIL) continue;
        double d_out = space.metric(P(u), P(v)) + res.dist[u];
        if (d_out < d_near) {
          double w;
          if (can_be_connected(u, v, &w)) {
            x_near = u;
            d_near = d_out;
            eweight = w;
          }
        }
      }
    }
    res.dist[v] = eweight + res.dist[x_near];  // create_pred_edge
    res.pred[v] = x_near;
    res.weight[v] = eweight;
    children[x_near].push_back(v);
    if (res.pred[1] != NIL && res.dist[v] + dist_to_goal > res.dist[1]) {  // :311-317
      remove_vertex(v);
      continue;
    }
    key[v] = res.dist[v] + dist_to_goal;
    Q.push(v);
    for (uint32_t u : Nc) {  // connect_successors (lazy_connector.hpp:230-275)
      if (u == x_near) continue;
      double d_in = space.metric(P(v), P(u)) + res.dist[v];
      if (d_in < res.dist[u]) {
        double w;
        if (can_be_connected(v, u, &w)) {
          res.dist[u] = d_in;
          uint32_t old_pred = res.pred[u];
          res.pred[u] = v;
          res.weight[u] = w;
          children[v].push_back(u);
          if (old_pred != u && old_pred != NIL && !out.removed[old_pred]) {
            auto& ch = children[old_pred];
            auto it = std::find(ch.begin(), ch.end(), u);
            if (it != ch.end()) ch.erase(it);
          }
          ++res.rewires;
        }
      }
    }
    {  // branch_and_bound_connector::update_successors (:142-185)
      std::vector<uint32_t> incons(1, v);
      while (!incons.empty()) {
        uint32_t s = incons.back();
        incons.pop_back();
        for (uint32_t t : children[s]) {
          if (res.pred[t] != s) continue;
          res.dist[t] = res.dist[s] + res.weight[t];
          key[t] = res.dist[t] + space.metric(P(t), P(1));
          Q.push_or_update(t);
          incons.push_back(t);
        }
      }
      if (res.pred[1] != NIL) {  // prune all the worst nodes
        while (!Q.empty() && key[Q.top()] > res.dist[1]) {
          remove_vertex(Q.top());
          Q.pop();
        }
      }
    }
  }
  res.cnt = space.cnt;
}

}  // namespace oracle
