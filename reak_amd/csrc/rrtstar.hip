// rrtstar.hip -- RRT* (unidirectional, LINEAR_SEARCH_KNN, undirected motion graph) over the quasi-static free
// space, for a batch of P independent problems:
//   rrtstar_planner::solve_planning_query_impl (ctrl/path_planning/rrtstar_path_planner.tpp:298-)
//   -> generate_rrt_star (ctrl/graph_alg/rrt_star.hpp:530-570) -> generate_rrt_star_loop (:169-190)
//   with rrg_node_generator (node_generators.hpp:137-172), star_neighborhood (neighborhood_functors.hpp:95-102),
//   lazy_node_connector (lazy_connector.hpp:79-123,230-275,332-372), pruned_node_connector::{create_pred_edge,
//   update_successors} (pruned_connector.hpp:310-332,366-382).
//
// Rewiring makes every loop iteration depend on the previous one, so iterations stay sequential; what runs on the
// device is the work *inside* an iteration, batched:
//   * the two k-NN sweeps of an iteration (knn_sweep.hip; HBM-bound at one query per tree),
//   * every candidate edge of the node generator (steer from each neighbour, the first success in order wins) and of
//     the connector (can_be_connected in both directions for every neighbour) in one edge_check launch each; the
//     sequential rules (running d_near, strict comparisons, neighbour order) are then applied on the host to the
//     precomputed verdicts, which are independent of one another.
// Every problem is a two-state machine (GENERATE: sample + k-NN + steer candidates; CONNECT: append the vertex +
// k-NN + can_be_connected candidates) and each device step serves the next state of all P problems at once
// (graph_batch.h: table-driven launches, one command upload and one result download per step).
// Bookkeeping (predecessors, accumulated distances, children lists, the DFS of update_successors) is host code.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <random>

#include "graph_batch.h"
#include "rkh_internal.h"

using namespace rkh;

namespace {
constexpr uint32_t NIL = 0xFFFFFFFFu;

// math::highest_set_bit (core/base/misc_math.hpp:50-59) and star_neighborhood::operator()
// (ctrl/graph_alg/neighborhood_functors.hpp:95-102): integer log2 and the fp64 radius, host side
size_t highest_set_bit(size_t N) {
  size_t temp = 0;
  for (size_t shift = sizeof(size_t) * 4; (shift && (N != 1)); shift >>= 1) {
    if (N >> shift) {
      temp |= shift;
      N >>= shift;
    }
  }
  return temp;
}

enum StarState { ST_IDLE, ST_GENERATE, ST_CONNECT };

struct StarProblem {
  rkh_rrt_params prm;
  std::mt19937 eng;
  // motion graph (host)
  std::vector<double> pos;
  std::vector<uint32_t> pred;
  std::vector<double> dist, weight;
  std::vector<std::vector<uint32_t>> children;
  std::vector<uint32_t> near_seq;
  double gamma = 0.0;
  // counters
  uint64_t iteration_count = 0, samples = 0, loop_iterations = 0, num_solutions = 0, rewires = 0, edges_checked = 0;
  double best_cost = std::numeric_limits<double>::infinity();
  // state machine
  StarState state = ST_IDLE;
  std::vector<double> p_new;
  uint32_t x_near = NIL;
  double eweight = 0.0;
  int tries = 0;
};
}  // namespace

struct rkh_rrtstar {
  GraphBatch gb;
  int D = 0;
  uint32_t P = 0;
  double lower[RKH_MAX_DOF], upper[RKH_MAX_DOF];
  std::vector<StarProblem> prob;
};

namespace {

double euclid(const double* a, const double* b, int D) {  // vect_distance_metrics.hpp:126-137
  double r = 0.0;
  for (int i = 0; i < D; ++i) {
    const double d = a[i] - b[i];
    r += d * d;
  }
  return std::sqrt(r);
}

bool keep_going(const StarProblem& q) {
  return (q.iteration_count < q.prm.max_vertices) && (q.prm.max_results > q.num_solutions);
}

uint32_t add_vertex(int D, StarProblem& q, const double* pt, double d, uint32_t pr) {
  q.pos.insert(q.pos.end(), pt, pt + D);
  q.dist.push_back(d);
  q.pred.push_back(pr);
  q.weight.push_back(0.0);
  q.children.emplace_back();
  return uint32_t(q.pred.size() - 1);
}

void star_params(const StarProblem& q, int D, uint32_t* k, double* radius) {
  const size_t N = q.pred.size();
  const size_t log_N = highest_set_bit(N) + 1;
  *k = uint32_t(4 * log_N);
  *radius = q.gamma * std::pow(log_N / double(N), 1.0 / double(D));
}

void draw_sample(rkh_rrtstar* p, StarProblem& q) {  // hyperbox_topology::random_point (hyperbox_topology.hpp:97-103)
  for (int d = 0; d < p->D; ++d) {
    double u;
    do {
      u = double(q.eng()) * (1.0 / 4294967296.0);
    } while (!(u < 1.0));
    q.p_new[d] = p->lower[d] + u * (p->upper[d] - p->lower[d]);
  }
  ++q.samples;
}

// lazy_node_connector::operator() (lazy_connector.hpp:332-372) on the verdicts of the CONNECT step
void connect_vertex(rkh_rrtstar* p, uint32_t i) {
  StarProblem& q = p->prob[i];
  const GraphBatch& gb = p->gb;
  const int D = p->D;
  const uint32_t v = uint32_t(q.pred.size() - 1);
  const uint32_t K = gb.kcnt(i);
  const uint32_t* kidx = gb.kidx(i);
  const uint8_t* accept = gb.accept(i);
  const double* x_out = gb.x_out(i);
  uint32_t x_near = q.x_near;
  double eweight = q.eweight;
  // connect_best_predecessor (:79-123)
  {
    const uint32_t x_near_original = x_near;
    double d_near = q.dist[x_near] + eweight;
    for (uint32_t e = 0; e < K; ++e) {
      const uint32_t u = kidx[e];
      if (u == x_near_original || q.pred[u] == NIL) continue;
      const double tentative_weight = euclid(&q.pos[size_t(u) * D], &q.pos[size_t(v) * D], D);
      const double d_out = tentative_weight + q.dist[u];
      if (d_out < d_near) {
        ++q.edges_checked;
        if (accept[e]) {  // can_be_connected(u, v)
          x_near = u;
          d_near = d_out;
          eweight = euclid(&q.pos[size_t(u) * D], &x_out[size_t(e) * D], D);
        }
      }
    }
  }
  // create_pred_edge (pruned_connector.hpp:366-382)
  q.dist[v] = eweight + q.dist[x_near];
  q.pred[v] = x_near;
  q.weight[v] = eweight;
  q.children[x_near].push_back(v);
  // connect_successors (:230-275)
  for (uint32_t e = 0; e < K; ++e) {
    const uint32_t u = kidx[e];
    if (u == x_near) continue;
    const double tentative_weight = euclid(&q.pos[size_t(v) * D], &q.pos[size_t(u) * D], D);
    const double d_in = tentative_weight + q.dist[v];
    if (d_in < q.dist[u]) {
      ++q.edges_checked;
      if (accept[K + e]) {  // can_be_connected(v, u)
        q.dist[u] = d_in;
        const uint32_t old_pred = q.pred[u];
        q.pred[u] = v;
        q.weight[u] = euclid(&q.pos[size_t(v) * D], &x_out[size_t(K + e) * D], D);
        q.children[v].push_back(u);
        if (old_pred != u && old_pred != NIL) {
          std::vector<uint32_t>& ch = q.children[old_pred];
          ch.erase(std::find(ch.begin(), ch.end(), u));
        }
        ++q.rewires;
      }
    }
  }
  // update_successors (pruned_connector.hpp:310-332)
  std::vector<uint32_t> incons(1, v);
  while (!incons.empty()) {
    const uint32_t s = incons.back();
    incons.pop_back();
    for (uint32_t t : q.children[s]) {
      if (q.pred[t] != s) continue;
      q.dist[t] = q.dist[s] + q.weight[t];
      incons.push_back(t);
    }
  }
}

}  // namespace

extern "C" {

}  // extern "C"

namespace {
// qs != nullptr: quasi-static free space (vertices = joint positions); dyn != nullptr: steerable dynamic free space
// (vertices = states (q, qd), D = 2 n_dof; edges are RK4 propagations)
rkh_status rrtstar_create(rkh_scene* scene, const rkh_qs_space* qs, const rkh_dyn_space* dyn, const rkh_rrt_params* prms,
                          uint32_t n_problems, rkh_rrtstar** out) {
  rkh_rrtstar* p = new rkh_rrtstar();
  p->D = qs ? qs->n_dof : 2 * dyn->n_dof;
  p->P = n_problems;
  for (int d = 0; d < p->D; ++d) {
    p->lower[d] = qs ? qs->lower[d] : dyn->lower[d];
    p->upper[d] = qs ? qs->upper[d] : dyn->upper[d];
  }
  uint32_t max_v = 0;
  std::vector<uint64_t> caps(n_problems);
  for (uint32_t i = 0; i < n_problems; ++i) {
    max_v = std::max(max_v, prms[i].max_vertices);
    caps[i] = uint64_t(prms[i].max_vertices) + 2;
  }
  const uint32_t kmax = uint32_t(4 * (highest_set_bit(size_t(max_v) + 2) + 1));
  rkh_status st = qs ? p->gb.init(scene, qs, n_problems, caps.data(), kmax)
                     : p->gb.init_dynamic(scene, dyn, n_problems, caps.data(), kmax);
  if (st != RKH_OK) {
    p->gb.destroy();
    delete p;
    return st;
  }
  p->prob.resize(n_problems);
  const int D = p->D;
  for (uint32_t i = 0; i < n_problems; ++i) {
    StarProblem& q = p->prob[i];
    q.prm = prms[i];
    q.eng.seed(prms[i].seed);
    q.p_new.resize(D);
    // init_motion_graph (rrtstar_path_planner.tpp:188-203): vertex 0 = start, vertex 1 = goal;
    // generate_rrt_star (rrt_star.hpp:563-564): distance[start] = 0, predecessor[start] = start
    add_vertex(D, q, prms[i].start, 0.0, 0);
    add_vertex(D, q, prms[i].goal, std::numeric_limits<double>::infinity(), NIL);
    q.gamma = 3.0 * euclid(prms[i].start, prms[i].goal, D);  // 3 * heuristic(start -> goal) (:303,322)
  }
  // the two initial rows: one append per step
  for (int r = 0; r < 2; ++r) {
    p->gb.begin();
    for (uint32_t i = 0; i < n_problems; ++i) {
      st = p->gb.cmd_append(i, &p->prob[i].pos[size_t(r) * D]);
      if (st != RKH_OK) break;
    }
    if (st == RKH_OK) st = p->gb.run();
    if (st != RKH_OK) {
      p->gb.destroy();
      delete p;
      return st;
    }
  }
  *out = p;
  return RKH_OK;
}
}  // namespace

extern "C" {

rkh_status rkh_rrtstar_create_qs_batch(rkh_scene* scene, const rkh_qs_space* space, const rkh_rrt_params* prms,
                                       uint32_t n_problems, rkh_rrtstar** out) {
  if (!scene || !space || !prms || !out || n_problems < 1) return RKH_ERR_BAD_ARG;
  if (space->n_dof != scene->host.n_dof || !(space->min_interval > 0.0)) {
    set_error("rkh_rrtstar_create: n_dof mismatch or min_interval <= 0");
    return RKH_ERR_BAD_ARG;
  }
  return rrtstar_create(scene, space, nullptr, prms, n_problems, out);
}

rkh_status rkh_rrtstar_create_batch(rkh_scene* scene, const rkh_dyn_space* space, const rkh_rrt_params* prms,
                                    uint32_t n_problems, rkh_rrtstar** out) {
  if (!scene || !space || !prms || !out || n_problems < 1) return RKH_ERR_BAD_ARG;
  if (space->n_dof != scene->host.n_dof) {
    set_error("rkh_rrtstar_create: n_dof mismatch");
    return RKH_ERR_BAD_ARG;
  }
  return rrtstar_create(scene, nullptr, space, prms, n_problems, out);
}

rkh_status rkh_rrtstar_destroy(rkh_rrtstar* p) {
  if (!p) return RKH_OK;
  p->gb.destroy();
  delete p;
  return RKH_OK;
}

// Run every problem until keep_going() is false (or max_loop_iterations loop passes, < 0 = unlimited).
rkh_status rkh_rrtstar_solve(rkh_rrtstar* p, int64_t max_loop_iterations, rkh_rrtstar_stats* stats) {
  if (!p) return RKH_ERR_BAD_ARG;
  const int D = p->D;
  const double inf = std::numeric_limits<double>::infinity();
  GraphBatch& gb = p->gb;
  // start (or resume) every problem at the top of generate_rrt_star_loop (rrt_star.hpp:169-190)
  auto next_iteration = [&](StarProblem& q) {
    if (keep_going(q) && (max_loop_iterations < 0 || int64_t(q.loop_iterations) < max_loop_iterations)) {
      ++q.loop_iterations;
      q.tries = 0;
      q.x_near = NIL;
      q.state = ST_GENERATE;
    } else {
      q.state = ST_IDLE;
    }
  };
  for (StarProblem& q : p->prob) next_iteration(q);
  for (;;) {
    // ---- build the next device step of every running problem
    gb.begin();
    bool any = false;
    for (uint32_t i = 0; i < p->P; ++i) {
      StarProblem& q = p->prob[i];
      if (q.state == ST_IDLE) continue;
      any = true;
      uint32_t k;
      double radius;
      star_params(q, D, &k, &radius);
      rkh_status st = RKH_OK;
      if (q.state == ST_GENERATE) {
        // rrg_node_generator (node_generators.hpp:137-172): sample, neighbourhood, steer from each neighbour in order
        draw_sample(p, q);
        st = gb.cmd_knn(i, q.p_new.data(), q.pred.size(), k, radius);
        gb.cmd_edges(i, GB_LIST_KNN_TO_QUERY, 0, EDGE_STEER_ACCEPT, q.prm.steer_tol);
      } else {
        // lazy_node_connector::operator() (lazy_connector.hpp:332-372): select_neighborhood(p) before create_vertex
        // (:347-350); then every neighbour in both directions: (u -> v) for connect_best_predecessor, (v -> u)
        // for connect_successors
        const uint64_t n_before = q.pred.size();
        st = gb.cmd_knn(i, q.p_new.data(), n_before, k, radius);
        if (st == RKH_OK) st = gb.cmd_append(i, q.p_new.data());
        gb.cmd_edges(i, GB_LIST_KNN_BIDIR, uint32_t(n_before), EDGE_CONNECT, q.prm.conn_tol);
      }
      if (st != RKH_OK) return st;
    }
    if (!any) break;
    rkh_status st = gb.run();
    if (st != RKH_OK) return st;
    // ---- apply the sequential rules to the verdicts
    for (uint32_t i = 0; i < p->P; ++i) {
      StarProblem& q = p->prob[i];
      if (q.state == ST_GENERATE) {
        const uint32_t K = gb.kcnt(i);
        const uint32_t* kidx = gb.kidx(i);
        const uint8_t* accept = gb.accept(i);
        const double* x_out = gb.x_out(i);
        bool was_expanded = false;
        for (uint32_t e = 0; e < K; ++e) {  // rrg_node_puller::expand_to_nearest (:61-77): first success wins
          ++q.edges_checked;
          if (accept[e]) {
            const uint32_t u = kidx[e];
            q.x_near = u;
            q.eweight = euclid(&q.pos[size_t(u) * D], &x_out[size_t(e) * D], D);  // traveled_dist
            std::memcpy(q.p_new.data(), &x_out[size_t(e) * D], D * sizeof(double));
            was_expanded = true;
            break;
          }
        }
        bool gen_done = was_expanded;
        if (!was_expanded) {
          if (q.tries >= 10) {
            gen_done = true;
            q.x_near = NIL;
          } else {
            ++q.tries;
          }
        }
        if (gen_done) {
          q.near_seq.push_back(q.x_near);
          if (q.x_near == NIL || q.dist[q.x_near] == inf) next_iteration(q);  // rrt_star.hpp:181-182
          else q.state = ST_CONNECT;
        }
      } else if (q.state == ST_CONNECT) {
        add_vertex(D, q, q.p_new.data(), inf, NIL);          // rrt_conn_visitor::create_vertex
        ++q.iteration_count;                                 // vis.vertex_added -> report_progress
        if (q.pred[1] != NIL && q.dist[1] < q.best_cost) {   // dispatched_register_solution (optimal graph)
          q.best_cost = q.dist[1];
          ++q.num_solutions;
        }
        connect_vertex(p, i);
        next_iteration(q);
      }
    }
  }
  if (stats)
    for (uint32_t i = 0; i < p->P; ++i) {
      const StarProblem& q = p->prob[i];
      rkh_rrtstar_stats& o = stats[i];
      o.num_vertices = q.pred.size();
      o.samples = q.samples;
      o.loop_iterations = q.loop_iterations;
      o.num_solutions = q.num_solutions;
      o.rewires = q.rewires;
      o.edges_checked = q.edges_checked;
      o.best_cost = q.best_cost;
    }
  return RKH_OK;
}

// The current best solution: the predecessor chain of the goal vertex (vertex 1), start first
// (register_optimal_solution_path_impl, solution_path_factories.hpp:226-270).
rkh_status rkh_rrtstar_get_solution(rkh_rrtstar* p, uint32_t problem, uint32_t* path, uint32_t capacity, uint32_t* n_path,
                                    double* cost) {
  if (!p || problem >= p->P || !n_path) return RKH_ERR_BAD_ARG;
  const StarProblem& q = p->prob[problem];
  *n_path = 0;
  if (cost) *cost = q.dist[1];
  if (q.pred[1] == NIL) return RKH_OK;  // the goal is not connected
  std::vector<uint32_t> rev;
  for (uint32_t v = 1; rev.size() <= q.pred.size(); v = q.pred[v]) {
    rev.push_back(v);
    if (v == 0) break;
  }
  *n_path = uint32_t(rev.size());
  if (path) {
    if (capacity < rev.size()) {
      set_error("rkh_rrtstar_get_solution: path buffer too small");
      return RKH_ERR_CAPACITY;
    }
    for (size_t i = 0; i < rev.size(); ++i) path[i] = rev[rev.size() - 1 - i];
  }
  return RKH_OK;
}

rkh_status rkh_rrtstar_get_graph(rkh_rrtstar* p, uint32_t problem, double* pos, uint32_t* pred, double* dist,
                                 uint32_t* near_seq) {
  if (!p || problem >= p->P) return RKH_ERR_BAD_ARG;
  const StarProblem& q = p->prob[problem];
  if (pos) std::memcpy(pos, q.pos.data(), q.pos.size() * sizeof(double));
  if (pred) std::memcpy(pred, q.pred.data(), q.pred.size() * sizeof(uint32_t));
  if (dist) std::memcpy(dist, q.dist.data(), q.dist.size() * sizeof(double));
  if (near_seq) std::memcpy(near_seq, q.near_seq.data(), q.near_seq.size() * sizeof(uint32_t));
  return RKH_OK;
}

}  // extern "C"
