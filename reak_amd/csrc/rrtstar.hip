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

enum StarState { ST_IDLE, ST_GENERATE, ST_CONNECT, ST_CONNECT_PRED, ST_CONNECT_SUCC };

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
  // bidirectional RRT*: the backward tree (successor links towards the goal) and the generator's second result
  std::vector<uint32_t> succ;
  std::vector<double> fwd_dist, fwd_weight;
  std::vector<std::vector<uint32_t>> parents;
  std::vector<uint32_t> near_pred, near_succ;
  std::vector<double> p_succ;
  uint32_t x_succ = NIL;
  double eweight_succ = 0.0;
  uint64_t fwd_rewires = 0, joins = 0;
  double best_join_cost = std::numeric_limits<double>::infinity();
  // branch-and-bound pruning: tombstones, the queue keyed distance_accum + distance to the goal (largest on top)
  std::vector<uint8_t> removed;
  std::vector<double> key;
  Heap4 Q;
  uint64_t pruned = 0, skipped = 0;
  double dist_to_goal = 0.0;  // of the point being connected
};
}  // namespace

struct rkh_rrtstar {
  GraphBatch gb;
  bool bidirectional = false;
  bool branch_and_bound = false;
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
  q.succ.push_back(NIL);
  q.fwd_dist.push_back(std::numeric_limits<double>::infinity());
  q.fwd_weight.push_back(0.0);
  q.parents.emplace_back();
  q.removed.push_back(0);
  q.key.push_back(0.0);
  const uint32_t v = uint32_t(q.pred.size() - 1);
  q.Q.idx(v) = size_t(-1);  // put(index_in_heap, v, -1) (branch_and_bound_connector.hpp:137,301)
  return v;
}

// vertex_to_be_removed + clear_vertex + remove_vertex: a tombstone (ids are append-only; the row on the device becomes
// +inf so that no sweep returns it, any_knn_synchro::removed_vertex).  The children of a removed vertex keep their
// predecessor field and their cost, as in the reference.
rkh_status remove_vertex(rkh_rrtstar* p, uint32_t i, uint32_t v) {
  StarProblem& q = p->prob[i];
  q.removed[v] = 1;
  ++q.pruned;
  const uint32_t pv = q.pred[v];
  if (pv != NIL && pv != v && !q.removed[pv]) {
    std::vector<uint32_t>& ch = q.children[pv];
    auto it = std::find(ch.begin(), ch.end(), v);
    if (it != ch.end()) ch.erase(it);
  }
  q.children[v].clear();
  return p->gb.remove_row(i, v);
}

void star_params(const StarProblem& q, int D, uint32_t* k, double* radius) {
  const size_t N = q.pred.size() - size_t(q.pruned);  // num_vertices(g)
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
rkh_status connect_vertex(rkh_rrtstar* p, uint32_t i) {
  StarProblem& q = p->prob[i];
  const int D = p->D;
  const uint32_t v = uint32_t(q.pred.size() - 1);
  GraphBatch::Verdicts nb;
  rkh_status vst = p->gb.verdicts(i, q.pos.data(), &q.pos[size_t(v) * D], &nb, q.removed.data());
  if (vst != RKH_OK) return vst;
  const uint32_t K = nb.K;
  const uint32_t* kidx = nb.id.data();
  const uint8_t* accept = nb.accept.data();
  const double* x_out = nb.x_out.data();
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
  const bool bnb = p->branch_and_bound;
  if (bnb) {  // branch_and_bound_connector::operator() (branch_and_bound_connector.hpp:311-320)
    if (q.pred[1] != NIL && q.dist[v] + q.dist_to_goal > q.dist[1]) return remove_vertex(p, i, v);
    q.key[v] = q.dist[v] + q.dist_to_goal;
    q.Q.push(v);
  }
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
        if (old_pred != u && old_pred != NIL && !q.removed[old_pred]) {
          std::vector<uint32_t>& ch = q.children[old_pred];
          auto it = std::find(ch.begin(), ch.end(), u);
          if (it != ch.end()) ch.erase(it);
        }
        ++q.rewires;
      }
    }
  }
  // update_successors (pruned_connector.hpp:310-332; with pruning: branch_and_bound_connector.hpp:142-185 -- the vertices
  // whose cost changed are re-keyed (sift-up only), then every vertex whose key exceeds the goal's cost is removed)
  std::vector<uint32_t> incons(1, v);
  while (!incons.empty()) {
    const uint32_t s = incons.back();
    incons.pop_back();
    for (uint32_t t : q.children[s]) {
      if (q.pred[t] != s) continue;
      q.dist[t] = q.dist[s] + q.weight[t];
      if (bnb) {
        q.key[t] = q.dist[t] + euclid(&q.pos[size_t(t) * D], &q.pos[size_t(1) * D], D);
        q.Q.push_or_update(t);
      }
      incons.push_back(t);
    }
  }
  if (bnb && q.pred[1] != NIL) {
    while (!q.Q.data.empty() && q.key[q.Q.data[0]] > q.dist[1]) {
      const rkh_status st = remove_vertex(p, i, q.Q.data[0]);
      if (st != RKH_OK) return st;
      q.Q.pop();
    }
  }
  return RKH_OK;
}

// the bidirectional lazy_node_connector::operator() (lazy_connector.hpp:465-518) on the verdicts of a CONNECT step:
// verdict e = can_be_connected(u_e, v), verdict K + e = can_be_connected(v, u_e)
rkh_status connect_vertex_bidir(rkh_rrtstar* p, uint32_t i, uint32_t x_pred, double ep_pred, uint32_t x_succ,
                                double ep_succ) {
  StarProblem& q = p->prob[i];
  const int D = p->D;
  const double inf = std::numeric_limits<double>::infinity();
  const uint32_t v = uint32_t(q.pred.size() - 1);
  GraphBatch::Verdicts nb;
  rkh_status vst = p->gb.verdicts(i, q.pos.data(), &q.pos[size_t(v) * D], &nb);
  if (vst != RKH_OK) return vst;
  const uint32_t K = nb.K;
  const uint32_t* kidx = nb.id.data();
  const uint8_t* accept = nb.accept.data();
  const double* x_out = nb.x_out.data();
  const double* pv = &q.pos[size_t(v) * D];
  auto P = [&](uint32_t u) { return &q.pos[size_t(u) * D]; };
  {  // connect_best_predecessor (:79-123)
    const uint32_t orig = x_pred;
    double d_near = inf;
    if (x_pred != NIL) d_near = q.dist[x_pred] + ep_pred;
    for (uint32_t e = 0; e < K; ++e) {
      const uint32_t u = kidx[e];
      if (u == orig || q.pred[u] == NIL) continue;
      const double d_out = euclid(P(u), pv, D) + q.dist[u];
      if (d_out < d_near) {
        ++q.edges_checked;
        if (accept[e]) {
          x_pred = u;
          d_near = d_out;
          ep_pred = euclid(P(u), &x_out[size_t(e) * D], D);
        }
      }
    }
  }
  {  // connect_best_successor (:125-168)
    const uint32_t orig = x_succ;
    double d_near = inf;
    if (x_succ != NIL) d_near = q.fwd_dist[x_succ] + ep_succ;
    for (uint32_t e = 0; e < K; ++e) {
      const uint32_t u = kidx[e];
      if (u == orig || q.succ[u] == NIL) continue;
      const double d_in = euclid(pv, P(u), D) + q.fwd_dist[u];
      if (d_in < d_near) {
        ++q.edges_checked;
        if (accept[K + e]) {
          x_succ = u;
          d_near = d_in;
          ep_succ = euclid(pv, &x_out[size_t(K + e) * D], D);
        }
      }
    }
  }
  if (x_pred == NIL && x_succ == NIL) return RKH_OK;  // (the reference removes the vertex here; unreachable from the loop)
  if (x_pred != NIL) {  // create_pred_edge (pruned_connector.hpp:366-382)
    q.dist[v] = ep_pred + q.dist[x_pred];
    q.pred[v] = x_pred;
    q.weight[v] = ep_pred;
    q.children[x_pred].push_back(v);
  }
  if (x_succ != NIL) {  // create_succ_edge (:388-404)
    q.fwd_dist[v] = ep_succ + q.fwd_dist[x_succ];
    q.succ[v] = x_succ;
    q.fwd_weight[v] = ep_succ;
    q.parents[x_succ].push_back(v);
  }
  if (q.pred[v] != NIL && q.succ[v] != NIL) {  // a joining vertex (the reference registers nothing for it)
    ++q.joins;
    if (q.dist[v] + q.fwd_dist[v] < q.best_join_cost) q.best_join_cost = q.dist[v] + q.fwd_dist[v];
  }
  // connect_successors (:230-275): vertices that have a successor belong to the backward tree and are left alone
  for (uint32_t e = 0; e < K; ++e) {
    const uint32_t u = kidx[e];
    if (u == x_pred || q.succ[u] != NIL) continue;
    const double d_in = euclid(pv, P(u), D) + q.dist[v];
    if (d_in < q.dist[u]) {
      ++q.edges_checked;
      if (accept[K + e]) {
        q.dist[u] = d_in;
        const uint32_t old_pred = q.pred[u];
        q.pred[u] = v;
        q.weight[u] = euclid(pv, &x_out[size_t(K + e) * D], D);
        q.children[v].push_back(u);
        if (old_pred != u && old_pred != NIL) {
          std::vector<uint32_t>& ch = q.children[old_pred];
          ch.erase(std::find(ch.begin(), ch.end(), u));
        }
        ++q.rewires;
      }
    }
  }
  {  // update_successors (pruned_connector.hpp:310-332)
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
  // connect_predecessors (:170-227): vertices that have a predecessor belong to the forward tree and are left alone
  for (uint32_t e = 0; e < K; ++e) {
    const uint32_t u = kidx[e];
    if (u == x_succ || q.pred[u] != NIL) continue;
    const double d_in = euclid(P(u), pv, D) + q.fwd_dist[v];
    if (d_in < q.fwd_dist[u]) {
      ++q.edges_checked;
      if (accept[e]) {
        q.fwd_dist[u] = d_in;
        const uint32_t old_succ = q.succ[u];
        q.succ[u] = v;
        q.fwd_weight[u] = euclid(P(u), &x_out[size_t(e) * D], D);
        q.parents[v].push_back(u);
        if (old_succ != u && old_succ != NIL) {
          std::vector<uint32_t>& pa = q.parents[old_succ];
          pa.erase(std::find(pa.begin(), pa.end(), u));
        }
        ++q.fwd_rewires;
      }
    }
  }
  {  // update_predecessors (pruned_connector.hpp:338-360)
    std::vector<uint32_t> incons(1, v);
    while (!incons.empty()) {
      const uint32_t t = incons.back();
      incons.pop_back();
      for (uint32_t s : q.parents[t]) {
        if (q.succ[s] != t) continue;
        q.fwd_dist[s] = q.fwd_dist[t] + q.fwd_weight[s];
        incons.push_back(s);
      }
    }
  }
  return RKH_OK;
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
  if (!p || p->bidirectional) return RKH_ERR_BAD_ARG;
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
        GraphBatch::Verdicts nb;
        st = gb.verdicts(i, q.pos.data(), q.p_new.data(), &nb, q.removed.data());
        if (st != RKH_OK) return st;
        const uint32_t K = nb.K;
        const uint32_t* kidx = nb.id.data();
        const uint8_t* accept = nb.accept.data();
        const double* x_out = nb.x_out.data();
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
          if (q.x_near == NIL || q.dist[q.x_near] == inf) {
            next_iteration(q);  // rrt_star.hpp:181-182
          } else if (p->branch_and_bound) {
            // branch_and_bound_connector::operator() (:284-293): a point that cannot lie on a better path is dropped
            const double dist_from_start = euclid(&q.pos[0], q.p_new.data(), D);
            q.dist_to_goal = euclid(q.p_new.data(), &q.pos[size_t(1) * D], D);
            if (q.pred[1] != NIL && dist_from_start + q.dist_to_goal > q.dist[1]) {
              ++q.skipped;
              next_iteration(q);
            } else {
              q.state = ST_CONNECT;
            }
          } else {
            q.state = ST_CONNECT;
          }
        }
      } else if (q.state == ST_CONNECT) {
        add_vertex(D, q, q.p_new.data(), inf, NIL);          // rrt_conn_visitor::create_vertex
        ++q.iteration_count;                                 // vis.vertex_added -> report_progress
        if (q.pred[1] != NIL && q.dist[1] < q.best_cost) {   // dispatched_register_solution (optimal graph)
          q.best_cost = q.dist[1];
          ++q.num_solutions;
        }
        st = connect_vertex(p, i);
        if (st != RKH_OK) return st;
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
      o.pruned = q.pruned;
      o.skipped = q.skipped;
    }
  return RKH_OK;
}

// USE_BRANCH_AND_BOUND_PRUNING_FLAG (rrtstar_path_planner.tpp:270-283): generate_bnb_rrt_star instead of
// generate_rrt_star; to be chosen before the first rkh_rrtstar_solve
rkh_status rkh_rrtstar_set_branch_and_bound(rkh_rrtstar* p, int enabled) {
  if (!p || p->bidirectional) return RKH_ERR_BAD_ARG;
  for (const StarProblem& q : p->prob)
    if (q.loop_iterations != 0) {
      set_error("rkh_rrtstar_set_branch_and_bound: the planner has already run");
      return RKH_ERR_BAD_ARG;
    }
  p->branch_and_bound = enabled != 0;
  for (StarProblem& q : p->prob) {
    q.Q.greater = true;
    q.Q.key = &q.key;
  }
  return RKH_OK;
}

// removed[num_vertices]: 1 for the vertices taken out of the graph (branch-and-bound pruning)
rkh_status rkh_rrtstar_get_removed(rkh_rrtstar* p, uint32_t problem, uint8_t* removed) {
  if (!p || problem >= p->P || !removed) return RKH_ERR_BAD_ARG;
  const StarProblem& q = p->prob[problem];
  std::memcpy(removed, q.removed.data(), q.removed.size());
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

// ---------------------------------------------------------------------------------------------------------------
// Bidirectional RRT*: generate_rrt_star_bidir (rrt_star.hpp:612-659) -> generate_rrt_star_bidir_loop (:197-236) with
// rrg_bidir_generator (node_generators.hpp:215-277) and the bidirectional lazy_node_connector (lazy_connector.hpp:465-518).
// Device steps: GENERATE = sample + k-NN + one walk per neighbour (EDGE_STEER_BOTH: the verdict of
// steer_towards_position and, from the same walk, of steer_back_to_position); CONNECT = k-NN + append + both
// directions of every neighbour, once for the expanded point and once for the retracted one.
extern "C" {

rkh_status rkh_birrtstar_create_qs_batch(rkh_scene* scene, const rkh_qs_space* space, const rkh_rrt_params* prms,
                                         uint32_t n_problems, rkh_rrtstar** out) {
  if (!scene || !space || !prms || !out || n_problems < 1) return RKH_ERR_BAD_ARG;
  if (space->n_dof != scene->host.n_dof || !(space->min_interval > 0.0)) {
    set_error("rkh_birrtstar_create: n_dof mismatch or min_interval <= 0");
    return RKH_ERR_BAD_ARG;
  }
  // two vertices can be added per loop iteration, the last iteration may start one short of max_vertices
  std::vector<rkh_rrt_params> grown(prms, prms + n_problems);
  for (rkh_rrt_params& g : grown) g.max_vertices += 1;
  rkh_status st = rrtstar_create(scene, space, nullptr, grown.data(), n_problems, out);
  if (st != RKH_OK) return st;
  rkh_rrtstar* p = *out;
  p->bidirectional = true;
  for (uint32_t i = 0; i < n_problems; ++i) {
    StarProblem& q = p->prob[i];
    q.prm = prms[i];
    q.p_succ.resize(p->D);
    q.fwd_dist[1] = 0.0;  // rrt_star.hpp:650-651: the goal is the root of the backward tree
    q.succ[1] = 1;
  }
  return RKH_OK;
}

rkh_status rkh_birrtstar_solve(rkh_rrtstar* p, int64_t max_loop_iterations, rkh_birrtstar_stats* stats) {
  if (!p || !p->bidirectional) return RKH_ERR_BAD_ARG;
  const int D = p->D;
  GraphBatch& gb = p->gb;
  auto keep = [&](const StarProblem& q) { return q.iteration_count < q.prm.max_vertices && q.prm.max_results > 0; };
  auto next_iteration = [&](StarProblem& q) {
    if (keep(q) && (max_loop_iterations < 0 || int64_t(q.loop_iterations) < max_loop_iterations)) {
      ++q.loop_iterations;
      q.tries = 0;
      q.x_near = NIL;
      q.x_succ = NIL;
      q.state = ST_GENERATE;
    } else {
      q.state = ST_IDLE;
    }
  };
  for (StarProblem& q : p->prob) next_iteration(q);
  for (;;) {
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
        draw_sample(p, q);
        st = gb.cmd_knn(i, q.p_new.data(), q.pred.size(), k, radius);
        gb.cmd_edges(i, GB_LIST_KNN_TO_QUERY, 0, EDGE_STEER_BOTH, q.prm.steer_tol);
      } else {
        const double* pt = q.state == ST_CONNECT_PRED ? q.p_new.data() : q.p_succ.data();
        const uint64_t n_before = q.pred.size();
        st = gb.cmd_knn(i, pt, n_before, k, radius);
        if (st == RKH_OK) st = gb.cmd_append(i, pt);
        gb.cmd_edges(i, GB_LIST_KNN_BIDIR, uint32_t(n_before), EDGE_CONNECT, q.prm.conn_tol);
      }
      if (st != RKH_OK) return st;
    }
    if (!any) break;
    rkh_status st = gb.run();
    if (st != RKH_OK) return st;
    for (uint32_t i = 0; i < p->P; ++i) {
      StarProblem& q = p->prob[i];
      if (q.state == ST_GENERATE) {
        // rrg_bidir_generator (node_generators.hpp:244-277)
        GraphBatch::Verdicts nb;
        st = gb.verdicts(i, q.pos.data(), q.p_new.data(), &nb);
        if (st != RKH_OK) return st;
        const uint32_t K = nb.K;
        const uint32_t* kidx = nb.id.data();
        const uint8_t* accept = nb.accept.data();
        const double* x_out = nb.x_out.data();
        bool was_expanded = false, was_retracted = false;
        q.x_near = NIL;
        q.x_succ = NIL;
        for (uint32_t e = 0; e < K; ++e) {  // expand_to_nearest (:84-100): neighbours of the forward tree, in order
          const uint32_t u = kidx[e];
          if (q.pred[u] == NIL) continue;
          ++q.edges_checked;
          if (accept[e] & 1) {
            q.x_near = u;
            q.eweight = euclid(&q.pos[size_t(u) * D], &x_out[size_t(e) * D], D);
            was_expanded = true;
            break;
          }
        }
        for (uint32_t e = 0; e < K; ++e) {  // retract_from_nearest (:102-118): neighbours of the backward tree
          const uint32_t u = kidx[e];
          if (q.succ[u] == NIL) continue;
          ++q.edges_checked;
          if ((accept[e] & 1) && !(accept[e] & 2)) {  // a completed walk back returns its own start: never accepted
            q.x_succ = u;
            q.eweight_succ = euclid(&x_out[size_t(e) * D], &q.pos[size_t(u) * D], D);
            std::memcpy(q.p_succ.data(), &x_out[size_t(e) * D], D * sizeof(double));
            was_retracted = true;
            break;
          }
        }
        if (was_expanded) {  // after the retraction: both pulls start from the sample
          for (uint32_t e = 0; e < K; ++e)
            if (kidx[e] == q.x_near) {
              std::memcpy(q.p_new.data(), &x_out[size_t(e) * D], D * sizeof(double));
              break;
            }
        }
        bool gen_done = was_expanded || was_retracted;
        if (!gen_done) {
          if (q.tries >= 10) gen_done = true;
          else ++q.tries;
        }
        if (gen_done) {
          q.near_pred.push_back(q.x_near);
          q.near_succ.push_back(q.x_succ);
          if (q.x_near != NIL) q.state = ST_CONNECT_PRED;
          else if (q.x_succ != NIL) q.state = ST_CONNECT_SUCC;
          else next_iteration(q);
        }
      } else if (q.state == ST_CONNECT_PRED || q.state == ST_CONNECT_SUCC) {
        const bool first = q.state == ST_CONNECT_PRED;
        add_vertex(D, q, first ? q.p_new.data() : q.p_succ.data(), std::numeric_limits<double>::infinity(), NIL);
        ++q.iteration_count;
        st = first ? connect_vertex_bidir(p, i, q.x_near, q.eweight, NIL, 0.0)
                   : connect_vertex_bidir(p, i, NIL, 0.0, q.x_succ, q.eweight_succ);
        if (st != RKH_OK) return st;
        if (first && q.x_succ != NIL) q.state = ST_CONNECT_SUCC;
        else next_iteration(q);
      }
    }
  }
  if (stats)
    for (uint32_t i = 0; i < p->P; ++i) {
      const StarProblem& q = p->prob[i];
      rkh_birrtstar_stats& o = stats[i];
      o.num_vertices = q.pred.size();
      o.samples = q.samples;
      o.loop_iterations = q.loop_iterations;
      o.rewires = q.rewires;
      o.fwd_rewires = q.fwd_rewires;
      o.joins = q.joins;
      o.edges_checked = q.edges_checked;
      o.best_join_cost = q.best_join_cost;
    }
  return RKH_OK;
}

rkh_status rkh_birrtstar_get_graph(rkh_rrtstar* p, uint32_t problem, double* pos, uint32_t* pred, double* dist, uint32_t* succ,
                                   double* fwd_dist, uint32_t* near_pred, uint32_t* near_succ) {
  if (!p || !p->bidirectional || problem >= p->P) return RKH_ERR_BAD_ARG;
  const StarProblem& q = p->prob[problem];
  if (pos) std::memcpy(pos, q.pos.data(), q.pos.size() * sizeof(double));
  if (pred) std::memcpy(pred, q.pred.data(), q.pred.size() * sizeof(uint32_t));
  if (dist) std::memcpy(dist, q.dist.data(), q.dist.size() * sizeof(double));
  if (succ) std::memcpy(succ, q.succ.data(), q.succ.size() * sizeof(uint32_t));
  if (fwd_dist) std::memcpy(fwd_dist, q.fwd_dist.data(), q.fwd_dist.size() * sizeof(double));
  if (near_pred) std::memcpy(near_pred, q.near_pred.data(), q.near_pred.size() * sizeof(uint32_t));
  if (near_succ) std::memcpy(near_succ, q.near_succ.data(), q.near_succ.size() * sizeof(uint32_t));
  return RKH_OK;
}

}  // extern "C"
