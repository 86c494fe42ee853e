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
// The P problems advance in lock step and share the edge launches (blockIdx.y = problem).
// Bookkeeping (predecessors, accumulated distances, children lists, the DFS of update_successors) is host code.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <random>

#include "rkh_internal.h"

namespace rkh {
int nn_padded_dims(int D);
}
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
  // device mirror of the vertex rows + k-NN plumbing
  NnStore tree;
  uint64_t n_dev = 0;
  void* d_knn_ws = nullptr;
  size_t knn_ws_bytes = 0;
  double* d_q = nullptr;          // [D] query / steer target
  uint32_t* d_kidx = nullptr;     // [kmax]
  double* d_kdist = nullptr;
  uint32_t* d_kcnt = nullptr;
  // edge batch buffers
  uint32_t* d_src_idx = nullptr;  // [emax]
  uint32_t* d_tgt_idx = nullptr;
  double* d_x_out = nullptr;      // [emax][D]
  uint32_t* d_nchk = nullptr;
  uint8_t* d_accept = nullptr;
  // host staging
  std::vector<uint32_t> h_kidx;
  std::vector<double> h_kdist;
  uint32_t h_kcnt = 0;
  std::vector<uint32_t> h_src, h_tgt;
  std::vector<double> h_x_out;
  std::vector<uint8_t> h_accept;
  // counters
  uint64_t iteration_count = 0, samples = 0, loop_iterations = 0, num_solutions = 0, rewires = 0, edges_checked = 0;
  double best_cost = std::numeric_limits<double>::infinity();
  // per-iteration state
  std::vector<double> p_new;
  uint32_t x_near = NIL;
  double eweight = 0.0;
  int tries = 0;
  bool expanded = false, gen_done = false;
};
}  // namespace

struct rkh_rrtstar {
  rkh_scene* scene = nullptr;
  hipStream_t stream = nullptr;
  QsDev qs;
  int n_dof = 0, D = 0, DP = 0;
  uint32_t P = 0, kmax = 0, emax = 0;
  double lower[RKH_MAX_DOF], upper[RKH_MAX_DOF];
  std::vector<StarProblem> prob;
  EdgeIO* d_io = nullptr;
  std::vector<EdgeIO> h_io;
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

rkh_status sync_tree(rkh_rrtstar* p, StarProblem& q) {  // append rows the device does not have yet
  const uint64_t n = q.pred.size();
  if (q.n_dev == n) return RKH_OK;
  const int D = p->D, DP = p->DP;
  std::vector<double> rows((n - q.n_dev) * DP, 0.0);
  for (uint64_t i = q.n_dev; i < n; ++i) std::memcpy(&rows[(i - q.n_dev) * DP], &q.pos[i * D], D * sizeof(double));
  RKH_HIP(hipMemcpyAsync(q.tree.d_pos + q.n_dev * DP, rows.data(), rows.size() * sizeof(double), hipMemcpyHostToDevice,
                         p->stream));
  RKH_HIP(hipStreamSynchronize(p->stream));
  q.n_dev = n;
  return RKH_OK;
}

// k-NN of one point per listed problem (star_neighborhood): results land in h_kidx / h_kdist / h_kcnt
rkh_status knn_step(rkh_rrtstar* p, const std::vector<uint32_t>& who, const std::vector<const double*>& pts) {
  hipStream_t s = p->stream;
  std::vector<uint32_t> ks(who.size());
  for (size_t w = 0; w < who.size(); ++w) {
    StarProblem& q = p->prob[who[w]];
    rkh_status st = sync_tree(p, q);
    if (st != RKH_OK) return st;
    const size_t N = q.pred.size();
    const size_t log_N = highest_set_bit(N) + 1;
    const uint32_t k = uint32_t(4 * log_N);
    const double radius = q.gamma * std::pow(log_N / double(N), 1.0 / double(p->D));
    ks[w] = k;
    RKH_HIP(hipMemcpyAsync(q.d_q, pts[w], p->D * sizeof(double), hipMemcpyHostToDevice, s));
    KnnWorkspace ws;
    size_t bytes = 0;
    st = knn_plan(N, 1, k, &ws, &bytes);
    if (st != RKH_OK) return st;
    if (bytes > q.knn_ws_bytes) {
      (void)hipFree(q.d_knn_ws);
      q.d_knn_ws = nullptr;
      RKH_HIP(hipMalloc(&q.d_knn_ws, bytes));
      q.knn_ws_bytes = bytes;
    }
    knn_carve(q.d_knn_ws, 1, &ws);
    st = launch_nnk(s, q.tree, N, q.d_q, 1, k, radius, q.d_kidx, q.d_kdist, q.d_kcnt, ws);
    if (st != RKH_OK) return st;
    RKH_HIP(hipMemcpyAsync(q.h_kidx.data(), q.d_kidx, k * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    RKH_HIP(hipMemcpyAsync(q.h_kdist.data(), q.d_kdist, k * sizeof(double), hipMemcpyDeviceToHost, s));
    RKH_HIP(hipMemcpyAsync(&q.h_kcnt, q.d_kcnt, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  }
  RKH_HIP(hipStreamSynchronize(s));
  for (size_t w = 0; w < who.size(); ++w) {
    StarProblem& q = p->prob[who[w]];
    uint32_t overflow = 0;
    RKH_HIP(hipMemcpy(&overflow, q.d_knn_ws, sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (overflow) {
      set_error("rrtstar: k-NN candidate capacity exceeded");
      return RKH_ERR_CAPACITY;
    }
  }
  return RKH_OK;
}

// one edge_check launch over the (src, tgt) lists of the listed problems; results in h_x_out / h_accept
rkh_status edge_step(rkh_rrtstar* p, const std::vector<uint32_t>& who, int mode, bool tgt_is_query) {
  hipStream_t s = p->stream;
  uint32_t max_e = 0;
  for (uint32_t w = 0; w < who.size(); ++w) {
    StarProblem& q = p->prob[who[w]];
    const uint32_t E = uint32_t(q.h_src.size());
    max_e = std::max(max_e, E);
    EdgeIO io;
    io.src = q.tree.d_pos;
    io.src_idx = q.d_src_idx;
    io.src_stride = p->DP;
    if (tgt_is_query) {
      io.tgt = q.d_q;
      io.tgt_stride = 0;
    } else {
      io.tgt = q.tree.d_pos;
      io.tgt_idx = q.d_tgt_idx;
      io.tgt_stride = p->DP;
    }
    io.B = E;
    io.x_out = q.d_x_out;
    io.steps_free = q.d_nchk;
    io.mode = mode;
    io.steer_tol = (mode == EDGE_CONNECT) ? q.prm.conn_tol : q.prm.steer_tol;
    io.accept = q.d_accept;
    io.err_flag = p->scene->d_err;
    p->h_io[w] = io;
    if (E) {
      RKH_HIP(hipMemcpyAsync(q.d_src_idx, q.h_src.data(), E * sizeof(uint32_t), hipMemcpyHostToDevice, s));
      if (!tgt_is_query) RKH_HIP(hipMemcpyAsync(q.d_tgt_idx, q.h_tgt.data(), E * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    }
  }
  if (max_e == 0) return RKH_OK;
  RKH_HIP(hipMemcpyAsync(p->d_io, p->h_io.data(), who.size() * sizeof(EdgeIO), hipMemcpyHostToDevice, s));
  rkh_status st = launch_edge_check(s, p->n_dof, p->scene->host.n_env, p->scene->d_scene, p->scene->d_pairs,
                                    p->scene->n_pairs, p->qs, EdgeIO(), max_e, nullptr, 0, p->d_io, nullptr,
                                    uint32_t(who.size()));
  if (st != RKH_OK) return st;
  for (uint32_t w = 0; w < who.size(); ++w) {
    StarProblem& q = p->prob[who[w]];
    const uint32_t E = uint32_t(q.h_src.size());
    if (!E) continue;
    RKH_HIP(hipMemcpyAsync(q.h_x_out.data(), q.d_x_out, size_t(E) * p->D * sizeof(double), hipMemcpyDeviceToHost, s));
    RKH_HIP(hipMemcpyAsync(q.h_accept.data(), q.d_accept, E, hipMemcpyDeviceToHost, s));
  }
  RKH_HIP(hipStreamSynchronize(s));
  return RKH_OK;
}

uint32_t add_vertex(rkh_rrtstar* p, StarProblem& q, const double* pt, double d, uint32_t pr) {
  q.pos.insert(q.pos.end(), pt, pt + p->D);
  q.dist.push_back(d);
  q.pred.push_back(pr);
  q.weight.push_back(0.0);
  q.children.emplace_back();
  return uint32_t(q.pred.size() - 1);
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
  rkh_rrtstar* p = new rkh_rrtstar();
  p->scene = scene;
  p->n_dof = space->n_dof;
  p->D = space->n_dof;
  p->DP = nn_padded_dims(p->D);
  p->P = n_problems;
  std::memset(&p->qs, 0, sizeof(p->qs));
  p->qs.min_interval = space->min_interval;
  p->qs.fraction = 1.0;
  for (int d = 0; d < p->D; ++d) {
    p->lower[d] = p->qs.lower[d] = space->lower[d];
    p->upper[d] = p->qs.upper[d] = space->upper[d];
  }
  RKH_HIP(hipSetDevice(scene->ctx->device));
  RKH_HIP(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking));
  uint32_t max_v = 0;
  for (uint32_t i = 0; i < n_problems; ++i) max_v = std::max(max_v, prms[i].max_vertices);
  p->kmax = uint32_t(4 * (highest_set_bit(size_t(max_v) + 2) + 1));
  p->emax = 2 * p->kmax;
  p->prob.resize(n_problems);
  p->h_io.resize(n_problems);
  RKH_HIP(hipMalloc(&p->d_io, n_problems * sizeof(EdgeIO)));
  const int D = p->D, DP = p->DP;
  for (uint32_t i = 0; i < n_problems; ++i) {
    StarProblem& q = p->prob[i];
    q.prm = prms[i];
    q.eng.seed(prms[i].seed);
    q.tree.D = D;
    q.tree.capacity = (uint64_t(prms[i].max_vertices) + 2 + 255) / 256 * 256;
    RKH_HIP(hipMalloc(&q.tree.d_pos, q.tree.capacity * DP * sizeof(double)));
    RKH_HIP(hipMalloc(&q.d_q, D * sizeof(double)));
    RKH_HIP(hipMalloc(&q.d_kidx, p->kmax * sizeof(uint32_t)));
    RKH_HIP(hipMalloc(&q.d_kdist, p->kmax * sizeof(double)));
    RKH_HIP(hipMalloc(&q.d_kcnt, sizeof(uint32_t)));
    RKH_HIP(hipMalloc(&q.d_src_idx, p->emax * sizeof(uint32_t)));
    RKH_HIP(hipMalloc(&q.d_tgt_idx, p->emax * sizeof(uint32_t)));
    RKH_HIP(hipMalloc(&q.d_x_out, size_t(p->emax) * D * sizeof(double)));
    RKH_HIP(hipMalloc(&q.d_nchk, p->emax * sizeof(uint32_t)));
    RKH_HIP(hipMalloc(&q.d_accept, p->emax));
    q.h_kidx.resize(p->kmax);
    q.h_kdist.resize(p->kmax);
    q.h_x_out.resize(size_t(p->emax) * D);
    q.h_accept.resize(p->emax);
    q.p_new.resize(D);
    // init_motion_graph (rrtstar_path_planner.tpp:188-203): vertex 0 = start, vertex 1 = goal;
    // generate_rrt_star (rrt_star.hpp:563-564): distance[start] = 0, predecessor[start] = start
    add_vertex(p, q, prms[i].start, 0.0, 0);
    add_vertex(p, q, prms[i].goal, std::numeric_limits<double>::infinity(), NIL);
    q.gamma = 3.0 * euclid(prms[i].start, prms[i].goal, D);  // 3 * heuristic(start -> goal) (:303,322)
  }
  *out = p;
  return RKH_OK;
}

rkh_status rkh_rrtstar_destroy(rkh_rrtstar* p) {
  if (!p) return RKH_OK;
  (void)hipStreamSynchronize(p->stream);
  for (StarProblem& q : p->prob) {
    void* bufs[] = {q.tree.d_pos, q.d_knn_ws, q.d_q, q.d_kidx, q.d_kdist, q.d_kcnt, q.d_src_idx, q.d_tgt_idx, q.d_x_out,
                    q.d_nchk, q.d_accept};
    for (void* b : bufs) (void)hipFree(b);
  }
  (void)hipFree(p->d_io);
  (void)hipStreamDestroy(p->stream);
  delete p;
  return RKH_OK;
}

// Run every problem until keep_going() is false (or max_loop_iterations loop passes, < 0 = unlimited).
rkh_status rkh_rrtstar_solve(rkh_rrtstar* p, int64_t max_loop_iterations, rkh_rrtstar_stats* stats) {
  if (!p) return RKH_ERR_BAD_ARG;
  const int D = p->D;
  const double inf = std::numeric_limits<double>::infinity();
  for (;;) {
    // problems that run another loop iteration
    std::vector<uint32_t> active;
    for (uint32_t i = 0; i < p->P; ++i) {
      StarProblem& q = p->prob[i];
      if (keep_going(q) && (max_loop_iterations < 0 || int64_t(q.loop_iterations) < max_loop_iterations)) {
        active.push_back(i);
        ++q.loop_iterations;
        q.tries = 0;
        q.gen_done = false;
        q.expanded = false;
        q.x_near = NIL;
      }
    }
    if (active.empty()) break;
    // ---- rrg_node_generator (node_generators.hpp:137-172): sample, neighbourhood, steer from each neighbour in order
    for (;;) {
      std::vector<uint32_t> who;
      std::vector<const double*> pts;
      for (uint32_t i : active) {
        StarProblem& q = p->prob[i];
        if (q.gen_done) continue;
        for (int d = 0; d < D; ++d) {  // hyperbox_topology::random_point (hyperbox_topology.hpp:97-103)
          double u;
          do {
            u = double(q.eng()) * (1.0 / 4294967296.0);
          } while (!(u < 1.0));
          q.p_new[d] = p->lower[d] + u * (p->upper[d] - p->lower[d]);
        }
        ++q.samples;
        who.push_back(i);
        pts.push_back(q.p_new.data());
      }
      if (who.empty()) break;
      rkh_status st = knn_step(p, who, pts);
      if (st != RKH_OK) return st;
      for (uint32_t i : who) {
        StarProblem& q = p->prob[i];
        q.h_src.assign(q.h_kidx.begin(), q.h_kidx.begin() + q.h_kcnt);
        q.h_tgt.clear();
      }
      st = edge_step(p, who, EDGE_STEER_ACCEPT, true);
      if (st != RKH_OK) return st;
      for (uint32_t i : who) {
        StarProblem& q = p->prob[i];
        bool was_expanded = false;
        for (uint32_t e = 0; e < q.h_kcnt; ++e) {  // rrg_node_puller::expand_to_nearest (:61-77): first success wins
          ++q.edges_checked;
          if (q.h_accept[e]) {
            const uint32_t u = q.h_src[e];
            q.x_near = u;
            q.eweight = euclid(&q.pos[size_t(u) * D], &q.h_x_out[size_t(e) * D], D);  // traveled_dist
            std::memcpy(q.p_new.data(), &q.h_x_out[size_t(e) * D], D * sizeof(double));
            was_expanded = true;
            break;
          }
        }
        if (was_expanded) {
          q.gen_done = true;
          q.expanded = true;
        } else if (q.tries >= 10) {
          q.gen_done = true;
          q.x_near = NIL;
        } else {
          ++q.tries;
        }
      }
    }
    // ---- lazy_node_connector::operator() (lazy_connector.hpp:332-372) for the problems that got a vertex
    std::vector<uint32_t> conn;
    std::vector<const double*> cpts;
    for (uint32_t i : active) {
      StarProblem& q = p->prob[i];
      q.near_seq.push_back(q.x_near);
      if (q.x_near == NIL || q.dist[q.x_near] == inf) continue;  // rrt_star.hpp:181-182
      conn.push_back(i);
      cpts.push_back(q.p_new.data());
    }
    if (conn.empty()) continue;
    rkh_status st = knn_step(p, conn, cpts);  // select_neighborhood(p) before create_vertex (:347-350)
    if (st != RKH_OK) return st;
    for (uint32_t i : conn) {
      StarProblem& q = p->prob[i];
      const uint32_t v = add_vertex(p, q, q.p_new.data(), inf, NIL);  // rrt_conn_visitor::create_vertex
      ++q.iteration_count;                                            // vis.vertex_added -> report_progress
      if (q.pred[1] != NIL && q.dist[1] < q.best_cost) {              // dispatched_register_solution (optimal graph)
        q.best_cost = q.dist[1];
        ++q.num_solutions;
      }
      // every neighbour in both directions: (u -> v) for connect_best_predecessor, (v -> u) for connect_successors
      q.h_src.clear();
      q.h_tgt.clear();
      for (uint32_t e = 0; e < q.h_kcnt; ++e) {
        q.h_src.push_back(q.h_kidx[e]);
        q.h_tgt.push_back(v);
      }
      for (uint32_t e = 0; e < q.h_kcnt; ++e) {
        q.h_src.push_back(v);
        q.h_tgt.push_back(q.h_kidx[e]);
      }
      rkh_status s2 = sync_tree(p, q);  // the new row must be on the device for the edge batch
      if (s2 != RKH_OK) return s2;
    }
    st = edge_step(p, conn, EDGE_CONNECT, false);
    if (st != RKH_OK) return st;
    for (uint32_t i : conn) {
      StarProblem& q = p->prob[i];
      const uint32_t v = uint32_t(q.pred.size() - 1);
      const uint32_t K = q.h_kcnt;
      uint32_t x_near = q.x_near;
      double eweight = q.eweight;
      // connect_best_predecessor (:79-123)
      {
        const uint32_t x_near_original = x_near;
        double d_near = q.dist[x_near] + eweight;
        for (uint32_t e = 0; e < K; ++e) {
          const uint32_t u = q.h_kidx[e];
          if (u == x_near_original || q.pred[u] == NIL) continue;
          const double tentative_weight = euclid(&q.pos[size_t(u) * D], &q.pos[size_t(v) * D], D);
          const double d_out = tentative_weight + q.dist[u];
          if (d_out < d_near) {
            ++q.edges_checked;
            if (q.h_accept[e]) {  // can_be_connected(u, v)
              x_near = u;
              d_near = d_out;
              eweight = euclid(&q.pos[size_t(u) * D], &q.h_x_out[size_t(e) * D], D);
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
        const uint32_t u = q.h_kidx[e];
        if (u == x_near) continue;
        const double tentative_weight = euclid(&q.pos[size_t(v) * D], &q.pos[size_t(u) * D], D);
        const double d_in = tentative_weight + q.dist[v];
        if (d_in < q.dist[u]) {
          ++q.edges_checked;
          if (q.h_accept[K + e]) {  // can_be_connected(v, u)
            q.dist[u] = d_in;
            const uint32_t old_pred = q.pred[u];
            q.pred[u] = v;
            q.weight[u] = euclid(&q.pos[size_t(v) * D], &q.h_x_out[size_t(K + e) * D], D);
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
