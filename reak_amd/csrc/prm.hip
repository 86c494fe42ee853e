// prm.hip -- PRM (LINEAR_SEARCH_KNN, ADJ_LIST_MOTION_GRAPH, undirected motion graph) over the quasi-static free
// space, for a batch of P independent problems:
//   prm_planner::solve_planning_query (ctrl/path_planning/prm_path_planner.tpp:131-365)
//   -> generate_prm (ctrl/graph_alg/probabilistic_roadmap.hpp:309-404) -> generate_prm_impl (:211-249)
//   with prm_node_connector (prm_connector.hpp:68-182), prm_conn_visitor (probabilistic_roadmap.hpp:75-196),
//   density_plan_visitor<prm_density_calculator> (density_plan_visitors.hpp:50-224, density_calculators.hpp:45-73),
//   random_walk (planning_visitors.hpp:403-432), star_neighborhood (neighborhood_functors.hpp:95-102).
//
// The control flow of generate_prm_impl depends on the random stream (construct / expand branch, rejection
// sampling, up to 11 random-walk attempts), but the number of draws each alternative consumes is fixed, so one loop
// iteration is ONE device step (graph_batch.h) per problem:
//   construct : the next M samples are tested for is_free together; the first free one is selected on the device,
//               becomes the k-NN query and the new vertex row, and can_be_connected runs for every neighbour;
//   expand    : all 11 random-walk attempts from Q.top() are walked together; the first that travels far enough is
//               selected on the device, its end point becomes the query / new vertex, then as above.
// The host rewinds its copy of the random stream to the draw after the selected alternative, so the stream is
// consumed exactly as by the sequential reference.  Roadmap bookkeeping (4-ary indirect heap keyed by density,
// union-find of connected components, densities) is host code.
//
// Reference behaviour kept: solve_planning_query passes a density_plan_visitor (not prm_planner_visitor), whose
// publish_path tests the goal's distance_accum -- never updated on this path -- so no solution is registered and
// the roadmap grows to max_vertex_count; the start/goal component merge is reported in the stats instead.
// Restated third-party pieces (Boost / BGL-Extra are not in the reference tree): boost::d_ary_heap_indirect
// <V, 4, ..., std::less<double>> (push, push_or_update = insert or sift up only, pop); out_edges order of a vertex
// taken as insertion order.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <random>
#include <set>

#include "graph_batch.h"
#include "rkh_internal.h"

using namespace rkh;

namespace {
constexpr uint32_t NIL = 0xFFFFFFFFu;
constexpr uint32_t kConstructBatch = 4;  // samples tested per construct step
constexpr uint32_t kWalkAttempts = 11;   // do { ... } while (++i <= 10)

size_t highest_set_bit(size_t N) {  // core/base/misc_math.hpp:50-59
  size_t temp = 0;
  for (size_t shift = sizeof(size_t) * 4; (shift && (N != 1)); shift >>= 1) {
    if (N >> shift) {
      temp |= shift;
      N >>= shift;
    }
  }
  return temp;
}

double euclid(const double* a, const double* b, int D) {  // vect_distance_metrics.hpp:126-137
  double r = 0.0;
  for (int i = 0; i < D; ++i) {
    const double d = a[i] - b[i];
    r += d * d;
  }
  return std::sqrt(r);
}

// the global mt19937 with a replay window: speculated draws can be handed back
struct RngStream {
  std::mt19937 eng;
  std::vector<uint32_t> buf;
  size_t cur = 0;
  uint32_t next() {
    if (cur == buf.size()) buf.push_back(uint32_t(eng()));
    return buf[cur++];
  }
  double uniform_01() {  // boost::uniform_01 on a 32-bit engine
    for (;;) {
      const double r = double(next()) * (1.0 / 4294967296.0);
      if (r < 1.0) return r;
    }
  }
  void compact() {  // only between iterations (no stream position of a step in flight is held)
    if (cur > 65536) {
      buf.erase(buf.begin(), buf.begin() + cur);
      cur = 0;
    }
  }
};

enum PrmPending { PD_NONE, PD_CONSTRUCT, PD_EXPAND };

struct PrmProblem {
  rkh_prm_params prm;
  RngStream rng;
  // roadmap (host)
  std::vector<double> pos;
  std::vector<uint32_t> edge_u, edge_v;
  std::vector<double> edge_w;
  std::vector<std::vector<uint32_t>> incident;
  std::vector<double> density;
  std::vector<uint32_t> cc_root;
  std::set<uint32_t> cc_set;
  Heap4 Q;
  std::vector<uint8_t> kind;
  std::vector<uint32_t> expanded;
  double gamma = 0.0;
  // counters
  uint64_t iteration_count = 0, samples = 0, rejected = 0, loop_iterations = 0, publish_calls = 0, edges_checked = 0;
  int64_t merged_at_vertex = -1;
  // the step in flight
  PrmPending pending = PD_NONE;
  bool in_construct = false;  // rejection sampling of the current construct iteration continues
  uint32_t v_top = NIL;
  size_t cursor_after[kGbStageA];       // stream position if candidate c is the selected one
  size_t cursor_all_failed = 0;
  double target_dist[kGbStageA];
  bool done = false;
};
}  // namespace

struct rkh_prm {
  GraphBatch gb;
  int D = 0;
  uint32_t P = 0;
  double lower[RKH_MAX_DOF], upper[RKH_MAX_DOF];
  std::vector<PrmProblem> prob;
};

namespace {

bool keep_going(const PrmProblem& q) {
  return (q.iteration_count < q.prm.base.max_vertices) && (q.prm.base.max_results > 0u);
}

void random_point(rkh_prm* p, PrmProblem& q, double* out) {  // hyperbox_topology::random_point
  for (int d = 0; d < p->D; ++d) out[d] = p->lower[d] + q.rng.uniform_01() * (p->upper[d] - p->lower[d]);
  ++q.samples;
}

void update_density(int D, PrmProblem& q, uint32_t u) {  // prm_density_calculator::update_density
  const size_t deg = q.incident[u].size();
  if (deg == 0) {
    q.density[u] = 0.0;
    return;
  }
  const size_t max_node_degree = size_t(D) + 1;
  double sum = 0.0;
  for (uint32_t e : q.incident[u]) sum += q.edge_w[e] / q.prm.sampling_radius;
  sum /= double(deg) * double(deg) / double(max_node_degree);
  q.density[u] = std::exp(-sum * sum);
}

void requeue(int D, PrmProblem& q, uint32_t u) {  // prm_conn_visitor::requeue_vertex
  update_density(D, q, u);
  q.Q.push_or_update(u);
}

uint32_t raw_add_vertex(int D, PrmProblem& q, const double* pt) {
  q.pos.insert(q.pos.end(), pt, pt + D);
  q.density.push_back(0.0);
  q.incident.emplace_back();
  q.cc_root.push_back(0);
  return uint32_t(q.density.size() - 1);
}

void shortcut_cc_root(PrmProblem& q, uint32_t u) {  // probabilistic_roadmap.hpp:109-121
  std::vector<uint32_t> trace(1, u);
  while (q.cc_root[u] != u) {
    u = q.cc_root[u];
    trace.push_back(u);
  }
  for (uint32_t t : trace) q.cc_root[t] = u;
}

void add_edge(PrmProblem& q, uint32_t u, uint32_t v, double w) {
  const uint32_t e = uint32_t(q.edge_w.size());
  q.edge_u.push_back(u);
  q.edge_v.push_back(v);
  q.edge_w.push_back(w);
  q.incident[u].push_back(e);
  q.incident[v].push_back(e);
  // prm_conn_visitor::edge_added (:123-143)
  shortcut_cc_root(q, u);
  shortcut_cc_root(q, v);
  if (q.cc_root[v] != q.cc_root[u]) {
    const uint32_t r1 = q.cc_root[u], r2 = q.cc_root[v];
    q.cc_root[r2] = r1;
    q.cc_root[v] = r1;
    q.cc_set.erase(r2);
    if (q.cc_set.size() < 2) ++q.publish_calls;
  }
  if (q.merged_at_vertex < 0) {
    uint32_t a = 0, b = 1;
    while (q.cc_root[a] != a) a = q.cc_root[a];
    while (q.cc_root[b] != b) b = q.cc_root[b];
    if (a == b) q.merged_at_vertex = int64_t(q.density.size());
  }
}

// prm_node_connector::operator() (prm_connector.hpp:136-182) on the verdicts of the finished step
rkh_status connect_vertex(rkh_prm* p, uint32_t i, const double* pt, uint32_t x_near, double eweight) {
  PrmProblem& q = p->prob[i];
  const int D = p->D;
  GraphBatch::Verdicts nb;
  rkh_status vst = p->gb.verdicts(i, q.pos.data(), pt, &nb);
  if (vst != RKH_OK) return vst;
  const uint32_t K = nb.K;
  const uint32_t* kidx = nb.id.data();
  const uint8_t* accept = nb.accept.data();
  const double* x_out = nb.x_out.data();
  // prm_conn_visitor::create_vertex (:92-107)
  const uint32_t v = raw_add_vertex(D, q, pt);
  q.cc_root[v] = v;
  q.cc_set.insert(v);
  update_density(D, q, v);
  ++q.iteration_count;
  q.Q.idx(v) = size_t(-1);
  if (x_near != NIL) {  // connect_to_first_pred (:71-91)
    add_edge(q, x_near, v, eweight);
    requeue(D, q, x_near);
  }
  requeue(D, q, v);
  for (uint32_t e = 0; e < K; ++e) {
    const uint32_t u = kidx[e];
    if (u == x_near) continue;
    ++q.edges_checked;
    if (accept[e]) add_edge(q, u, v, euclid(&q.pos[size_t(u) * D], &x_out[size_t(e) * D], D));
    requeue(D, q, u);
  }
  requeue(D, q, v);
  return RKH_OK;
}

}  // namespace

extern "C" {

}  // extern "C"

namespace {
// qs != nullptr: quasi-static free space; dyn != nullptr: steerable dynamic free space (vertices = states, D = 2 n_dof)
rkh_status prm_create(rkh_scene* scene, const rkh_qs_space* space, const rkh_dyn_space* dyn, const rkh_prm_params* prms,
                      uint32_t n_problems, rkh_prm** out) {
  for (uint32_t i = 0; i < n_problems; ++i)
    if (!(prms[i].sampling_radius > 0.0)) {
      set_error("rkh_prm_create: sampling_radius must be positive");
      return RKH_ERR_BAD_ARG;
    }
  rkh_prm* p = new rkh_prm();
  p->D = space ? space->n_dof : 2 * dyn->n_dof;
  p->P = n_problems;
  const int D = p->D;
  for (int d = 0; d < D; ++d) {
    p->lower[d] = space ? space->lower[d] : dyn->lower[d];
    p->upper[d] = space ? space->upper[d] : dyn->upper[d];
  }
  uint32_t max_v = 0;
  std::vector<uint64_t> caps(n_problems);
  for (uint32_t i = 0; i < n_problems; ++i) {
    max_v = std::max(max_v, prms[i].base.max_vertices);
    caps[i] = uint64_t(prms[i].base.max_vertices) + 2;
  }
  const uint32_t kmax = uint32_t(4 * (highest_set_bit(size_t(max_v) + 2) + 1));
  rkh_status st = space ? p->gb.init(scene, space, n_problems, caps.data(), kmax)
                        : p->gb.init_dynamic(scene, dyn, n_problems, caps.data(), kmax);
  if (st != RKH_OK) {
    p->gb.destroy();
    delete p;
    return st;
  }
  p->prob.resize(n_problems);
  for (uint32_t i = 0; i < n_problems; ++i) {
    PrmProblem& q = p->prob[i];
    q.prm = prms[i];
    q.rng.eng.seed(prms[i].base.seed);
    q.Q.key = &q.density;
    // RK_PRM_PLANNER_INITIALIZE_START_AND_GOAL, then generate_prm on a non-empty graph (:368-397)
    raw_add_vertex(D, q, prms[i].base.start);
    raw_add_vertex(D, q, prms[i].base.goal);
    for (uint32_t u = 0; u < 2; ++u) {
      update_density(D, q, u);
      q.Q.push(u);
      q.cc_root[u] = u;
    }
    q.cc_set.insert(0);
    q.cc_set.insert(1);
    q.gamma = 3.0 * euclid(prms[i].base.start, prms[i].base.goal, D);
  }
  for (int r = 0; r < 2; ++r) {  // the two initial rows: one append per step
    p->gb.begin();
    for (uint32_t i = 0; i < n_problems && st == RKH_OK; ++i) st = p->gb.cmd_append(i, &p->prob[i].pos[size_t(r) * D]);
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

rkh_status rkh_prm_create_qs_batch(rkh_scene* scene, const rkh_qs_space* space, const rkh_prm_params* prms,
                                   uint32_t n_problems, rkh_prm** out) {
  if (!scene || !space || !prms || !out || n_problems < 1) return RKH_ERR_BAD_ARG;
  if (space->n_dof != scene->host.n_dof || !(space->min_interval > 0.0)) {
    set_error("rkh_prm_create: n_dof mismatch or min_interval <= 0");
    return RKH_ERR_BAD_ARG;
  }
  return prm_create(scene, space, nullptr, prms, n_problems, out);
}

rkh_status rkh_prm_create_batch(rkh_scene* scene, const rkh_dyn_space* space, const rkh_prm_params* prms, uint32_t n_problems,
                                rkh_prm** out) {
  if (!scene || !space || !prms || !out || n_problems < 1) return RKH_ERR_BAD_ARG;
  if (space->n_dof != scene->host.n_dof) {
    set_error("rkh_prm_create: n_dof mismatch");
    return RKH_ERR_BAD_ARG;
  }
  return prm_create(scene, nullptr, space, prms, n_problems, out);
}

rkh_status rkh_prm_destroy(rkh_prm* p) {
  if (!p) return RKH_OK;
  p->gb.destroy();
  delete p;
  return RKH_OK;
}

// Run every problem until keep_going() is false (or max_loop_iterations loop passes, < 0 = unlimited).
rkh_status rkh_prm_solve(rkh_prm* p, int64_t max_loop_iterations, rkh_prm_stats* stats) {
  if (!p) return RKH_ERR_BAD_ARG;
  const int D = p->D;
  GraphBatch& gb = p->gb;
  for (;;) {
    gb.begin();
    bool any = false;
    for (uint32_t i = 0; i < p->P; ++i) {
      PrmProblem& q = p->prob[i];
      q.pending = PD_NONE;
      if (!q.in_construct) {
        // top of the while (vis.keep_going()) loop of generate_prm_impl
        if (!(keep_going(q) && (max_loop_iterations < 0 || int64_t(q.loop_iterations) < max_loop_iterations))) continue;
        ++q.loop_iterations;
        q.rng.compact();
        const double rand_value = q.rng.uniform_01();
        if (rand_value > q.prm.expand_probability) {
          q.in_construct = true;
        } else if (q.Q.data.empty()) {
          set_error("rkh_prm_solve: expansion queue ran empty (the reference calls top() on an empty heap here)");
          return RKH_ERR_CAPACITY;
        } else {
          q.pending = PD_EXPAND;
        }
      }
      any = true;
      GbAux& a = gb.h_aux[i];
      const size_t N = q.density.size();
      const size_t log_N = highest_set_bit(N) + 1;  // star_neighborhood (neighborhood_functors.hpp:95-102)
      const uint32_t k = uint32_t(4 * log_N);
      const double radius = q.gamma * std::pow(log_N / double(N), 1.0 / double(D));
      rkh_status st;
      if (q.in_construct) {
        // construction node (:229-235): the next kConstructBatch samples of the rejection loop
        q.pending = PD_CONSTRUCT;
        for (uint32_t c = 0; c < kConstructBatch; ++c) {
          random_point(p, q, a.pts[c]);
          q.cursor_after[c] = q.rng.cur;
        }
        st = gb.cmd_stage_a(i, GB_SELECT_POINT, kConstructBatch, 0, 0.0);
      } else {
        // expansion node (:237-246): the 11 attempts of random_walk (planning_visitors.hpp:403-432) from Q.top()
        const uint32_t v = q.Q.data[0];
        q.v_top = v;
        q.expanded.push_back(v);
        const double* pv = &q.pos[size_t(v) * D];
        double p_rnd[RKH_MAX_DOF], dp[RKH_MAX_DOF];
        random_point(p, q, p_rnd);
        for (int d = 0; d < D; ++d) dp[d] = p_rnd[d] - (p->lower[d] + 0.5 * (p->upper[d] - p->lower[d]));
        for (uint32_t c = 0; c < kWalkAttempts; ++c) {
          for (int d = 0; d < D; ++d) a.pts[c][d] = pv[d] + dp[d];
          const double dist = euclid(pv, a.pts[c], D);
          const double target_dist = q.rng.uniform_01() * q.prm.sampling_radius;
          a.frac[c] = target_dist / dist;
          a.target_dist[c] = target_dist;
          q.target_dist[c] = target_dist;
          q.cursor_after[c] = q.rng.cur;
          random_point(p, q, p_rnd);  // drawn by the failure branch of this attempt
          for (int d = 0; d < D; ++d) dp[d] = p_rnd[d] - (p->lower[d] + 0.5 * (p->upper[d] - p->lower[d]));
        }
        q.cursor_all_failed = q.rng.cur;
        st = gb.cmd_stage_a(i, GB_SELECT_WALK, kWalkAttempts, v, q.prm.base.steer_tol);
      }
      if (st == RKH_OK) st = gb.cmd_knn(i, a.pts[0], N, k, radius);  // the query is written by the select kernel
      if (st != RKH_OK) return st;
      gb.cmd_edges(i, GB_LIST_KNN_TO_VERTEX, uint32_t(N), EDGE_CONNECT, q.prm.base.conn_tol);
    }
    if (!any) break;
    rkh_status st = gb.run();
    if (st != RKH_OK) return st;
    for (uint32_t i = 0; i < p->P; ++i) {
      PrmProblem& q = p->prob[i];
      if (q.pending == PD_NONE) continue;
      const uint32_t sel = gb.selected(i);
      if (q.pending == PD_CONSTRUCT) {
        if (sel == NIL) {  // all rejected: the rejection loop goes on with the next samples
          q.rejected += kConstructBatch;
          continue;
        }
        // hand the samples after the accepted one back to the stream
        q.samples -= (kConstructBatch - 1 - sel);
        q.rejected += sel;
        q.rng.cur = q.cursor_after[sel];
        gb.confirm_selected(i);
        double pt[RKH_MAX_DOF];
        std::memcpy(pt, gb.h_aux[i].pts[sel], D * sizeof(double));
        st = connect_vertex(p, i, pt, NIL, 0.0);
        if (st != RKH_OK) return st;
        q.kind.push_back(0);
        q.expanded.push_back(NIL);
        q.in_construct = false;
      } else {
        if (sel == NIL) {  // random_walk failed: Q.pop()
          q.edges_checked += kWalkAttempts;
          q.rng.cur = q.cursor_all_failed;
          q.Q.pop();
          q.kind.push_back(2);
          continue;
        }
        q.edges_checked += sel + 1;
        q.samples -= (kWalkAttempts - sel);  // samples of the attempts that were never made
        q.rng.cur = q.cursor_after[sel];
        gb.confirm_selected(i);
        double pt[RKH_MAX_DOF];
        std::memcpy(pt, gb.a_x_out(i) + size_t(sel) * D, D * sizeof(double));
        const double traveled = euclid(&q.pos[size_t(q.v_top) * D], pt, D);
        st = connect_vertex(p, i, pt, q.v_top, traveled);
        if (st != RKH_OK) return st;
        q.kind.push_back(1);
      }
    }
  }
  if (stats)
    for (uint32_t i = 0; i < p->P; ++i) {
      const PrmProblem& q = p->prob[i];
      rkh_prm_stats& o = stats[i];
      o.num_vertices = q.density.size();
      o.num_edges = q.edge_w.size();
      o.samples = q.samples;
      o.rejected = q.rejected;
      o.loop_iterations = q.loop_iterations;
      o.num_components = q.cc_set.size();
      o.publish_calls = q.publish_calls;
      o.merged_at_vertex = q.merged_at_vertex;
      o.edges_checked = q.edges_checked;
      o.device_steps = p->gb.steps;
    }
  return RKH_OK;
}

rkh_status rkh_prm_get_graph(rkh_prm* p, uint32_t problem, double* pos, uint32_t* edge_u, uint32_t* edge_v,
                             double* edge_w, double* density, uint32_t* cc_root, uint8_t* kind, uint32_t* expanded) {
  if (!p || problem >= p->P) return RKH_ERR_BAD_ARG;
  const PrmProblem& q = p->prob[problem];
  if (pos) std::memcpy(pos, q.pos.data(), q.pos.size() * sizeof(double));
  if (edge_u) std::memcpy(edge_u, q.edge_u.data(), q.edge_u.size() * sizeof(uint32_t));
  if (edge_v) std::memcpy(edge_v, q.edge_v.data(), q.edge_v.size() * sizeof(uint32_t));
  if (edge_w) std::memcpy(edge_w, q.edge_w.data(), q.edge_w.size() * sizeof(double));
  if (density) std::memcpy(density, q.density.data(), q.density.size() * sizeof(double));
  if (cc_root) std::memcpy(cc_root, q.cc_root.data(), q.cc_root.size() * sizeof(uint32_t));
  if (kind) std::memcpy(kind, q.kind.data(), q.kind.size());
  if (expanded) std::memcpy(expanded, q.expanded.data(), q.expanded.size() * sizeof(uint32_t));
  return RKH_OK;
}

}  // extern "C"
