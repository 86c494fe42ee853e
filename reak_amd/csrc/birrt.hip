// birrt.hip -- bidirectional RRT over the quasi-static free space, for a batch of P independent problems:
//   generate_bidirectional_rrt (ctrl/graph_alg/rr_tree.hpp:256-317) with detail::expand_rrt_vertex (:86-112),
//   planning_visitor_base::steer_towards_position / joining_vertex_found (ctrl/path_planning/planning_visitors.hpp:
//   349-360, 223-231), register_basic_solution_path_impl for two graphs (solution_path_factories.hpp:359-408).
//
// The loop is a strict alternation -- expand tree 1 towards its target, then tree 2 towards the vertex tree 1 just
// added (or a fresh sample), and so on -- so one expansion is one device step (graph_batch.h): nearest neighbour of the
// target in the tree (the k-NN sweep with k = 1: first minimum wins, like the linear search) and the edge walk from it,
// for all problems of the batch at once.  Each problem owns two trees = two slots of the batch; the vertex an
// expansion adds is appended to its tree's device rows during the other tree's next step.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <limits>
#include <random>

#include "graph_batch.h"
#include "rkh_internal.h"

using namespace rkh;

namespace {
constexpr uint32_t NIL = 0xFFFFFFFFu;

double euclid(const double* a, const double* b, int D) {  // vect_distance_metrics.hpp:126-137
  double r = 0.0;
  for (int i = 0; i < D; ++i) {
    const double d = a[i] - b[i];
    r += d * d;
  }
  return std::sqrt(r);
}

struct BiProblem {
  rkh_rrt_params prm;
  std::mt19937 eng;
  std::vector<double> pos[2];
  std::vector<uint32_t> parent[2];
  std::vector<uint32_t> nn_seq;
  std::vector<uint8_t> accept;
  uint64_t iteration_count = 0, loop_iterations = 0, samples = 0, num_solutions = 0, joins = 0, edges_checked = 0;
  double best_cost = std::numeric_limits<double>::infinity();
  uint32_t best_join[2] = {NIL, NIL};  // joining vertices (tree 1, tree 2) of the best registered solution
  // loop state (rr_tree.hpp:272-281)
  uint32_t v_target[2] = {0, 0};     // v_target1 (a vertex of tree 2), v_target2 (a vertex of tree 1)
  bool target_is_vertex[2] = {true, true};
  std::vector<double> p_target[2];   // p_target1 (tree 1 grows towards it), p_target2
  int next_tree = 0;                 // which expansion comes next (0: tree 1)
  bool running = false;
  bool pending_append[2] = {false, false};  // the tree's newest vertex is not on the device yet
};
}  // namespace

struct rkh_birrt {
  GraphBatch gb;
  int D = 0;
  uint32_t P = 0;
  double lower[RKH_MAX_DOF], upper[RKH_MAX_DOF];
  std::vector<BiProblem> prob;
};

namespace {

bool keep_going(const BiProblem& q) {
  return (q.iteration_count < q.prm.max_vertices) && (q.prm.max_results > q.num_solutions);
}

void draw_sample(rkh_birrt* p, BiProblem& q, std::vector<double>& out) {  // hyperbox_topology::random_point
  for (int d = 0; d < p->D; ++d) {
    double u;
    do {
      u = double(q.eng()) * (1.0 / 4294967296.0);
    } while (!(u < 1.0));
    out[d] = p->lower[d] + u * (p->upper[d] - p->lower[d]);
  }
  ++q.samples;
}

// joining_vertex_found -> register_joining_point -> register_basic_solution_path_impl (two graphs)
void joining_vertex_found(BiProblem& q, int D, uint32_t u1, uint32_t u2) {
  ++q.joins;
  double total = euclid(&q.pos[0][size_t(u1) * D], &q.pos[1][size_t(u2) * D], D);
  for (uint32_t j = u1; q.parent[0][j] != NIL;) {
    const uint32_t v = q.parent[0][j];
    total += euclid(&q.pos[0][size_t(v) * D], &q.pos[0][size_t(j) * D], D);
    j = v;
  }
  for (uint32_t j = u2; q.parent[1][j] != NIL;) {
    const uint32_t v = q.parent[1][j];
    total += euclid(&q.pos[1][size_t(j) * D], &q.pos[1][size_t(v) * D], D);
    j = v;
  }
  if (q.num_solutions == 0 || total < q.best_cost) {
    q.best_cost = total;
    q.best_join[0] = u1;
    q.best_join[1] = u2;
    ++q.num_solutions;
  }
}

}  // namespace

extern "C" {

rkh_status rkh_birrt_create_qs_batch(rkh_scene* scene, const rkh_qs_space* space, const rkh_rrt_params* prms,
                                     uint32_t n_problems, rkh_birrt** out) {
  if (!scene || !space || !prms || !out || n_problems < 1) return RKH_ERR_BAD_ARG;
  if (space->n_dof != scene->host.n_dof || !(space->min_interval > 0.0)) {
    set_error("rkh_birrt_create: n_dof mismatch or min_interval <= 0");
    return RKH_ERR_BAD_ARG;
  }
  rkh_birrt* p = new rkh_birrt();
  p->D = space->n_dof;
  p->P = n_problems;
  const int D = p->D;
  for (int d = 0; d < D; ++d) {
    p->lower[d] = space->lower[d];
    p->upper[d] = space->upper[d];
  }
  std::vector<uint64_t> caps(2 * size_t(n_problems));
  for (uint32_t i = 0; i < n_problems; ++i) caps[2 * i] = caps[2 * i + 1] = uint64_t(prms[i].max_vertices) + 2;
  rkh_status st = p->gb.init(scene, space, 2 * n_problems, caps.data(), 1);
  if (st != RKH_OK) {
    p->gb.destroy();
    delete p;
    return st;
  }
  p->prob.resize(n_problems);
  p->gb.begin();
  for (uint32_t i = 0; i < n_problems && st == RKH_OK; ++i) {
    BiProblem& q = p->prob[i];
    q.prm = prms[i];
    q.eng.seed(prms[i].seed);
    // the planner creates the roots (start in tree 1, goal in tree 2) before the loop: not counted as iterations
    q.pos[0].assign(prms[i].start, prms[i].start + D);
    q.parent[0].push_back(NIL);
    q.pos[1].assign(prms[i].goal, prms[i].goal + D);
    q.parent[1].push_back(NIL);
    q.p_target[0].assign(prms[i].goal, prms[i].goal + D);   // p_target1 = position of tree 2's root
    q.p_target[1].assign(prms[i].start, prms[i].start + D);  // p_target2 = position of tree 1's root
    st = p->gb.cmd_append(2 * i, q.pos[0].data());
    if (st == RKH_OK) st = p->gb.cmd_append(2 * i + 1, q.pos[1].data());
  }
  if (st == RKH_OK) st = p->gb.run();
  if (st != RKH_OK) {
    p->gb.destroy();
    delete p;
    return st;
  }
  *out = p;
  return RKH_OK;
}

rkh_status rkh_birrt_destroy(rkh_birrt* p) {
  if (!p) return RKH_OK;
  p->gb.destroy();
  delete p;
  return RKH_OK;
}

// Run every problem until keep_going() is false (or max_loop_iterations loop passes, < 0 = unlimited).
rkh_status rkh_birrt_solve(rkh_birrt* p, int64_t max_loop_iterations, rkh_birrt_stats* stats) {
  if (!p) return RKH_ERR_BAD_ARG;
  const int D = p->D;
  GraphBatch& gb = p->gb;
  const double inf = std::numeric_limits<double>::infinity();
  for (;;) {
    gb.begin();
    bool any = false;
    for (uint32_t i = 0; i < p->P; ++i) {
      BiProblem& q = p->prob[i];
      if (q.next_tree == 0) {  // top of the while (vis.keep_going()) loop
        q.running = keep_going(q) && (max_loop_iterations < 0 || int64_t(q.loop_iterations) < max_loop_iterations);
        if (q.running) ++q.loop_iterations;
      }
      if (!q.running) continue;
      any = true;
      const int t = q.next_tree;
      const uint32_t slot = 2 * i + t, other = 2 * i + (1 - t);
      rkh_status st = RKH_OK;
      if (q.pending_append[1 - t]) {  // the vertex the previous expansion added to the other tree
        st = gb.cmd_append(other, &q.pos[1 - t][q.pos[1 - t].size() - D]);
        q.pending_append[1 - t] = false;
      }
      // find_nearest_neighbor(p_target, g_t) + expand_rrt_vertex: steer from it towards the target
      if (st == RKH_OK) st = gb.cmd_knn(slot, q.p_target[t].data(), q.parent[t].size(), 1, inf);
      if (st != RKH_OK) return st;
      gb.cmd_edges(slot, GB_LIST_KNN_TO_QUERY, 0, EDGE_STEER_ACCEPT, q.prm.steer_tol);
    }
    if (!any) break;
    rkh_status st = gb.run();
    if (st != RKH_OK) return st;
    for (uint32_t i = 0; i < p->P; ++i) {
      BiProblem& q = p->prob[i];
      if (!q.running) continue;
      const int t = q.next_tree;
      const uint32_t slot = 2 * i + t;
      const uint32_t u = gb.kidx(slot)[0];
      const bool reached_new = gb.accept(slot)[0] != 0;
      q.nn_seq.push_back(u);
      q.accept.push_back(reached_new ? 1 : 0);
      ++q.edges_checked;
      uint32_t v = u;
      if (reached_new) {  // add_child_vertex; vis.vertex_added (report_progress); vis.edge_added returns early
        const double* pv = gb.x_out(slot);
        q.pos[t].insert(q.pos[t].end(), pv, pv + D);
        q.parent[t].push_back(u);
        v = uint32_t(q.parent[t].size() - 1);
        ++q.iteration_count;
        q.pending_append[t] = true;
      }
      // rr_tree.hpp:286-299 (t = 0) / :304-317 (t = 1): what the other tree aims at next
      const int o = 1 - t;
      if (reached_new && q.target_is_vertex[t]) {
        if (t == 0) joining_vertex_found(q, D, v, q.v_target[0]);
        else joining_vertex_found(q, D, q.v_target[1], v);
        draw_sample(p, q, q.p_target[o]);
        q.target_is_vertex[o] = false;
      } else if (!reached_new) {
        draw_sample(p, q, q.p_target[o]);
        q.target_is_vertex[o] = false;
      } else {
        std::memcpy(q.p_target[o].data(), &q.pos[t][size_t(v) * D], D * sizeof(double));
        q.v_target[o] = v;
        q.target_is_vertex[o] = true;
      }
      q.next_tree = o;
    }
  }
  if (stats)
    for (uint32_t i = 0; i < p->P; ++i) {
      const BiProblem& q = p->prob[i];
      rkh_birrt_stats& o = stats[i];
      o.num_vertices_1 = q.parent[0].size();
      o.num_vertices_2 = q.parent[1].size();
      o.loop_iterations = q.loop_iterations;
      o.samples = q.samples;
      o.num_solutions = q.num_solutions;
      o.joins = q.joins;
      o.edges_checked = q.edges_checked;
      o.best_cost = q.best_cost;
    }
  return RKH_OK;
}

// The best registered solution: vertices of tree 1 from the start to the joining vertex, then vertices of tree 2 from
// its joining vertex to the goal (register_basic_solution_path_impl for two graphs, solution_path_factories.hpp:359-408).
rkh_status rkh_birrt_get_solution(rkh_birrt* p, uint32_t problem, uint32_t* path1, uint32_t* n_path1, uint32_t* path2,
                                  uint32_t* n_path2, uint32_t capacity, double* cost) {
  if (!p || problem >= p->P || !n_path1 || !n_path2) return RKH_ERR_BAD_ARG;
  const BiProblem& q = p->prob[problem];
  *n_path1 = *n_path2 = 0;
  if (cost) *cost = q.best_cost;
  if (q.best_join[0] == NIL) return RKH_OK;
  std::vector<uint32_t> rev;
  for (uint32_t v = q.best_join[0]; v != NIL; v = q.parent[0][v]) rev.push_back(v);
  std::vector<uint32_t> fwd;
  for (uint32_t v = q.best_join[1]; v != NIL; v = q.parent[1][v]) fwd.push_back(v);
  *n_path1 = uint32_t(rev.size());
  *n_path2 = uint32_t(fwd.size());
  if (capacity < rev.size() || capacity < fwd.size()) {
    if (path1 || path2) {
      set_error("rkh_birrt_get_solution: path buffer too small");
      return RKH_ERR_CAPACITY;
    }
    return RKH_OK;
  }
  if (path1)
    for (size_t i = 0; i < rev.size(); ++i) path1[i] = rev[rev.size() - 1 - i];
  if (path2)
    for (size_t i = 0; i < fwd.size(); ++i) path2[i] = fwd[i];
  return RKH_OK;
}

rkh_status rkh_birrt_get_trees(rkh_birrt* p, uint32_t problem, double* pos1, uint32_t* parent1, double* pos2,
                               uint32_t* parent2, uint32_t* nn_seq, uint8_t* accept) {
  if (!p || problem >= p->P) return RKH_ERR_BAD_ARG;
  const BiProblem& q = p->prob[problem];
  if (pos1) std::memcpy(pos1, q.pos[0].data(), q.pos[0].size() * sizeof(double));
  if (parent1) std::memcpy(parent1, q.parent[0].data(), q.parent[0].size() * sizeof(uint32_t));
  if (pos2) std::memcpy(pos2, q.pos[1].data(), q.pos[1].size() * sizeof(double));
  if (parent2) std::memcpy(parent2, q.parent[1].data(), q.parent[1].size() * sizeof(uint32_t));
  if (nn_seq) std::memcpy(nn_seq, q.nn_seq.data(), q.nn_seq.size() * sizeof(uint32_t));
  if (accept) std::memcpy(accept, q.accept.data(), q.accept.size());
  return RKH_OK;
}

}  // extern "C"
