// graph_batch.h -- device plumbing shared by the motion-graph planners whose loop iterations are sequential
// (RRT*, PRM): P independent problems, each a small state machine on the host; one "step" of every problem
// (append a vertex row -> k-NN of one point -> candidate edge list from the k-NN result -> quasi-static edge
// walk of every candidate) runs as ONE set of launches for all problems (device tables, blockIdx.z / .y =
// problem) with one command upload and one result download -- the cost of a step does not grow with P until
// the chip is full.  The sequential rules of the reference (neighbour order, strict comparisons, running
// minima) are applied by the host to the downloaded, mutually independent verdicts.
//
// Reference pieces served: star_neighborhood + min_dist_linear_search k-NN (ctrl/graph_alg/neighborhood_functors.hpp:95-102,
// ctrl/path_planning/topological_search.hpp:244-274); steer_towards_position / can_be_connected over the
// quasi-static free space (ctrl/path_planning/planning_visitors.hpp:349-360,385-395 ->
// ctrl/interpolation/interpolated_topologies.hpp:137-163 -> manip_free_workspace.hpp:154-156).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "rkh_internal.h"

namespace rkh {

int nn_padded_dims(int D);

enum GbListMode : uint32_t {
  GB_LIST_NONE = 0,          // no edges this step
  GB_LIST_KNN_TO_QUERY = 1,  // (u -> query point) for every k-NN result u, in k-NN order
  GB_LIST_KNN_BIDIR = 2,     // (u -> v) for every u, then (v -> u) for every u
  GB_LIST_KNN_TO_VERTEX = 3, // (u -> v) for every u
  GB_LIST_POINT = 4,         // one edge (v -> query point)
};

constexpr uint32_t kGbStageA = 16;  // candidate points / walks of the first stage

enum GbSelect : uint32_t {
  GB_SELECT_NONE = 0,
  GB_SELECT_POINT = 1,  // the first accepted stage-A candidate point becomes the query point and the new vertex row
  GB_SELECT_WALK = 2,   // the end point of the first accepted stage-A walk becomes the query point and the new vertex row
};

struct GbAux {  // per-problem command fields read by the prep / select / list kernels
  double query[RKH_MAX_DOF];
  double append_row[RKH_MAX_DOF];
  double* append_dst;      // null: nothing to append
  // stage A (PRM): candidate points / random-walk targets, tested before the k-NN of the same step
  double pts[kGbStageA][RKH_MAX_DOF];
  double frac[kGbStageA];
  double target_dist[kGbStageA];
  uint32_t a_src[kGbStageA];
  uint32_t a_count;
  uint32_t select_mode;    // GbSelect
  double* select_dst;      // row the selected point is appended to
  const uint8_t* a_accept; // stage-A verdicts / end points (device result block)
  const double* a_xout;
  uint32_t* sel;           // selected candidate or 0xFFFFFFFF (device result block)
  uint32_t list_mode;
  uint32_t v;              // vertex id used by the list modes
  const uint32_t* kidx;    // k-NN result (device)
  const uint32_t* kcnt;
  uint32_t* src_idx;       // edge lists (device)
  uint32_t* tgt_idx;
  uint32_t* n_edges;       // device-side edge count (read by the edge kernel)
};

static __global__ void gb_prep_kernel(const GbAux* __restrict__ aux, int DP) {
  const GbAux& a = aux[blockIdx.x];
  if (a.append_dst && int(threadIdx.x) < DP) a.append_dst[threadIdx.x] = threadIdx.x < RKH_MAX_DOF ? a.append_row[threadIdx.x] : 0.0;
}

// first accepted stage-A candidate -> query point of the k-NN and new vertex row (one block of 64 per problem)
static __global__ void gb_select_kernel(GbAux* __restrict__ aux, int D, int DP) {
  GbAux& a = aux[blockIdx.x];
  if (a.select_mode == GB_SELECT_NONE) return;
  uint32_t j = 0xFFFFFFFFu;
  for (uint32_t c = 0; c < a.a_count; ++c)
    if (a.a_accept[c]) {
      j = c;
      break;
    }
  if (threadIdx.x == 0) *a.sel = j;
  if (j == 0xFFFFFFFFu) return;
  const int d = threadIdx.x;
  if (d < DP) {
    double v = 0.0;
    if (d < D) v = (a.select_mode == GB_SELECT_POINT) ? a.pts[j][d] : a.a_xout[size_t(j) * D + d];
    if (d < RKH_MAX_DOF) a.query[d] = v;
    a.select_dst[d] = v;
  }
}

// Result blocks -> pinned host memory, written by the device itself, then a step number behind a system-scope fence:
// the host spins on that word instead of sleeping in hipStreamSynchronize (whose wake-up costs more than a whole
// device step of a single planner is worth).  One block per problem; the block that arrives last publishes the step.
static __global__ void gb_download_kernel(const uint4* __restrict__ d_res, uint4* __restrict__ h_res, uint32_t words16,
                                          uint32_t* __restrict__ arrivals, volatile uint32_t* __restrict__ h_flag, uint32_t step) {
  const uint4* src = d_res + size_t(blockIdx.x) * words16;
  uint4* dst = h_res + size_t(blockIdx.x) * words16;
  for (uint32_t i = threadIdx.x; i < words16; i += blockDim.x) dst[i] = src[i];
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    if (atomicAdd(arrivals, 1u) == gridDim.x - 1) {
      *arrivals = 0u;
      __threadfence_system();
      *h_flag = step;
    }
  }
}

static __global__ void gb_list_kernel(const GbAux* __restrict__ aux) {
  const GbAux& a = aux[blockIdx.x];
  const bool dropped = (a.select_mode != GB_SELECT_NONE) && (*a.sel == 0xFFFFFFFFu);  // nothing was selected
  const uint32_t K = (a.list_mode == GB_LIST_NONE || a.list_mode == GB_LIST_POINT || dropped) ? 0u : *a.kcnt;
  uint32_t E = 0;
  switch (dropped ? uint32_t(GB_LIST_NONE) : a.list_mode) {
    case GB_LIST_KNN_TO_QUERY:
      E = K;
      for (uint32_t e = threadIdx.x; e < K; e += blockDim.x) a.src_idx[e] = a.kidx[e];
      break;
    case GB_LIST_KNN_BIDIR:
      E = 2 * K;
      for (uint32_t e = threadIdx.x; e < K; e += blockDim.x) {
        a.src_idx[e] = a.kidx[e];
        a.tgt_idx[e] = a.v;
        a.src_idx[K + e] = a.v;
        a.tgt_idx[K + e] = a.kidx[e];
      }
      break;
    case GB_LIST_KNN_TO_VERTEX:
      E = K;
      for (uint32_t e = threadIdx.x; e < K; e += blockDim.x) {
        a.src_idx[e] = a.kidx[e];
        a.tgt_idx[e] = a.v;
      }
      break;
    case GB_LIST_POINT:
      E = 1;
      if (threadIdx.x == 0) a.src_idx[0] = a.v;
      break;
    default: break;
  }
  if (threadIdx.x == 0) *a.n_edges = E;
}

// boost::d_ary_heap_indirect<Vertex, 4, IndexInHeapMap, KeyMap, Compare> restated from its published definition (push,
// push_or_update = insert-or-sift-UP-only, pop, top).  greater = false: std::less (PRM's density queue, smallest key on
// top); greater = true: std::greater (branch_and_bound_connector's queue, largest key on top).
struct Heap4 {
  bool greater = false;
  bool before(double a, double b) const { return greater ? a > b : a < b; }
  std::vector<uint32_t> data;
  std::vector<size_t> index;
  const std::vector<double>* key = nullptr;
  size_t& idx(uint32_t v) {
    if (index.size() <= v) index.resize(size_t(v) + 1, 0);
    return index[v];
  }
  void sift_up(size_t i) {
    if (i == 0) return;
    const size_t orig = i;
    const uint32_t moving = data[i];
    const double moving_key = (*key)[moving];
    size_t levels = 0;
    while (i != 0) {
      const size_t parent = (i - 1) / 4;
      if (before(moving_key, (*key)[data[parent]])) {
        ++levels;
        i = parent;
      } else {
        break;
      }
    }
    i = orig;
    for (size_t l = 0; l < levels; ++l) {
      const size_t parent = (i - 1) / 4;
      const uint32_t pv = data[parent];
      idx(pv) = i;
      data[i] = pv;
      i = parent;
    }
    data[i] = moving;
    idx(moving) = i;
  }
  void sift_down() {
    if (data.empty()) return;
    size_t i = 0;
    const double moving_key = (*key)[data[0]];
    const size_t n = data.size();
    for (;;) {
      const size_t first = 4 * i + 1;
      if (first >= n) break;
      const size_t nc = (first + 4 <= n) ? 4 : n - first;
      size_t best = 0;
      double best_key = (*key)[data[first]];
      for (size_t c = 1; c < nc; ++c) {
        const double k = (*key)[data[first + c]];
        if (before(k, best_key)) {
          best = c;
          best_key = k;
        }
      }
      if (before(best_key, moving_key)) {
        const size_t c = first + best;
        std::swap(data[c], data[i]);
        idx(data[i]) = i;
        idx(data[c]) = c;
        i = c;
      } else {
        break;
      }
    }
  }
  void push(uint32_t v) {
    const size_t i = data.size();
    data.push_back(v);
    idx(v) = i;
    sift_up(i);
  }
  void push_or_update(uint32_t v) {
    size_t i = idx(v);
    if (i == size_t(-1)) {
      i = data.size();
      data.push_back(v);
      idx(v) = i;
    }
    sift_up(i);
  }
  void pop() {
    idx(data[0]) = size_t(-1);
    if (data.size() != 1) {
      data[0] = data.back();
      idx(data[0]) = 0;
      data.pop_back();
      sift_down();
    } else {
      data.pop_back();
    }
  }
};


struct GbProblem {
  NnStore tree;
  uint64_t n_dev = 0;            // rows on the device
  void* d_knn_ws = nullptr;
  uint32_t* d_src_idx = nullptr;
  uint32_t* d_tgt_idx = nullptr;
};

struct GraphBatch {
  rkh_scene* scene = nullptr;
  hipStream_t stream = nullptr;
  QsDev qs;
  bool dynamic = false;  // edges are RK4 propagations through the steerable dynamic space (D = 2 n_dof states)
  DynDev dyn;
  int n_dof = 0, D = 0, DP = 0;
  uint32_t P = 0, kmax = 0, emax = 0;
  std::vector<GbProblem> prob;
  static constexpr size_t kKnnWsBytes = 128 * 1024;
  // command block: [KnnArgs x P][EdgeIO x P][EdgeIO (stage A) x P][GbAux x P], pinned host copy + device copy
  unsigned char *h_cmd = nullptr, *d_cmd = nullptr;
  size_t cmd_bytes = 0;
  KnnArgs *h_knn = nullptr, *d_knn = nullptr;
  EdgeIO *h_io = nullptr, *d_io = nullptr;
  EdgeIO *h_ioa = nullptr, *d_ioa = nullptr;
  GbAux *h_aux = nullptr, *d_aux = nullptr;
  // result block per problem: {kcnt, overflow, n_edges, sel} kidx[kmax] kdist[kmax] nchk[emax] accept[emax] x_out[emax][D]
  //                           a_nchk[16] a_accept[16] a_xout[16][D]
  unsigned char *h_res = nullptr, *d_res = nullptr;
  unsigned char* h_res_dev = nullptr;            // h_res as the device sees it
  uint32_t *h_flag = nullptr, *h_flag_dev = nullptr, *d_arrivals = nullptr;
  uint32_t flag_seq = 0;
  bool spin_download = true;
  size_t res_stride = 0, off_kidx = 0, off_kdist = 0, off_nchk = 0, off_accept = 0, off_xout = 0, off_anchk = 0,
         off_aaccept = 0, off_axout = 0;
  bool any_knn = false, any_edges = false, any_append = false, any_stage_a = false;
  uint64_t steps = 0;
  std::vector<uint32_t> knn_k;       // per problem: k, n and radius of the last cmd_knn
  std::vector<uint64_t> knn_n;
  std::vector<double> knn_radius;
  uint64_t tie_replays = 0;
  std::vector<double> inf_row;

  // the steerable dynamic free space: vertices are states (q, qd), edges steer_position_toward (planner.hip's space)
  rkh_status init_dynamic(rkh_scene* sc, const rkh_dyn_space* space, uint32_t n_problems, const uint64_t* capacities,
                          uint32_t kmax_) {
    if (2 * space->n_dof > RKH_MAX_DOF) {
      set_error("graph batch: state dimension exceeds RKH_MAX_DOF");
      return RKH_ERR_UNSUPPORTED;
    }
    rkh_status st = build_dyn_dev(*space, 1.0, &dyn);
    if (st != RKH_OK) return st;
    dynamic = true;
    std::memset(&qs, 0, sizeof(qs));
    return init_common(sc, space->n_dof, 2 * space->n_dof, n_problems, capacities, kmax_);
  }
  rkh_status init(rkh_scene* sc, const rkh_qs_space* space, uint32_t n_problems, const uint64_t* capacities,
                  uint32_t kmax_) {
    std::memset(&qs, 0, sizeof(qs));
    qs.min_interval = space->min_interval;
    qs.fraction = 1.0;
    qs_set_speed(qs, space->speed_limits, space->n_dof);
    for (int d = 0; d < space->n_dof; ++d) {
      qs.lower[d] = space->lower[d];
      qs.upper[d] = space->upper[d];
    }
    return init_common(sc, space->n_dof, space->n_dof, n_problems, capacities, kmax_);
  }
  rkh_status init_common(rkh_scene* sc, int n_dof_, int D_, uint32_t n_problems, const uint64_t* capacities, uint32_t kmax_) {
    scene = sc;
    n_dof = n_dof_;
    D = D_;
    DP = nn_padded_dims(D);
    P = n_problems;
    kmax = kmax_ + 1;  // one neighbour more than asked for: a tie across the k-th place must be visible (neighbours())
    emax = 2 * kmax;
    knn_k.assign(P, 0);
    knn_n.assign(P, 0);
    knn_radius.assign(P, 0.0);
    RKH_HIP(hipSetDevice(sc->ctx->device));
    RKH_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    cmd_bytes = size_t(P) * (sizeof(KnnArgs) + 2 * sizeof(EdgeIO) + sizeof(GbAux));
    RKH_HIP(hipHostMalloc(reinterpret_cast<void**>(&h_cmd), cmd_bytes, hipHostMallocDefault));
    RKH_HIP(hipMalloc(reinterpret_cast<void**>(&d_cmd), cmd_bytes));
    auto carve = [&](unsigned char* base, KnnArgs** k, EdgeIO** io, EdgeIO** ioa, GbAux** ax) {
      *k = reinterpret_cast<KnnArgs*>(base);
      *io = reinterpret_cast<EdgeIO*>(base + size_t(P) * sizeof(KnnArgs));
      *ioa = reinterpret_cast<EdgeIO*>(base + size_t(P) * (sizeof(KnnArgs) + sizeof(EdgeIO)));
      *ax = reinterpret_cast<GbAux*>(base + size_t(P) * (sizeof(KnnArgs) + 2 * sizeof(EdgeIO)));
    };
    carve(h_cmd, &h_knn, &h_io, &h_ioa, &h_aux);
    carve(d_cmd, &d_knn, &d_io, &d_ioa, &d_aux);
    auto up8 = [](size_t v) { return (v + 7) / 8 * 8; };
    off_kidx = 16;
    off_kdist = up8(off_kidx + size_t(kmax) * 4);
    off_nchk = off_kdist + size_t(kmax) * 8;
    off_accept = off_nchk + size_t(emax) * 4;
    off_xout = up8(off_accept + emax);
    off_anchk = up8(off_xout + size_t(emax) * D * 8);
    off_aaccept = off_anchk + kGbStageA * 4;
    off_axout = up8(off_aaccept + kGbStageA);
    res_stride = (off_axout + size_t(kGbStageA) * D * 8 + 15) / 16 * 16;  // gb_download_kernel moves 16-byte words
    RKH_HIP(hipHostMalloc(reinterpret_cast<void**>(&h_res), res_stride * P, hipHostMallocDefault));
    RKH_HIP(hipMalloc(reinterpret_cast<void**>(&d_res), res_stride * P));
    RKH_HIP(hipMemset(d_res, 0, res_stride * P));
    std::memset(h_res, 0, res_stride * P);
    {  // results written to the host by the device + a flag word the host spins on (gb_download_kernel)
      const char* env = getenv("RKH_GB_SPIN");
      spin_download = !env || atoi(env) != 0;
      RKH_HIP(hipHostMalloc(reinterpret_cast<void**>(&h_flag), 64, hipHostMallocDefault));
      *h_flag = 0u;
      RKH_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&h_flag_dev), h_flag, 0));
      RKH_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&h_res_dev), h_res, 0));
      RKH_HIP(hipMalloc(reinterpret_cast<void**>(&d_arrivals), sizeof(uint32_t)));
      RKH_HIP(hipMemset(d_arrivals, 0, sizeof(uint32_t)));
      flag_seq = 0;
    }
    prob.resize(P);
    for (uint32_t i = 0; i < P; ++i) {
      GbProblem& q = prob[i];
      q.tree.D = D;
      q.tree.capacity = (capacities[i] + 255) / 256 * 256;
      RKH_HIP(hipMalloc(&q.tree.d_pos, q.tree.capacity * DP * sizeof(double)));
      RKH_HIP(hipMalloc(&q.d_knn_ws, kKnnWsBytes));
      RKH_HIP(hipMalloc(&q.d_src_idx, emax * sizeof(uint32_t)));
      RKH_HIP(hipMalloc(&q.d_tgt_idx, emax * sizeof(uint32_t)));
    }
    begin();
    return RKH_OK;
  }

  void destroy() {
    if (stream) (void)hipStreamSynchronize(stream);
    for (GbProblem& q : prob) {
      (void)hipFree(q.tree.d_pos);
      (void)hipFree(q.d_knn_ws);
      (void)hipFree(q.d_src_idx);
      (void)hipFree(q.d_tgt_idx);
    }
    prob.clear();
    (void)hipHostFree(h_cmd);
    (void)hipFree(d_cmd);
    (void)hipHostFree(h_res);
    if (h_flag) (void)hipHostFree(h_flag);
    (void)hipFree(d_arrivals);
    h_flag = nullptr;
    d_arrivals = nullptr;
    (void)hipFree(d_res);
    if (stream) (void)hipStreamDestroy(stream);
    stream = nullptr;
  }

  unsigned char* dres(uint32_t i) const { return d_res + size_t(i) * res_stride; }
  const unsigned char* hres(uint32_t i) const { return h_res + size_t(i) * res_stride; }
  // ---- results of the last run()
  uint32_t kcnt(uint32_t i) const { return reinterpret_cast<const uint32_t*>(hres(i))[0]; }
  uint32_t overflow(uint32_t i) const { return reinterpret_cast<const uint32_t*>(hres(i))[1]; }
  uint32_t n_edges(uint32_t i) const { return reinterpret_cast<const uint32_t*>(hres(i))[2]; }
  const uint32_t* kidx(uint32_t i) const { return reinterpret_cast<const uint32_t*>(hres(i) + off_kidx); }
  const double* kdist(uint32_t i) const { return reinterpret_cast<const double*>(hres(i) + off_kdist); }
  const uint32_t* nchk(uint32_t i) const { return reinterpret_cast<const uint32_t*>(hres(i) + off_nchk); }
  const uint8_t* accept(uint32_t i) const { return hres(i) + off_accept; }
  const double* x_out(uint32_t i) const { return reinterpret_cast<const double*>(hres(i) + off_xout); }
  uint32_t selected(uint32_t i) const { return reinterpret_cast<const uint32_t*>(hres(i))[3]; }
  const uint8_t* a_accept(uint32_t i) const { return hres(i) + off_aaccept; }
  const double* a_x_out(uint32_t i) const { return reinterpret_cast<const double*>(hres(i) + off_axout); }

  // The neighbourhood of the last cmd_knn of problem i, in the order the reference's search returns it.
  // min_dist_linear_search (topological_search.hpp:244-274) keeps a bounded max-heap ordered by distance alone, so among
  // exactly equal distances both the order of the result and -- when a tie straddles the k-th place -- its members are
  // decided by the heap's history (std::push_heap / pop_heap / sort_heap), while the device returns ascending
  // (distance, index).  Equal distances only arise from coincident vertices (the bidirectional RRT* pulls the same
  // point from both trees at a joining vertex); then, and only then, the search is replayed here on the host copy of
  // the positions with the reference's own sequence of heap operations.  id[e] = vertex, slot[e] = where its edge
  // verdicts sit in accept() / x_out() (a vertex the device did not list borrows the slot of a listed vertex at the
  // same position: its walks are the same walks); the second direction of a GB_LIST_KNN_BIDIR list is at stride + slot.
  struct Neighbours {
    uint32_t K = 0, stride = 0;
    std::vector<uint32_t> id, slot;
  };
  rkh_status neighbours(uint32_t i, const double* host_pos, const double* query, Neighbours* out,
                        const uint8_t* removed = nullptr) {
    const uint32_t kc = kcnt(i), k = knn_k[i];
    const uint32_t* ki = kidx(i);
    const double* kd = kdist(i);
    out->stride = kc;
    out->K = kc < k ? kc : k;
    out->id.assign(ki, ki + out->K);
    out->slot.resize(out->K);
    for (uint32_t e = 0; e < out->K; ++e) out->slot[e] = e;
    bool tie = false;
    for (uint32_t e = 0; e + 1 < kc && e < k; ++e) tie = tie || (kd[e] == kd[e + 1]);
    if (!tie) return RKH_OK;
    ++tie_replays;
    // linear k-NN with the reference's heap (the distance is the left-to-right fp64 sum of the device kernels)
    typedef std::pair<double, uint32_t> Entry;
    auto cmp = [](const Entry& a, const Entry& b) { return a.first < b.first; };
    std::vector<Entry> heap;
    double radius = knn_radius[i];
    for (uint64_t v = 0; v < knn_n[i]; ++v) {
      if (removed && removed[v]) continue;  // not a vertex of the graph any more
      double r = 0.0;
      for (int d = 0; d < D; ++d) {
        const double df = query[d] - host_pos[v * D + d];
        r += df * df;
      }
      const double dist = std::sqrt(r);
      if (!(dist < radius)) continue;
      heap.push_back(Entry(dist, uint32_t(v)));
      std::push_heap(heap.begin(), heap.end(), cmp);
      if (heap.size() > k) {
        std::pop_heap(heap.begin(), heap.end(), cmp);
        heap.pop_back();
        radius = heap.front().first;
      }
    }
    std::sort_heap(heap.begin(), heap.end(), cmp);
    out->K = uint32_t(heap.size());
    out->id.resize(out->K);
    out->slot.resize(out->K);
    for (uint32_t e = 0; e < out->K; ++e) {
      const uint32_t v = heap[e].second;
      out->id[e] = v;
      uint32_t found = 0xFFFFFFFFu;
      for (uint32_t j = 0; j < kc && found == 0xFFFFFFFFu; ++j)
        if (ki[j] == v) found = j;
      for (uint32_t j = 0; j < kc && found == 0xFFFFFFFFu; ++j)
        if (std::memcmp(&host_pos[size_t(ki[j]) * D], &host_pos[size_t(v) * D], D * sizeof(double)) == 0) found = j;
      if (found == 0xFFFFFFFFu) {
        set_error("graph batch: equal neighbour distances between distinct positions (tie order not reproducible)");
        return RKH_ERR_UNSUPPORTED;
      }
      out->slot[e] = found;
    }
    return RKH_OK;
  }

  // the neighbourhood with its edge verdicts gathered in the reference's order: accept[e] / x_out[e] belong to the first
  // direction of neighbour id[e], accept[K + e] / x_out[K + e] to the second one of a GB_LIST_KNN_BIDIR list
  struct Verdicts {
    uint32_t K = 0;
    std::vector<uint32_t> id;
    std::vector<uint8_t> accept;
    std::vector<double> x_out;
  };
  rkh_status verdicts(uint32_t i, const double* host_pos, const double* query, Verdicts* out,
                      const uint8_t* removed = nullptr) {
    Neighbours nb;
    rkh_status st = neighbours(i, host_pos, query, &nb, removed);
    if (st != RKH_OK) return st;
    const uint32_t K = nb.K;
    out->K = K;
    out->id = nb.id;
    out->accept.resize(2 * size_t(K));
    out->x_out.resize(2 * size_t(K) * D);
    const uint8_t* acc = accept(i);
    const double* xo = x_out(i);
    for (uint32_t e = 0; e < K; ++e) {
      const uint32_t a = nb.slot[e], b = nb.stride + nb.slot[e];  // b < 2 * kmax = emax: inside the result block
      out->accept[e] = acc[a];
      out->accept[K + e] = acc[b];
      std::memcpy(&out->x_out[size_t(e) * D], &xo[size_t(a) * D], D * sizeof(double));
      std::memcpy(&out->x_out[size_t(K + e) * D], &xo[size_t(b) * D], D * sizeof(double));
    }
    return RKH_OK;
  }

  // ---- command building
  void begin() {
    for (uint32_t i = 0; i < P; ++i) {
      h_knn[i] = KnnArgs();
      h_io[i] = EdgeIO();
      h_ioa[i] = EdgeIO();
      GbAux& a = h_aux[i];
      a.append_dst = nullptr;
      a.a_count = 0;
      a.select_mode = GB_SELECT_NONE;
      a.select_dst = nullptr;
      a.a_accept = dres(i) + off_aaccept;
      a.a_xout = reinterpret_cast<const double*>(dres(i) + off_axout);
      a.sel = reinterpret_cast<uint32_t*>(dres(i)) + 3;
      a.list_mode = GB_LIST_NONE;
      a.v = 0;
      a.kidx = reinterpret_cast<const uint32_t*>(dres(i) + off_kidx);
      a.kcnt = reinterpret_cast<const uint32_t*>(dres(i));
      a.src_idx = prob[i].d_src_idx;
      a.tgt_idx = prob[i].d_tgt_idx;
      a.n_edges = reinterpret_cast<uint32_t*>(dres(i)) + 2;
    }
    any_knn = any_edges = any_append = any_stage_a = false;
  }
  // Stage A: `count` candidates of problem i, tested before this step's k-NN.
  //   GB_SELECT_POINT: is_free(pts[c]);  GB_SELECT_WALK: walk from vertex v towards pts[c] by frac[c], accepted if
  //   the distance travelled exceeds tol * target_dist[c] (random_walk).  The first accepted candidate (point, or
  //   end point of the walk) becomes the query of cmd_knn and is appended as vertex row n_dev.  The caller fills
  //   h_aux[i].pts / frac / target_dist and calls confirm_selected() after run() if selected(i) is valid.
  rkh_status cmd_stage_a(uint32_t i, uint32_t select_mode, uint32_t count, uint32_t v, double tol) {
    GbProblem& q = prob[i];
    if (count > kGbStageA || q.n_dev >= q.tree.capacity) {
      set_error("graph batch: stage-A candidate count or vertex capacity exceeded");
      return RKH_ERR_CAPACITY;
    }
    GbAux& a = h_aux[i];
    a.a_count = count;
    a.select_mode = select_mode;
    a.select_dst = q.tree.d_pos + q.n_dev * DP;
    for (uint32_t c = 0; c < kGbStageA; ++c) a.a_src[c] = v;
    EdgeIO& io = h_ioa[i];
    io.src = q.tree.d_pos;
    io.src_idx = d_aux[i].a_src;
    io.src_stride = DP;
    io.tgt = &d_aux[i].pts[0][0];
    io.tgt_stride = RKH_MAX_DOF;
    io.B = count;
    io.x_out = reinterpret_cast<double*>(dres(i) + off_axout);
    io.steps_free = reinterpret_cast<uint32_t*>(dres(i) + off_anchk);
    io.accept = dres(i) + off_aaccept;
    io.err_flag = scene->d_err;
    if (select_mode == GB_SELECT_POINT) {
      io.mode = EDGE_POINT;
    } else {
      if (dynamic)  // a walk's travel time is its fraction of the edge time; the kernel's step budget is kMaxSteps
        for (uint32_t c = 0; c < count; ++c)
          if (!(a.frac[c] * dyn.full_time <= kMaxSteps * dyn.dt)) {
            set_error("graph batch: a random walk over the dynamic space asks for more than the step budget of an edge");
            return RKH_ERR_UNSUPPORTED;
          }
      io.mode = EDGE_WALK_ACCEPT;
      io.frac = d_aux[i].frac;
      io.best_case = d_aux[i].target_dist;
      io.steer_tol = tol;
    }
    any_stage_a = true;
    return RKH_OK;
  }
  void confirm_selected(uint32_t i) { ++prob[i].n_dev; }
  // any_knn_synchro::removed_vertex: the row keeps its index and is overwritten with +inf (no sweep returns it);
  // stream-ordered before the next step's kernels
  rkh_status remove_row(uint32_t i, uint32_t row) {
    if (inf_row.empty()) inf_row.assign(64, INFINITY);
    GbProblem& q = prob[i];
    if (row >= q.n_dev) {
      set_error("graph batch: no such vertex row");
      return RKH_ERR_BAD_ARG;
    }
    RKH_HIP(hipMemcpyAsync(q.tree.d_pos + uint64_t(row) * DP, inf_row.data(), DP * sizeof(double), hipMemcpyHostToDevice,
                           stream));
    return RKH_OK;
  }
  // vertex row n_dev of problem i (the caller's vertex ids are row numbers)
  rkh_status cmd_append(uint32_t i, const double* row) {
    GbProblem& q = prob[i];
    if (q.n_dev >= q.tree.capacity || h_aux[i].append_dst) {
      set_error("graph batch: vertex capacity exceeded (or two appends in one step)");
      return RKH_ERR_CAPACITY;
    }
    GbAux& a = h_aux[i];
    for (int d = 0; d < RKH_MAX_DOF; ++d) a.append_row[d] = d < D ? row[d] : 0.0;
    a.append_dst = q.tree.d_pos + q.n_dev * DP;
    ++q.n_dev;
    any_append = true;
    return RKH_OK;
  }
  // k nearest of `query` among the first n rows, strictly inside `radius`
  rkh_status cmd_knn(uint32_t i, const double* query, uint64_t n, uint32_t k, double radius) {
    GbProblem& q = prob[i];
    if (k + 1 > kmax) {
      set_error("graph batch: k exceeds the planned maximum");
      return RKH_ERR_CAPACITY;
    }
    knn_k[i] = k;
    knn_n[i] = n;
    knn_radius[i] = radius;
    k += 1;  // see neighbours()
    for (int d = 0; d < RKH_MAX_DOF; ++d) h_aux[i].query[d] = d < D ? query[d] : 0.0;
    KnnArgs& a = h_knn[i];
    size_t bytes = 0;
    rkh_status st = knn_plan(n, 1, k, &a.ws, &bytes);
    if (st != RKH_OK) return st;
    if (bytes > kKnnWsBytes) {
      set_error("graph batch: k-NN workspace too small");
      return RKH_ERR_CAPACITY;
    }
    knn_carve(q.d_knn_ws, 1, &a.ws);
    a.ws.overflow = reinterpret_cast<uint32_t*>(dres(i)) + 1;
    a.pos = q.tree.d_pos;
    a.n = n;
    a.q = d_aux[i].query;
    a.D = D;
    a.B = 1;
    a.k = k;
    a.radius = radius;
    a.m_pow2 = next_pow2(a.ws.m_sub);
    a.out_idx = reinterpret_cast<uint32_t*>(dres(i) + off_kidx);
    a.out_dist = reinterpret_cast<double*>(dres(i) + off_kdist);
    a.out_cnt = reinterpret_cast<uint32_t*>(dres(i));
    any_knn = true;
    return RKH_OK;
  }
  void cmd_query_point(uint32_t i, const double* query) {
    for (int d = 0; d < RKH_MAX_DOF; ++d) h_aux[i].query[d] = d < D ? query[d] : 0.0;
  }
  // candidate edges of this step (see GbListMode); mode / tol as in EdgeIO
  void cmd_edges(uint32_t i, uint32_t list_mode, uint32_t v, int mode, double tol) {
    GbProblem& q = prob[i];
    GbAux& a = h_aux[i];
    a.list_mode = list_mode;
    a.v = v;
    EdgeIO& io = h_io[i];
    io.src = q.tree.d_pos;
    io.src_idx = q.d_src_idx;
    io.src_stride = DP;
    if (list_mode == GB_LIST_KNN_TO_QUERY || list_mode == GB_LIST_POINT) {
      io.tgt = d_aux[i].query;
      io.tgt_stride = 0;
    } else {
      io.tgt = q.tree.d_pos;
      io.tgt_idx = q.d_tgt_idx;
      io.tgt_stride = DP;
    }
    io.B = 0;
    io.d_B = reinterpret_cast<const uint32_t*>(dres(i)) + 2;
    io.x_out = reinterpret_cast<double*>(dres(i) + off_xout);
    io.steps_free = reinterpret_cast<uint32_t*>(dres(i) + off_nchk);
    io.accept = dres(i) + off_accept;
    io.mode = mode;
    io.steer_tol = tol;
    io.err_flag = scene->d_err;
    any_edges = true;
  }

  // the steer mapping of a step's launch: one wave per edge, or two (state_derivative_duo) while even the launch's upper
  // bound of edges leaves half the SIMDs idle (RKH_DUO_THRESHOLD, as in the batch planner)
  int duo_lanes(uint32_t edges_per_problem) const {
    static const uint32_t thr = [] { const char* e = getenv("RKH_DUO_THRESHOLD"); return e ? uint32_t(std::max(0, atoi(e))) : 512u; }();
    return (uint64_t(edges_per_problem) * P <= thr && !scene->host.has_meshes) ? 128 : 64;
  }

  rkh_status run() {
    hipStream_t s = stream;
    RKH_HIP(hipMemcpyAsync(d_cmd, h_cmd, cmd_bytes, hipMemcpyHostToDevice, s));
    if (any_append) hipLaunchKernelGGL(gb_prep_kernel, dim3(P), dim3(64), 0, s, d_aux, DP);
    if (any_stage_a) {
      rkh_status st = dynamic ? launch_propagate(s, n_dof, scene->host.n_env, scene->d_scene, scene->d_pairs, scene->n_pairs_verdict,
                                                 dyn, EdgeIO(), kGbStageA, nullptr, 0, duo_lanes(kGbStageA), d_ioa, nullptr, P)
                              : launch_edge_check(s, n_dof, scene->host.n_env, scene->d_scene, scene->d_pairs,
                                                  scene->n_pairs_verdict, qs, EdgeIO(), kGbStageA, nullptr, 0, d_ioa, nullptr, P);
      if (st != RKH_OK) return st;
      hipLaunchKernelGGL(gb_select_kernel, dim3(P), dim3(64), 0, s, d_aux, D, DP);
    }
    if (any_knn) {
      rkh_status st = launch_nnk_table(s, D, d_knn, h_knn, P);
      if (st != RKH_OK) return st;
    }
    hipLaunchKernelGGL(gb_list_kernel, dim3(P), dim3(64), 0, s, d_aux);
    if (any_edges) {
      // one wave per edge: a step holds at most 2 k candidates per problem, far from filling the two-lanes mappings
      rkh_status st = dynamic ? launch_propagate(s, n_dof, scene->host.n_env, scene->d_scene, scene->d_pairs, scene->n_pairs_verdict,
                                                 dyn, EdgeIO(), emax, nullptr, 0, duo_lanes(emax), d_io, nullptr, P)
                              : launch_edge_check(s, n_dof, scene->host.n_env, scene->d_scene, scene->d_pairs,
                                                  scene->n_pairs_verdict, qs, EdgeIO(), emax, nullptr, 0, d_io, nullptr, P);
      if (st != RKH_OK) return st;
    }
    if (spin_download) {
      const uint32_t tag = ++flag_seq;
      hipLaunchKernelGGL(gb_download_kernel, dim3(P), dim3(256), 0, s, reinterpret_cast<const uint4*>(d_res),
                         reinterpret_cast<uint4*>(h_res_dev), uint32_t(res_stride / 16), d_arrivals, h_flag_dev, tag);
      RKH_HIP(hipGetLastError());
      for (uint32_t spins = 0; *reinterpret_cast<volatile uint32_t*>(h_flag) != tag; ++spins) {
        __builtin_ia32_pause();
        if ((spins & 0xFFFFu) == 0xFFFFu) {  // a failed launch or a device fault must not hang the host
          const hipError_t q = hipStreamQuery(s);
          if (q != hipSuccess && q != hipErrorNotReady) RKH_HIP(q);
          if (q == hipSuccess && *reinterpret_cast<volatile uint32_t*>(h_flag) != tag) {
            set_error("graph batch: the step's results never arrived");
            return RKH_ERR_DEVICE;
          }
        }
      }
    } else {
      RKH_HIP(hipMemcpyAsync(h_res, d_res, res_stride * P, hipMemcpyDeviceToHost, s));
      RKH_HIP(hipStreamSynchronize(s));
    }
    ++steps;
    for (uint32_t i = 0; i < P; ++i)
      if (h_knn[i].B && overflow(i)) {
        set_error("graph batch: k-NN candidate capacity exceeded");
        return RKH_ERR_CAPACITY;
      }
    return RKH_OK;
  }
};

}  // namespace rkh
