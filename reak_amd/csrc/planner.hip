// planner.hip -- speculative-batch RRT driver: rrt_planner::solve_planning_query with
// LINEAR_SEARCH_KNN / UNIDIRECTIONAL_PLANNING (ctrl/path_planning/rrt_path_planner.tpp:66-145)
// -> generate_rrt (ctrl/graph_alg/rr_tree.hpp:179-199) over the steerable dynamic space.
//
// generate_rrt is a strict recurrence (sample i's nearest neighbour depends on the vertices added
// by samples < i), but with linear-search NN every RNG draw is a sample coordinate, so the sample
// stream is known in advance.  One round takes the next B samples and, against the tree snapshot:
//   1. nn1 sweep            (nn_sweep.hip)    nearest snapshot vertex of every sample
//   2. propagate            (propagate.hip)   steer + collision-check all B candidate edges, accept test
//   3. fixup  (this file)   candidate b is INVALID iff a vertex that an earlier accepted candidate of the
//                           same round would add is strictly closer to sample b than its snapshot NN
//                           (strict '<' = first-minimum-wins, new vertices have higher indices)
//   4. commit (this file)   everything before the first invalid candidate is exactly what the sequential
//                           algorithm does: append accepted end states in order, log nn/accept per sample
//   5. goal probes          (propagate.hip)   edge_added's query.get_distance_to_goal for the new vertices
// The next round restarts at the first invalid sample.  Vertex ids, parents, sample consumption and the
// stop condition therefore equal the sequential planner's; all state (vertex count, stream offset,
// batch size) lives on the device so rounds can be enqueued back to back without host round trips.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <random>

#include "rkh_internal.h"

namespace rkh {

int nn_padded_dims(int D);

struct PlannerState {  // device-resident
  uint32_t n;            // vertices in the tree
  uint32_t s0;           // next sample (== generate_rrt iterations so far)
  uint32_t B;            // candidates of the current round
  uint32_t F;            // first invalid candidate of the current round
  uint32_t n_before;     // first vertex whose goal probe is still pending
  uint32_t n_new;        // number of vertices whose goal probe is pending (they ride in the next propagate launch)
  uint32_t probed_n;     // vertices [1, probed_n) have their goal probe result in goal_dist
  uint32_t done;         // keep_going() == false (vertex budget reached) or samples exhausted (2)
  uint32_t max_total;    // max_vertices + 1 (root is not counted by m_iteration_count)
  uint32_t samples_ready;  // samples uploaded so far
  uint32_t b_max;
  float batch_factor;
  uint32_t b_min;
  unsigned long long rounds, edges_speculated, fixup_cut;
};

__global__ void round_begin_kernel(PlannerState* st, uint32_t* round_n, uint32_t round_slot) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (round_n) round_n[round_slot] = st->done ? 0u : st->n;
  uint32_t B = 0;
  if (!st->done) {
    const float want = st->batch_factor * sqrtf(float(st->n));
    B = uint32_t(want);
    if (B < st->b_min) B = st->b_min;
    if (B > st->b_max) B = st->b_max;
    const uint32_t avail = st->samples_ready - st->s0;
    if (B > avail) B = avail;
    if (B == 0) st->done = 2;  // sample stream exhausted: host must upload more
  }
  st->B = B;
  st->F = B;
  if (B) {
    st->rounds += 1;
    st->edges_speculated += B;
  }
}

// One wave per candidate b: smallest squared distance from sample b to the end states of accepted
// candidates j < b.  min_j sqrt(s_j) == sqrt(min_j s_j) (sqrt is monotone), so one sqrt decides.
template <int DP>
__global__ __launch_bounds__(256) void fixup_kernel(PlannerState* __restrict__ st, const double* __restrict__ samples,
                                                     int D, const double* __restrict__ x_out,
                                                     const uint8_t* __restrict__ accept,
                                                     const double* __restrict__ nn_dist) {
  const uint32_t B = st->B;
  const uint32_t b = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (b >= B || b == 0) return;
  const double* q = samples + (uint64_t(st->s0) + b) * D;
  double qv[DP];
#pragma unroll
  for (int d = 0; d < DP; ++d) qv[d] = d < D ? q[d] : 0.0;
  double smin = INFINITY;
  for (uint32_t j = lane; j < b; j += 64) {
    if (!accept[j]) continue;
    const double* p = x_out + uint64_t(j) * D;
    double df = qv[0] - p[0];
    double s = df * df;
#pragma unroll
    for (int d = 1; d < DP; ++d) {
      df = qv[d] - (d < D ? p[d] : 0.0);
      s = s + df * df;
    }
    if (s < smin) smin = s;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const double o = __shfl_xor(smin, off, 64);
    if (o < smin) smin = o;
  }
  if (lane == 0 && sqrt(smin) < nn_dist[b]) atomicMin(&st->F, b);
}

// Single block: commit candidates [0, F) in order (prefix scan of the accept flags), honouring the
// vertex budget of keep_going() (motion_planner_base.hpp:355-374).
__global__ __launch_bounds__(1024) void commit_kernel(PlannerState* __restrict__ st, double* __restrict__ tree, int D,
                                                       int DP, const double* __restrict__ x_out,
                                                       const uint8_t* __restrict__ accept,
                                                       const uint32_t* __restrict__ nn_idx, uint32_t* __restrict__ parent,
                                                       uint32_t* __restrict__ node_sample, uint32_t* __restrict__ nn_seq,
                                                       uint8_t* __restrict__ accept_log) {
  __shared__ uint32_t scan[1024];
  __shared__ uint32_t carry;
  __shared__ uint32_t cut;  // number of candidates actually consumed
  const uint32_t F = st->F;
  const uint32_t n0 = st->n;
  const uint32_t s0 = st->s0;
  const uint32_t budget = st->max_total - n0;  // vertices that may still be added
  if (threadIdx.x == 0) {
    carry = 0;
    cut = F;
  }
  __syncthreads();
  for (uint32_t base = 0; base < F; base += 1024) {
    const uint32_t b = base + threadIdx.x;
    const uint32_t a = (b < F && accept[b]) ? 1u : 0u;
    scan[threadIdx.x] = a;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) {  // Hillis-Steele inclusive scan
      uint32_t v = 0;
      if (threadIdx.x >= off) v = scan[threadIdx.x - off];
      __syncthreads();
      scan[threadIdx.x] += v;
      __syncthreads();
    }
    const uint32_t incl = carry + scan[threadIdx.x];  // accepted among [0, b]
    if (b < F) {
      // the candidate whose vertex exhausts the budget is the last one the sequential loop runs
      if (a && incl == budget) atomicMin(&cut, b + 1);
      if (incl <= budget && (incl < budget || a)) {
        // consumed by the sequential loop (it stops right after the budget-filling vertex)
        nn_seq[s0 + b] = nn_idx[b];
        accept_log[s0 + b] = uint8_t(a);
        if (a) {
          const uint32_t row = n0 + incl - 1;
          for (int d = 0; d < DP; ++d) tree[uint64_t(row) * DP + d] = d < D ? x_out[uint64_t(b) * D + d] : 0.0;
          parent[row] = nn_idx[b];
          node_sample[row] = s0 + b;
        }
      }
    }
    __syncthreads();
    if (threadIdx.x == 1023) carry = incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    uint32_t added = carry < budget ? carry : budget;
    st->n = n0 + added;
    // the propagate launch of this round also ran the goal probes that were pending: [n_before, n_before+n_new)
    st->probed_n = st->n_before + st->n_new;
    st->n_new = added;  // this round's vertices are probed by the next launch
    st->n_before = n0;
    st->s0 = s0 + cut;
    st->fixup_cut += (st->B - F);
    if (st->n >= st->max_total) st->done = 1;
  }
}

// after a probe-only flush launch: nothing is pending any more
__global__ void probes_flushed_kernel(PlannerState* st) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  st->probed_n = st->n_before + st->n_new;
  st->n_before = st->n;
  st->n_new = 0;
}

}  // namespace rkh

using namespace rkh;

struct rkh_planner {
  rkh_scene* scene = nullptr;
  hipStream_t stream = nullptr;
  rkh_dyn_space space;
  rkh_rrt_params prm;
  DynDev dyn;
  int n_dof = 0, D = 0, DP = 0;
  // tree
  NnStore tree;
  uint32_t* d_parent = nullptr;
  uint32_t* d_node_sample = nullptr;
  double* d_goal_dist = nullptr;
  uint64_t capacity = 0;
  // sample stream
  std::mt19937 eng;
  std::vector<double> h_chunk;
  double* d_samples = nullptr;
  uint64_t sample_cap = 0, samples_ready = 0;
  uint32_t* d_nn_seq = nullptr;
  uint8_t* d_accept_log = nullptr;
  // per-round scratch
  uint32_t b_max = 1024;
  uint32_t* d_nn_idx = nullptr;
  double* d_nn_dist = nullptr;
  double* d_x_out = nullptr;
  uint32_t* d_steps = nullptr;
  uint8_t* d_accept = nullptr;
  double* d_probe_x = nullptr;
  uint32_t* d_probe_steps = nullptr;
  double* d_goal = nullptr;
  double* d_part_dist = nullptr;
  uint32_t* d_part_idx = nullptr;
  uint32_t part_blocks = 0;
  PlannerState* d_state = nullptr;
  PlannerState h_state;
  // host-side solution bookkeeping (register_basic_solution_path_impl, solution_path_factories.hpp:58-110)
  uint64_t goal_checked = 0;  // vertices whose goal probe has been examined
  uint64_t num_solutions = 0;
  double best_cost = INFINITY;
  bool truncated = false;
  uint64_t final_n = 0, final_iterations = 0;
  // optional HIP-event timing of the NN sweep kernel (RKH_PROFILE_NN=1)
  bool profile_nn = false;
  std::vector<hipEvent_t> ev;  // pairs
  uint32_t* d_round_n = nullptr;
  uint32_t prof_rounds = 0;
  static constexpr uint32_t kProfMax = 8192;
};

namespace {

rkh_status upload_samples(rkh_planner* p, uint64_t upto) {
  // hyperbox_topology::random_point (hyperbox_topology.hpp:97-103): D draws of uniform_01 per sample,
  // uniform_01<mt19937&,double> = eng() * 2^-32 (Boost.Random; one 32-bit draw per coordinate)
  if (upto > p->sample_cap) upto = p->sample_cap;
  if (upto <= p->samples_ready) return RKH_OK;
  const uint64_t cnt = upto - p->samples_ready;
  const int D = p->D;
  p->h_chunk.resize(cnt * D);
  for (uint64_t i = 0; i < cnt; ++i)
    for (int d = 0; d < D; ++d) {
      double u;
      do {
        u = double(p->eng()) * (1.0 / 4294967296.0);
      } while (!(u < 1.0));
      p->h_chunk[i * D + d] = p->space.lower[d] + u * (p->space.upper[d] - p->space.lower[d]);
    }
  RKH_HIP(hipMemcpyAsync(p->d_samples + p->samples_ready * D, p->h_chunk.data(), cnt * D * sizeof(double),
                         hipMemcpyHostToDevice, p->stream));
  RKH_HIP(hipStreamSynchronize(p->stream));  // h_chunk is reused
  p->samples_ready = upto;
  const uint32_t sr = uint32_t(upto);
  RKH_HIP(hipMemcpyAsync(&p->d_state->samples_ready, &sr, sizeof(uint32_t), hipMemcpyHostToDevice, p->stream));
  RKH_HIP(hipStreamSynchronize(p->stream));
  return RKH_OK;
}

template <int DP>
void launch_fixup(rkh_planner* p) {
  hipLaunchKernelGGL((fixup_kernel<DP>), dim3((p->b_max + 3) / 4), dim3(256), 0, p->stream, p->d_state, p->d_samples, p->D,
                     p->d_x_out, p->d_accept, p->d_nn_dist);
}

void make_edge_ios(rkh_planner* p, EdgeIO* io_out, EdgeIO* gp_out) {
  EdgeIO io;
  io.src = p->tree.d_pos;
  io.src_idx = p->d_nn_idx;
  io.src_stride = p->DP;
  io.tgt = p->d_samples;
  io.d_tgt_off = &p->d_state->s0;
  io.tgt_stride = p->D;
  io.B = p->b_max;
  io.d_B = &p->d_state->B;
  io.x_out = p->d_x_out;
  io.steps_free = p->d_steps;
  io.mode = EDGE_STEER_ACCEPT;
  io.best_case = p->d_nn_dist;
  io.steer_tol = p->prm.steer_tol;
  io.accept = p->d_accept;
  io.err_flag = p->scene->d_err;
  EdgeIO gp;
  gp.src = p->tree.d_pos;
  gp.d_src_first = &p->d_state->n_before;
  gp.src_stride = p->DP;
  gp.tgt = p->d_goal;
  gp.tgt_stride = 0;
  gp.B = p->b_max;
  gp.d_B = &p->d_state->n_new;
  gp.x_out = p->d_probe_x;
  gp.steps_free = p->d_probe_steps;
  gp.mode = EDGE_GOAL_PROBE;
  gp.goal_dist = p->d_goal_dist;
  gp.err_flag = p->scene->d_err;
  *io_out = io;
  *gp_out = gp;
}

// goal probes still pending after the last enqueued round
rkh_status flush_probes(rkh_planner* p) {
  EdgeIO io, gp;
  make_edge_ios(p, &io, &gp);
  rkh_status st = launch_propagate(p->stream, p->n_dof, p->scene->host.n_env, p->scene->d_scene, p->scene->d_pairs,
                                   p->scene->n_pairs, p->dyn, gp, p->b_max);
  if (st != RKH_OK) return st;
  hipLaunchKernelGGL(probes_flushed_kernel, dim3(1), dim3(1), 0, p->stream, p->d_state);
  RKH_HIP(hipGetLastError());
  return RKH_OK;
}

rkh_status enqueue_round(rkh_planner* p) {
  hipStream_t s = p->stream;
  const int D = p->D;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  uint32_t slot = 0;
  if (p->profile_nn && p->prof_rounds < rkh_planner::kProfMax) {
    slot = p->prof_rounds++;
    if (p->ev.size() < 2 * size_t(slot + 1)) {
      hipEvent_t a, b;
      RKH_HIP(hipEventCreate(&a));
      RKH_HIP(hipEventCreate(&b));
      p->ev.push_back(a);
      p->ev.push_back(b);
    }
    ev0 = p->ev[2 * slot];
    ev1 = p->ev[2 * slot + 1];
  }
  hipLaunchKernelGGL(round_begin_kernel, dim3(1), dim3(1), 0, s, p->d_state, ev0 ? p->d_round_n : nullptr, slot);
  // 1. NN sweep of the round's samples over the snapshot
  const uint64_t n_upper = std::min<uint64_t>(p->capacity, uint64_t(p->prm.max_vertices) + 1);
  rkh_status st = launch_nn1(s, p->tree, n_upper, &p->d_state->n, p->d_samples, &p->d_state->s0, p->b_max,
                             &p->d_state->B, p->d_nn_idx, p->d_nn_dist, p->d_part_dist, p->d_part_idx, p->part_blocks, ev0, ev1);
  if (st != RKH_OK) return st;
  // 2. speculative steer of all candidates; the same launch carries the goal probes (edge_added,
  //    planning_visitors.hpp:194-200) of the vertices the previous round committed
  EdgeIO io, gp;
  make_edge_ios(p, &io, &gp);
  st = launch_propagate(s, p->n_dof, p->scene->host.n_env, p->scene->d_scene, p->scene->d_pairs, p->scene->n_pairs,
                        p->dyn, io, p->b_max, &gp, p->b_max);
  if (st != RKH_OK) return st;
  // 3. fix-up against the vertices this round itself would add
  switch (p->DP) {
    case 2: launch_fixup<2>(p); break;
    case 4: launch_fixup<4>(p); break;
    case 6: launch_fixup<6>(p); break;
    case 8: launch_fixup<8>(p); break;
    case 12: launch_fixup<12>(p); break;
    case 16: launch_fixup<16>(p); break;
    default: set_error("planner: unsupported state dimension"); return RKH_ERR_UNSUPPORTED;
  }
  // 4. commit the valid prefix
  hipLaunchKernelGGL(commit_kernel, dim3(1), dim3(1024), 0, s, p->d_state, p->tree.d_pos, D, p->DP, p->d_x_out,
                     p->d_accept, p->d_nn_idx, p->d_parent, p->d_node_sample, p->d_nn_seq, p->d_accept_log);
  RKH_HIP(hipGetLastError());
  return RKH_OK;
}

}  // namespace

extern "C" {

rkh_status rkh_planner_create(rkh_scene* scene, const rkh_dyn_space* space, const rkh_rrt_params* prm,
                              rkh_planner** out) {
  if (!scene || !space || !prm || !out) return RKH_ERR_BAD_ARG;
  if (space->n_dof != scene->host.n_dof || prm->max_vertices < 1) {
    set_error("rkh_planner_create: n_dof mismatch or max_vertices < 1");
    return RKH_ERR_BAD_ARG;
  }
  rkh_planner* p = new rkh_planner();
  p->scene = scene;
  p->space = *space;
  p->prm = *prm;
  p->n_dof = space->n_dof;
  p->D = 2 * space->n_dof;
  p->DP = nn_padded_dims(p->D);
  rkh_status st = build_dyn_dev(*space, 1.0, &p->dyn);
  if (st != RKH_OK) { delete p; return st; }
  RKH_HIP(hipSetDevice(scene->ctx->device));
  RKH_HIP(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking));
  if (const char* e = getenv("RKH_BATCH_MAX")) p->b_max = std::max(8, atoi(e));
  p->b_max = std::min<uint32_t>(p->b_max, 4096);
  const uint64_t max_total = uint64_t(prm->max_vertices) + 1;
  p->capacity = (max_total + 255) / 256 * 256;
  p->tree.D = p->D;
  p->tree.capacity = p->capacity;
  RKH_HIP(hipMalloc(&p->tree.d_pos, p->capacity * p->DP * sizeof(double)));
  RKH_HIP(hipMalloc(&p->d_parent, p->capacity * sizeof(uint32_t)));
  RKH_HIP(hipMalloc(&p->d_node_sample, p->capacity * sizeof(uint32_t)));
  RKH_HIP(hipMalloc(&p->d_goal_dist, p->capacity * sizeof(double)));
  p->sample_cap = std::max<uint64_t>(4 * max_total + 4 * p->b_max, 1u << 14);
  if (const char* e = getenv("RKH_SAMPLE_CAP")) p->sample_cap = std::max<uint64_t>(p->sample_cap, strtoull(e, nullptr, 10));
  RKH_HIP(hipMalloc(&p->d_samples, p->sample_cap * p->D * sizeof(double)));
  RKH_HIP(hipMalloc(&p->d_nn_seq, p->sample_cap * sizeof(uint32_t)));
  RKH_HIP(hipMalloc(&p->d_accept_log, p->sample_cap));
  RKH_HIP(hipMalloc(&p->d_nn_idx, p->b_max * sizeof(uint32_t)));
  RKH_HIP(hipMalloc(&p->d_nn_dist, p->b_max * sizeof(double)));
  RKH_HIP(hipMalloc(&p->d_x_out, uint64_t(p->b_max) * p->D * sizeof(double)));
  RKH_HIP(hipMalloc(&p->d_steps, p->b_max * sizeof(uint32_t)));
  RKH_HIP(hipMalloc(&p->d_accept, p->b_max));
  RKH_HIP(hipMalloc(&p->d_probe_x, uint64_t(p->b_max) * p->D * sizeof(double)));
  RKH_HIP(hipMalloc(&p->d_probe_steps, p->b_max * sizeof(uint32_t)));
  RKH_HIP(hipMalloc(&p->d_goal, p->D * sizeof(double)));
  p->part_blocks = nn1_partial_blocks(p->capacity, p->b_max);
  // the grid may be re-derived for smaller B with more blocks: size for the worst case (gy = 1)
  p->part_blocks = std::max<uint32_t>(p->part_blocks, nn1_partial_blocks(p->capacity, 1));
  RKH_HIP(hipMalloc(&p->d_part_dist, uint64_t(p->part_blocks) * p->b_max * sizeof(double)));
  RKH_HIP(hipMalloc(&p->d_part_idx, uint64_t(p->part_blocks) * p->b_max * sizeof(uint32_t)));
  RKH_HIP(hipMalloc(&p->d_state, sizeof(PlannerState)));
  if (const char* e = getenv("RKH_PROFILE_NN")) p->profile_nn = atoi(e) != 0;
  if (p->profile_nn) RKH_HIP(hipMalloc(&p->d_round_n, rkh_planner::kProfMax * sizeof(uint32_t)));
  // root vertex = query start (create_root, rrt_path_planner.tpp:131-133)
  std::vector<double> row(p->DP, 0.0);
  for (int d = 0; d < p->D; ++d) row[d] = prm->start[d];
  RKH_HIP(hipMemcpy(p->tree.d_pos, row.data(), p->DP * sizeof(double), hipMemcpyHostToDevice));
  const uint32_t no_parent = 0xFFFFFFFFu;
  RKH_HIP(hipMemcpy(p->d_parent, &no_parent, sizeof(uint32_t), hipMemcpyHostToDevice));
  RKH_HIP(hipMemcpy(p->d_goal, prm->goal, p->D * sizeof(double), hipMemcpyHostToDevice));
  PlannerState& hs = p->h_state;
  std::memset(&hs, 0, sizeof(hs));
  hs.n = 1;
  hs.n_before = 1;
  hs.probed_n = 1;
  hs.max_total = uint32_t(max_total);
  hs.b_max = p->b_max;
  hs.b_min = 8;
  hs.batch_factor = 2.0f;
  if (const char* e = getenv("RKH_BATCH_FACTOR")) hs.batch_factor = float(atof(e));
  if (const char* e = getenv("RKH_BATCH_MIN")) hs.b_min = std::max(1, atoi(e));
  RKH_HIP(hipMemcpy(p->d_state, &hs, sizeof(hs), hipMemcpyHostToDevice));
  p->eng.seed(prm->seed);
  *out = p;
  return RKH_OK;
}

rkh_status rkh_planner_destroy(rkh_planner* p) {
  if (!p) return RKH_OK;
  hipStreamSynchronize(p->stream);
  void* bufs[] = {p->tree.d_pos, p->d_parent, p->d_node_sample, p->d_goal_dist, p->d_samples, p->d_nn_seq,
                  p->d_accept_log, p->d_nn_idx, p->d_nn_dist, p->d_x_out, p->d_steps, p->d_accept, p->d_probe_x,
                  p->d_probe_steps, p->d_goal, p->d_part_dist, p->d_part_idx, p->d_state};
  for (void* b : bufs) hipFree(b);
  hipFree(p->d_round_n);
  for (hipEvent_t e : p->ev) hipEventDestroy(e);
  hipStreamDestroy(p->stream);
  delete p;
  return RKH_OK;
}

void* rkh_planner_stream(rkh_planner* p) { return p ? (void*)p->stream : nullptr; }

rkh_status rkh_planner_nn_profile(rkh_planner* p, double* total_ms, uint64_t* total_bytes, uint64_t* launches) {
  if (!p || !total_ms || !total_bytes || !launches) return RKH_ERR_BAD_ARG;
  *total_ms = 0.0;
  *total_bytes = 0;
  *launches = 0;
  if (!p->profile_nn || p->prof_rounds == 0) return RKH_OK;
  RKH_HIP(hipStreamSynchronize(p->stream));
  std::vector<uint32_t> rn(p->prof_rounds);
  RKH_HIP(hipMemcpy(rn.data(), p->d_round_n, rn.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
  for (uint32_t r = 0; r < p->prof_rounds; ++r) {
    if (rn[r] == 0) continue;  // no-op round after completion
    float ms = 0.f;
    RKH_HIP(hipEventElapsedTime(&ms, p->ev[2 * r], p->ev[2 * r + 1]));
    *total_ms += ms;
    *total_bytes += uint64_t(rn[r]) * p->DP * sizeof(double);  // algorithmic bytes of one sweep: n * D * 8
    *launches += 1;
  }
  return RKH_OK;
}

rkh_status rkh_planner_enqueue(rkh_planner* p, uint32_t rounds) {
  if (!p) return RKH_ERR_BAD_ARG;
  if (p->truncated) return RKH_OK;
  // make sure the enqueued rounds cannot run out of samples
  const uint64_t need = uint64_t(p->h_state.s0) + uint64_t(rounds) * p->b_max;
  rkh_status st = upload_samples(p, need);
  if (st != RKH_OK) return st;
  for (uint32_t r = 0; r < rounds; ++r) {
    st = enqueue_round(p);
    if (st != RKH_OK) return st;
  }
  return RKH_OK;
}

rkh_status rkh_planner_sync(rkh_planner* p, rkh_planner_stats* stats) {
  if (!p) return RKH_ERR_BAD_ARG;
  RKH_HIP(hipMemcpyAsync(&p->h_state, p->d_state, sizeof(PlannerState), hipMemcpyDeviceToHost, p->stream));
  RKH_HIP(hipStreamSynchronize(p->stream));
  int flag = 0;
  RKH_HIP(hipMemcpy(&flag, p->scene->d_err, sizeof(int), hipMemcpyDeviceToHost));
  if (flag != 0) {
    RKH_HIP(hipMemset(p->scene->d_err, 0, sizeof(int)));
    set_error("planner: mass matrix is singular (Cholesky pivot < 1e-8)");
    return rkh_status(flag);
  }
  PlannerState& hs = p->h_state;
  if (hs.done == 1 && hs.probed_n < hs.n) {  // finished: run the goal probes of the last committed vertices
    rkh_status fs = flush_probes(p);
    if (fs != RKH_OK) return fs;
    RKH_HIP(hipMemcpyAsync(&p->h_state, p->d_state, sizeof(PlannerState), hipMemcpyDeviceToHost, p->stream));
    RKH_HIP(hipStreamSynchronize(p->stream));
  }
  if (hs.done == 2 && p->samples_ready < p->sample_cap) {  // sample stream ran dry mid-enqueue: refill and carry on
    hs.done = 0;
    RKH_HIP(hipMemcpy(&p->d_state->done, &hs.done, sizeof(uint32_t), hipMemcpyHostToDevice));
  }
  // edge_added: a finite goal-probe distance registers a solution if it beats the best so far
  // (planning_visitors.hpp:194-200, solution_path_factories.hpp:58-110); keep_going() then also checks
  // max_num_results (p2p_planning_query.hpp:121-123).
  const uint64_t probed = hs.probed_n < 1 ? 1 : hs.probed_n;  // vertices [1, probed) have a goal-probe result
  if (!p->truncated && probed > 1 && p->goal_checked < probed - 1) {
    const uint64_t first = p->goal_checked, cnt = probed - 1 - first;
    std::vector<double> gd(cnt);
    RKH_HIP(hipMemcpy(gd.data(), p->d_goal_dist + first, cnt * sizeof(double), hipMemcpyDeviceToHost));
    std::vector<double> pos;
    std::vector<uint32_t> par;
    for (uint64_t i = 0; i < cnt; ++i) {
      if (!(gd[i] < INFINITY)) continue;
      if (pos.empty()) {
        pos.resize(uint64_t(hs.n) * p->DP);
        par.resize(hs.n);
        RKH_HIP(hipMemcpy(pos.data(), p->tree.d_pos, pos.size() * sizeof(double), hipMemcpyDeviceToHost));
        RKH_HIP(hipMemcpy(par.data(), p->d_parent, par.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
      }
      double total = gd[i];
      uint64_t v = first + i + 1;
      while (par[v] != 0xFFFFFFFFu) {
        const uint64_t pv = par[v];
        double acc = 0.0;
        for (int d = 0; d < p->D; ++d) {
          const double df = pos[pv * p->DP + d] - pos[v * p->DP + d];
          acc += df * df;
        }
        total += std::sqrt(acc);
        v = pv;
      }
      if (p->num_solutions == 0 || total < p->best_cost) {
        p->best_cost = total;
        ++p->num_solutions;
        if (p->num_solutions >= p->prm.max_results) {
          // the sequential planner stops right after this vertex: drop what speculation added beyond it
          p->truncated = true;
          p->final_n = first + i + 2;
          uint32_t smp = 0;
          RKH_HIP(hipMemcpy(&smp, p->d_node_sample + (first + i + 1), sizeof(uint32_t), hipMemcpyDeviceToHost));
          p->final_iterations = uint64_t(smp) + 1;
          break;
        }
      }
    }
    p->goal_checked = probed - 1;
  }
  if (stats) {
    std::memset(stats, 0, sizeof(*stats));
    stats->num_vertices = p->truncated ? p->final_n : hs.n;
    stats->iterations = p->truncated ? p->final_iterations : hs.s0;
    stats->edges_checked = stats->iterations + (stats->num_vertices - 1);
    stats->edges_speculated = hs.edges_speculated + (hs.n - 1);
    stats->rounds = hs.rounds;
    stats->num_solutions = p->num_solutions;
    stats->best_cost = p->best_cost;
    stats->done = (p->truncated || hs.done == 1) ? 1u : 0u;
  }
  return RKH_OK;
}

rkh_status rkh_planner_solve(rkh_planner* p, rkh_planner_stats* stats) {
  if (!p) return RKH_ERR_BAD_ARG;
  rkh_planner_stats local;
  for (;;) {
    rkh_status st = rkh_planner_enqueue(p, 16);
    if (st != RKH_OK) return st;
    st = rkh_planner_sync(p, &local);
    if (st != RKH_OK) return st;
    if (local.done) break;
    if (p->samples_ready >= p->sample_cap && p->h_state.s0 + p->b_max > p->sample_cap) {
      set_error("planner: sample stream capacity exhausted (raise RKH_SAMPLE_CAP)");
      return RKH_ERR_CAPACITY;
    }
  }
  if (stats) *stats = local;
  return RKH_OK;
}

rkh_status rkh_planner_get_tree(rkh_planner* p, double* pos, uint32_t* parent, uint32_t* nn_seq, uint8_t* accept,
                                double* goal_dist) {
  if (!p) return RKH_ERR_BAD_ARG;
  if (goal_dist) {  // make sure no goal probe is pending
    rkh_status fs = flush_probes(p);
    if (fs != RKH_OK) return fs;
  }
  RKH_HIP(hipStreamSynchronize(p->stream));
  const uint64_t n = p->truncated ? p->final_n : p->h_state.n;
  const uint64_t it = p->truncated ? p->final_iterations : p->h_state.s0;
  if (pos) {
    if (p->DP == p->D) {
      RKH_HIP(hipMemcpy(pos, p->tree.d_pos, n * p->D * sizeof(double), hipMemcpyDeviceToHost));
    } else {
      std::vector<double> tmp(n * p->DP);
      RKH_HIP(hipMemcpy(tmp.data(), p->tree.d_pos, tmp.size() * sizeof(double), hipMemcpyDeviceToHost));
      for (uint64_t i = 0; i < n; ++i) std::memcpy(pos + i * p->D, &tmp[i * p->DP], p->D * sizeof(double));
    }
  }
  if (parent) RKH_HIP(hipMemcpy(parent, p->d_parent, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
  if (nn_seq && it) RKH_HIP(hipMemcpy(nn_seq, p->d_nn_seq, it * sizeof(uint32_t), hipMemcpyDeviceToHost));
  if (accept && it) RKH_HIP(hipMemcpy(accept, p->d_accept_log, it, hipMemcpyDeviceToHost));
  if (goal_dist && n > 1) RKH_HIP(hipMemcpy(goal_dist, p->d_goal_dist, (n - 1) * sizeof(double), hipMemcpyDeviceToHost));
  return RKH_OK;
}

}  // extern "C"
