// planner.hip -- speculative-batch RRT driver: rrt_planner::solve_planning_query with
// LINEAR_SEARCH_KNN / UNIDIRECTIONAL_PLANNING (ctrl/path_planning/rrt_path_planner.tpp:66-145)
// -> generate_rrt (ctrl/graph_alg/rr_tree.hpp:179-199) over the steerable dynamic space, for a batch of
// P independent planning problems (seeds / queries) on one scene.
//
// generate_rrt is a strict recurrence (sample i's nearest neighbour depends on the vertices added
// by samples < i), but with linear-search NN every RNG draw is a sample coordinate, so the sample
// stream is known in advance.  One round takes, per problem, the next B samples and, against the tree
// snapshot:
//   1. nn1 sweep            (nn_sweep.hip)    nearest snapshot vertex of every sample
//   2. propagate            (propagate.hip)   steer + collision-check all B candidate edges, accept test;
//                                             the same launch runs the goal probes (edge_added's
//                                             query.get_distance_to_goal) of the previous round's new vertices
//   3. fixup  (this file)   candidate b is INVALID iff a vertex that an earlier accepted candidate of the
//                           same round would add is strictly closer to sample b than its snapshot NN
//                           (strict '<' = first-minimum-wins, new vertices have higher indices)
//   4. commit (this file)   everything before the first invalid candidate is exactly what the sequential
//                           algorithm does: append accepted end states in order, log nn/accept per sample
// The next round restarts at the first invalid sample.  Vertex ids, parents, sample consumption and the
// stop condition therefore equal the sequential planner's.  All P problems share each kernel launch
// (blockIdx.y/z = problem) and all per-problem state (vertex count, stream offset, batch size) lives on the
// device, so rounds are enqueued back to back without host round trips and the grids are large enough to fill
// the chip.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>

#include "nn_mirror.h"
#include "rkh_internal.h"

namespace rkh {

int nn_padded_dims(int D);

struct PlannerState {  // device-resident, one per problem
  uint32_t n;            // vertices in the tree
  uint32_t s0;           // next sample (== generate_rrt iterations so far)
  uint32_t B;            // candidates of the current round
  uint32_t F;            // first invalid candidate of the current round
  uint32_t n_before;     // first vertex whose goal probe is still pending
  uint32_t n_new;        // number of vertices whose goal probe is pending (they ride in the next propagate launch)
  uint32_t probed_n;     // vertices [1, probed_n) have their goal probe result in goal_dist
  uint32_t done;         // 1: keep_going() == false (vertex budget reached); 2: sample stream exhausted
  uint32_t max_total;    // max_vertices + 1 (root is not counted by m_iteration_count)
  uint32_t samples_ready;  // samples uploaded so far
  uint32_t b_max;
  float batch_factor;
  uint32_t b_min;
  uint32_t pad;
  unsigned long long rounds, edges_speculated, fixup_cut;
};

struct ProblemDev {  // device pointers of one problem
  PlannerState* st;
  double* tree;
  uint32_t* parent;
  uint32_t* node_sample;
  const double* samples;
  uint32_t* nn_seq;
  uint8_t* accept_log;
  uint32_t* nn_idx;
  double* nn_dist;
  double* x_out;
  uint8_t* accept;
  uint32_t* round_n;  // profiling: vertex count at the start of each round (may be null)
  uint4* mirror;      // half-precision mirror of the tree rows (nn_mirror.h), or null
  uint32_t* dx_max_bits;  // its running maximum of |x - x_h|
};

// sel: {edge counter of even rounds, of odd rounds}.  Every problem adds its candidates + pending goal probes to the
// counter of this round's parity and block 0 clears the other one for the next round; the two steer kernels of the
// round compare the sum with their threshold (see launch_edges).
constexpr uint32_t kProfRounds = 8192;  // profiled rounds per planner (RKH_PROFILE_NN)
constexpr uint32_t kProbeGranule = 32;  // goal probes ride in whole steer waves when the wave fit is on (commit_kernel)
// One block for all problems.  Batch sizes: B = scale * batch_factor * sqrt(n) (results do not depend on them).  With
// fit_fill > 0 the scale of the round is chosen here, from the exact counts: the two-lanes steer kernel runs one wave of
// 32 edges per SIMD (`slots` waves at a time), so the steer time of a round is its number of waves divided by `slots`,
// rounded UP; the scale (0.75 .. 1.4) is bisected so that the round's waves -- candidates plus the pending goal probes
// of every problem -- fill fit_fill of a whole number of such passes.
__global__ __launch_bounds__(256) void round_begin_kernel(const ProblemDev* __restrict__ probs, uint32_t P,
                                                           uint32_t round_slot, uint32_t* __restrict__ sel, uint32_t parity,
                                                           float fit_fill, uint32_t slots, uint32_t* __restrict__ wave_base,
                                                           uint32_t* __restrict__ nn_base, uint32_t nn_queries,
                                                           uint32_t* __restrict__ edge_base, uint32_t epw,
                                                           uint32_t* __restrict__ step_cnt) {
  __shared__ unsigned int s_waves;
  // live-edge counters of the step-wise steer launches of this round (launch_propagate_pair_steps)
  if (step_cnt && threadIdx.x <= uint32_t(kMaxSteps) + 2u) step_cnt[threadIdx.x] = 0u;  // + the pool cursor (last word)
  // per-problem inputs of the batch formula, cached once (the bisection below evaluates it ten times per problem) and the
  // three count arrays, scanned in LDS; problems beyond the cache capacity fall back to global memory
  constexpr uint32_t kCache = 1024;
  __shared__ float s_bf[kCache], s_sq[kCache];
  __shared__ uint32_t s_bmin[kCache], s_bcap[kCache], s_probe_waves[kCache];
  __shared__ uint32_t s_scan[3][2 * kCache + 1];
  const bool cached = P <= kCache;
  const uint32_t tid = threadIdx.x;
  // With the wave fit on, a problem's candidates are a whole number of 32-edge steer waves: every (problem, candidates)
  // segment of the steer grid otherwise ends in a wave that is half empty on average (256 such waves per round of ~2700).
  const bool wave_round = fit_fill > 0.0f && epw > 1u;
  auto batch_of = [&](const PlannerState* st, float sc) -> uint32_t {
    if (st->done) return 0u;
    const float want = sc * st->batch_factor * sqrtf(float(st->n));
    uint32_t B = uint32_t(want);
    if (wave_round && B >= epw) B -= B % epw;  // whole steer waves: no half-empty last wave per problem
    if (B < st->b_min) B = st->b_min;
    if (B > st->b_max) B = st->b_max;
    const uint32_t avail = st->samples_ready - st->s0;
    if (B > avail) B = avail;
    return B;
  };
  if (cached) {
    for (uint32_t i = tid; i < P; i += blockDim.x) {
      const PlannerState* st = probs[i].st;
      const bool done = st->done != 0u;
      const uint32_t avail = st->samples_ready - st->s0;
      s_bf[i] = st->batch_factor;
      s_sq[i] = sqrtf(float(st->n));
      s_bmin[i] = done ? 0u : st->b_min;
      s_bcap[i] = done ? 0u : (st->b_max < avail ? st->b_max : avail);  // min(b_max, avail): the two upper clamps
      s_probe_waves[i] = (st->n_new + epw - 1u) / epw;
    }
    __syncthreads();
  }
  auto batch_cached = [&](uint32_t i, float sc) -> uint32_t {  // same value as batch_of(probs[i].st, sc)
    const float want = sc * s_bf[i] * s_sq[i];
    uint32_t B = uint32_t(want);
    if (wave_round && B >= epw) B -= B % epw;
    if (B < s_bmin[i]) B = s_bmin[i];
    if (B > s_bcap[i]) B = s_bcap[i];
    return B;
  };
  auto waves_at = [&](float sc) -> uint32_t {  // block-wide sum, same value in every thread
    uint32_t w = 0;
    for (uint32_t i = tid; i < P; i += blockDim.x) {
      if (cached) {
        w += (batch_cached(i, sc) + epw - 1u) / epw + s_probe_waves[i];
      } else {
        const PlannerState* st = probs[i].st;
        w += (batch_of(st, sc) + epw - 1u) / epw + (st->n_new + epw - 1u) / epw;
      }
    }
    __syncthreads();
    if (tid == 0) s_waves = 0u;
    __syncthreads();
    if (w) atomicAdd(&s_waves, w);
    __syncthreads();
    return s_waves;
  };
  float scale = 1.0f;
  if (fit_fill > 0.0f) {
    const float w1 = float(waves_at(1.0f));
    if (w1 > 0.75f * float(slots)) {
      const float passes = ceilf(w1 / float(slots) - 0.15f);
      const float target = passes * float(slots) * fit_fill;
      float lo = 0.75f, hi = 1.4f;
      for (int it = 0; it < 10; ++it) {
        const float mid = 0.5f * (lo + hi);
        if (float(waves_at(mid)) > target) hi = mid;
        else lo = mid;
      }
      scale = lo;
    }
  }
  if (tid == 0) sel[parity ^ 1u] = 0u;
  uint32_t edges = 0;
  for (uint32_t i = tid; i < P; i += blockDim.x) {
    const ProblemDev pr = probs[i];
    PlannerState* st = pr.st;
    if (pr.round_n) pr.round_n[round_slot] = st->done ? 0u : st->n;
    const uint32_t B = batch_of(st, scale);
    if (!st->done && B == 0) st->done = 2;  // sample stream exhausted: host must upload more
    st->B = B;
    st->F = B;
    if (pr.round_n) pr.round_n[kProfRounds + round_slot] = B;  // second half of the profile array: queries of the round
    if (B) {
      st->rounds += 1;
      st->edges_speculated += B;
    }
    edges += B + st->n_new;
    // counts of the three compact launch mappings (scanned below): steer waves (lane_kernel_edges_per_wave() edges each) per (candidates | probes) segment of
    // the two-lanes steer kernel, single edges per segment of the one-wave-per-edge kernel, query blocks of the NN sweep
    uint32_t* wb = cached ? s_scan[0] : wave_base;
    uint32_t* eb = cached ? s_scan[1] : edge_base;
    uint32_t* nb = cached ? s_scan[2] : nn_base;
    if (wave_base) {
      wb[2 * i + 1] = (B + epw - 1u) / epw;
      wb[2 * i + 2] = (st->n_new + epw - 1u) / epw;
    }
    if (nn_base) nb[i + 1] = (B + nn_queries - 1u) / nn_queries;
    if (edge_base) {
      eb[2 * i + 1] = B;
      eb[2 * i + 2] = st->n_new;
    }
  }
  if (edges) atomicAdd(&sel[parity], edges);
  __syncthreads();
  // exclusive prefixes in place (entry 0 = 0, last entry = total), one wave each; in LDS when cached, then copied out
  {
    uint32_t* wb = cached ? s_scan[0] : wave_base;
    uint32_t* eb = cached ? s_scan[1] : edge_base;
    uint32_t* nb = cached ? s_scan[2] : nn_base;
    auto scan = [](uint32_t* a, uint32_t n) {
      uint32_t acc = 0;
      a[0] = 0;
      for (uint32_t k = 1; k <= n; ++k) {
        acc += a[k];
        a[k] = acc;
      }
    };
    if (wave_base && tid == 0) scan(wb, 2 * P);
    if (edge_base && tid == 128) scan(eb, 2 * P);
    if (nn_base && tid == 64) scan(nb, P);
    if (cached) {
      __syncthreads();
      for (uint32_t k = tid; k <= 2 * P; k += blockDim.x) {
        if (wave_base) wave_base[k] = s_scan[0][k];
        if (edge_base) edge_base[k] = s_scan[1][k];
        if (nn_base && k <= P) nn_base[k] = s_scan[2][k];
      }
    }
  }
}

// One wave per candidate b: smallest squared distance from sample b to the end states of accepted
// candidates j < b.  min_j sqrt(s_j) == sqrt(min_j s_j) (sqrt is monotone), so one sqrt decides.
template <int DP>
__global__ __launch_bounds__(256) void fixup_kernel(const ProblemDev* __restrict__ probs, int D) {
  const ProblemDev pr = probs[blockIdx.y];
  PlannerState* st = pr.st;
  const uint32_t B = st->B;
  const uint32_t b = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (b >= B || b == 0) return;
  const double* q = pr.samples + (uint64_t(st->s0) + b) * D;
  double qv[DP];
#pragma unroll
  for (int d = 0; d < DP; ++d) qv[d] = d < D ? q[d] : 0.0;
  double smin = INFINITY;
  for (uint32_t j = lane; j < b; j += 64) {
    if (!pr.accept[j]) continue;
    const double* p = pr.x_out + uint64_t(j) * D;
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
  if (lane == 0 && sqrt(smin) < pr.nn_dist[b]) atomicMin(&st->F, b);
}

// One 256-thread block per problem: commit candidates [0, F) in order (prefix scan of the accept flags),
// honouring the vertex budget of keep_going() (motion_planner_base.hpp:355-374).
// probe_granule: goal probes ride in the next steer launch in whole groups of this many (32 = one steer wave of the
// two-lanes mapping: a (problem, probes) segment of the grid then has no half-empty last wave; what is left over waits
// for the vertices of the next round, the last ones for flush_probes; 1 = every pending probe rides along).
__global__ __launch_bounds__(256) void commit_kernel(const ProblemDev* __restrict__ probs, int D, int DP,
                                                      uint32_t probe_granule) {
  __shared__ uint32_t scan[256];
  __shared__ uint32_t carry;
  __shared__ uint32_t cut;  // number of candidates actually consumed
  const ProblemDev pr = probs[blockIdx.x];
  PlannerState* st = pr.st;
  const uint32_t F = st->F;
  const uint32_t n0 = st->n;
  const uint32_t s0 = st->s0;
  const uint32_t budget = st->max_total - n0;  // vertices that may still be added
  if (threadIdx.x == 0) {
    carry = 0;
    cut = F;
  }
  __syncthreads();
  for (uint32_t base = 0; base < F; base += 256) {
    const uint32_t b = base + threadIdx.x;
    const uint32_t a = (b < F && pr.accept[b]) ? 1u : 0u;
    scan[threadIdx.x] = a;
    __syncthreads();
    for (uint32_t off = 1; off < 256; off <<= 1) {  // Hillis-Steele inclusive scan
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
        pr.nn_seq[s0 + b] = pr.nn_idx[b];
        pr.accept_log[s0 + b] = uint8_t(a);
        if (a) {
          const uint32_t row = n0 + incl - 1;
          for (int d = 0; d < DP; ++d) pr.tree[uint64_t(row) * DP + d] = d < D ? pr.x_out[uint64_t(b) * D + d] : 0.0;
          if (pr.mirror) mirror_store_row(pr.mirror, row, pr.x_out + uint64_t(b) * D, D, pr.dx_max_bits);
          pr.parent[row] = pr.nn_idx[b];
          pr.node_sample[row] = s0 + b;
        }
      }
    }
    __syncthreads();
    if (threadIdx.x == 255) carry = incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const uint32_t added = carry < budget ? carry : budget;
    st->n = n0 + added;
    // the propagate launch of this round also ran the first n_new of the pending goal probes: [n_before, n_before+n_new)
    const uint32_t first_pending = st->n_before + st->n_new;
    st->probed_n = first_pending;
    st->n_before = first_pending;
    // the next launch takes the pending ones (this round's vertices included) in whole granules
    const uint32_t pending = n0 + added - first_pending;
    st->n_new = pending - pending % probe_granule;
    st->s0 = s0 + cut;
    st->fixup_cut += (st->B - F);
    if (st->n >= st->max_total) st->done = 1;
  }
}

// after a probe-only flush launch: nothing is pending any more
// The sample stream is generated ON THE DEVICE: hyperbox_topology::random_point (hyperbox_topology.hpp:97-103) draws D
// times uniform_01<mt19937&, double> = eng() * 2^-32 per sample (Boost.Random on a 32-bit engine: one draw per
// coordinate), so draw number k * D + d is coordinate d of sample k whatever happens in the planner.  One block per
// problem keeps that problem's mt19937 state (624 words + position, global_rng.hpp:44-54 seeded like std::mt19937) in LDS,
// regenerates it 624 words at a time in the three data-parallel stretches of the recurrence (words 0..226 read only old
// words, 227..453 the new 0..226, 454..622 the new 227..395, word 623 the new 396 and 0) and writes the samples behind the
// ones the rounds may read; only then it raises the problem's samples_ready (a round that still sees the old value just
// takes a smaller batch).  (Round 1 generated the stream on one host thread and uploaded it: ~15 M draws per 16 rounds,
// as long as the GPU needed for those rounds.)
struct SampleSeg {
  double* dst;           // first new sample of the problem
  uint64_t count;        // doubles to produce (a multiple of D)
  uint32_t* mt;          // the problem's generator: 624 state words + position
  uint32_t* ready_ptr;   // &PlannerState::samples_ready of the problem
  uint32_t ready_new;
  uint32_t pad;
};
constexpr int kMtN = 624, kMtM = 397;
__device__ __forceinline__ uint32_t mt_twist(uint32_t cur, uint32_t nxt, uint32_t far) {
  const uint32_t y = (cur & 0x80000000u) | (nxt & 0x7fffffffu);
  return far ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}
__global__ __launch_bounds__(256) void generate_samples_kernel(const SampleSeg* __restrict__ tab,
                                                                const double* __restrict__ bounds, int D) {
  __shared__ uint32_t mt[kMtN];
  const SampleSeg sg = tab[blockIdx.x];
  const uint32_t tid = threadIdx.x;
  for (uint32_t i = tid; i < uint32_t(kMtN); i += 256) mt[i] = sg.mt[i];
  uint32_t idx = sg.mt[kMtN];
  __syncthreads();
  uint64_t produced = 0;
  while (produced < sg.count) {  // uniform
    if (idx == uint32_t(kMtN)) {
      // words [a, b): new[i] = twist(old[i], old[i + 1], word (i + 397) mod 624 as it stands at this point)
      auto stretch = [&](uint32_t a, uint32_t b) {
        const uint32_t i = a + tid;
        uint32_t v = 0;
        if (i < b) v = mt_twist(mt[i], mt[i + 1], mt[i + kMtM < uint32_t(kMtN) ? i + kMtM : i + kMtM - kMtN]);
        __syncthreads();  // every thread has read its old neighbour before anybody overwrites it
        if (i < b) mt[i] = v;
        __syncthreads();
      };
      stretch(0, kMtN - kMtM);                  // 0 .. 226
      stretch(kMtN - kMtM, 2 * (kMtN - kMtM));  // 227 .. 453
      stretch(2 * (kMtN - kMtM), kMtN - 1);     // 454 .. 622
      if (tid == 0) mt[kMtN - 1] = mt_twist(mt[kMtN - 1], mt[0], mt[kMtM - 1]);
      __syncthreads();
      idx = 0;
    }
    const uint64_t left = sg.count - produced;
    const uint32_t chunk = (uint64_t(kMtN - idx) < left) ? uint32_t(kMtN - idx) : uint32_t(left);
    for (uint32_t t = tid; t < chunk; t += 256) {
      uint32_t y = mt[idx + t];
      y ^= (y >> 11);
      y ^= (y << 7) & 0x9d2c5680u;
      y ^= (y << 15) & 0xefc60000u;
      y ^= (y >> 18);
      const double u = double(y) * (1.0 / 4294967296.0);  // < 1 for every 32-bit y: uniform_01 never redraws
      const int d = int((produced + t) % uint64_t(D));
      sg.dst[produced + t] = bounds[d] + u * (bounds[D + d] - bounds[d]);
    }
    produced += chunk;
    idx += chunk;
    __syncthreads();
  }
  for (uint32_t i = tid; i < uint32_t(kMtN); i += 256) sg.mt[i] = mt[i];
  if (tid == 0) sg.mt[kMtN] = idx;
  __threadfence();  // the samples are visible device-wide before the count says so
  __syncthreads();
  if (tid == 0) *sg.ready_ptr = sg.ready_new;
}

// Goal-probe results since the last sync: block b copies problem b's new stretch into one buffer (one device->host copy
// for all problems instead of one per problem).
struct GoalSeg {
  const double* src;
  uint64_t dst_off, count;
};
__global__ __launch_bounds__(256) void gather_goal_dist_kernel(const GoalSeg* __restrict__ tab, double* __restrict__ out) {
  const GoalSeg g = tab[blockIdx.x];
  for (uint64_t i = threadIdx.x; i < g.count; i += 256) out[g.dst_off + i] = g.src[i];
}

// Phase boundary of a split steer launch (launch_edges): the edges of a (problem, candidates | probes) segment whose
// first `k_split` steps were all free go on, their ids are written to the segment's list (any order: edges are
// independent and their results are indexed by the edge).  One block per segment.  Runs only when the two-lanes
// mapping ran the first phase (the same gate); otherwise the lists are empty and the second launch finds no work.
__global__ __launch_bounds__(256) void phase_compact_kernel(const EdgeIO* __restrict__ tab_a, const EdgeIO* __restrict__ tab_b,
                                                             const EdgeIO* __restrict__ tab2_a,
                                                             const EdgeIO* __restrict__ tab2_b, uint32_t k_split,
                                                             uint32_t* __restrict__ cnt2, KernelGate gate) {
  __shared__ uint32_t s_cnt;
  const uint32_t prob = blockIdx.x, g = blockIdx.y;
  const EdgeIO io = g ? tab_b[prob] : tab_a[prob];
  uint32_t* ids = const_cast<uint32_t*>((g ? tab2_b[prob] : tab2_a[prob]).edge_ids);
  if (threadIdx.x == 0) s_cnt = 0u;
  __syncthreads();
  bool run = true;
  if (gate.count) {
    const uint32_t c = *gate.count;
    run = c >= gate.lo && c < gate.hi;
  }
  const uint32_t B = run ? (io.d_B ? *io.d_B : io.B) : 0u;
  const int lane = threadIdx.x & 63;
  for (uint32_t base = 0; base < B; base += 256) {
    const uint32_t e = base + threadIdx.x;
    const bool on = e < B && io.steps_free[e] == k_split;
    const unsigned long long m = __ballot(on);
    uint32_t off = 0;
    if (lane == 0 && m) off = atomicAdd(&s_cnt, uint32_t(__popcll(m)));
    off = __shfl(off, 0, 64);
    if (on) ids[off + uint32_t(__popcll(m & ((1ull << lane) - 1ull)))] = e;
  }
  __syncthreads();
  if (threadIdx.x == 0) cnt2[2 * prob + g] = s_cnt;
}

// exclusive prefix of the second phase's working waves per segment (the KernelGate::wave_base of its launch)
__global__ void phase_scan_kernel(const uint32_t* __restrict__ cnt2, uint32_t n_segments, uint32_t epw,
                                  uint32_t* __restrict__ wave_base2) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  uint32_t acc = 0;
  wave_base2[0] = 0;
  for (uint32_t k = 0; k < n_segments; ++k) {
    acc += (cnt2[k] + epw - 1u) / epw;
    wave_base2[k + 1] = acc;
  }
}

// before the flush launch: every pending probe rides, whatever the granule
__global__ void probes_take_all_kernel(const ProblemDev* __restrict__ probs) {
  if (threadIdx.x != 0) return;
  PlannerState* st = probs[blockIdx.x].st;
  st->n_new = st->n - st->n_before;
}

__global__ void probes_flushed_kernel(const ProblemDev* __restrict__ probs) {
  if (threadIdx.x != 0) return;
  PlannerState* st = probs[blockIdx.x].st;
  st->probed_n = st->n_before + st->n_new;
  st->n_before = st->n;
  st->n_new = 0;
}

}  // namespace rkh

using namespace rkh;

namespace {
struct Problem {  // host view of one planning problem
  rkh_rrt_params prm;
  uint32_t* d_mt = nullptr;  // the problem's mt19937 on the device: 624 state words + position
  // device buffers
  double* d_tree = nullptr;
  uint32_t* d_parent = nullptr;
  uint32_t* d_node_sample = nullptr;
  double* d_goal_dist = nullptr;
  double* d_samples = nullptr;
  uint32_t* d_nn_seq = nullptr;
  uint8_t* d_accept_log = nullptr;
  uint32_t* d_nn_idx = nullptr;
  double* d_nn_dist = nullptr;
  double* d_x_out = nullptr;
  uint32_t* d_steps = nullptr;
  uint8_t* d_accept = nullptr;
  double* d_probe_x = nullptr;
  uint32_t* d_probe_steps = nullptr;
  void* d_mirror = nullptr;        // half-precision mirror of d_tree (nn_mirror.h)
  void* d_cand = nullptr;          // per-query scratch of the mirror sweep (nn1_mirror_carve), then one word: dx_max_bits
  uint32_t* d_ids_c = nullptr;  // survivors of the first steer phase: candidates, goal probes (launch_edges)
  uint32_t* d_ids_p = nullptr;
  double* d_goal = nullptr;
  double* d_part_dist = nullptr;
  uint32_t* d_part_idx = nullptr;
  uint32_t* d_round_n = nullptr;
  uint64_t capacity = 0, sample_cap = 0, samples_ready = 0;
  PlannerState h_state;
  // solution bookkeeping (register_basic_solution_path_impl, solution_path_factories.hpp:58-110)
  uint64_t goal_checked = 0;
  uint64_t num_solutions = 0;
  double best_cost = INFINITY;
  uint32_t best_vertex = 0xFFFFFFFFu;  // vertex whose goal probe gave the best registered solution
  bool truncated = false;
  uint64_t final_n = 0, final_iterations = 0;
};
}  // namespace

struct rkh_planner {
  rkh_scene* scene = nullptr;
  hipStream_t stream = nullptr;
  double* h_gd = nullptr;     // pinned read-back buffer of the goal-probe results (rkh_planner_sync)
  uint64_t h_gd_cap = 0;
  hipStream_t copy_stream = nullptr;  // sample-stream uploads (beside the rounds enqueued on `stream`)
  bool quasi_static = false;  // false: steerable dynamic space (propagate kernel); true: manip_quasi_static_env (edge_check)
  double lower[RKH_MAX_STATE], upper[RKH_MAX_STATE];  // hyperbox the samples are drawn from
  DynDev dyn;
  QsDev qs;
  int n_dof = 0, D = 0, DP = 0;
  uint32_t P = 0;
  std::vector<Problem> prob;
  uint32_t b_max = 1024;
  int lanes_per_edge = 64;  // 64: one wavefront per candidate edge; 16: four candidates per wave; 1 / 2: two lanes per edge
  int lane_variant = 2;     // throughput mapping of the automatic mode: 2 = propagate_pair.hip, 1 = propagate_lane.hip
  double* d_lane_ws = nullptr;  // workspace of the two-lanes-per-edge kernel
  double coord_bound = 0.0;     // max |coordinate| of vertices and samples (hyperbox bounds), 0 = unknown
  uint32_t* d_sel = nullptr;    // [2] edges of the current round (by round parity), see round_begin_kernel
  uint32_t* d_nn_base = nullptr;    // [P + 1] prefix of the NN sweep's query blocks per problem (matrix-core kernel)
  uint32_t* d_wave_base = nullptr;  // [2 P + 1] prefix of the working waves per (problem, candidates | probes) segment
  uint32_t round_parity = 0;
  // host-side upper bounds that size the launches of a round (the exact counts live on the device): n_ub[i] >= vertex
  // count of problem i (exact after every sync, + the round's batch bound per enqueued round)
  std::vector<uint64_t> n_ub;
  uint32_t prev_batch_ub = 0;  // batch bound of the previous round = bound on this round's goal probes
  int wave_fit = 1;          // per-round batch scale chosen on the device (round_begin_kernel); RKH_WAVE_FIT=0: off
  double wave_fill = 0.99;   // target fill of the last pass of steer waves (RKH_WAVE_FILL)
  uint32_t wave_slots = 1024;    // SIMDs of the device = concurrent waves of the two-lanes steer kernel
  uint32_t duo_threshold = 512;   // rounds below this many edges: two waves per edge (RKH_DUO_THRESHOLD; 0 = never)
  uint32_t lane_threshold = 1024;  // rounds with at least this many edges go to the two-lanes-per-edge kernel (one wave-per-edge pass fills the 1024 SIMDs; measured optimum at 4, 16 and 32 problems, tests/diag_lane_threshold.sh)
  uint32_t part_blocks = 0;
  uint64_t max_capacity = 0;
  // device tables (P entries each)
  PlannerState* d_states = nullptr;
  ProblemDev* d_probs = nullptr;
  NnArgs* d_nn_args = nullptr;
  EdgeIO* d_io_steer = nullptr;
  EdgeIO* d_io_probe = nullptr;
  // second phase of a split steer launch (launch_edges): the same records with the survivors' lists
  EdgeIO* d_io_steer2 = nullptr;
  EdgeIO* d_io_probe2 = nullptr;
  uint32_t* d_cnt2 = nullptr;        // [2 P] survivors per (problem, candidates | probes) segment
  uint32_t* d_wave_base2 = nullptr;  // [2 P + 1] prefix of their waves
  uint32_t steer_split = 5;          // steps of the first phase (RKH_STEER_SPLIT; 0 = one launch for the whole edge)
  // Step-wise steer launches (propagate_pair_step_kernel, the default; RKH_STEER_STEPWISE=0: the two-phase launch above):
  // one launch per RK4 step over the live edges of all problems, survivors handed on through two ping-pong lists.
  bool nn_mirror = false;  // the NN search of a round runs over the trees' half-precision mirrors (nn_mirror.hip)
  double x_norm_bound = 0.0;  // >= |x| of every vertex (hyperbox corners, start states)
  int steer_stepwise = 1;
  // RKH_STEER_POOL=1: the first steer launch of a round in its POOL form (lane pairs refill from the round's pool).  Same
  // results (test_stepwise_and_two_phase_...), measured SLOWER than one LIST launch per step (512 x 100 000: 6.45 against
  // 7.57 M expansions/s): the lanes stay full while the pool lasts, but an edge started when it runs dry can still need
  // 20 steps of >= 115 us each, and that tail is no shorter than the one every step-wise round has anyway.
  int steer_pool = 0;
  uint32_t pool_waves = 0;  // its grid: the resident steer waves of the device (RKH_STEER_POOL_WAVES: tests)
  uint2* d_step_list[2] = {nullptr, nullptr};  // (segment, edge) of the edges alive after step k (k odd / even)
  uint32_t* d_step_cnt = nullptr;              // [kMaxSteps + 2] entries of the list launch k reads, + the pool cursor (cleared by round_begin_kernel)
  unsigned long long* d_steps_exec = nullptr;  // edge-steps integrated by the steer kernels (diagnostics: rkh_planner_steer_steps)
  uint32_t step_blocks_cap = 0;                // grid bound of a step launch (its blocks stride over the chunks beyond it)
  // rounds below this many edges keep the single whole-edge launch of the two-lanes mapping (RKH_STEER_SPLIT_MIN_EDGES;
  // default: what leaves every SIMD at most one 32-edge wave -- such a round gains nothing from shedding waves)
  uint32_t split_min_edges = 0;
  uint64_t max_n_ub = 1;  // largest vertex-count bound over the problems (sizes the mirror sweep's row slices)
  uint64_t sum_batch_ub = 0, prev_sum_batch_ub = 0;  // host-side bounds on the candidates of this / the previous round, all problems
  // segment tables of the sample generator: [0] what the enqueued rounds need, [1] the next call's share, generated
  // while the GPU works on the rounds just enqueued
  double* d_bounds = nullptr;  // lower[D], upper[D] of the sampled hyperbox
  struct Staging {
    SampleSeg* h_tab = nullptr; // pinned segment table [P]
    SampleSeg* d_tab = nullptr;
    hipEvent_t done = nullptr;
    bool pending = false;
  } staging[2];
  GoalSeg* h_gd_tab = nullptr;  // pinned [P]
  GoalSeg* d_gd_tab = nullptr;
  double* d_gd = nullptr;       // gathered goal-probe results (rkh_planner_sync)
  uint64_t d_gd_cap = 0;
  // optional HIP-event timing of the NN sweep kernel (RKH_PROFILE_NN=1)
  bool profile_nn = false;
  std::vector<hipEvent_t> ev;  // pairs
  std::vector<hipEvent_t> ev_steer;  // pairs around the steer launches of the same rounds
  uint32_t prof_rounds = 0;
  static constexpr uint32_t kProfMax = kProfRounds;
};

namespace {

// Extend every problem's device-resident sample stream by `ahead` samples beyond its (last known) cursor
// (generate_samples_kernel).  The launch goes to the planner's second stream, beside the rounds already enqueued on the
// first one; the table's previous use is awaited through its event, so no stream synchronisation happens here.
rkh_status upload_samples_all(rkh_planner* p, uint64_t ahead, int which) {
  rkh_planner::Staging& sg = p->staging[which];
  const int D = p->D;
  std::vector<uint64_t> upto(p->P, 0);
  bool any = false;
  for (uint32_t i = 0; i < p->P; ++i) {
    Problem& q = p->prob[i];
    if (q.truncated || q.h_state.done == 1) continue;
    uint64_t want = uint64_t(q.h_state.s0) + ahead;
    if (want > q.sample_cap) want = q.sample_cap;
    if (want <= q.samples_ready) continue;
    upto[i] = want;
    any = true;
  }
  if (!any) return RKH_OK;
  if (sg.pending) {
    RKH_HIP(hipEventSynchronize(sg.done));
    sg.pending = false;
  }
  if (!sg.done) RKH_HIP(hipEventCreateWithFlags(&sg.done, hipEventDisableTiming));
  if (!sg.h_tab) {
    RKH_HIP(hipHostMalloc(reinterpret_cast<void**>(&sg.h_tab), p->P * sizeof(SampleSeg), hipHostMallocDefault));
    RKH_HIP(hipMalloc(&sg.d_tab, p->P * sizeof(SampleSeg)));
  }
  uint32_t n_seg = 0;
  for (uint32_t i = 0; i < p->P; ++i) {
    if (!upto[i]) continue;
    Problem& q = p->prob[i];
    SampleSeg& seg = sg.h_tab[n_seg++];
    seg.dst = q.d_samples + q.samples_ready * D;
    seg.count = (upto[i] - q.samples_ready) * D;
    seg.mt = q.d_mt;
    seg.ready_ptr = &p->d_states[i].samples_ready;
    seg.ready_new = uint32_t(upto[i]);
    seg.pad = 0;
    q.samples_ready = upto[i];
  }
  RKH_HIP(hipMemcpyAsync(sg.d_tab, sg.h_tab, n_seg * sizeof(SampleSeg), hipMemcpyHostToDevice, p->copy_stream));
  hipLaunchKernelGGL(generate_samples_kernel, dim3(n_seg), dim3(256), 0, p->copy_stream, sg.d_tab, p->d_bounds, D);
  RKH_HIP(hipGetLastError());
  // The generator runs on its own stream, beside the rounds already enqueued on the planner stream: it writes beyond every
  // problem's samples_ready (no round reads there) and then raises samples_ready.  Work enqueued on the planner stream
  // from here on waits for it.
  RKH_HIP(hipEventRecord(sg.done, p->copy_stream));
  RKH_HIP(hipStreamWaitEvent(p->stream, sg.done, 0));
  sg.pending = true;
  return RKH_OK;
}

template <int DP>
void launch_fixup(rkh_planner* p, uint32_t batch_ub) {
  hipLaunchKernelGGL((fixup_kernel<DP>), dim3((batch_ub + 3) / 4, p->P), dim3(256), 0, p->stream, p->d_probs, p->D);
}

// upper bound of the batch size round_begin_kernel will choose for a problem with at most n_ub vertices (same float
// formula, monotone in n)
uint32_t batch_upper_bound(const PlannerState& st, uint64_t n_ub, float batch_scale = 1.0f) {
  const float want = batch_scale * st.batch_factor * sqrtf(float(n_ub));
  uint32_t B = uint32_t(want);
  if (B < st.b_min) B = st.b_min;
  if (B > st.b_max) B = st.b_max;
  return B;
}

// steer / probe edges of all problems: RK4 propagation (dynamic space) or the min_interval walk (quasi-static space)
rkh_status launch_edges(rkh_planner* p, uint32_t grid_a, uint32_t grid_b, const EdgeIO* tab_a, const EdgeIO* tab_b,
                        bool compact = false) {
  if (p->quasi_static)
    return launch_edge_check(p->stream, p->n_dof, p->scene->host.n_env, p->scene->d_scene, p->scene->d_pairs,
                             p->scene->n_pairs_verdict, p->qs, EdgeIO(), grid_a, nullptr, grid_b, tab_a, tab_b, p->P);
  if (p->lanes_per_edge != 0)
    return launch_propagate(p->stream, p->n_dof, p->scene->host.n_env, p->scene->d_scene, p->scene->d_pairs,
                            p->scene->n_pairs_verdict, p->dyn, EdgeIO(), grid_a, nullptr, grid_b, p->lanes_per_edge, tab_a, tab_b,
                            p->P, p->d_lane_ws);
  // automatic: both mappings are launched; on the device each compares the round's edge count with the threshold and
  // the one that is not chosen exits at once.  Small rounds -> one wave per edge (latency), large -> 32 edges per wave.
  KernelGate gate_wave{p->d_sel + p->round_parity, 0u, p->lane_threshold};
  KernelGate gate_lane{p->d_sel + p->round_parity, p->lane_threshold, 0xFFFFFFFFu};
  gate_wave.steps_exec = gate_lane.steps_exec = p->d_steps_exec;
  if (compact && p->d_wave_base) {  // a regular round: (candidates, probes) segments as round_begin_kernel counted them
    gate_lane.wave_base = p->d_wave_base;
    gate_lane.n_segments = 2 * p->P;
    gate_wave.wave_base = p->d_wave_base + (2 * p->P + 1);
    gate_wave.n_segments = 2 * p->P;
  }
  rkh_status st = RKH_OK;
  if (p->duo_threshold > 0 && compact && p->d_wave_base) {
    // the smallest rounds (at most half the chip's SIMDs at one wave per edge: a single problem, a few young trees):
    // two waves per edge (state_derivative_duo), the f-eval's critical path instead of its instruction count
    KernelGate gate_duo = gate_wave;
    gate_duo.hi = std::min(p->duo_threshold, p->lane_threshold);
    gate_wave.lo = gate_duo.hi;
    st = launch_propagate(p->stream, p->n_dof, p->scene->host.n_env, p->scene->d_scene, p->scene->d_pairs,
                          p->scene->n_pairs_verdict, p->dyn, EdgeIO(), grid_a, nullptr, grid_b, 128, tab_a, tab_b, p->P, nullptr,
                          gate_duo);
    if (st != RKH_OK) return st;
  }
  if (gate_wave.lo < gate_wave.hi)
    st = launch_propagate(p->stream, p->n_dof, p->scene->host.n_env, p->scene->d_scene, p->scene->d_pairs,
                          p->scene->n_pairs_verdict, p->dyn, EdgeIO(), grid_a, nullptr, grid_b, 64, tab_a, tab_b, p->P, nullptr,
                          gate_wave);
  if (st != RKH_OK) return st;
  // The two-lanes mapping in two phases when the round is a regular one: half of the edges of a round end within a few
  // steps (tests/diag_edge_lifetimes.py) and leave their lanes idle for the rest of their wave, so the first
  // steer_split steps run for every edge, the survivors are compacted per segment and only they run the remaining
  // steps -- in fewer waves.  Same arithmetic per edge, same results.
  const bool stepwise = p->steer_stepwise && p->d_step_cnt;
  const bool split = compact && p->d_wave_base && p->d_io_steer2 && p->lane_variant == 2 && p->dyn.n_steps > 1 &&
                     (stepwise || (p->steer_split > 0 && int(p->steer_split) < p->dyn.n_steps)) &&
                     tab_a == p->d_io_steer && tab_b == p->d_io_probe;
  if (!split)
    return launch_propagate(p->stream, p->n_dof, p->scene->host.n_env, p->scene->d_scene, p->scene->d_pairs,
                            p->scene->n_pairs_verdict, p->dyn, EdgeIO(), grid_a, nullptr, grid_b, p->lane_variant, tab_a, tab_b,
                            p->P, p->d_lane_ws, gate_lane);
  // ... when the round is large enough; below that the extra launches and tails cost more than the idle lanes (64
  // problems x 100 000 with the two-phase launch: 2.94 against 3.04 M expansions/s): such rounds take one launch
  const uint32_t split_edges = p->split_min_edges;
  // host-side bound on the edges of this round (candidates + pending probes of all problems)
  const uint64_t edges_ub = std::min<uint64_t>(p->sum_batch_ub + p->prev_sum_batch_ub + uint64_t(p->P) * kProbeGranule,
                                               uint64_t(grid_a + grid_b) * p->P);
  if (split_edges > gate_lane.lo) {
    KernelGate whole = gate_lane;
    whole.hi = split_edges;
    whole.steps_exec = p->d_steps_exec;
    if (edges_ub >= whole.lo) {  // (a round that cannot reach the gate needs no launch at all)
      st = launch_propagate(p->stream, p->n_dof, p->scene->host.n_env, p->scene->d_scene, p->scene->d_pairs,
                            p->scene->n_pairs_verdict, p->dyn, EdgeIO(), grid_a, nullptr, grid_b, p->lane_variant, tab_a, tab_b,
                            p->P, p->d_lane_ws, whole);
      if (st != RKH_OK) return st;
    }
    gate_lane.lo = split_edges;
  }
  if (edges_ub < gate_lane.lo) return RKH_OK;
  if (stepwise) {
    const uint32_t epw = pair_kernel_edges_per_wave();
    const uint32_t blocks = uint32_t(std::min<uint64_t>((edges_ub + epw - 1) / epw, p->step_blocks_cap));
    // (the pool cursor is the last word of the step counters: round_begin_kernel clears it with them)
    return launch_propagate_pair_steps(p->stream, p->n_dof, p->scene->d_scene, p->dyn, tab_a, tab_b, p->P,
                                       p->d_wave_base + (2 * p->P + 1), p->d_step_list[0], p->d_step_list[1],
                                       p->d_step_cnt, p->d_lane_ws, blocks, gate_lane, p->d_steps_exec,
                                       p->steer_pool ? p->pool_waves : 0u, p->d_step_cnt + kMaxSteps + 2);
  }
  KernelGate g1 = gate_lane;
  g1.step1 = p->steer_split;
  st = launch_propagate(p->stream, p->n_dof, p->scene->host.n_env, p->scene->d_scene, p->scene->d_pairs,
                        p->scene->n_pairs_verdict, p->dyn, EdgeIO(), grid_a, nullptr, grid_b, p->lane_variant, tab_a, tab_b, p->P,
                        p->d_lane_ws, g1);
  if (st != RKH_OK) return st;
  hipLaunchKernelGGL(phase_compact_kernel, dim3(p->P, 2), dim3(256), 0, p->stream, p->d_io_steer, p->d_io_probe,
                     p->d_io_steer2, p->d_io_probe2, p->steer_split, p->d_cnt2, gate_lane);
  hipLaunchKernelGGL(phase_scan_kernel, dim3(1), dim3(64), 0, p->stream, p->d_cnt2, 2 * p->P,
                     pair_kernel_edges_per_wave(), p->d_wave_base2);
  RKH_HIP(hipGetLastError());
  KernelGate g2 = gate_lane;
  g2.wave_base = p->d_wave_base2;
  g2.n_segments = 2 * p->P;
  g2.step0 = p->steer_split;
  return launch_propagate(p->stream, p->n_dof, p->scene->host.n_env, p->scene->d_scene, p->scene->d_pairs,
                          p->scene->n_pairs_verdict, p->dyn, EdgeIO(), grid_a, nullptr, grid_b, p->lane_variant, p->d_io_steer2,
                          p->d_io_probe2, p->P, p->d_lane_ws, g2);
}

// goal probes still pending after the last enqueued round
rkh_status flush_probes(rkh_planner* p) {
  hipLaunchKernelGGL(probes_take_all_kernel, dim3(p->P), dim3(64), 0, p->stream, p->d_probs);
  rkh_status st = launch_edges(p, p->b_max + kProbeGranule, 0, p->d_io_probe, nullptr);
  if (st != RKH_OK) return st;
  hipLaunchKernelGGL(probes_flushed_kernel, dim3(p->P), dim3(64), 0, p->stream, p->d_probs);
  RKH_HIP(hipGetLastError());
  return RKH_OK;
}

rkh_status enqueue_round(rkh_planner* p) {
  hipStream_t s = p->stream;
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
    if (p->ev_steer.size() < 2 * size_t(slot + 1)) {
      hipEvent_t a, b;
      RKH_HIP(hipEventCreate(&a));
      RKH_HIP(hipEventCreate(&b));
      p->ev_steer.push_back(a);
      p->ev_steer.push_back(b);
    }
  }
  // the round's batch scale is chosen on the device (round_begin_kernel); the launches are sized for its upper end
  const bool fit = p->wave_fit && !p->quasi_static && p->lanes_per_edge == 0;
  const float scale = fit ? 1.4f : 1.0f;
  // launch sizes of this round from the host-side bounds
  uint32_t batch_ub = 1;
  p->prev_sum_batch_ub = p->sum_batch_ub ? p->sum_batch_ub : uint64_t(p->b_max) * p->P;
  p->sum_batch_ub = 0;
  p->max_n_ub = 1;
  for (uint32_t i = 0; i < p->P; ++i) {
    const PlannerState& hs = p->prob[i].h_state;
    const uint32_t b = batch_upper_bound(hs, p->n_ub[i], scale);
    batch_ub = std::max(batch_ub, b);
    p->sum_batch_ub += b;
    p->max_n_ub = std::max(p->max_n_ub, p->n_ub[i]);
    p->n_ub[i] = std::min<uint64_t>(p->n_ub[i] + b, uint64_t(hs.max_total));
  }
  const uint32_t probe_ub = p->prev_batch_ub ? p->prev_batch_ub : p->b_max;
  p->prev_batch_ub = batch_ub + kProbeGranule;  // next round's probes: this round's vertices + what was left over
  p->round_parity ^= 1u;
  hipLaunchKernelGGL(round_begin_kernel, dim3(1), dim3(256), 0, s, p->d_probs, p->P, slot, p->d_sel, p->round_parity,
                     fit ? float(p->wave_fill) : 0.0f, p->wave_slots, p->d_wave_base, p->d_nn_base,
                     p->nn_mirror ? nn1_mirror_queries() : nn1_mfma_queries(),
                     p->d_wave_base ? p->d_wave_base + (2 * p->P + 1) : nullptr, lane_kernel_edges_per_wave(),
                     p->d_step_cnt);
  // 1. NN sweep of every problem's samples over its snapshot
  rkh_status st = p->nn_mirror
                      ? launch_nn1_mirror(s, p->D, p->d_nn_args, p->P, p->max_n_ub, batch_ub, p->x_norm_bound, p->d_nn_base,
                                          ev0, ev1)
                      : launch_nn1(s, p->D, NnArgs(), p->d_nn_args, p->P, p->max_capacity, batch_ub, p->part_blocks, ev0,
                                   ev1, p->coord_bound, p->d_nn_base, true);
  if (st != RKH_OK) return st;
  // 2. speculative steer of all candidates + the goal probes of the vertices the previous round committed
  if (ev0) (void)hipEventRecord(p->ev_steer[2 * slot], s);
  st = launch_edges(p, batch_ub, probe_ub, p->d_io_steer, p->d_io_probe, true);
  if (st != RKH_OK) return st;
  if (ev0) (void)hipEventRecord(p->ev_steer[2 * slot + 1], s);
  // 3. fix-up against the vertices this round itself would add
  switch (p->DP) {
    case 2: launch_fixup<2>(p, batch_ub); break;
    case 4: launch_fixup<4>(p, batch_ub); break;
    case 6: launch_fixup<6>(p, batch_ub); break;
    case 8: launch_fixup<8>(p, batch_ub); break;
    case 12: launch_fixup<12>(p, batch_ub); break;
    case 16: launch_fixup<16>(p, batch_ub); break;
    case 24: launch_fixup<24>(p, batch_ub); break;
    case 32: launch_fixup<32>(p, batch_ub); break;
    default: set_error("planner: unsupported state dimension"); return RKH_ERR_UNSUPPORTED;
  }
  // 4. commit the valid prefix
  hipLaunchKernelGGL(commit_kernel, dim3(p->P), dim3(256), 0, s, p->d_probs, p->D, p->DP, fit ? kProbeGranule : 1u);
  RKH_HIP(hipGetLastError());
  return RKH_OK;
}

void free_problem(Problem& q) {
  void* bufs[] = {q.d_tree, q.d_parent, q.d_node_sample, q.d_goal_dist, q.d_samples, q.d_nn_seq, q.d_accept_log,
                  q.d_nn_idx, q.d_nn_dist, q.d_x_out, q.d_steps, q.d_accept, q.d_probe_x, q.d_probe_steps, q.d_goal,
                  q.d_part_dist, q.d_part_idx, q.d_round_n, q.d_mt, q.d_ids_c, q.d_ids_p, q.d_mirror, q.d_cand};
  for (void* b : bufs) (void)hipFree(b);
}

// generate_rrt has no iteration cap (rr_tree.hpp:192-196: keep_going() looks at the vertex count only), so the device-
// resident sample stream and its per-iteration logs must not have one either: when a problem's cursor comes near the
// end of its buffers they are re-allocated at twice the size.  Called with both streams idle (rkh_planner_sync).
rkh_status grow_sample_buffers(rkh_planner* p, uint32_t i, uint64_t new_cap) {
  Problem& q = p->prob[i];
  const int D = p->D;
  double* ns = nullptr;
  uint32_t* nq = nullptr;
  uint8_t* na = nullptr;
  RKH_HIP(hipMalloc(&ns, new_cap * D * sizeof(double)));
  RKH_HIP(hipMalloc(&nq, new_cap * sizeof(uint32_t)));
  RKH_HIP(hipMalloc(&na, new_cap));
  RKH_HIP(hipMemcpy(ns, q.d_samples, q.samples_ready * D * sizeof(double), hipMemcpyDeviceToDevice));
  RKH_HIP(hipMemcpy(nq, q.d_nn_seq, q.sample_cap * sizeof(uint32_t), hipMemcpyDeviceToDevice));
  RKH_HIP(hipMemcpy(na, q.d_accept_log, q.sample_cap, hipMemcpyDeviceToDevice));
  (void)hipFree(q.d_samples);
  (void)hipFree(q.d_nn_seq);
  (void)hipFree(q.d_accept_log);
  q.d_samples = ns;
  q.d_nn_seq = nq;
  q.d_accept_log = na;
  q.sample_cap = new_cap;
  // the device tables that point into these buffers
  const double* cs = ns;
  RKH_HIP(hipMemcpy(&p->d_probs[i].samples, &cs, sizeof(cs), hipMemcpyHostToDevice));
  RKH_HIP(hipMemcpy(&p->d_probs[i].nn_seq, &nq, sizeof(nq), hipMemcpyHostToDevice));
  RKH_HIP(hipMemcpy(&p->d_probs[i].accept_log, &na, sizeof(na), hipMemcpyHostToDevice));
  RKH_HIP(hipMemcpy(&p->d_nn_args[i].q, &cs, sizeof(cs), hipMemcpyHostToDevice));
  RKH_HIP(hipMemcpy(&p->d_io_steer[i].tgt, &cs, sizeof(cs), hipMemcpyHostToDevice));
  if (p->d_io_steer2) RKH_HIP(hipMemcpy(&p->d_io_steer2[i].tgt, &cs, sizeof(cs), hipMemcpyHostToDevice));
  return RKH_OK;
}

rkh_status read_states(rkh_planner* p) {
  std::vector<PlannerState> hs(p->P);
  RKH_HIP(hipMemcpyAsync(hs.data(), p->d_states, p->P * sizeof(PlannerState), hipMemcpyDeviceToHost, p->stream));
  RKH_HIP(hipStreamSynchronize(p->stream));
  for (uint32_t i = 0; i < p->P; ++i) {
    p->prob[i].h_state = hs[i];
    p->n_ub[i] = hs[i].n;  // the stream is idle: the bound is exact again
  }
  uint32_t pending = 1;
  for (uint32_t i = 0; i < p->P; ++i) pending = std::max(pending, hs[i].n_new);
  p->prev_batch_ub = pending;
  return RKH_OK;
}

}  // namespace

extern "C" {

static rkh_status planner_create_common(rkh_scene* scene, const rkh_dyn_space* space, const rkh_qs_space* qspace,
                                        const rkh_rrt_params* prms, uint32_t n_problems, rkh_planner** out) {
  if (!scene || (!space && !qspace) || !prms || !out || n_problems < 1) return RKH_ERR_BAD_ARG;
  const int space_dof = space ? space->n_dof : qspace->n_dof;
  if (space_dof != scene->host.n_dof) {
    set_error("rkh_planner_create: the space's n_dof does not match the scene");
    return RKH_ERR_BAD_ARG;
  }
  for (uint32_t i = 0; i < n_problems; ++i)
    if (prms[i].max_vertices < 1) {
      set_error("rkh_planner_create: max_vertices < 1");
      return RKH_ERR_BAD_ARG;
    }
  rkh_planner* p = new rkh_planner();
  p->scene = scene;
  p->n_dof = space_dof;
  p->P = n_problems;
  rkh_status st = RKH_OK;
  if (space) {
    p->D = 2 * space->n_dof;
    for (int d = 0; d < p->D; ++d) {
      p->lower[d] = space->lower[d];
      p->upper[d] = space->upper[d];
    }
    st = build_dyn_dev(*space, 1.0, &p->dyn);
    if (st != RKH_OK) { delete p; return st; }
  } else {
    if (!(qspace->min_interval > 0.0) || qspace->n_dof > kMaxDof) {
      delete p;
      set_error("rkh_qs_space: min_interval must be positive");
      return RKH_ERR_BAD_ARG;
    }
    p->quasi_static = true;
    p->D = qspace->n_dof;
    std::memset(&p->qs, 0, sizeof(p->qs));
    p->qs.min_interval = qspace->min_interval;
    p->qs.fraction = 1.0;
    qs_set_speed(p->qs, qspace->speed_limits, qspace->n_dof);
    for (int d = 0; d < p->D; ++d) {
      p->lower[d] = p->qs.lower[d] = qspace->lower[d];
      p->upper[d] = p->qs.upper[d] = qspace->upper[d];
    }
  }
  p->DP = nn_padded_dims(p->D);
  // vertices and samples lie inside the hyperbox (is_free / random_point): bound for the NN sweep's float pre-filter
  if (!getenv("RKH_NN_F64_ONLY"))
    for (int d = 0; d < p->D; ++d) {
      p->coord_bound = std::max(p->coord_bound, std::max(std::fabs(p->lower[d]), std::fabs(p->upper[d])));
      for (uint32_t i = 0; i < n_problems; ++i)  // the root is a vertex too
        p->coord_bound = std::max(p->coord_bound, std::fabs(prms[i].start[d]));
    }
  RKH_HIP(hipSetDevice(scene->ctx->device));
  RKH_HIP(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking));
  RKH_HIP(hipStreamCreateWithFlags(&p->copy_stream, hipStreamNonBlocking));
  {
    std::vector<double> bounds(2 * p->D);
    for (int d = 0; d < p->D; ++d) {
      bounds[d] = p->lower[d];
      bounds[p->D + d] = p->upper[d];
    }
    RKH_HIP(hipMalloc(&p->d_bounds, bounds.size() * sizeof(double)));
    RKH_HIP(hipMemcpy(p->d_bounds, bounds.data(), bounds.size() * sizeof(double), hipMemcpyHostToDevice));
  }
  if (const char* e = getenv("RKH_LANE_VARIANT")) p->lane_variant = (atoi(e) == 1) ? 1 : 2;
  if (const char* e = getenv("RKH_WAVE_FIT")) p->wave_fit = atoi(e);
  if (const char* e = getenv("RKH_STEER_SPLIT")) p->steer_split = uint32_t(std::max(0, atoi(e)));
  if (const char* e = getenv("RKH_WAVE_FILL")) p->wave_fill = atof(e);
  if (const char* e = getenv("RKH_STEER_STEPWISE")) p->steer_stepwise = atoi(e);
  if (const char* e = getenv("RKH_STEER_POOL")) p->steer_pool = atoi(e);
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, scene->ctx->device) == hipSuccess && prop.multiProcessorCount > 0)
      p->wave_slots = uint32_t(prop.multiProcessorCount) * (p->lane_variant == 2 ? pair_kernel_waves_per_cu(scene->host.n_dof)
                                                                                 : lane_kernel_waves_per_cu(scene->host.n_dof));
    if (getenv("RKH_VERBOSE")) fprintf(stderr, "rkh planner: %d CUs, %u resident steer waves\n", prop.multiProcessorCount, p->wave_slots);
  }
  // whole-edge launches below: step-wise, one 32-edge wave per SIMD; two-phase, one pass of steer waves (its optimum)
  p->split_min_edges = (p->steer_stepwise ? p->wave_slots / 2 : p->wave_slots) * pair_kernel_edges_per_wave();
  if (const char* e = getenv("RKH_STEER_SPLIT_MIN_EDGES")) p->split_min_edges = uint32_t(std::max(0, atoi(e)));
  p->step_blocks_cap = 2 * p->wave_slots;
  p->pool_waves = p->wave_slots;
  if (const char* e = getenv("RKH_STEER_POOL_WAVES")) p->pool_waves = uint32_t(std::max(1, atoi(e)));
  // Many problems per planner: a round's candidates per problem stay within ONE query block of the mirror sweep (a
  // second block re-reads the whole tree for a handful of queries; 512 problems x 100 000: 7.45 -> 7.62 M expansions/s).
  // The batch rule only reaches the cap late in a run (1.25 sqrt(n) = 384 at n = 94 k) or through the wave fit's scale.
  if (nn1_mirror_applies(p->D, p->coord_bound) && n_problems >= 64) p->b_max = std::min(p->b_max, nn1_mirror_queries());
  if (const char* e = getenv("RKH_BATCH_MAX")) p->b_max = std::max(8, atoi(e));
  p->b_max = std::min<uint32_t>(p->b_max, 4096);
  if (const char* e = getenv("RKH_LANE_THRESHOLD")) p->lane_threshold = uint32_t(std::max(0, atoi(e)));
  if (const char* e = getenv("RKH_DUO_THRESHOLD")) p->duo_threshold = uint32_t(std::max(0, atoi(e)));
  if (p->scene->host.has_meshes) p->duo_threshold = 0;  // (instantiated without the support-map query only)
  if (const char* e = getenv("RKH_LANES_PER_EDGE")) {
    p->lanes_per_edge = (atoi(e) == 1) ? 1 : ((atoi(e) == 2) ? 2 : ((atoi(e) == 16) ? 16 : (atoi(e) == 0 ? 0 : 64)));
  } else if (p->n_dof <= 6 && scene_fits_lane_kernel(scene->host, p->lane_variant)) {
    p->lanes_per_edge = 0;  // automatic, per round
  } else {
    // one wavefront per candidate is the latency-optimal mapping; once a round can offer more waves than the chip
    // has slots (256 CUs x 4 SIMDs x 2 waves) four candidates share a wave
    p->lanes_per_edge = (uint64_t(n_problems) * 2 * p->b_max > 4096) ? 16 : 64;
  }
  if ((p->lanes_per_edge == 1 || p->lanes_per_edge == 2 || p->lanes_per_edge == 0) &&
      !(p->n_dof <= 7 && scene_fits_lane_kernel(scene->host, p->lanes_per_edge == 1 ? 1 : p->lane_variant)))
    p->lanes_per_edge = 64;  // the two-lanes-per-edge kernel does not take this scene
  if (p->lanes_per_edge == 16 && 2 * p->n_dof > 16) p->lanes_per_edge = 64;  // a 16-lane group holds at most 16 components
  if (scene->host.planar) p->lanes_per_edge = 64;  // planar chains have one mapping (one lane per edge, propagate_planar.hip)
  if (const char* e = getenv("RKH_PROFILE_NN")) p->profile_nn = atoi(e) != 0;
  // candidates per round = batch_factor * sqrt(n) per problem (results do not depend on it).  More candidates per
  // round mean fewer rounds but more discarded speculation (0.89 of the propagated edges are committed at 1.25, 0.72 at
  // 2, 0.55 at 3, 0.45 at 4), and a round is only cheap to enlarge while the chip is not full.  Measured optimum
  // (tests/diag_bench_sweep.sh, tests/diag_single.py): 1.25 for 256 problems x 100 000 vertices (5.6 M expansions/s),
  // 2 for 32 ... 128 problems (64 x 100 000: 3.03 M against 2.86 at 1.25 and 2.51 at 4), 4 for 16 (444 k against 322 k
  // at 1.25), 2 = 4 for a single problem (bound by the latency of one edge; 2 checks fewer edges).  The rule: what
  // brings a mid-run round (n = max_vertices / 2) of all problems to ~32 k edges -- one 32-edge steer wave per SIMD --
  // within [1.25, 2], up to 4 for at most 16 problems.
  double mid_sqrt_sum = 0.0;
  for (uint32_t i = 0; i < n_problems; ++i) mid_sqrt_sum += std::sqrt(0.5 * double(prms[i].max_vertices));
  const double factor_cap = n_problems == 1 ? 2.0 : (n_problems <= 16 ? 4.0 : 2.0);
  float batch_factor = float(std::min(factor_cap, std::max(1.25, 32768.0 / mid_sqrt_sum)));
  uint32_t b_min = 8;
  if (const char* e = getenv("RKH_BATCH_FACTOR")) batch_factor = float(atof(e));
  if (const char* e = getenv("RKH_BATCH_MIN")) b_min = std::max(1, atoi(e));
  p->nn_mirror = nn1_mirror_applies(p->D, p->coord_bound);
  for (int d = 0; d < p->D; ++d) {
    double m = std::max(std::fabs(p->lower[d]), std::fabs(p->upper[d]));
    for (uint32_t i = 0; i < n_problems; ++i) m = std::max(m, std::fabs(prms[i].start[d]));
    p->x_norm_bound += m * m;
  }
  p->x_norm_bound = std::sqrt(p->x_norm_bound) * (1.0 + 1e-9);
  const uint32_t P = n_problems;
  p->prob.resize(P);
  RKH_HIP(hipMalloc(&p->d_states, P * sizeof(PlannerState)));
  RKH_HIP(hipMalloc(&p->d_probs, P * sizeof(ProblemDev)));
  RKH_HIP(hipMalloc(&p->d_nn_args, P * sizeof(NnArgs)));
  RKH_HIP(hipMalloc(&p->d_io_steer, P * sizeof(EdgeIO)));
  RKH_HIP(hipMalloc(&p->d_io_probe, P * sizeof(EdgeIO)));
  if (!p->quasi_static && (p->lanes_per_edge == 1 || p->lanes_per_edge == 2 || p->lanes_per_edge == 0))
    RKH_HIP(hipMalloc(&p->d_lane_ws, std::max(propagate_lanes_workspace_bytes(p->n_dof, p->b_max, p->b_max, P),
                                              propagate_pairs_workspace_bytes(p->n_dof, p->b_max, p->b_max, P))));
  if (p->d_lane_ws) {
    // two prefix arrays of 2 P + 1 entries: waves of the two-lanes kernel, then single edges (one-wave-per-edge kernel)
    RKH_HIP(hipMalloc(&p->d_wave_base, 2 * (2 * size_t(P) + 1) * sizeof(uint32_t)));
    RKH_HIP(hipMemset(p->d_wave_base, 0, 2 * (2 * size_t(P) + 1) * sizeof(uint32_t)));
  }
  RKH_HIP(hipMalloc(&p->d_nn_base, (size_t(P) + 1) * sizeof(uint32_t)));
  RKH_HIP(hipMemset(p->d_nn_base, 0, (size_t(P) + 1) * sizeof(uint32_t)));
  RKH_HIP(hipMalloc(&p->d_sel, 2 * sizeof(uint32_t)));
  RKH_HIP(hipMemset(p->d_sel, 0, 2 * sizeof(uint32_t)));
  RKH_HIP(hipMalloc(&p->d_steps_exec, sizeof(unsigned long long)));
  RKH_HIP(hipMemset(p->d_steps_exec, 0, sizeof(unsigned long long)));
  for (uint32_t i = 0; i < P; ++i) {
    const uint64_t cap = (uint64_t(prms[i].max_vertices) + 1 + 255) / 256 * 256;
    p->max_capacity = std::max(p->max_capacity, cap);
  }
  p->part_blocks = std::max(nn1_partial_blocks(p->max_capacity, p->b_max, P), nn1_partial_blocks(p->max_capacity, 1, P));
  p->n_ub.assign(P, 1);
  std::vector<PlannerState> hs(P);
  std::vector<ProblemDev> hp(P);
  std::vector<NnArgs> hn(P);
  std::vector<EdgeIO> hio(P), hgp(P);
  const int D = p->D, DP = p->DP;
  for (uint32_t i = 0; i < P; ++i) {
    Problem& q = p->prob[i];
    q.prm = prms[i];
    {  // get_global_rng().seed(s): std::mt19937 / boost::mt19937 seeding, position at the end of the state
      std::vector<uint32_t> mt(kMtN + 1);
      mt[0] = uint32_t(prms[i].seed);
      for (int k = 1; k < kMtN; ++k) mt[k] = 1812433253u * (mt[k - 1] ^ (mt[k - 1] >> 30)) + uint32_t(k);
      mt[kMtN] = uint32_t(kMtN);
      RKH_HIP(hipMalloc(&q.d_mt, mt.size() * sizeof(uint32_t)));
      RKH_HIP(hipMemcpy(q.d_mt, mt.data(), mt.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    const uint64_t max_total = uint64_t(prms[i].max_vertices) + 1;
    q.capacity = (max_total + 255) / 256 * 256;
    RKH_HIP(hipMalloc(&q.d_tree, q.capacity * DP * sizeof(double)));
    RKH_HIP(hipMalloc(&q.d_parent, q.capacity * sizeof(uint32_t)));
    RKH_HIP(hipMalloc(&q.d_node_sample, q.capacity * sizeof(uint32_t)));
    RKH_HIP(hipMalloc(&q.d_goal_dist, q.capacity * sizeof(double)));
    q.sample_cap = std::max<uint64_t>(4 * max_total + 4 * p->b_max, 1u << 14);
    if (const char* e = getenv("RKH_SAMPLE_CAP")) q.sample_cap = std::max<uint64_t>(q.sample_cap, strtoull(e, nullptr, 10));
    RKH_HIP(hipMalloc(&q.d_samples, q.sample_cap * D * sizeof(double)));
    RKH_HIP(hipMalloc(&q.d_nn_seq, q.sample_cap * sizeof(uint32_t)));
    RKH_HIP(hipMalloc(&q.d_accept_log, q.sample_cap));
    RKH_HIP(hipMalloc(&q.d_nn_idx, p->b_max * sizeof(uint32_t)));
    RKH_HIP(hipMalloc(&q.d_nn_dist, p->b_max * sizeof(double)));
    RKH_HIP(hipMalloc(&q.d_x_out, uint64_t(p->b_max) * D * sizeof(double)));
    RKH_HIP(hipMalloc(&q.d_steps, p->b_max * sizeof(uint32_t)));
    RKH_HIP(hipMalloc(&q.d_accept, p->b_max));
    RKH_HIP(hipMalloc(&q.d_probe_x, uint64_t(p->b_max + kProbeGranule) * D * sizeof(double)));
    RKH_HIP(hipMalloc(&q.d_probe_steps, (p->b_max + kProbeGranule) * sizeof(uint32_t)));
    RKH_HIP(hipMalloc(&q.d_goal, D * sizeof(double)));
    RKH_HIP(hipMalloc(&q.d_part_dist, uint64_t(p->part_blocks) * p->b_max * sizeof(double)));
    // one more row than the partials need: NnArgs::seed (sampled minima of the matrix-core sweep, "none" = all ones)
    RKH_HIP(hipMalloc(&q.d_part_idx, uint64_t(p->part_blocks + 1) * p->b_max * sizeof(uint32_t)));
    RKH_HIP(hipMemset(q.d_part_idx + uint64_t(p->part_blocks) * p->b_max, 0xFF, uint64_t(p->b_max) * sizeof(uint32_t)));
    if (p->profile_nn) RKH_HIP(hipMalloc(&q.d_round_n, 2 * rkh_planner::kProfMax * sizeof(uint32_t)));
    if (p->nn_mirror) {
      RKH_HIP(hipMalloc(&q.d_mirror, nn1_mirror_bytes(q.capacity)));
      rkh_status ms = launch_mirror_fill(p->stream, q.d_mirror, q.capacity);
      if (ms != RKH_OK) return ms;
      const size_t cand_bytes = nn1_mirror_query_bytes() * p->b_max + 256;
      RKH_HIP(hipMalloc(&q.d_cand, cand_bytes));
      RKH_HIP(hipMemsetAsync(q.d_cand, 0, cand_bytes, p->stream));
    }
    // root vertex = query start (create_root, rrt_path_planner.tpp:131-133)
    std::vector<double> row(DP, 0.0);
    for (int d = 0; d < D; ++d) row[d] = prms[i].start[d];
    RKH_HIP(hipMemcpy(q.d_tree, row.data(), DP * sizeof(double), hipMemcpyHostToDevice));
    if (p->nn_mirror) {
      uint32_t* dxw = reinterpret_cast<uint32_t*>(static_cast<char*>(q.d_cand) + nn1_mirror_query_bytes() * p->b_max);
      rkh_status ms = launch_mirror_build(p->stream, q.d_mirror, q.d_tree, 1, D, DP, dxw);
      if (ms != RKH_OK) return ms;
    }
    const uint32_t no_parent = 0xFFFFFFFFu;
    RKH_HIP(hipMemcpy(q.d_parent, &no_parent, sizeof(uint32_t), hipMemcpyHostToDevice));
    RKH_HIP(hipMemcpy(q.d_goal, prms[i].goal, D * sizeof(double), hipMemcpyHostToDevice));
    PlannerState& s0 = hs[i];
    std::memset(&s0, 0, sizeof(s0));
    s0.n = 1;
    s0.n_before = 1;
    s0.probed_n = 1;
    s0.max_total = uint32_t(max_total);
    s0.b_max = p->b_max;
    s0.b_min = b_min;
    s0.batch_factor = batch_factor;
    q.h_state = s0;
    p->n_ub[i] = 1;
    PlannerState* dst = p->d_states + i;
    ProblemDev& pd = hp[i];
    pd.st = dst;
    pd.tree = q.d_tree;
    pd.parent = q.d_parent;
    pd.node_sample = q.d_node_sample;
    pd.samples = q.d_samples;
    pd.nn_seq = q.d_nn_seq;
    pd.accept_log = q.d_accept_log;
    pd.nn_idx = q.d_nn_idx;
    pd.nn_dist = q.d_nn_dist;
    pd.x_out = q.d_x_out;
    pd.accept = q.d_accept;
    pd.round_n = q.d_round_n;
    pd.mirror = static_cast<uint4*>(q.d_mirror);
    pd.dx_max_bits = q.d_cand ? reinterpret_cast<uint32_t*>(static_cast<char*>(q.d_cand) + nn1_mirror_query_bytes() * p->b_max)
                              : nullptr;
    NnArgs& na = hn[i];
    na.pos = q.d_tree;
    na.d_n = &dst->n;
    na.q = q.d_samples;
    na.d_qoff = &dst->s0;
    na.B = p->b_max;
    na.d_B = &dst->B;
    na.part_dist = q.d_part_dist;
    na.part_idx = q.d_part_idx;
    na.seed = q.d_part_idx + uint64_t(p->part_blocks) * p->b_max;
    na.idx = q.d_nn_idx;
    na.dist = q.d_nn_dist;
    na.mirror = q.d_mirror;
    if (q.d_cand) {
      nn1_mirror_carve(q.d_cand, p->b_max, &na);
      na.dx_max_bits = pd.dx_max_bits;
    }
    EdgeIO& io = hio[i];
    io.src = q.d_tree;
    io.src_idx = q.d_nn_idx;
    io.src_stride = DP;
    io.tgt = q.d_samples;
    io.d_tgt_off = &dst->s0;
    io.tgt_stride = D;
    io.B = p->b_max;
    io.d_B = &dst->B;
    io.x_out = q.d_x_out;
    io.steps_free = q.d_steps;
    io.mode = EDGE_STEER_ACCEPT;
    io.best_case = q.d_nn_dist;
    io.steer_tol = prms[i].steer_tol;
    io.accept = q.d_accept;
    io.err_flag = scene->d_err;
    EdgeIO& gp = hgp[i];
    gp.src = q.d_tree;
    gp.d_src_first = &dst->n_before;
    gp.src_stride = DP;
    gp.tgt = q.d_goal;
    gp.tgt_stride = 0;
    gp.B = p->b_max + kProbeGranule;
    gp.d_B = &dst->n_new;
    gp.x_out = q.d_probe_x;
    gp.steps_free = q.d_probe_steps;
    gp.mode = EDGE_GOAL_PROBE;
    gp.goal_dist = q.d_goal_dist;
    gp.err_flag = scene->d_err;
  }
  RKH_HIP(hipMemcpy(p->d_states, hs.data(), P * sizeof(PlannerState), hipMemcpyHostToDevice));
  RKH_HIP(hipMemcpy(p->d_probs, hp.data(), P * sizeof(ProblemDev), hipMemcpyHostToDevice));
  RKH_HIP(hipMemcpy(p->d_nn_args, hn.data(), P * sizeof(NnArgs), hipMemcpyHostToDevice));
  RKH_HIP(hipMemcpy(p->d_io_steer, hio.data(), P * sizeof(EdgeIO), hipMemcpyHostToDevice));
  RKH_HIP(hipMemcpy(p->d_io_probe, hgp.data(), P * sizeof(EdgeIO), hipMemcpyHostToDevice));
  if (p->d_wave_base && !p->quasi_static) {  // second phase of a split steer launch: the same edges through the survivors' lists
    RKH_HIP(hipMalloc(&p->d_io_steer2, P * sizeof(EdgeIO)));
    RKH_HIP(hipMalloc(&p->d_io_probe2, P * sizeof(EdgeIO)));
    RKH_HIP(hipMalloc(&p->d_cnt2, 2 * size_t(P) * sizeof(uint32_t)));
    RKH_HIP(hipMemset(p->d_cnt2, 0, 2 * size_t(P) * sizeof(uint32_t)));
    RKH_HIP(hipMalloc(&p->d_wave_base2, (2 * size_t(P) + 1) * sizeof(uint32_t)));
    RKH_HIP(hipMemset(p->d_wave_base2, 0, (2 * size_t(P) + 1) * sizeof(uint32_t)));
    if (p->steer_stepwise && p->lane_variant == 2) {
      const size_t cap = size_t(P) * (2 * size_t(p->b_max) + kProbeGranule);
      for (auto& l : p->d_step_list) RKH_HIP(hipMalloc(&l, cap * sizeof(uint2)));
      RKH_HIP(hipMalloc(&p->d_step_cnt, (kMaxSteps + 3) * sizeof(uint32_t)));
      RKH_HIP(hipMemset(p->d_step_cnt, 0, (kMaxSteps + 3) * sizeof(uint32_t)));
      // the step kernel's blocks take their RK4 workspace out of the two-lanes workspace
      if (propagate_pair_step_workspace_bytes(p->n_dof, p->step_blocks_cap) >
          propagate_pairs_workspace_bytes(p->n_dof, p->b_max, p->b_max, P))
        p->step_blocks_cap = uint32_t(propagate_pairs_workspace_bytes(p->n_dof, p->b_max, p->b_max, P) /
                                      propagate_pair_step_workspace_bytes(p->n_dof, 1));
    }
    std::vector<EdgeIO> hio2 = hio, hgp2 = hgp;
    for (uint32_t i = 0; i < P; ++i) {
      Problem& q = p->prob[i];
      RKH_HIP(hipMalloc(&q.d_ids_c, size_t(p->b_max) * sizeof(uint32_t)));
      RKH_HIP(hipMalloc(&q.d_ids_p, size_t(p->b_max + kProbeGranule) * sizeof(uint32_t)));
      hio2[i].edge_ids = q.d_ids_c;
      hio2[i].resume = hio[i].x_out;
      hio2[i].d_B = p->d_cnt2 + 2 * i;
      hgp2[i].edge_ids = q.d_ids_p;
      hgp2[i].resume = hgp[i].x_out;
      hgp2[i].d_B = p->d_cnt2 + 2 * i + 1;
    }
    RKH_HIP(hipMemcpy(p->d_io_steer2, hio2.data(), P * sizeof(EdgeIO), hipMemcpyHostToDevice));
    RKH_HIP(hipMemcpy(p->d_io_probe2, hgp2.data(), P * sizeof(EdgeIO), hipMemcpyHostToDevice));
  }
  *out = p;
  return RKH_OK;
}

rkh_status rkh_planner_create_batch(rkh_scene* scene, const rkh_dyn_space* space, const rkh_rrt_params* prms,
                                    uint32_t n_problems, rkh_planner** out) {
  if (!space) return RKH_ERR_BAD_ARG;
  if (scene && scene->host.planar && !scene->host.planar_dynamics) {
    set_error("this planar (2D) chain was given at position level (no actuators / inertias): quasi-static spaces only");
    return RKH_ERR_UNSUPPORTED;
  }
  return planner_create_common(scene, space, nullptr, prms, n_problems, out);
}

rkh_status rkh_planner_create_qs_batch(rkh_scene* scene, const rkh_qs_space* space, const rkh_rrt_params* prms,
                                       uint32_t n_problems, rkh_planner** out) {
  if (!space) return RKH_ERR_BAD_ARG;
  return planner_create_common(scene, nullptr, space, prms, n_problems, out);
}

rkh_status rkh_planner_create(rkh_scene* scene, const rkh_dyn_space* space, const rkh_rrt_params* prm,
                              rkh_planner** out) {
  return rkh_planner_create_batch(scene, space, prm, 1, out);
}

uint32_t rkh_planner_num_problems(const rkh_planner* p) { return p ? p->P : 0; }

rkh_status rkh_planner_destroy(rkh_planner* p) {
  if (!p) return RKH_OK;
  (void)hipStreamSynchronize(p->stream);
  for (Problem& q : p->prob) free_problem(q);
  (void)hipFree(p->d_states);
  (void)hipFree(p->d_probs);
  (void)hipFree(p->d_nn_args);
  (void)hipFree(p->d_io_steer);
  (void)hipFree(p->d_io_probe);
  (void)hipFree(p->d_io_steer2);
  (void)hipFree(p->d_io_probe2);
  (void)hipFree(p->d_cnt2);
  (void)hipFree(p->d_wave_base2);
  (void)hipFree(p->d_step_list[0]);
  (void)hipFree(p->d_step_list[1]);
  (void)hipFree(p->d_step_cnt);
  (void)hipFree(p->d_steps_exec);
  for (auto& sg : p->staging) {
    if (sg.pending) (void)hipEventSynchronize(sg.done);
    if (sg.h_tab) (void)hipHostFree(sg.h_tab);
    if (sg.d_tab) (void)hipFree(sg.d_tab);
    if (sg.done) (void)hipEventDestroy(sg.done);
  }
  (void)hipFree(p->d_lane_ws);
  (void)hipFree(p->d_sel);
  (void)hipFree(p->d_wave_base);
  (void)hipFree(p->d_nn_base);
  for (hipEvent_t e : p->ev) (void)hipEventDestroy(e);
  for (hipEvent_t e : p->ev_steer) (void)hipEventDestroy(e);
  (void)hipStreamSynchronize(p->copy_stream);
  (void)hipStreamDestroy(p->copy_stream);
  if (p->d_bounds) (void)hipFree(p->d_bounds);
  if (p->h_gd) (void)hipHostFree(p->h_gd);
  if (p->h_gd_tab) (void)hipHostFree(p->h_gd_tab);
  if (p->d_gd_tab) (void)hipFree(p->d_gd_tab);
  if (p->d_gd) (void)hipFree(p->d_gd);
  (void)hipStreamDestroy(p->stream);
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
  std::vector<uint64_t> rows(p->prof_rounds, 0);
  std::vector<uint32_t> rn(p->prof_rounds);
  for (Problem& q : p->prob) {
    RKH_HIP(hipMemcpy(rn.data(), q.d_round_n, rn.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    for (uint32_t r = 0; r < p->prof_rounds; ++r) rows[r] += rn[r];
  }
  for (uint32_t r = 0; r < p->prof_rounds; ++r) {
    if (rows[r] == 0) continue;  // no-op round after completion
    float ms = 0.f;
    RKH_HIP(hipEventElapsedTime(&ms, p->ev[2 * r], p->ev[2 * r + 1]));
    *total_ms += ms;
    *total_bytes += rows[r] * p->DP * sizeof(double);  // algorithmic bytes of one launch: sum over problems of n * D * 8
    *launches += 1;
  }
  return RKH_OK;
}

rkh_status rkh_planner_nn_pairs(rkh_planner* p, uint64_t* pairs) {
  if (!p || !pairs) return RKH_ERR_BAD_ARG;
  *pairs = 0;
  if (!p->profile_nn || p->prof_rounds == 0) return RKH_OK;
  RKH_HIP(hipStreamSynchronize(p->stream));
  std::vector<uint32_t> rn(p->prof_rounds), rb(p->prof_rounds);
  for (Problem& q : p->prob) {
    RKH_HIP(hipMemcpy(rn.data(), q.d_round_n, rn.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    RKH_HIP(hipMemcpy(rb.data(), q.d_round_n + rkh_planner::kProfMax, rb.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    for (uint32_t r = 0; r < p->prof_rounds; ++r) *pairs += uint64_t(rn[r]) * rb[r];
  }
  return RKH_OK;
}

rkh_status rkh_planner_steer_profile(rkh_planner* p, double* total_ms, uint64_t* launches) {
  if (!p || !total_ms || !launches) return RKH_ERR_BAD_ARG;
  *total_ms = 0.0;
  *launches = 0;
  if (!p->profile_nn || p->prof_rounds == 0) return RKH_OK;
  RKH_HIP(hipStreamSynchronize(p->stream));
  for (uint32_t r = 0; r < p->prof_rounds; ++r) {
    float ms = 0.f;
    RKH_HIP(hipEventElapsedTime(&ms, p->ev_steer[2 * r], p->ev_steer[2 * r + 1]));
    *total_ms += ms;
    *launches += 1;
  }
  return RKH_OK;
}

rkh_status rkh_planner_steer_steps(rkh_planner* p, uint64_t* executed_steps) {
  if (!p || !executed_steps) return RKH_ERR_BAD_ARG;
  *executed_steps = 0;
  if (!p->d_steps_exec) return RKH_OK;
  RKH_HIP(hipStreamSynchronize(p->stream));
  unsigned long long v = 0;
  RKH_HIP(hipMemcpy(&v, p->d_steps_exec, sizeof(v), hipMemcpyDeviceToHost));
  *executed_steps = v;
  return RKH_OK;
}

rkh_status rkh_planner_enqueue(rkh_planner* p, uint32_t rounds) {
  if (!p) return RKH_ERR_BAD_ARG;
  // make sure the enqueued rounds cannot run out of samples (usually already there: see below)
  const uint64_t share = uint64_t(rounds ? rounds : 1) * p->b_max;
  rkh_status st = upload_samples_all(p, share, 0);
  if (st != RKH_OK) return st;
  for (uint32_t r = 0; r < rounds; ++r) {
    st = enqueue_round(p);
    if (st != RKH_OK) return st;
  }
  // while the GPU works on these rounds: the next call's share of the stream (generated on the host, copied behind the
  // rounds on the same stream)
  if (rounds) st = upload_samples_all(p, 2 * share, 1);
  return st;
}

rkh_status rkh_planner_sync(rkh_planner* p, rkh_planner_stats* stats) {
  if (!p) return RKH_ERR_BAD_ARG;
  rkh_status st = read_states(p);
  if (st != RKH_OK) return st;
  int flag = 0;
  RKH_HIP(hipMemcpy(&flag, p->scene->d_err, sizeof(int), hipMemcpyDeviceToHost));
  if (flag != 0) {
    RKH_HIP(hipMemset(p->scene->d_err, 0, sizeof(int)));
    set_error("planner: mass matrix is singular (Cholesky pivot < 1e-8)");
    return rkh_status(flag);
  }
  bool all_done = true, pending = false;
  for (Problem& q : p->prob) {
    if (!(q.truncated || q.h_state.done == 1)) all_done = false;
    if (q.h_state.probed_n < q.h_state.n) pending = true;
  }
  if (all_done && pending) {  // finished: run the goal probes of the last committed vertices
    st = flush_probes(p);
    if (st != RKH_OK) return st;
    st = read_states(p);
    if (st != RKH_OK) return st;
  }
  // goal-probe results since the last call, all problems: asynchronous copies into one pinned buffer, one wait
  std::vector<uint64_t> gd_off(p->P, 0), gd_cnt(p->P, 0);
  {
    uint64_t total = 0;
    for (uint32_t i = 0; i < p->P; ++i) {
      Problem& q = p->prob[i];
      const uint64_t probed = q.h_state.probed_n < 1 ? 1 : q.h_state.probed_n;
      if (!q.truncated && probed > 1 && q.goal_checked < probed - 1) {
        gd_off[i] = total;
        gd_cnt[i] = probed - 1 - q.goal_checked;
        total += gd_cnt[i];
      }
    }
    if (total > p->h_gd_cap) {
      if (p->h_gd) (void)hipHostFree(p->h_gd);
      p->h_gd = nullptr;
      p->h_gd_cap = total + total / 2 + 1024;
      RKH_HIP(hipHostMalloc(reinterpret_cast<void**>(&p->h_gd), p->h_gd_cap * sizeof(double), hipHostMallocDefault));
    }
    if (total) {
      if (!p->h_gd_tab) {
        RKH_HIP(hipHostMalloc(reinterpret_cast<void**>(&p->h_gd_tab), p->P * sizeof(GoalSeg), hipHostMallocDefault));
        RKH_HIP(hipMalloc(&p->d_gd_tab, p->P * sizeof(GoalSeg)));
      }
      if (total > p->d_gd_cap) {
        if (p->d_gd) (void)hipFree(p->d_gd);
        p->d_gd = nullptr;
        p->d_gd_cap = total + total / 2 + 1024;
        RKH_HIP(hipMalloc(&p->d_gd, p->d_gd_cap * sizeof(double)));
      }
      uint32_t n_seg = 0;
      for (uint32_t i = 0; i < p->P; ++i)
        if (gd_cnt[i]) {
          GoalSeg& g = p->h_gd_tab[n_seg++];
          g.src = p->prob[i].d_goal_dist + p->prob[i].goal_checked;
          g.dst_off = gd_off[i];
          g.count = gd_cnt[i];
        }
      RKH_HIP(hipMemcpyAsync(p->d_gd_tab, p->h_gd_tab, n_seg * sizeof(GoalSeg), hipMemcpyHostToDevice, p->stream));
      hipLaunchKernelGGL(gather_goal_dist_kernel, dim3(n_seg), dim3(256), 0, p->stream, p->d_gd_tab, p->d_gd);
      RKH_HIP(hipGetLastError());
      RKH_HIP(hipMemcpyAsync(p->h_gd, p->d_gd, total * sizeof(double), hipMemcpyDeviceToHost, p->stream));
      RKH_HIP(hipStreamSynchronize(p->stream));
    }
  }
  for (uint32_t i = 0; i < p->P; ++i) {
    Problem& q = p->prob[i];
    PlannerState& hs = q.h_state;
    if (!q.truncated && hs.done != 1 && uint64_t(hs.s0) + 64ull * p->b_max > q.sample_cap) {
      RKH_HIP(hipStreamSynchronize(p->copy_stream));
      const rkh_status gs = grow_sample_buffers(p, i, std::max<uint64_t>(2 * q.sample_cap, uint64_t(hs.s0) + 256ull * p->b_max));
      if (gs != RKH_OK) return gs;
    }
    if (hs.done == 2 && q.samples_ready < q.sample_cap) {  // sample stream ran dry mid-enqueue: refill and carry on
      hs.done = 0;
      RKH_HIP(hipMemcpy(&p->d_states[i].done, &hs.done, sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    // edge_added: a finite goal-probe distance registers a solution if it beats the best so far
    // (planning_visitors.hpp:194-200, solution_path_factories.hpp:58-110); keep_going() then also checks
    // max_num_results (p2p_planning_query.hpp:121-123).
    const uint64_t probed = hs.probed_n < 1 ? 1 : hs.probed_n;  // vertices [1, probed) have a goal-probe result
    if (!q.truncated && probed > 1 && q.goal_checked < probed - 1) {
      const uint64_t first = q.goal_checked, cnt = probed - 1 - first;
      const double* gd = p->h_gd + gd_off[i];
      std::vector<double> pos;
      std::vector<uint32_t> par;
      for (uint64_t k = 0; k < cnt; ++k) {
        if (!(gd[k] < INFINITY)) continue;
        if (pos.empty()) {
          pos.resize(uint64_t(hs.n) * p->DP);
          par.resize(hs.n);
          RKH_HIP(hipMemcpy(pos.data(), q.d_tree, pos.size() * sizeof(double), hipMemcpyDeviceToHost));
          RKH_HIP(hipMemcpy(par.data(), q.d_parent, par.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
        }
        double total = gd[k];
        uint64_t v = first + k + 1;
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
        if (q.num_solutions == 0 || total < q.best_cost) {
          q.best_cost = total;
          q.best_vertex = uint32_t(first + k + 1);
          ++q.num_solutions;
          if (q.num_solutions >= q.prm.max_results) {
            // the sequential planner stops right after this vertex: drop what speculation added beyond it
            q.truncated = true;
            q.final_n = first + k + 2;
            uint32_t smp = 0;
            RKH_HIP(hipMemcpy(&smp, q.d_node_sample + (first + k + 1), sizeof(uint32_t), hipMemcpyDeviceToHost));
            q.final_iterations = uint64_t(smp) + 1;
            const uint32_t one = 1;  // freeze the problem on the device as well
            RKH_HIP(hipMemcpy(&p->d_states[i].done, &one, sizeof(uint32_t), hipMemcpyHostToDevice));
            break;
          }
        }
      }
      q.goal_checked = probed - 1;
    }
    if (stats) {
      rkh_planner_stats& o = stats[i];
      std::memset(&o, 0, sizeof(o));
      o.num_vertices = q.truncated ? q.final_n : hs.n;
      o.iterations = q.truncated ? q.final_iterations : hs.s0;
      o.edges_checked = o.iterations + (o.num_vertices - 1);
      o.edges_speculated = hs.edges_speculated + (hs.n - 1);
      o.rounds = hs.rounds;
      o.num_solutions = q.num_solutions;
      o.best_cost = q.best_cost;
      o.done = (q.truncated || hs.done == 1) ? 1u : 0u;
    }
  }
  return RKH_OK;
}

// The best registered solution as a vertex path root -> ... -> v (the motion then goes on to the goal, which the goal
// probe of v reached): register_basic_solution_path_impl (solution_path_factories.hpp:58-110) walks the same parents.
rkh_status rkh_planner_get_solution(rkh_planner* p, uint32_t problem, uint32_t* path, uint32_t capacity,
                                    uint32_t* n_path, double* cost) {
  if (!p || problem >= p->P || !n_path) return RKH_ERR_BAD_ARG;
  Problem& q = p->prob[problem];
  *n_path = 0;
  if (cost) *cost = q.best_cost;
  if (q.best_vertex == 0xFFFFFFFFu) return RKH_OK;  // no solution registered
  RKH_HIP(hipStreamSynchronize(p->stream));
  std::vector<uint32_t> par(size_t(q.best_vertex) + 1);
  RKH_HIP(hipMemcpy(par.data(), q.d_parent, par.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
  std::vector<uint32_t> rev;
  for (uint32_t v = q.best_vertex; v != 0xFFFFFFFFu; v = par[v]) rev.push_back(v);
  *n_path = uint32_t(rev.size());
  if (path) {
    if (capacity < rev.size()) {
      set_error("rkh_planner_get_solution: path buffer too small");
      return RKH_ERR_CAPACITY;
    }
    for (size_t i = 0; i < rev.size(); ++i) path[i] = rev[rev.size() - 1 - i];
  }
  return RKH_OK;
}

rkh_status rkh_planner_solve(rkh_planner* p, rkh_planner_stats* stats) {
  if (!p) return RKH_ERR_BAD_ARG;
  std::vector<rkh_planner_stats> local(p->P);
  for (;;) {
    rkh_status st = rkh_planner_enqueue(p, 16);
    if (st != RKH_OK) return st;
    st = rkh_planner_sync(p, local.data());
    if (st != RKH_OK) return st;
    bool all = true;
    for (uint32_t i = 0; i < p->P; ++i) {
      if (!local[i].done) all = false;
      Problem& q = p->prob[i];
      if (!local[i].done && q.samples_ready >= q.sample_cap && q.h_state.s0 + p->b_max > q.sample_cap) {
        set_error("planner: sample stream buffers did not grow (out of device memory?)");  // rkh_planner_sync grows them
        return RKH_ERR_CAPACITY;
      }
    }
    if (all) break;
  }
  if (stats) std::memcpy(stats, local.data(), p->P * sizeof(rkh_planner_stats));
  return RKH_OK;
}

rkh_status rkh_planner_get_tree(rkh_planner* p, uint32_t problem, double* pos, uint32_t* parent, uint32_t* nn_seq,
                                uint8_t* accept, double* goal_dist) {
  if (!p || problem >= p->P) return RKH_ERR_BAD_ARG;
  if (goal_dist) {  // make sure no goal probe is pending
    rkh_status fs = flush_probes(p);
    if (fs != RKH_OK) return fs;
  }
  RKH_HIP(hipStreamSynchronize(p->stream));
  Problem& q = p->prob[problem];
  const uint64_t n = q.truncated ? q.final_n : q.h_state.n;
  const uint64_t it = q.truncated ? q.final_iterations : q.h_state.s0;
  if (pos) {
    if (p->DP == p->D) {
      RKH_HIP(hipMemcpy(pos, q.d_tree, n * p->D * sizeof(double), hipMemcpyDeviceToHost));
    } else {
      std::vector<double> tmp(n * p->DP);
      RKH_HIP(hipMemcpy(tmp.data(), q.d_tree, tmp.size() * sizeof(double), hipMemcpyDeviceToHost));
      for (uint64_t i = 0; i < n; ++i) std::memcpy(pos + i * p->D, &tmp[i * p->DP], p->D * sizeof(double));
    }
  }
  if (parent) RKH_HIP(hipMemcpy(parent, q.d_parent, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
  if (nn_seq && it) RKH_HIP(hipMemcpy(nn_seq, q.d_nn_seq, it * sizeof(uint32_t), hipMemcpyDeviceToHost));
  if (accept && it) RKH_HIP(hipMemcpy(accept, q.d_accept_log, it, hipMemcpyDeviceToHost));
  if (goal_dist && n > 1) RKH_HIP(hipMemcpy(goal_dist, q.d_goal_dist, (n - 1) * sizeof(double), hipMemcpyDeviceToHost));
  return RKH_OK;
}

}  // extern "C"
