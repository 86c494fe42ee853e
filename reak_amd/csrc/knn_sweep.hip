// knn_sweep.hip -- exact batched k-nearest-neighbour (with radius) sweeps over the tree (gfx950).
//
// Replaces min_dist_linear_search k-NN (ctrl/path_planning/topological_search.hpp:244-274: bounded max-heap,
// candidates need d < radius strictly, output ascending) with euclidean_distance_metric
// (ctrl/topologies/vect_distance_metrics.hpp:113-150), as used by star_neighborhood
// (ctrl/graph_alg/neighborhood_functors.hpp:95-102) for RRT* / PRM.
//
// A per-thread top-k does not fit registers (k = 4(floor(log2 n)+1) = 84 at n = 1M), so the search is
//   A. bound sweep   : every thread keeps only the minimum of its own row subset; the k-th smallest of those
//                      M = blocks x R subset minima is an upper bound tau on the k-th smallest distance overall
//                      (k disjoint subsets each hold a vertex at least that close)
//   B. collect sweep : every vertex with d <= tau and d < radius is appended to the query's candidate list
//                      (about 1.2 k entries when M >= 4k); exact: the true k nearest are all in the list
//   C. select        : one block per query sorts its candidates by (distance, index) and writes the first k.
// Distances are the reference's left-to-right fp64 sums with a correctly rounded sqrt, so the returned set and
// its order equal the CPU search's (among exactly equal distances the reference order is std::heap-defined; here
// it is ascending index).  Same tile / broadcast structure as nn_sweep.hip; traffic is two passes over the rows.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <cmath>

#include "rkh_internal.h"

namespace rkh {

int nn_padded_dims(int D);

static constexpr int kTileRows = 256;
static constexpr int kThreads = 256;
static constexpr int kSelMax = 8;  // a block contributes the ksel <= 8 smallest of its row-subset minima per query

static inline int knn_query_block(uint32_t B) { return B <= 8 ? 8 : 32; }  // queries per block

// MODE 0: bound sweep (per query, the kSel smallest of the block's R = 256/QB row-subset minima).
// MODE 1: collect sweep (appends candidates).
// Thread (ql, r) keeps query ql in registers and scans rows r, r+R, .. of every tile; with few queries (RRT* / PRM:
// one per tree) QB = 8 puts 32 threads on each query so the whole block works on the rows.  The next tile is fetched
// into registers while the current one is scanned (as in nn_sweep.hip).
template <int DP, int QB, int MODE>
__global__ __launch_bounds__(kThreads) void knn_sweep_kernel(KnnArgs single, const KnnArgs* __restrict__ tab) {
  constexpr int R = kThreads / QB;
  constexpr int ROWS_PER_THREAD = kTileRows / R;
  const KnnArgs a = tab ? tab[blockIdx.z] : single;
  if (blockIdx.x >= a.ws.gx || blockIdx.y * QB >= a.B) return;  // a table's grid is sized for its largest job
  const double* __restrict__ pos = a.pos;
  const uint64_t n = a.n;
  const double* __restrict__ q = a.q;
  const int D = a.D;
  const uint32_t B = a.B, Bpad = a.B, cmax = a.ws.cmax;
  const double radius = a.radius;
  double* __restrict__ sub = a.ws.sub;
  const double* __restrict__ tau = a.ws.tau;
  uint32_t* __restrict__ cnt = a.ws.cnt;
  double* __restrict__ cand_d = a.ws.cand_d;
  uint32_t* __restrict__ cand_i = a.ws.cand_i;
  uint32_t* __restrict__ overflow = a.ws.overflow;
  const uint32_t gx = a.ws.gx;
  __shared__ __attribute__((aligned(16))) double tile[kTileRows * DP];
  __shared__ double red[kThreads];
  const int tid = threadIdx.x;
  const int ql = tid % QB;
  const int r = tid / QB;
  const uint32_t qi = blockIdx.y * QB + ql;
  const bool q_valid = qi < B;
  double qv[DP];
  {
    const uint64_t qsrc = q_valid ? qi : (B - 1);
#pragma unroll
    for (int d = 0; d < DP; ++d) qv[d] = d < D ? q[qsrc * D + d] : 0.0;
  }
  const uint64_t tiles_total = (n + kTileRows - 1) / kTileRows;
  const uint64_t tiles_per_block = (tiles_total + gx - 1) / gx;
  const uint64_t tile0 = uint64_t(blockIdx.x) * tiles_per_block;
  uint64_t tile1 = tile0 + tiles_per_block;
  if (tile1 > tiles_total) tile1 = tiles_total;

  double best_s = INFINITY;  // MODE 0: smallest square of this thread's subset
  double t_q = INFINITY, thr_s = INFINITY;
  if (MODE == 1) {
    t_q = tau[q_valid ? qi : 0];
    // squares above this cannot give sqrt(s) <= tau (4 ulp of slack for the rounding of tau*tau and of sqrt)
    thr_s = (t_q == INFINITY) ? INFINITY : (t_q * t_q) * (1.0 + 8.0 * DBL_EPSILON);
  }
  constexpr int N2 = kTileRows * DP / 2;
  constexpr int PF = N2 / kThreads;
  static_assert(N2 % kThreads == 0, "tile must split evenly");
  double2 pf[PF];
  auto fetch = [&](uint64_t t) {
    const uint64_t row_base = t * kTileRows;
    const double2* src = reinterpret_cast<const double2*>(pos + row_base * DP);
    const uint64_t valid2 = (n - row_base >= uint64_t(kTileRows)) ? uint64_t(N2) : (n - row_base) * DP / 2;
#pragma unroll
    for (int j = 0; j < PF; ++j) {
      const int i = tid + j * kThreads;
      pf[j] = (uint64_t(i) < valid2) ? src[i] : make_double2(INFINITY, INFINITY);  // rows beyond n never qualify
    }
  };
  if (tile0 < tile1) fetch(tile0);
  for (uint64_t t = tile0; t < tile1; ++t) {
    const uint64_t row_base = t * kTileRows;
    {
      double2* dst = reinterpret_cast<double2*>(tile);
#pragma unroll
      for (int j = 0; j < PF; ++j) dst[tid + j * kThreads] = pf[j];
    }
    if (t + 1 < tile1) fetch(t + 1);
    __syncthreads();
#pragma unroll 2
    for (int k = 0; k < ROWS_PER_THREAD; ++k) {
      const int row = k * R + r;
      const double* p = tile + row * DP;
      double df = qv[0] - p[0];
      double s = df * df;
#pragma unroll
      for (int d = 1; d < DP; ++d) {
        df = qv[d] - p[d];
        s = s + df * df;
      }
      if (MODE == 0) {
        if (s < best_s) best_s = s;
      } else if (s <= thr_s && q_valid) {
        const double dd = sqrt(s);
        if (dd <= t_q && dd < radius) {
          const uint32_t slot = atomicAdd(&cnt[qi], 1u);
          if (slot < cmax) {
            cand_d[uint64_t(qi) * cmax + slot] = dd;
            cand_i[uint64_t(qi) * cmax + slot] = uint32_t(row_base + row);
          } else {
            *overflow = 1u;
          }
        }
      }
    }
    __syncthreads();
  }
  if (MODE == 0) {
    // the ksel smallest of the R subset minima of every query (ranks by (value, subset) are a permutation; any k
    // disjoint subsets with minimum <= tau prove that k vertices lie within tau, so dropping the larger ones is safe)
    const double mine = sqrt(best_s);
    red[tid] = mine;
    __syncthreads();
    int rank = 0;
#pragma unroll 8
    for (int rr = 0; rr < R; ++rr) {
      const double v = red[rr * QB + ql];
      rank += (v < mine || (v == mine && rr < r)) ? 1 : 0;
    }
    const int ksel = int(a.ws.ksel);
    if (q_valid && rank < ksel) sub[(uint64_t(blockIdx.x) * ksel + rank) * Bpad + qi] = mine;
  }
}

// bitonic sort of (key, idx) pairs in LDS, ascending lexicographic; n_pow2 elements, blockDim.x threads
__device__ __forceinline__ void bitonic_sort_lds(double* key, uint32_t* idx, uint32_t n_pow2) {
  for (uint32_t size = 2; size <= n_pow2; size <<= 1) {
    for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
      __syncthreads();
      for (uint32_t t = threadIdx.x; t < n_pow2 / 2; t += blockDim.x) {
        const uint32_t lo = 2 * t - (t & (stride - 1));
        const uint32_t hi = lo + stride;
        const bool up = ((lo & size) == 0);
        const double ka = key[lo], kb = key[hi];
        const uint32_t ia = idx[lo], ib = idx[hi];
        const bool a_gt_b = (ka > kb) || (ka == kb && ia > ib);
        if (a_gt_b == up) {
          key[lo] = kb; key[hi] = ka;
          idx[lo] = ib; idx[hi] = ia;
        }
      }
    }
  }
  __syncthreads();
}

// one block per query: tau = k-th smallest subset minimum (or +inf if there are fewer than k subsets)
__global__ __launch_bounds__(256) void knn_tau_kernel(KnnArgs single, const KnnArgs* __restrict__ tab) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const KnnArgs a = tab ? tab[blockIdx.z] : single;
  const uint32_t qi = blockIdx.x;
  if (qi >= a.B) return;
  const double* __restrict__ sub = a.ws.sub;
  const uint32_t m_sub = a.ws.m_sub, m_pow2 = a.m_pow2, Bpad = a.B, k = a.k;
  double* __restrict__ tau = a.ws.tau;
  uint32_t* __restrict__ cnt = a.ws.cnt;
  uint32_t* __restrict__ overflow = a.ws.overflow;
  double* key = reinterpret_cast<double*>(smem);
  uint32_t* idx = reinterpret_cast<uint32_t*>(key + m_pow2);
  for (uint32_t i = threadIdx.x; i < m_pow2; i += blockDim.x) {
    key[i] = i < m_sub ? sub[uint64_t(i) * Bpad + qi] : INFINITY;
    idx[i] = i;
  }
  bitonic_sort_lds(key, idx, m_pow2);
  if (threadIdx.x == 0) {
    tau[qi] = (m_sub >= k) ? key[k - 1] : INFINITY;
    cnt[qi] = 0;
    if (qi == 0) *overflow = 0;
  }
}

// one block per query: sort the candidates, write the first k (nearest first), pad the rest
__global__ __launch_bounds__(256) void knn_select_kernel(KnnArgs single, const KnnArgs* __restrict__ tab) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const KnnArgs a = tab ? tab[blockIdx.z] : single;
  const uint32_t qi = blockIdx.x;
  if (qi >= a.B) return;
  const double* __restrict__ cand_d = a.ws.cand_d;
  const uint32_t* __restrict__ cand_i = a.ws.cand_i;
  const uint32_t* __restrict__ cnt = a.ws.cnt;
  const uint32_t cmax = a.ws.cmax, c_pow2 = a.ws.cmax, k = a.k;
  uint32_t* __restrict__ out_idx = a.out_idx;
  double* __restrict__ out_dist = a.out_dist;
  uint32_t* __restrict__ out_cnt = a.out_cnt;
  double* key = reinterpret_cast<double*>(smem);
  uint32_t* idx = reinterpret_cast<uint32_t*>(key + c_pow2);
  uint32_t nc = cnt[qi];
  if (nc > cmax) nc = cmax;
  uint32_t n_sort = 1;  // sort only as many slots as there are candidates (power of two, <= c_pow2)
  while (n_sort < nc) n_sort <<= 1;
  for (uint32_t i = threadIdx.x; i < n_sort; i += blockDim.x) {
    key[i] = i < nc ? cand_d[uint64_t(qi) * cmax + i] : INFINITY;
    idx[i] = i < nc ? cand_i[uint64_t(qi) * cmax + i] : 0xFFFFFFFFu;
  }
  bitonic_sort_lds(key, idx, n_sort);
  const uint32_t found = nc < k ? nc : k;
  for (uint32_t i = threadIdx.x; i < k; i += blockDim.x) {
    out_idx[uint64_t(qi) * k + i] = i < found ? idx[i] : 0xFFFFFFFFu;
    out_dist[uint64_t(qi) * k + i] = i < found ? key[i] : INFINITY;
  }
  if (threadIdx.x == 0) out_cnt[qi] = found;
}

uint32_t next_pow2(uint32_t v) {
  uint32_t p = 1;
  while (p < v) p <<= 1;
  return p;
}

rkh_status knn_plan(uint64_t n, uint32_t B, uint32_t k, KnnWorkspace* ws, size_t* bytes) {
  if (k == 0 || k > 1024) {
    set_error("k-NN: k must be in [1, 1024]");
    return RKH_ERR_BAD_ARG;
  }
  const uint32_t qb = knn_query_block(B);
  const uint32_t gy = (B + qb - 1) / qb;
  uint64_t tiles = (n + kTileRows - 1) / kTileRows;
  if (tiles < 1) tiles = 1;
  // enough subsets for a tight bound (M >= 4k) and enough blocks to fill the chip, but M <= 4096 (LDS sort)
  uint64_t gx = std::max<uint64_t>((4ull * k + kSelMax - 1) / kSelMax, 2048 / gy);
  if (gx > tiles) gx = tiles;
  if (gx > 4096 / kSelMax) gx = 4096 / kSelMax;
  ws->gx = uint32_t(gx);
  // about 4k subset minima in total are plenty for a tight bound (the k-th smallest of M minima leaves ~ -M ln(1 - k/M)
  // candidates); fewer minima = a shorter sort in knn_tau_kernel
  uint32_t ksel = uint32_t((4ull * k + gx - 1) / gx);
  if (ksel < 1) ksel = 1;
  if (ksel > uint32_t(kSelMax)) ksel = kSelMax;
  ws->ksel = ksel;
  ws->m_sub = uint32_t(gx) * ksel;
  ws->cmax = std::max<uint32_t>(2048, next_pow2(8 * k));
  if (ws->cmax > 4096) ws->cmax = 4096;
  *bytes = 256 + size_t(ws->m_sub) * B * 8 + size_t(B) * 8 + size_t(B) * 4 + size_t(B) * ws->cmax * 12 + 64;
  return RKH_OK;
}

void knn_carve(void* base, uint32_t B, KnnWorkspace* ws) {
  unsigned char* p = static_cast<unsigned char*>(base);
  ws->overflow = reinterpret_cast<uint32_t*>(p);
  p += 256;
  ws->sub = reinterpret_cast<double*>(p);
  p += size_t(ws->m_sub) * B * 8;
  ws->tau = reinterpret_cast<double*>(p);
  p += size_t(B) * 8;
  ws->cand_d = reinterpret_cast<double*>(p);
  p += size_t(B) * ws->cmax * 8;
  ws->cand_i = reinterpret_cast<uint32_t*>(p);
  p += size_t(B) * ws->cmax * 4;
  ws->cnt = reinterpret_cast<uint32_t*>(p);
}

template <int DP>
static void launch_nnk_dp(hipStream_t s, const KnnArgs& single, const KnnArgs* d_tab, uint32_t gx, uint32_t gy,
                          uint32_t b_max, uint32_t m_pow2, uint32_t cmax, uint32_t n_jobs) {
  dim3 grid(gx, gy, n_jobs), block(kThreads);
  const bool few = knn_query_block(b_max) == 8;
  if (few) hipLaunchKernelGGL((knn_sweep_kernel<DP, 8, 0>), grid, block, 0, s, single, d_tab);
  else hipLaunchKernelGGL((knn_sweep_kernel<DP, 32, 0>), grid, block, 0, s, single, d_tab);
  hipLaunchKernelGGL(knn_tau_kernel, dim3(b_max, 1, n_jobs), dim3(256), size_t(m_pow2) * 12, s, single, d_tab);
  if (few) hipLaunchKernelGGL((knn_sweep_kernel<DP, 8, 1>), grid, block, 0, s, single, d_tab);
  else hipLaunchKernelGGL((knn_sweep_kernel<DP, 32, 1>), grid, block, 0, s, single, d_tab);
  hipLaunchKernelGGL(knn_select_kernel, dim3(b_max, 1, n_jobs), dim3(256), size_t(cmax) * 12, s, single, d_tab);
}

static rkh_status launch_nnk_any(hipStream_t s, int D, const KnnArgs& single, const KnnArgs* d_tab, uint32_t gx,
                                 uint32_t gy, uint32_t b_max, uint32_t m_pow2, uint32_t cmax, uint32_t n_jobs) {
  switch (nn_padded_dims(D)) {
#define RKH_CASE(DP) \
  case DP: launch_nnk_dp<DP>(s, single, d_tab, gx, gy, b_max, m_pow2, cmax, n_jobs); break
    RKH_CASE(2);
    RKH_CASE(4);
    RKH_CASE(6);
    RKH_CASE(8);
    RKH_CASE(12);
    RKH_CASE(16);
    RKH_CASE(24);
    RKH_CASE(32);
#undef RKH_CASE
    default: set_error("k-NN: unsupported dimension"); return RKH_ERR_BAD_ARG;
  }
  RKH_HIP(hipGetLastError());
  return RKH_OK;
}

rkh_status launch_nnk(hipStream_t s, const NnStore& st, uint64_t n, const double* d_q, uint32_t B, uint32_t k,
                      double radius, uint32_t* d_idx, double* d_dist, uint32_t* d_count, const KnnWorkspace& ws) {
  if (B == 0) return RKH_OK;
  KnnArgs a;
  a.pos = st.d_pos;
  a.n = n;
  a.q = d_q;
  a.D = st.D;
  a.B = B;
  a.k = k;
  a.radius = radius;
  a.ws = ws;
  a.m_pow2 = next_pow2(ws.m_sub);
  a.out_idx = d_idx;
  a.out_dist = d_dist;
  a.out_cnt = d_count;
  return launch_nnk_any(s, st.D, a, nullptr, ws.gx, (B + knn_query_block(B) - 1) / knn_query_block(B), B, a.m_pow2, ws.cmax, 1);
}

rkh_status launch_nnk_table(hipStream_t s, int D, const KnnArgs* d_table, const KnnArgs* h_table, uint32_t n_jobs) {
  uint32_t gx = 0, b_max = 0, m_pow2 = 0, cmax = 0;
  for (uint32_t i = 0; i < n_jobs; ++i) {
    const KnnArgs& a = h_table[i];
    if (a.B == 0) continue;
    gx = std::max(gx, a.ws.gx);
    b_max = std::max(b_max, a.B);
    m_pow2 = std::max(m_pow2, a.m_pow2);
    cmax = std::max(cmax, a.ws.cmax);
  }
  if (b_max == 0) return RKH_OK;
  return launch_nnk_any(s, D, KnnArgs(), d_table, gx, (b_max + knn_query_block(b_max) - 1) / knn_query_block(b_max), b_max, m_pow2, cmax, n_jobs);
}

}  // namespace rkh
