// nn_sweep.hip -- exact batched nearest-neighbour sweeps over the growing tree (gfx950).
//
// Replaces min_dist_linear_search (ctrl/path_planning/topological_search.hpp:95-118 1-NN,
// :244-274 k-NN) with euclidean_distance_metric (ctrl/topologies/vect_distance_metrics.hpp:113-150).
//
// Layout: vertex positions are row-major [n][Dp] fp64 in HBM (Dp = D rounded up to a supported
// width, pad = 0.0 which leaves the left-to-right sum of squares bit-identical).  A 256-thread block
// streams its slice of the rows linearly through a 256-row LDS tile (contiguous 16-B/lane global
// loads, no transpose); thread (q, r) keeps query q in registers and walks rows r, r+R, r+2R ...
// of the tile through LDS broadcast reads, so the inner loop has no cross-lane traffic at all.
// Bit-exact argmin: squares are compared first; sqrt (correctly rounded) is taken only for
// candidates within 4 ulp of the running best square, and ties resolve to the lower vertex index
// ("first minimum wins", strict '<' in the reference).
//
// Roofline: one sweep reads n*Dp*8 bytes once; per (row, query) it issues 3*Dp-1 fp64 VALU ops, so
// with <= ~8 queries per sweep the kernel is HBM-bound, above ~16 it is fp64-VALU-bound.
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>
#include <cstdlib>

#include "rkh_internal.h"

namespace rkh {

static constexpr int kTileRows = 256;
static constexpr int kThreads = 256;

__device__ __forceinline__ bool lex_less(double da, uint32_t ia, double db, uint32_t ib) {
  return (da < db) || (da == db && ia < ib);
}

// One block: rows [row0, row1) x queries [blockIdx.y*QB, +QB).
// blockIdx.z selects the problem when a table of NnArgs is given (one launch sweeps many independent trees).
template <int DP, int QB>
__global__ __launch_bounds__(kThreads) void nn1_sweep_kernel(NnArgs single, const NnArgs* __restrict__ table, int D,
                                                              uint32_t Bpad) {
  constexpr int R = kThreads / QB;          // row sub-ranges per block
  constexpr int ROWS_PER_THREAD = kTileRows / R;
  __shared__ __attribute__((aligned(16))) double tile[kTileRows * DP];
  __shared__ double red_d[kThreads];
  __shared__ uint32_t red_i[kThreads];

  const NnArgs a = table ? table[blockIdx.z] : single;
  const double* __restrict__ pos = a.pos;
  const double* __restrict__ q = a.q;
  const uint32_t* __restrict__ d_qoff = a.d_qoff;
  double* __restrict__ part_dist = a.part_dist;
  uint32_t* __restrict__ part_idx = a.part_idx;
  const uint64_t n = a.d_n ? uint64_t(*a.d_n) : a.n;
  const uint32_t B = a.d_B ? *a.d_B : a.B;
  const int tid = threadIdx.x;
  const int ql = tid % QB;
  const int r = tid / QB;
  const uint32_t qi = blockIdx.y * QB + ql;
  if (blockIdx.y * QB >= B) return;  // whole block has no query (B read on device)

  // query -> registers (padded with zeros)
  double qv[DP];
  {
    const uint64_t qsrc = uint64_t(qi < B ? qi : (B - 1)) + (d_qoff ? uint64_t(*d_qoff) : 0ull);
#pragma unroll
    for (int d = 0; d < DP; ++d) qv[d] = d < D ? q[qsrc * D + d] : 0.0;
  }

  // balanced contiguous slice of rows for this block, in whole tiles
  const uint64_t tiles_total = (n + kTileRows - 1) / kTileRows;
  const uint64_t tiles_per_block = (tiles_total + gridDim.x - 1) / gridDim.x;
  const uint64_t tile0 = uint64_t(blockIdx.x) * tiles_per_block;
  uint64_t tile1 = tile0 + tiles_per_block;
  if (tile1 > tiles_total) tile1 = tiles_total;

  double best_d = INFINITY;     // sqrt of best_s
  double best_thr = INFINITY;   // squares above this cannot tie or beat best_d
  uint32_t best_i = 0xFFFFFFFFu;

  // Software pipeline: the rows of tile t+1 are fetched into registers (16 B/lane, contiguous) while tile t is
  // being scanned out of LDS, so every CU keeps HBM requests in flight during its compute phase.
  constexpr int N2 = kTileRows * DP / 2;          // double2 per tile
  constexpr int PF = N2 / kThreads;               // double2 per thread per tile
  static_assert(N2 % kThreads == 0, "tile must split evenly");
  double2 pf[PF];
  auto fetch = [&](uint64_t t) {
    const uint64_t row_base = t * kTileRows;
    const double2* src = reinterpret_cast<const double2*>(pos + row_base * DP);
    const uint64_t valid2 = (n - row_base >= uint64_t(kTileRows)) ? uint64_t(N2) : (n - row_base) * DP / 2;
#pragma unroll
    for (int j = 0; j < PF; ++j) {
      const int i = tid + j * kThreads;
      // rows beyond n become +inf so they never win
      pf[j] = (uint64_t(i) < valid2) ? src[i] : make_double2(INFINITY, INFINITY);
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
      double s;
      {
        double df = qv[0] - p[0];
        s = df * df;
      }
#pragma unroll
      for (int d = 1; d < DP; ++d) {
        double df = qv[d] - p[d];
        s = s + df * df;
      }
      if (s <= best_thr) {  // rare after the first few rows
        const double dd = sqrt(s);
        if (dd < best_d) {
          best_d = dd;
          best_i = uint32_t(row_base + row);
          best_thr = s * (1.0 + 4.0 * DBL_EPSILON);
        }
      }
    }
    __syncthreads();
  }

  // combine the R sub-ranges of each query (lexicographic (dist, index) = first minimum wins)
  red_d[tid] = best_d;
  red_i[tid] = best_i;
  __syncthreads();
  if (tid < QB) {
    double bd = red_d[tid];
    uint32_t bi = red_i[tid];
#pragma unroll
    for (int rr = 1; rr < R; ++rr) {
      const double od = red_d[rr * QB + tid];
      const uint32_t oi = red_i[rr * QB + tid];
      if (lex_less(od, oi, bd, bi)) {
        bd = od;
        bi = oi;
      }
    }
    if (qi < B) {
      part_dist[uint64_t(blockIdx.x) * Bpad + qi] = bd;
      part_idx[uint64_t(blockIdx.x) * Bpad + qi] = bi;
    }
  }
}

// The same sweep with a single-precision pre-filter, for the compute-bound regime (many queries per sweep, as in the
// planner's speculative rounds).  Per (row, query) the exact test costs 3*Dp-1 fp64 VALU operations; here a packed fp32
// estimate of the squared distance (Dp/2 v_pk_add_f32 + Dp/2 v_pk_fma_f32 on a float copy of the tile) rejects every
// row that provably cannot tie or beat the thread's running best, and only the survivors -- a handful per thread -- are
// evaluated with the exact fp64 sequence above, so the result is bit-identical.
// Error bound of the estimate (u = 2^-24, M = coord_bound >= every |coordinate|): each float difference is off by at
// most u (2M + 2|d|), hence |s32 - s| <= 4 u M sqrt(Dp s) + (4 + Dp) u s.  A row is skipped only if
//   s32 > best_thr (1 + 2 (4 + Dp) u) + 8 u M sqrt(Dp) best_d   (rounded up to float)
// which implies s > best_thr, i.e. the row could not have changed (best_d, best_i).
typedef float rkh_f2 __attribute__((ext_vector_type(2)));

template <int DP, int QB>
__global__ __launch_bounds__(kThreads) void nn1_sweep_f32_kernel(NnArgs single, const NnArgs* __restrict__ table, int D,
                                                                  uint32_t Bpad, double coord_bound) {
  constexpr int R = kThreads / QB;
  constexpr int ROWS_PER_THREAD = kTileRows / R;
  constexpr int H = DP / 2;
  __shared__ __attribute__((aligned(16))) double tile[kTileRows * DP];
  __shared__ __attribute__((aligned(16))) rkh_f2 tile32[kTileRows * H];
  __shared__ double red_d[kThreads];
  __shared__ uint32_t red_i[kThreads];

  const NnArgs a = table ? table[blockIdx.z] : single;
  const double* __restrict__ pos = a.pos;
  const double* __restrict__ q = a.q;
  const uint32_t* __restrict__ d_qoff = a.d_qoff;
  double* __restrict__ part_dist = a.part_dist;
  uint32_t* __restrict__ part_idx = a.part_idx;
  const uint64_t n = a.d_n ? uint64_t(*a.d_n) : a.n;
  const uint32_t B = a.d_B ? *a.d_B : a.B;
  const int tid = threadIdx.x;
  const int ql = tid % QB;
  const int r = tid / QB;
  const uint32_t qi = blockIdx.y * QB + ql;
  if (blockIdx.y * QB >= B) return;

  double qv[DP];
  rkh_f2 q2[H];
  {
    const uint64_t qsrc = uint64_t(qi < B ? qi : (B - 1)) + (d_qoff ? uint64_t(*d_qoff) : 0ull);
#pragma unroll
    for (int d = 0; d < DP; ++d) qv[d] = d < D ? q[qsrc * D + d] : 0.0;
#pragma unroll
    for (int j = 0; j < H; ++j) q2[j] = rkh_f2{float(qv[2 * j]), float(qv[2 * j + 1])};
  }
  const double u32 = 5.9604644775390625e-08;  // 2^-24
  const double slack_rel = 1.0 + 2.0 * double(4 + DP) * u32;
  const double slack_abs = 8.0 * u32 * coord_bound * sqrt(double(DP));

  const uint64_t tiles_total = (n + kTileRows - 1) / kTileRows;
  const uint64_t tiles_per_block = (tiles_total + gridDim.x - 1) / gridDim.x;
  const uint64_t tile0 = uint64_t(blockIdx.x) * tiles_per_block;
  uint64_t tile1 = tile0 + tiles_per_block;
  if (tile1 > tiles_total) tile1 = tiles_total;

  double best_d = INFINITY;
  double best_thr = INFINITY;
  float filt = INFINITY;       // skip rows whose float estimate exceeds this
  uint32_t best_i = 0xFFFFFFFFu;

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
      pf[j] = (uint64_t(i) < valid2) ? src[i] : make_double2(INFINITY, INFINITY);
    }
  };
  if (tile0 < tile1) fetch(tile0);
  for (uint64_t t = tile0; t < tile1; ++t) {
    const uint64_t row_base = t * kTileRows;
    {
      double2* dst = reinterpret_cast<double2*>(tile);
#pragma unroll
      for (int j = 0; j < PF; ++j) {
        dst[tid + j * kThreads] = pf[j];
        tile32[tid + j * kThreads] = rkh_f2{float(pf[j].x), float(pf[j].y)};  // the double2 index is the float2 index
      }
    }
    if (t + 1 < tile1) fetch(t + 1);
    __syncthreads();
#pragma unroll 4
    for (int k = 0; k < ROWS_PER_THREAD; ++k) {
      const int row = k * R + r;
      const rkh_f2* p2 = tile32 + row * H;
      rkh_f2 acc = rkh_f2{0.0f, 0.0f};
#pragma unroll
      for (int j = 0; j < H; ++j) {
        const rkh_f2 df = q2[j] - p2[j];
        acc = __builtin_elementwise_fma(df, df, acc);
      }
      const float s32 = acc.x + acc.y;
      if (!(s32 > filt)) {  // survivors only: the exact fp64 sequence of nn1_sweep_kernel
        const double* p = tile + row * DP;
        double s;
        {
          double df = qv[0] - p[0];
          s = df * df;
        }
#pragma unroll
        for (int d = 1; d < DP; ++d) {
          double df = qv[d] - p[d];
          s = s + df * df;
        }
        if (s <= best_thr) {
          const double dd = sqrt(s);
          if (dd < best_d) {
            best_d = dd;
            best_i = uint32_t(row_base + row);
            best_thr = s * (1.0 + 4.0 * DBL_EPSILON);
            filt = __double2float_ru(best_thr * slack_rel + slack_abs * best_d);
          }
        }
      }
    }
    __syncthreads();
  }

  red_d[tid] = best_d;
  red_i[tid] = best_i;
  __syncthreads();
  if (tid < QB) {
    double bd = red_d[tid];
    uint32_t bi = red_i[tid];
#pragma unroll
    for (int rr = 1; rr < R; ++rr) {
      const double od = red_d[rr * QB + tid];
      const uint32_t oi = red_i[rr * QB + tid];
      if (lex_less(od, oi, bd, bi)) {
        bd = od;
        bi = oi;
      }
    }
    if (qi < B) {
      part_dist[uint64_t(blockIdx.x) * Bpad + qi] = bd;
      part_idx[uint64_t(blockIdx.x) * Bpad + qi] = bi;
    }
  }
}

// The pre-filter on the matrix cores, for large query batches (more than 64 queries per sweep and tree).  The squared
// distance is expanded, s = |x|^2 - 2 x.q + |q|^2, so that the (rows x queries) block of estimates is a rank-Dp product:
// v_mfma_f32_32x32x2_f32 with A = a 32-row slab of the float tile, B = -2 q for the wave's 32 queries and C = |x|^2 gives
// c = |x|^2 - 2 x.q for 32 x 32 (row, query) pairs in Dp/2 instructions, at the packed-fp32 VALU rate but on the matrix
// pipe, which leaves the VALU with one min-tree and one compare per 16 estimates.  A lane holds ONE query (column
// l & 31) and 16 rows per slab (the C/D map of the 32x32 shapes: row = (reg & 3) + 8 (reg >> 2) + 4 (l >> 5)); the two
// lane halves hold different rows of the same queries and are merged at the end.
// No exact arithmetic inside the sweep: with E >= |c_r - (s_r - |q^|^2)| for every row r, the true nearest row r*
// satisfies c_{r*} <= c_min + 2 E, and the running minimum is never below the final one, so the rows with
//   c_r <= (running min of c, including r's slab) + 2 E
// are a superset of the candidates; they are appended, (row, c), to a short per-lane list in LDS, a handful per sweep.
// After the sweep the list is cut down with the final minimum (typically to one or two rows) and those rows are
// evaluated with the exact fp64 sequence of nn1_sweep_kernel straight from HBM; the lexicographic minimum of
// (distance, index) over them is the reference's "first minimum wins".  Bit-identical results.
// (Keeping whole slabs as candidates -- one compare per slab -- was measured: resolving a slab's 16 rows from HBM at
// the end costs far more than the per-row checks it saves.)
// Error bound (u = 2^-24, M = coord_bound, M' = M (1 + u), x^, q^ = the float-rounded coordinates): the MFMA is a
// k-ordered fmaf chain starting from C, so |c - (|x^|^2 - 2 x^.q^)| <= (Dp + 1) u (|x^|^2 + 2 sum|x^ q^|) plus the
// (Dp + 1) u |x^|^2 of the float evaluation of |x^|^2 itself, <= 4 (Dp + 1) u Dp M'^2; and |sum (x^ - q^)^2 - s| <=
// 4 u M sqrt(Dp s) + 4 u^2 M^2 Dp <= 8 u Dp M^2 + 4 u^2 M^2 Dp (s <= 4 Dp M^2).  E is their sum with a factor 2.
typedef float rkh_f16v __attribute__((ext_vector_type(16)));
typedef float rkh_f4v __attribute__((ext_vector_type(4)));
static constexpr int kMfmaThreads = 256;
static constexpr int kMfmaQueries = 128;  // 4 waves x 32 queries (8 waves x 32 per block measured 25 % slower)
static constexpr int kCandCap = 8;  // per-lane candidate list (compacted, then resolved exactly, when it fills)

template <int DP>
__global__ __launch_bounds__(kMfmaThreads, 4) void nn1_sweep_mfma_kernel(NnArgs single, const NnArgs* __restrict__ table,
                                                                      int D, uint32_t Bpad, double coord_bound,
                                                                      const uint32_t* __restrict__ yblock_base,
                                                                      uint32_t n_problems) {
  constexpr int H = DP / 2;
  constexpr int TS = kTileRows + 4;  // float stride of one coordinate's row of the transposed copy
  __shared__ __attribute__((aligned(16))) float tileT[DP * TS];
  __shared__ __attribute__((aligned(16))) float xn[kTileRows];
  __shared__ uint32_t cand_row[kCandCap][kMfmaThreads];
  __shared__ float cand_c[kCandCap][kMfmaThreads];

  // (row slice, query block, problem) of this block.  With a prefix of the query blocks per problem (yblock_base, written
  // by the planner's round_begin_kernel) the grid's blocks, in dispatch order, take the working (slice, query block)
  // pairs one after the other: no holes for problems with fewer queries than the grid was sized for, and an even spread
  // over the XCDs.
  uint32_t bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (yblock_base) {
    const uint32_t L = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const uint32_t yy = L / gridDim.x;
    if (yy >= yblock_base[n_problems]) return;
    uint32_t lo = 0, hi = n_problems;  // yblock_base[lo] <= yy < yblock_base[hi]
    while (hi - lo > 1) {
      const uint32_t mid = (lo + hi) >> 1;
      if (yblock_base[mid] <= yy) lo = mid;
      else hi = mid;
    }
    bx = L - yy * gridDim.x;
    by = yy - yblock_base[lo];
    bz = lo;
  }
  const NnArgs a = table ? table[bz] : single;
  const double* __restrict__ pos = a.pos;
  const double* __restrict__ q = a.q;
  const uint32_t* __restrict__ d_qoff = a.d_qoff;
  double* __restrict__ part_dist = a.part_dist;
  uint32_t* __restrict__ part_idx = a.part_idx;
  const uint64_t n = a.d_n ? uint64_t(*a.d_n) : a.n;
  const uint32_t B = a.d_B ? *a.d_B : a.B;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int col = lane & 31, hi = lane >> 5;
  const uint32_t qi = by * kMfmaQueries + wave * 32 + col;
  if (by * kMfmaQueries >= B) return;
  const uint64_t qsrc = uint64_t(qi < B ? qi : (B - 1)) + (d_qoff ? uint64_t(*d_qoff) : 0ull);

  float bop[H];  // B operand of step j: -2 q^[2 j + hi]
#pragma unroll
  for (int j = 0; j < H; ++j) {
    const int d = 2 * j + hi;
    bop[j] = -2.0f * float(d < D ? q[qsrc * D + d] : 0.0);
  }
  const double u32 = 5.9604644775390625e-08;  // 2^-24
  const double Mb = coord_bound * (1.0 + u32);
  const double e_one = 2.0 * (4.0 * double(DP + 1) * u32 * double(DP) * Mb * Mb + 8.0 * u32 * double(DP) * Mb * Mb +
                              4.0 * u32 * u32 * Mb * Mb * double(DP));
  // 2 E, plus the rounding of the float sum (running min + band) at the magnitude of the estimates (<= 3 Dp M'^2)
  const float band = __double2float_ru(2.0 * e_one + 8.0 * u32 * 3.0 * double(DP) * Mb * Mb);

  const uint64_t tiles_total = (n + kTileRows - 1) / kTileRows;
  const uint64_t tiles_per_block = (tiles_total + gridDim.x - 1) / gridDim.x;
  const uint64_t tile0 = uint64_t(bx) * tiles_per_block;
  uint64_t tile1 = tile0 + tiles_per_block;
  if (tile1 > tiles_total) tile1 = tiles_total;

  double best_d = INFINITY;         // champion of the candidates resolved so far (list overflow only)
  uint32_t best_i = 0xFFFFFFFFu;
  float cmin = INFINITY;            // running minimum of the estimates of this lane's query
  int cnt = 0;                      // candidates in the list

  // exact fp64 distance of vertex `row` (global index): the operation sequence of nn1_sweep_kernel
  auto resolve = [&](uint32_t row) {
    if (uint64_t(row) >= n) return;  // padding rows of the last tile (a half-wave that has seen nothing else)
    const double* p = pos + uint64_t(row) * DP;
    const double* qq = q + qsrc * D;
    double s;
    {
      const double df = qq[0] - p[0];
      s = df * df;
    }
#pragma unroll 3
    for (int d = 1; d < DP; ++d) {  // (a few coordinates per memory round trip; fully unrolled it spills into the sweep)
      const double df = (d < D ? qq[d] : 0.0) - p[d];
      s = s + df * df;
    }
    const double dd = sqrt(s);
    if (lex_less(dd, row, best_d, best_i)) {
      best_d = dd;
      best_i = row;
    }
  };
  // drop the candidates the current minimum rules out; if the list is still full, resolve it
  auto compact = [&]() {
    const float lim = cmin + band;
    int w = 0;
#pragma unroll 1
    for (int k = 0; k < cnt; ++k) {
      const float cc = cand_c[k][tid];
      const uint32_t rr = cand_row[k][tid];
      if (cc <= lim) {
        cand_c[w][tid] = cc;
        cand_row[w][tid] = rr;
        ++w;
      }
    }
    cnt = w;
    if (cnt == kCandCap) {
#pragma unroll 1
      for (int k = 0; k < cnt; ++k) resolve(cand_row[k][tid]);
      cnt = 0;
    }
  };

  constexpr int N2 = kTileRows * DP / 2;
  constexpr int PF = (N2 + kMfmaThreads - 1) / kMfmaThreads;
  double2 pf[PF];
  auto fetch = [&](uint64_t t) {
    const uint64_t row_base = t * kTileRows;
    const double2* src = reinterpret_cast<const double2*>(pos + row_base * DP);
    const uint64_t valid2 = (n - row_base >= uint64_t(kTileRows)) ? uint64_t(N2) : (n - row_base) * DP / 2;
#pragma unroll
    for (int j = 0; j < PF; ++j) {
      const int i = tid + j * kMfmaThreads;
      pf[j] = (uint64_t(i) < valid2 && i < N2) ? src[i] : make_double2(INFINITY, INFINITY);
    }
  };
  if (tile0 < tile1) fetch(tile0);
  for (uint64_t t = tile0; t < tile1; ++t) {
    const uint32_t row_base = uint32_t(t * kTileRows);
    {
#pragma unroll
      for (int j = 0; j < PF; ++j) {
        const int i = tid + j * kMfmaThreads;
        if (i >= N2) continue;
        const int row = i / H, dp = i - row * H;
        // rows past the end of the tree: a large finite float (estimate ~1e36: never a candidate)
        const bool pad = !(pf[j].x < INFINITY);
        tileT[(2 * dp) * TS + row] = pad ? 1e18f : float(pf[j].x);
        tileT[(2 * dp + 1) * TS + row] = pad ? 1e18f : float(pf[j].y);
      }
    }
    if (t + 1 < tile1) fetch(t + 1);
    __syncthreads();
    if (tid < kTileRows) {  // |x^|^2 of row tid, a float fmaf chain over the coordinates
      float acc = 0.0f;
#pragma unroll
      for (int d = 0; d < DP; ++d) {
        const float v = tileT[d * TS + tid];
        acc = __builtin_fmaf(v, v, acc);
      }
      xn[tid] = acc;
    }
    __syncthreads();
    cmin = fminf(cmin, __shfl_xor(cmin, 32, 64));  // the other half's minimum bounds the final one just as well
    // a wave whose 32 query slots all lie past the batch only helps staging the tiles
    const int n_slabs = (by * kMfmaQueries + wave * 32 < B) ? kTileRows / 32 : 0;
#pragma unroll 2
    for (int g = 0; g < n_slabs; ++g) {
      rkh_f16v c;
      {
        const rkh_f4v* x4 = reinterpret_cast<const rkh_f4v*>(xn + 32 * g + 4 * hi);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const rkh_f4v v = x4[2 * k];  // rows 32 g + 8 k + 4 hi .. +3
          c[4 * k] = v.x; c[4 * k + 1] = v.y; c[4 * k + 2] = v.z; c[4 * k + 3] = v.w;
        }
      }
#pragma unroll
      for (int j = 0; j < H; ++j) {
        const float aop = tileT[(2 * j + hi) * TS + 32 * g + col];
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(aop, bop[j], c, 0, 0, 0);
      }
      // minima of the four 4-row groups, then of the slab
      float gm[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) gm[k] = fminf(fminf(c[4 * k], c[4 * k + 1]), fminf(c[4 * k + 2], c[4 * k + 3]));
      const float m = fminf(fminf(gm[0], gm[1]), fminf(gm[2], gm[3]));
      // running minimum of the query, over both lane halves: the pair then meets a new minimum as often as ONE sequence
      // of twice the length would (ln 2 more often), not twice as often -- and every such event stalls the whole wave
      cmin = fminf(cmin, m);
      cmin = fminf(cmin, __shfl_xor(cmin, 32, 64));
      const float lim = cmin + band;
      if (m <= lim) {  // a new minimum, or a row within the band of the old one
        if (cnt > kCandCap - 3) compact();
        bool overflow = false;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (gm[k] <= lim) {  // usually one group, one row
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int i = 4 * k + j;
              if (c[i] <= lim) {
                if (cnt < kCandCap) {
                  cand_c[cnt][tid] = c[i];
                  cand_row[cnt][tid] = row_base + uint32_t(32 * g + 8 * (i >> 2) + 4 * hi + (i & 3));
                  ++cnt;
                } else {
                  overflow = true;
                }
              }
            }
          }
        }
        // more candidates in one slab than the list takes (many coincident vertices): settle this slab exactly
        if (overflow) {
#pragma unroll 1
          for (int i = 0; i < 16; ++i) resolve(row_base + uint32_t(32 * g + 8 * (i >> 2) + 4 * hi + (i & 3)));
        }
      }
    }
    __syncthreads();
  }
  // resolve what the final minimum (of both halves) leaves of the list
  cmin = fminf(cmin, __shfl_xor(cmin, 32, 64));
  {
    const float lim = cmin + band;
#pragma unroll 1
    for (int k = 0; k < cnt; ++k)
      if (cand_c[k][tid] <= lim) resolve(cand_row[k][tid]);
  }
  {  // the two halves of the wave hold different rows of the same 32 queries
    const double od = __shfl_xor(best_d, 32, 64);
    const uint32_t oi = __shfl_xor(best_i, 32, 64);
    if (lex_less(od, oi, best_d, best_i)) {
      best_d = od;
      best_i = oi;
    }
  }
  if (hi == 0 && qi < B) {
    part_dist[uint64_t(bx) * Bpad + qi] = best_d;
    part_idx[uint64_t(bx) * Bpad + qi] = best_i;
  }
}

// one wave per query: lanes stride over the per-block partials, then a shuffle reduction
__global__ __launch_bounds__(256) void nn1_reduce_kernel(NnArgs single, const NnArgs* __restrict__ table,
                                                          uint32_t nblocks, uint32_t Bpad) {
  const NnArgs a = table ? table[blockIdx.y] : single;
  const double* __restrict__ part_dist = a.part_dist;
  const uint32_t* __restrict__ part_idx = a.part_idx;
  uint32_t* __restrict__ idx = a.idx;
  double* __restrict__ dist = a.dist;
  const uint32_t B = a.d_B ? *a.d_B : a.B;
  const uint32_t qi = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (qi >= B) return;
  double bd = INFINITY;
  uint32_t bi = 0xFFFFFFFFu;
  for (uint32_t b = lane; b < nblocks; b += 64) {
    const double od = part_dist[uint64_t(b) * Bpad + qi];
    const uint32_t oi = part_idx[uint64_t(b) * Bpad + qi];
    if (lex_less(od, oi, bd, bi)) {
      bd = od;
      bi = oi;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const double od = __shfl_xor(bd, off, 64);
    const uint32_t oi = __shfl_xor(bi, off, 64);
    if (lex_less(od, oi, bd, bi)) {
      bd = od;
      bi = oi;
    }
  }
  if (lane == 0) {
    idx[qi] = bi;
    dist[qi] = bd;
  }
}

static int padded_dims(int D) {
  static const int sizes[] = {2, 4, 6, 8, 12, 16, 24, 32};
  for (int s : sizes)
    if (D <= s) return s;
  return -1;
}
int nn_padded_dims(int D) { return padded_dims(D); }

// matrix-core pre-filter for batches of more than 64 queries (RKH_NN_MFMA=0 keeps the packed-fp32 VALU pre-filter)
static bool mfma_enabled() {
  static const bool on = [] {
    const char* e = getenv("RKH_NN_MFMA");
    return !(e && e[0] == '0');
  }();
  return on;
}

static uint32_t pick_qb(uint32_t B) {
  // queries per block: the smallest padded query count wins (a block computes all its QB slots); ties go to the larger
  // block (fewer re-reads of the tiles)
  if (B <= 8) return 8;
  if (B <= 16) return 16;
  if (B <= 32) return 32;
  if (B <= 64) return 64;
  if (mfma_enabled()) return kMfmaQueries;
  const uint32_t pad128 = (B + 127) / 128 * 128, pad256 = (B + 255) / 256 * 256;
  return pad128 < pad256 ? 128 : 256;
}

static uint32_t pick_gx(uint64_t n_upper, uint32_t gy) {
  uint64_t tiles = (n_upper + kTileRows - 1) / kTileRows;
  if (tiles < 1) tiles = 1;
  static const long forced = [] {  // diagnostic override of the row-slice count
    const char* e = getenv("RKH_NN_BLOCKS");
    return e ? atol(e) : 0L;
  }();
  uint64_t want = forced > 0 ? uint64_t(forced) : 4096 / gy;  // ~16 blocks per CU over the whole grid
  if (want < 1) want = 1;
  // a block's fixed costs (query setup, resolving its candidates, one partial per query) are worth at least 4 tiles
  const uint64_t most = tiles >= 4 ? tiles / 4 : 1;
  if (want > most) want = most;
  return uint32_t(tiles < want ? tiles : want);
}

uint32_t nn1_partial_blocks(uint64_t n_upper, uint32_t B, uint32_t n_problems) {
  const uint32_t qb = pick_qb(B);
  const uint32_t gy = (B + qb - 1) / qb;
  return pick_gx(n_upper, gy * (n_problems ? n_problems : 1));
}

uint32_t nn1_mfma_queries() { return uint32_t(kMfmaQueries); }

static const char* g_last_kernel = "";
const char* nn_last_kernel_name() { return g_last_kernel; }

template <int DP>
static rkh_status launch_nn1_dp(hipStream_t s, int D, const NnArgs& single, const NnArgs* d_table, uint32_t n_problems,
                                uint64_t n_upper, uint32_t B, uint32_t part_capacity_blocks, hipEvent_t ev0,
                                hipEvent_t ev1, double coord_bound, const uint32_t* d_yblock_base) {
  const uint32_t qb = pick_qb(B);
  const uint32_t gy = (B + qb - 1) / qb;
  uint32_t gx = pick_gx(n_upper, gy * n_problems);
  if (gx > part_capacity_blocks) gx = part_capacity_blocks;
  const uint32_t Bpad = B;
  dim3 grid(gx, gy, n_problems), block(kThreads);
#define RKH_NN1_LAUNCH(QB) hipLaunchKernelGGL((nn1_sweep_kernel<DP, QB>), grid, block, 0, s, single, d_table, D, Bpad)
#define RKH_NN1_LAUNCH_F32(QB) \
  hipLaunchKernelGGL((nn1_sweep_f32_kernel<DP, QB>), grid, block, 0, s, single, d_table, D, Bpad, coord_bound)
  if (ev0) (void)hipEventRecord(ev0, s);
  const bool f32 = coord_bound > 0.0 && qb >= 32;  // compute-bound regime with known coordinate bounds
  const bool mfma = f32 && qb == kMfmaQueries && mfma_enabled() && DP <= 16;
  g_last_kernel = mfma ? "nn1_sweep_mfma_kernel" : (f32 ? "nn1_sweep_f32_kernel" : "nn1_sweep_kernel");
  if (mfma) {
    if constexpr (DP <= 16)
      hipLaunchKernelGGL((nn1_sweep_mfma_kernel<DP>), grid, dim3(kMfmaThreads), 0, s, single, d_table, D, Bpad, coord_bound,
                         d_table ? d_yblock_base : nullptr, n_problems);
  } else
  switch (qb) {
    case 8: RKH_NN1_LAUNCH(8); break;
    case 16: RKH_NN1_LAUNCH(16); break;
    case 32: if (f32) RKH_NN1_LAUNCH_F32(32); else RKH_NN1_LAUNCH(32); break;
    case 64: if (f32) RKH_NN1_LAUNCH_F32(64); else RKH_NN1_LAUNCH(64); break;
    case 128: if (f32) RKH_NN1_LAUNCH_F32(128); else RKH_NN1_LAUNCH(128); break;
    default: if (f32) RKH_NN1_LAUNCH_F32(256); else RKH_NN1_LAUNCH(256); break;
  }
#undef RKH_NN1_LAUNCH
#undef RKH_NN1_LAUNCH_F32
  if (ev1) (void)hipEventRecord(ev1, s);
  hipLaunchKernelGGL(nn1_reduce_kernel, dim3((B + 3) / 4, n_problems), dim3(256), 0, s, single, d_table, gx, Bpad);
  RKH_HIP(hipGetLastError());
  return RKH_OK;
}

// 1-NN of up to B queries per problem.  single: one problem given by value; d_table: n_problems NnArgs in HBM.
// n_upper (host bound on the vertex count) and B (host bound on the query count) only size the grid.
rkh_status launch_nn1(hipStream_t s, int D, const NnArgs& single, const NnArgs* d_table, uint32_t n_problems,
                      uint64_t n_upper, uint32_t B, uint32_t part_capacity_blocks, hipEvent_t ev0, hipEvent_t ev1,
                      double coord_bound, const uint32_t* d_yblock_base) {
  if (B == 0 || n_problems == 0) return RKH_OK;
  switch (padded_dims(D)) {
#define RKH_CASE(DP) \
  case DP: return launch_nn1_dp<DP>(s, D, single, d_table, n_problems, n_upper, B, part_capacity_blocks, ev0, ev1, coord_bound, \
                                    d_yblock_base)
    RKH_CASE(2);
    RKH_CASE(4);
    RKH_CASE(6);
    RKH_CASE(8);
    RKH_CASE(12);
    RKH_CASE(16);
    RKH_CASE(24);
    RKH_CASE(32);
#undef RKH_CASE
  }
  set_error("nn: unsupported dimension");
  return RKH_ERR_BAD_ARG;
}

// ---- synthetic fill: uniform points in the unit hypercube, splitmix64 per element ------------------
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
__global__ void fill_uniform_kernel(double* pos, uint64_t n, int D, int DP, uint64_t seed) {
  const uint64_t total = n * uint64_t(DP);
  for (uint64_t i = blockIdx.x * uint64_t(blockDim.x) + threadIdx.x; i < total; i += uint64_t(gridDim.x) * blockDim.x) {
    const int d = int(i % DP);
    const uint64_t row = i / DP;
    double v = 0.0;
    if (d < D) v = double(splitmix64(seed ^ (row * 64 + d)) >> 11) * (1.0 / 9007199254740992.0);
    pos[i] = v;
  }
}
rkh_status launch_fill_uniform(hipStream_t s, const NnStore& st, uint64_t n, uint64_t seed) {
  const int DP = padded_dims(st.D);
  hipLaunchKernelGGL(fill_uniform_kernel, dim3(2048), dim3(256), 0, s, st.d_pos, n, st.D, DP, seed);
  RKH_HIP(hipGetLastError());
  return RKH_OK;
}

}  // namespace rkh
