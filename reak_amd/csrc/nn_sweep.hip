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

// The HBM-bound regime (at most 8 queries per sweep, D equal to its padded width): rows go straight from HBM into the
// registers of ONE lane each (Dp/2 16-byte loads per lane; the wave's Dp/2 loads together cover whole cache lines) --
// no LDS tile, no block barrier in the loop.  The queries are uniform across the block, so they are never held in
// vector registers: they are re-read every row through the scalar cache (constant address space => s_load) and enter
// the fp64 operations as scalar operands.  A lane keeps (best distance, index, threshold) of every query for its own
// rows; lanes and waves are merged once at the end.  Same operation sequence per (row, query) as nn1_sweep_kernel.
typedef const __attribute__((address_space(4))) double* nn_cdouble_p;
typedef double nn_d2 __attribute__((ext_vector_type(2)));
// queries per basic block of the register-direct sweep (1, 2 and 4 measured within 2 % of one another)
static constexpr int kStreamGroup = 1;

template <int DP, int QB>
__global__ __launch_bounds__(kThreads, 2) void nn1_stream_kernel(NnArgs single, const NnArgs* __restrict__ table,
                                                                  uint32_t Bpad) {
  constexpr int H = DP / 2;
  __shared__ double red_d[kThreads / 64][QB];
  __shared__ uint32_t red_i[kThreads / 64][QB];

  // everything read from the table entry is block-uniform by construction; the compiler has to be told (the entry
  // arrives through vector loads), so that the row addresses get a scalar base and the queries scalar loads
  auto uniform64 = [](uint64_t v) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane(uint32_t(v)), hi = __builtin_amdgcn_readfirstlane(uint32_t(v >> 32));
    return (uint64_t(hi) << 32) | lo;
  };
  const NnArgs a = table ? table[blockIdx.z] : single;
  typedef const __attribute__((address_space(1))) nn_d2* nn_grow_p;  // global, not flat: flat loads would share
  const uint64_t pos = uniform64(reinterpret_cast<uint64_t>(a.pos));   // the queries' scalar-load counter
  const uint64_t n = uniform64(a.d_n ? uint64_t(*a.d_n) : a.n);
  const uint32_t B = __builtin_amdgcn_readfirstlane(a.d_B ? *a.d_B : a.B);
  if (B == 0) return;
  const int tid = threadIdx.x;
  nn_cdouble_p qc =
      (nn_cdouble_p)uniform64(reinterpret_cast<uint64_t>(a.q + (a.d_qoff ? uint64_t(*a.d_qoff) : 0ull) * DP));

  const uint64_t tiles_total = (n + kTileRows - 1) / kTileRows;
  const uint64_t tiles_per_block = (tiles_total + gridDim.x - 1) / gridDim.x;
  const uint64_t tile0 = uint64_t(blockIdx.x) * tiles_per_block;
  uint64_t tile1 = tile0 + tiles_per_block;
  if (tile1 > tiles_total) tile1 = tiles_total;

  double best_d[QB], best_thr[QB];
  uint32_t best_i[QB];
#pragma unroll
  for (int k = 0; k < QB; ++k) {
    best_d[k] = INFINITY;
    best_thr[k] = INFINITY;
    best_i[k] = 0xFFFFFFFFu;
  }

  auto fetch = [&](uint64_t t, nn_d2* dst) {
    uint64_t row = t * kTileRows + tid;
    if (row >= n) row = n - 1;  // re-reads the last row; the result is discarded below
    nn_grow_p src = (nn_grow_p)(pos + row * (DP * sizeof(double)));
#pragma unroll
    for (int j = 0; j < H; ++j) dst[j] = src[j];
  };
  auto scan = [&](uint64_t t, const nn_d2* cur) {
    const uint64_t row = t * kTileRows + tid;
    const bool valid = row < n;
    asm volatile("" : "+s"(qc));  // keep the query loads inside the loop (hoisted, they would not fit the scalar file)
    // queries in groups of G: the squared distances of a group are computed in one basic block (the scalar loads of its
    // later queries can then be issued under the arithmetic of the earlier ones), then tested together
    constexpr int G = QB < kStreamGroup ? QB : kStreamGroup;
#pragma unroll
    for (int g = 0; g < QB; g += G) {
      double sq[G];
#pragma unroll
      for (int j = 0; j < G; ++j) {
        const int k = g + j;
        nn_cdouble_p qk = qc + uint32_t(uint32_t(k) < B ? k : B - 1) * DP;
        double s;
        {
          const double df = qk[0] - cur[0].x;
          s = df * df;
        }
#pragma unroll
        for (int d = 1; d < DP; ++d) {
          const double df = qk[d] - ((d & 1) ? cur[d >> 1].y : cur[d >> 1].x);
          s = s + df * df;
        }
        sq[j] = s;
      }
      bool hit = false;
#pragma unroll
      for (int j = 0; j < G; ++j) hit = hit || (sq[j] <= best_thr[g + j]);
      if (valid && hit) {  // rare after the first few rows
#pragma unroll
        for (int j = 0; j < G; ++j) {
          const int k = g + j;
          if (sq[j] <= best_thr[k]) {
            const double dd = sqrt(sq[j]);
            if (dd < best_d[k]) {
              best_d[k] = dd;
              best_i[k] = uint32_t(row);
              best_thr[k] = sq[j] * (1.0 + 4.0 * DBL_EPSILON);
            }
          }
        }
      }
    }
  };
  // Three row buffers in turn: the loads of tiles t + 1 and t + 2 are in flight while tile t is scanned (a wave then
  // keeps 2 x 64 rows on the way to it; with one tile ahead the sweep was bound by the load latency, not by HBM).  The
  // prefetches inside the loop are unconditional (past the slice they re-read its last tile): behind a branch the
  // compiler's wait-count bookkeeping has to assume the shorter queue and waits for the loads it has just issued.
  nn_d2 ra[H], rb[H], rc[H];
  if (tile0 < tile1) {
    const uint64_t last = tile1 - 1;
    auto clamp = [&](uint64_t t) { return t < last ? t : last; };
    fetch(tile0, ra);
    fetch(clamp(tile0 + 1), rb);
    uint64_t t = tile0;
    for (; t + 3 <= tile1; t += 3) {
      fetch(clamp(t + 2), rc);
      scan(t, ra);
      fetch(clamp(t + 3), ra);
      scan(t + 1, rb);
      fetch(clamp(t + 4), rb);
      scan(t + 2, rc);
    }
    if (t < tile1) {
      scan(t, ra);
    }
    if (t + 1 < tile1) {
      scan(t + 1, rb);
    }
  }

  // lanes -> wave (minimum distance first, then the lowest index among the lanes that hold it), waves -> block
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int k = 0; k < QB; ++k) {
    double bd = best_d[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) bd = fmin(bd, __shfl_xor(bd, off, 64));
    uint32_t bi = best_d[k] == bd ? best_i[k] : 0xFFFFFFFFu;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const uint32_t oi = __shfl_xor(bi, off, 64);
      bi = oi < bi ? oi : bi;
    }
    if (lane == 0) {
      red_d[wave][k] = bd;
      red_i[wave][k] = bi;
    }
  }
  __syncthreads();
  if (tid < QB && uint32_t(tid) < B) {
    double bd = red_d[0][tid];
    uint32_t bi = red_i[0][tid];
#pragma unroll
    for (int w = 1; w < kThreads / 64; ++w) {
      if (lex_less(red_d[w][tid], red_i[w][tid], bd, bi)) {
        bd = red_d[w][tid];
        bi = red_i[w][tid];
      }
    }
    a.part_dist[uint64_t(blockIdx.x) * Bpad + tid] = bd;
    a.part_idx[uint64_t(blockIdx.x) * Bpad + tid] = bi;
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
// satisfies c_{r*} <= c_min + 2 E, and a running minimum over ANY rows of the tree is never below the final one, so the
// rows with
//   c_r <= (running min of c, including r's slab) + 2 E
// are a superset of the candidates.  A half-slab (16 rows of one lane) that holds such a row is recorded as ONE entry
// (slab, 16-bit row mask, its minimum) in a short per-lane list in LDS -- straight-line code, a compare per row, no
// nested divergent loops.  After the sweep the list is cut down with the final minimum (typically to one entry, one
// bit) and those rows are evaluated with the exact fp64 sequence of nn1_sweep_kernel straight from HBM; the
// lexicographic minimum of (distance, index) over them is the reference's "first minimum wins".  Bit-identical results.
// Seeding: every recorded entry stalls its whole wave, and a block that starts from +inf meets ~ln(rows) new minima per
// lane.  A first pass (SEED = true: same tiles, same instructions, minimum only) over every s-th tile of the tree leaves
// min c of that sample in NnArgs::seed (ordered-integer atomicMin); the sweep proper starts from it -- same rows, same
// bits, so it is a running minimum in the sense above -- and records ~(rows / sample) entries per query over the whole
// tree instead of ~ln(slice) per block.
// Slabs are software-pipelined: the MFMA chain of slab g + 1 is issued before the minimum tree / compare of slab g, so
// the matrix pipe has work while the wave's VALU settles the previous slab.
// (Tried and measured slower: whole slabs as candidates -- resolving 16 rows per entry from HBM costs more than the
// per-row compares save; an exact fp64 recheck inside the sweep.)
// Error bound (u = 2^-24, M = coord_bound, M' = M (1 + u), x^, q^ = the float-rounded coordinates): the MFMA is a
// k-ordered fmaf chain starting from C, so |c - (|x^|^2 - 2 x^.q^)| <= (Dp + 1) u (|x^|^2 + 2 sum|x^ q^|) plus the
// (Dp + 1) u |x^|^2 of the float evaluation of |x^|^2 itself, <= 4 (Dp + 1) u Dp M'^2; and |sum (x^ - q^)^2 - s| <=
// 4 u M sqrt(Dp s) + 4 u^2 M^2 Dp <= 8 u Dp M^2 + 4 u^2 M^2 Dp (s <= 4 Dp M^2).  E is their sum with a factor 2.
typedef float rkh_f16v __attribute__((ext_vector_type(16)));
typedef float rkh_f4v __attribute__((ext_vector_type(4)));
static constexpr int kMfmaThreads = 256;
static constexpr int kMfmaQueries = 128;  // 4 waves x 32 queries (8 waves x 32 per block measured 25 % slower)
static constexpr int kCandCap = 8;        // per-lane entry list (compacted, then resolved exactly, when it fills)
// Seeding pays when the sample is a small fraction of the tree: the sample is kSeedSample tiles (32 Ki rows), and trees
// of fewer than kSeedMinTiles tiles (sample stride below 16, i.e. a seeding pass of more than 1/16 of the sweep's matrix
// work) are swept unseeded -- measured on the planner's trees (<= 100 k rows) a stride-8 pass cost 125 us and saved 76.
static constexpr int kSeedSample = 128;
static constexpr int kSeedMinTiles = 16 * kSeedSample;

// order-preserving map float -> uint32 (atomicMin on the image = min on the floats); 0xFFFFFFFF decodes to NaN = "none"
__device__ __forceinline__ uint32_t seed_encode(float c) {
  const uint32_t u = __float_as_uint(c);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float seed_decode(uint32_t e) {
  return __uint_as_float((e & 0x80000000u) ? (e & 0x7FFFFFFFu) : ~e);
}
__device__ __forceinline__ float min3_raw(float a, float b, float c) {
  float r;
  asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
// The same through the compiler (v_min3_f32 as well; results of matrix instructions are not re-quieted).  Kernels whose
// matrix instructions do NOT hold the vector issue for their whole length (the bf16 forms: 8 of 32 cycles) must use
// this one: the hazard recognizer counts the wait states between a matrix instruction and a VALU read of its result
// only for instructions it knows, not for inline asm -- the asm form read accumulators that were still being written.
__device__ __forceinline__ float min3_f(float a, float b, float c) { return __builtin_fminf(__builtin_fminf(a, b), c); }
// minimum over the two lane halves (lanes l and l ^ 32), one VALU swap instead of an LDS round trip
__device__ __forceinline__ float min_over_halves(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  const float lo = __uint_as_float(r[0]), hi = __uint_as_float(r[1]);
  return min3_raw(lo, hi, hi);
}

// Grid: 1-D, 8 * ceil(W / 8) blocks for W = gx * (query blocks of all problems) work items (row slice, query block,
// problem).  Hardware deals consecutive blocks round-robin to the 8 XCDs, so block L runs on XCD L % 8 as that XCD's
// (L / 8)-th block: XCD x takes the items [x Wc, (x + 1) Wc) in order, and items are numbered with the query block
// fastest -- the query blocks that sweep the same row slice run back to back on one XCD and share its L2.
template <int DP, bool SEED>
__global__ __launch_bounds__(kMfmaThreads, 4) void nn1_sweep_mfma_kernel(NnArgs single, const NnArgs* __restrict__ table,
                                                                      int D, uint32_t Bpad, double coord_bound,
                                                                      const uint32_t* __restrict__ yblock_base,
                                                                      uint32_t n_problems, uint32_t gx, uint32_t gy) {
  constexpr int H = DP / 2;
  constexpr int TS = kTileRows + 4;  // float stride of one coordinate's row of the transposed copy
  constexpr int kSlabs = kTileRows / 32;
  __shared__ __attribute__((aligned(16))) float tileT[DP * TS];
  __shared__ __attribute__((aligned(16))) float xn[kTileRows];
  __shared__ uint32_t cand_key[SEED ? 1 : kCandCap][kMfmaThreads];
  __shared__ uint32_t cand_mask[SEED ? 1 : kCandCap][kMfmaThreads];
  __shared__ float cand_m[SEED ? 1 : kCandCap][kMfmaThreads];

  uint32_t bx, by, bz;
  {
    const uint32_t L = blockIdx.x;
    const uint32_t ytot = yblock_base ? yblock_base[n_problems] : gy * n_problems;
    const uint32_t W = ytot * gx, Wc = (W + 7) >> 3;
    const uint32_t slot = L >> 3, w = (L & 7) * Wc + slot;
    if (slot >= Wc || w >= W) return;
    const uint32_t yy = w / gx;
    uint32_t p = 0, y0, cnt;
    if (yblock_base) {
      uint32_t hi_p = n_problems;  // yblock_base[p] <= yy < yblock_base[hi_p]
      while (hi_p - p > 1) {
        const uint32_t mid = (p + hi_p) >> 1;
        if (yblock_base[mid] <= yy) p = mid;
        else hi_p = mid;
      }
      y0 = yblock_base[p];
      cnt = yblock_base[p + 1] - y0;
    } else {
      p = yy / gy;
      y0 = p * gy;
      cnt = gy;
    }
    const uint32_t r = w - y0 * gx;
    bx = r / cnt;
    by = r - bx * cnt;
    bz = p;
  }
  // everything read from the table entry is block-uniform; said explicitly, it lives in scalar registers
  auto uniform64 = [](uint64_t v) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane(uint32_t(v)), hi = __builtin_amdgcn_readfirstlane(uint32_t(v >> 32));
    return (uint64_t(hi) << 32) | lo;
  };
  const NnArgs a = table ? table[bz] : single;
  const double* __restrict__ pos = reinterpret_cast<const double*>(uniform64(reinterpret_cast<uint64_t>(a.pos)));
  const uint64_t n = uniform64(a.d_n ? uint64_t(*a.d_n) : a.n);
  const uint32_t B = __builtin_amdgcn_readfirstlane(a.d_B ? *a.d_B : a.B);
  const double* __restrict__ q = reinterpret_cast<const double*>(
      uniform64(reinterpret_cast<uint64_t>(a.q + (a.d_qoff ? uint64_t(*a.d_qoff) : 0ull) * D)));
  uint32_t* __restrict__ seed = reinterpret_cast<uint32_t*>(uniform64(reinterpret_cast<uint64_t>(a.seed)));
  double* __restrict__ part_dist = reinterpret_cast<double*>(uniform64(reinterpret_cast<uint64_t>(a.part_dist)));
  uint32_t* __restrict__ part_idx = reinterpret_cast<uint32_t*>(uniform64(reinterpret_cast<uint64_t>(a.part_idx)));
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int col = lane & 31, hi = lane >> 5;
  const uint32_t qi = by * kMfmaQueries + wave * 32 + col;
  if (by * kMfmaQueries >= B) return;
  const uint32_t qsrc = qi < B ? qi : (B - 1);

  float bop[H];  // B operand of step j: -2 q^[2 j + hi]
#pragma unroll
  for (int j = 0; j < H; ++j) {
    const int d = 2 * j + hi;
    bop[j] = -2.0f * float(d < D ? q[uint64_t(qsrc) * D + d] : 0.0);
  }
  const double u32 = 5.9604644775390625e-08;  // 2^-24
  const double Mb = coord_bound * (1.0 + u32);
  const double e_one = 2.0 * (4.0 * double(DP + 1) * u32 * double(DP) * Mb * Mb + 8.0 * u32 * double(DP) * Mb * Mb +
                              4.0 * u32 * u32 * Mb * Mb * double(DP));
  // 2 E, plus the rounding of the float sum (running min + band) at the magnitude of the estimates (<= 3 Dp M'^2)
  const float band = __double2float_ru(2.0 * e_one + 8.0 * u32 * 3.0 * double(DP) * Mb * Mb);

  // the tiles of this block: its contiguous slice of the tree, or (SEED) every stride-th tile of the whole tree dealt
  // round-robin to the gx seed blocks
  const uint64_t tiles_total = (n + kTileRows - 1) / kTileRows;
  uint64_t t_first, t_step, t_count;
  if (SEED) {
    if (tiles_total < uint64_t(kSeedMinTiles)) return;
    const uint64_t stride = tiles_total / kSeedSample;
    const uint64_t ns = (tiles_total + stride - 1) / stride;
    t_first = uint64_t(bx) * stride;
    t_step = uint64_t(gx) * stride;
    t_count = ns > bx ? (ns - bx + gx - 1) / gx : 0;
  } else {
    const uint64_t tiles_per_block = (tiles_total + gx - 1) / gx;
    t_first = uint64_t(bx) * tiles_per_block;
    uint64_t t_end = t_first + tiles_per_block;
    if (t_end > tiles_total) t_end = tiles_total;
    t_step = 1;
    t_count = t_end > t_first ? t_end - t_first : 0;
  }

  double best_d = INFINITY;         // champion of the entries resolved so far (list overflow only)
  uint32_t best_i = 0xFFFFFFFFu;
  // running minimum of the estimates of this lane's query; NaN ("no seed") is dropped by fminf
  float cmin = (!SEED && seed) ? fminf(INFINITY, seed_decode(seed[qsrc])) : INFINITY;
  int cnt = 0;                      // entries in the list

  // exact fp64 distance of vertex `row` (global index): the operation sequence of nn1_sweep_kernel
  auto resolve = [&](uint32_t row) {
    if (uint64_t(row) >= n) return;  // padding rows of the last tile
    const double* p = pos + uint64_t(row) * DP;
    const double* qq = q + uint64_t(qsrc) * D;
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
  // every flagged row of entry k: key = slab index counted from the block's first tile
  auto resolve_entry = [&](int k) {
    uint32_t mask = cand_mask[k][tid];
    const uint32_t base = uint32_t(t_first) * uint32_t(kTileRows) + cand_key[k][tid] * 32u + 4u * uint32_t(hi);
#pragma unroll 1
    while (mask) {
      const uint32_t i = uint32_t(__builtin_ctz(mask));
      mask &= mask - 1;
      resolve(base + 8u * (i >> 2) + (i & 3u));
    }
  };
  // drop the entries the current minimum rules out; if the list is still full, resolve it
  auto compact = [&](float lim) {
    int w = 0;
#pragma unroll 1
    for (int k = 0; k < cnt; ++k) {
      const float mm = cand_m[k][tid];
      if (mm <= lim) {
        const uint32_t kk = cand_key[k][tid], mk = cand_mask[k][tid];
        cand_m[w][tid] = mm;
        cand_key[w][tid] = kk;
        cand_mask[w][tid] = mk;
        ++w;
      }
    }
    cnt = w;
    if (cnt == kCandCap) {
#pragma unroll 1
      for (int k = 0; k < cnt; ++k) resolve_entry(k);
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
  // a wave whose 32 query slots all lie past the batch only helps staging the tiles
  const bool sweeping = by * kMfmaQueries + wave * 32 < B;
  if (t_count > 0) fetch(t_first);
  for (uint64_t it = 0; it < t_count; ++it) {
    // the staging indices are recomputed every tile (an opaque copy of the thread id): hoisted out of the loop they
    // stay live across the slab loop and push its registers into scratch
    int tl = tid;
    asm volatile("" : "+v"(tl));
    {
#pragma unroll
      for (int j = 0; j < PF; ++j) {
        const int i = tl + j * kMfmaThreads;
        if (i >= N2) continue;
        const int row = i / H, dp = i - row * H;
        // rows past the end of the tree: a large finite float (estimate ~1e36: never a candidate)
        const bool pad = !(pf[j].x < INFINITY);
        tileT[(2 * dp) * TS + row] = pad ? 1e18f : float(pf[j].x);
        tileT[(2 * dp + 1) * TS + row] = pad ? 1e18f : float(pf[j].y);
      }
    }
    if (it + 1 < t_count) fetch(t_first + (it + 1) * t_step);
    __syncthreads();
    if (tl < kTileRows) {  // |x^|^2 of row tl, a float fmaf chain over the coordinates
      float acc = 0.0f;
#pragma unroll
      for (int d = 0; d < DP; ++d) {
        const float v = tileT[d * TS + tl];
        acc = __builtin_fmaf(v, v, acc);
      }
      xn[tl] = acc;
    }
    __syncthreads();
    if (sweeping) {
      // operands of slab g: C = |x^|^2 of the lane's 16 rows, A = the slab's column of coordinate 2 j + hi
      auto load_ops = [&](int g, float (&aop)[H], rkh_f16v& c) {
        const rkh_f4v* x4 = reinterpret_cast<const rkh_f4v*>(xn + 32 * g + 4 * hi);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const rkh_f4v v = x4[2 * k];  // rows 32 g + 8 k + 4 hi .. +3
          c[4 * k] = v.x; c[4 * k + 1] = v.y; c[4 * k + 2] = v.z; c[4 * k + 3] = v.w;
        }
#pragma unroll
        for (int j = 0; j < H; ++j) aop[j] = tileT[(2 * j + hi) * TS + 32 * g + col];
      };
      auto chain = [&](const float (&aop)[H], rkh_f16v& c) {
#pragma unroll
        for (int j = 0; j < H; ++j) c = __builtin_amdgcn_mfma_f32_32x32x2f32(aop[j], bop[j], c, 0, 0, 0);
      };
      // the estimates of slab g are complete: running minimum, and (sweep proper) one entry if a row is within the band
      auto settle = [&](int g, const rkh_f16v& c) {
        // (through the compiler, so that the wait states between a matrix instruction and the VALU reads of its result
        // are counted: see min3_f)
        float m = min3_f(min3_f(c[0], c[1], c[2]), min3_f(c[3], c[4], c[5]), min3_f(c[6], c[7], c[8]));
        m = min3_f(m, min3_f(c[9], c[10], c[11]), min3_f(c[12], c[13], c[14]));
        m = __builtin_fminf(m, c[15]);
        cmin = __builtin_fminf(cmin, m);
        if (SEED) return;
        // over both lane halves: the pair then meets a new minimum as often as ONE sequence of twice the length would
        cmin = min_over_halves(cmin);
        const float lim = cmin + band;
        if (m <= lim) {
          if (cnt == kCandCap) compact(lim);
          uint32_t mask = 0;
#pragma unroll
          for (int i = 0; i < 16; ++i) mask |= (c[i] <= lim) ? (1u << i) : 0u;
          cand_key[cnt][tid] = uint32_t(it) * uint32_t(kSlabs) + uint32_t(g);
          cand_mask[cnt][tid] = mask;
          cand_m[cnt][tid] = m;
          ++cnt;
        }
      };
      rkh_f16v c0, c1;
      float a0[H], a1[H];
      load_ops(0, a0, c0);
      chain(a0, c0);
#pragma unroll
      for (int g = 0; g < kSlabs; g += 2) {
        load_ops(g + 1, a1, c1);
        chain(a1, c1);
        settle(g, c0);
        if (g + 2 < kSlabs) {
          load_ops(g + 2, a0, c0);
          chain(a0, c0);
        }
        settle(g + 1, c1);
      }
    }
    __syncthreads();
  }
  cmin = min_over_halves(cmin);
  if (SEED) {
    if (hi == 0 && qi < B && t_count > 0) atomicMin(seed + qi, seed_encode(cmin));
    return;
  }
  // resolve what the final minimum (of both halves) leaves of the list
  {
    const float lim = cmin + band;
#pragma unroll 1
    for (int k = 0; k < cnt; ++k)
      if (cand_m[k][tid] <= lim) resolve_entry(k);
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

// ---- the pre-filter on the bf16 matrix cores ---------------------------------------------------------------------------
// The f32-input matrix instruction runs at the vector rate and does not overlap with VALU work (tests/micro/
// mfma_valu_overlap.hip); v_mfma_f32_32x32x16_bf16 has 16x its rate and holds the vector issue for 8 of its 32 cycles.
// A float does not fit a bf16, but the estimate only has to be BOUNDED: every float is split x^ = x_hi + x_lo + r,
// x_hi = bf16(x^), x_lo = bf16(x^ - x_hi), |r| <= 2^-16 |x^| (bf16 keeps 8 significant bits, round to nearest), and
//   x^.q^ ~ sum (x_hi + x_lo) q_hi + sum x_hi q_lo      (products of two bf16 are exact in the f32 accumulator)
// leaves out x_lo q_lo, (x_hi + x_lo) r_q and r_x q^: at most 3.05 * 2^-16 |x^||q^| per coordinate.  |x^|^2 (the float
// fmaf chain of the lane half) enters as three bf16 pieces against ones, i.e. exactly.  Per lane half the k-slots of a
// chain are [x_hi | x_lo | x_hi | n_hi n_mid n_lo | 0..] against [-2q_hi | -2q_hi | -2q_lo | 1 1 1 | 0..]: 3 Dp/2 + 3
// slots, 8 per instruction (3 instructions at Dp = 12 where the f32 form needs 6 of twice the cycles).  Lane
// (r, h) = (l & 31, l >> 5) holds slots 8h .. 8h+7 of row / query r in each fragment: the half-row layout of
// nn1_few_mfma_kernel.
// Error bound (M' = coord_bound (1 + 2^-24)): 2 * 3.05 * 2^-16 Dp M'^2 for the split; the accumulation of the <= 64
// exact products of a chain in the matrix pipe is not specified to the bit -- taken as 2^-23 relative to the
// magnitude 3 Dp M'^2 PER PRODUCT SLOT of four instructions (256 roundings; an IEEE f32 sum of them would stay below a
// quarter of that); the float |x^|^2 chains and the double -> float rounding of the inputs as in the f32 kernel.  E is
// their sum with a factor 2, the band 2 E plus the rounding of (running minimum + band).
#ifndef RKH_BF16_PIPELINED
#define RKH_BF16_PIPELINED 0
#endif
#ifndef RKH_BF16_CAP
#define RKH_BF16_CAP 4
#endif
static constexpr int kBf16Cap = RKH_BF16_CAP;  // entries per lane of the bf16 sweep (LDS: a fourth block per CU)
typedef __bf16 rkh_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 rkh_bf16x2 __attribute__((ext_vector_type(2)));
typedef float rkh_f2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t bf16_bits(float a) {  // round to nearest even (v_cvt_pk_bf16_f32), in the low half
  const rkh_f2v v = {a, 0.0f};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, rkh_bf16x2)) & 0xFFFFu;
}
template <int DP>
__device__ __forceinline__ float bf16_band(double coord_bound) {
  const double u32 = 5.9604644775390625e-08;  // 2^-24
  const double Mb = coord_bound * (1.0 + u32);
  const double dm2 = double(DP) * Mb * Mb;
  const double e_split = 2.0 * 3.05 * 1.52587890625e-05 * dm2;          // 2^-16
  const double e_acc = 256.0 * 1.1920928955078125e-07 * 3.0 * dm2;       // 2^-23
  const double e_norm = double(DP / 2 + 2) * u32 * dm2;
  const double e_in = 8.0 * u32 * dm2 + 4.0 * u32 * u32 * dm2;
  const double e_one = 2.0 * (e_split + e_acc + e_norm + e_in);
  return __double2float_ru(2.0 * e_one + 8.0 * u32 * 3.0 * dm2);
}
template <int H>
struct Bf16Operand {
  static constexpr int K = 3 * H + 3;       // k-slots of a lane half
  static constexpr int NI = (K + 7) / 8;    // matrix instructions per chain
  // slots -> fragments (two bf16 per register, slot 2m in the low half)
  __device__ static __forceinline__ void pack(const uint32_t (&e)[8 * NI], uint4 (&frag)[NI]) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      frag[i].x = e[8 * i + 0] | (e[8 * i + 1] << 16);
      frag[i].y = e[8 * i + 2] | (e[8 * i + 3] << 16);
      frag[i].z = e[8 * i + 4] | (e[8 * i + 5] << 16);
      frag[i].w = e[8 * i + 6] | (e[8 * i + 7] << 16);
    }
  }
  __device__ static __forceinline__ void split(float v, uint32_t& hi, uint32_t& lo) {
    hi = bf16_bits(v);
    lo = bf16_bits(v - __uint_as_float(hi << 16));  // the difference is exact
  }
  __device__ static __forceinline__ void build_a(const float (&x)[H], float nrm, uint4 (&frag)[NI]) {
    uint32_t e[8 * NI];
#pragma unroll
    for (int k = 0; k < 8 * NI; ++k) e[k] = 0u;
#pragma unroll
    for (int j = 0; j < H; ++j) {
      uint32_t hi, lo;
      split(x[j], hi, lo);
      e[j] = hi;
      e[H + j] = lo;
      e[2 * H + j] = hi;
    }
    const uint32_t n0 = bf16_bits(nrm);
    const float r1 = nrm - __uint_as_float(n0 << 16);
    const uint32_t n1 = bf16_bits(r1);
    const float r2 = r1 - __uint_as_float(n1 << 16);
    e[3 * H] = n0;
    e[3 * H + 1] = n1;
    e[3 * H + 2] = bf16_bits(r2);
    pack(e, frag);
  }
  __device__ static __forceinline__ void build_b(const float (&qm2)[H], uint4 (&frag)[NI]) {  // qm2 = -2 q^
    uint32_t e[8 * NI];
#pragma unroll
    for (int k = 0; k < 8 * NI; ++k) e[k] = 0u;
#pragma unroll
    for (int j = 0; j < H; ++j) {
      uint32_t hi, lo;
      split(qm2[j], hi, lo);
      e[j] = hi;
      e[H + j] = hi;
      e[2 * H + j] = lo;
    }
    e[3 * H] = e[3 * H + 1] = e[3 * H + 2] = 0x3F80u;  // 1.0
    pack(e, frag);
  }
};

// nn1_sweep_mfma_kernel with the estimate on the bf16 matrix cores: same grid, same work items, same entry lists and
// exact resolution; the tile in LDS holds ready-made A fragments (staged by two half rows per thread), a slab costs a
// wave NI ds_read_b128 and NI matrix instructions.  A seed left by the f32 sampled pass is a valid running minimum here
// (it is within the f32 kernel's E, which is below this kernel's, of a true value).
template <int DP>
__global__ __launch_bounds__(kMfmaThreads, 4) void nn1_sweep_bf16_kernel(NnArgs single, const NnArgs* __restrict__ table,
                                                                      int D, uint32_t Bpad, double coord_bound,
                                                                      const uint32_t* __restrict__ yblock_base,
                                                                      uint32_t n_problems, uint32_t gx, uint32_t gy) {
  constexpr int H = DP / 2;
  constexpr int NI = Bf16Operand<H>::NI;
  constexpr int kSlabs = kTileRows / 32;
  // A operands of the tile: fragment i of (row, lane half) -- consecutive rows 16 bytes apart: conflict-free ds_read_b128
  __shared__ uint4 tileA[NI][2][kTileRows];
  __shared__ uint32_t cand_key[kBf16Cap][kMfmaThreads];
  __shared__ uint32_t cand_mask[kBf16Cap][kMfmaThreads];
  __shared__ float cand_m[kBf16Cap][kMfmaThreads];

  uint32_t bx, by, bz;
  {
    const uint32_t L = blockIdx.x;
    const uint32_t ytot = yblock_base ? yblock_base[n_problems] : gy * n_problems;
    const uint32_t W = ytot * gx, Wc = (W + 7) >> 3;
    const uint32_t slot = L >> 3, w = (L & 7) * Wc + slot;
    if (slot >= Wc || w >= W) return;
    const uint32_t yy = w / gx;
    uint32_t p = 0, y0, cnt;
    if (yblock_base) {
      uint32_t hi_p = n_problems;  // yblock_base[p] <= yy < yblock_base[hi_p]
      while (hi_p - p > 1) {
        const uint32_t mid = (p + hi_p) >> 1;
        if (yblock_base[mid] <= yy) p = mid;
        else hi_p = mid;
      }
      y0 = yblock_base[p];
      cnt = yblock_base[p + 1] - y0;
    } else {
      p = yy / gy;
      y0 = p * gy;
      cnt = gy;
    }
    const uint32_t r = w - y0 * gx;
    bx = r / cnt;
    by = r - bx * cnt;
    bz = p;
  }
  // everything read from the table entry is block-uniform; said explicitly, it lives in scalar registers
  auto uniform64 = [](uint64_t v) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane(uint32_t(v)), hi = __builtin_amdgcn_readfirstlane(uint32_t(v >> 32));
    return (uint64_t(hi) << 32) | lo;
  };
  const NnArgs a = table ? table[bz] : single;
  const double* __restrict__ pos = reinterpret_cast<const double*>(uniform64(reinterpret_cast<uint64_t>(a.pos)));
  const uint64_t n = uniform64(a.d_n ? uint64_t(*a.d_n) : a.n);
  const uint32_t B = __builtin_amdgcn_readfirstlane(a.d_B ? *a.d_B : a.B);
  const double* __restrict__ q = reinterpret_cast<const double*>(
      uniform64(reinterpret_cast<uint64_t>(a.q + (a.d_qoff ? uint64_t(*a.d_qoff) : 0ull) * D)));
  uint32_t* __restrict__ seed = reinterpret_cast<uint32_t*>(uniform64(reinterpret_cast<uint64_t>(a.seed)));
  double* __restrict__ part_dist = reinterpret_cast<double*>(uniform64(reinterpret_cast<uint64_t>(a.part_dist)));
  uint32_t* __restrict__ part_idx = reinterpret_cast<uint32_t*>(uniform64(reinterpret_cast<uint64_t>(a.part_idx)));
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int col = lane & 31, hi = lane >> 5;
  const uint32_t qi = by * kMfmaQueries + wave * 32 + col;
  if (by * kMfmaQueries >= B) return;
  const uint32_t qsrc = qi < B ? qi : (B - 1);

  uint4 bop[NI];  // B fragments of this lane's query: -2 q^ of the lane half's coordinates (hi, hi, lo), then ones
  {
    float qf[H];
#pragma unroll
    for (int j = 0; j < H; ++j) {
      const int d = H * hi + j;
      qf[j] = -2.0f * float(q[uint64_t(qsrc) * D + (d < D ? d : D - 1)]);
      qf[j] = d < D ? qf[j] : 0.0f;
    }
    Bf16Operand<H>::build_b(qf, bop);
  }
  const float band = bf16_band<DP>(coord_bound);

  // the tiles of this block: its contiguous slice of the tree
  const uint64_t tiles_total = (n + kTileRows - 1) / kTileRows;
  const uint64_t tiles_per_block = (tiles_total + gx - 1) / gx;
  const uint64_t t_first = uint64_t(bx) * tiles_per_block, t_step = 1;
  uint64_t t_end = t_first + tiles_per_block;
  if (t_end > tiles_total) t_end = tiles_total;
  const uint64_t t_count = t_end > t_first ? t_end - t_first : 0;

  double best_d = INFINITY;         // champion of the entries resolved so far (list overflow only)
  uint32_t best_i = 0xFFFFFFFFu;
  // running minimum of the estimates of this lane's query; NaN ("no seed") is dropped by fminf
  float cmin = seed ? fminf(INFINITY, seed_decode(seed[qsrc])) : INFINITY;
  int cnt = 0;                      // entries in the list

  // exact fp64 distance of vertex `row` (global index): the operation sequence of nn1_sweep_kernel
  auto resolve = [&](uint32_t row) {
    if (uint64_t(row) >= n) return;  // padding rows of the last tile
    const double* p = pos + uint64_t(row) * DP;
    const double* qq = q + uint64_t(qsrc) * D;
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
  // every flagged row of entry k: key = slab index counted from the block's first tile
  auto resolve_entry = [&](int k) {
    uint32_t mask = cand_mask[k][tid];
    const uint32_t base = uint32_t(t_first) * uint32_t(kTileRows) + cand_key[k][tid] * 32u + 4u * uint32_t(hi);
#pragma unroll 1
    while (mask) {
      const uint32_t i = uint32_t(__builtin_ctz(mask));
      mask &= mask - 1;
      resolve(base + 8u * (i >> 2) + (i & 3u));
    }
  };
  // drop the entries the current minimum rules out; if the list is still full, resolve it
  auto compact = [&](float lim) {
    int w = 0;
#pragma unroll 1
    for (int k = 0; k < cnt; ++k) {
      const float mm = cand_m[k][tid];
      if (mm <= lim) {
        const uint32_t kk = cand_key[k][tid], mk = cand_mask[k][tid];
        cand_m[w][tid] = mm;
        cand_key[w][tid] = kk;
        cand_mask[w][tid] = mk;
        ++w;
      }
    }
    cnt = w;
    if (cnt == kBf16Cap) {
#pragma unroll 1
      for (int k = 0; k < cnt; ++k) resolve_entry(k);
      cnt = 0;
    }
  };

  // a thread stages two half rows per tile: pairs p = tid and tid + 256, pair p = (row p >> 1, half p & 1)
  constexpr int PP = 2 * kTileRows / kMfmaThreads;
  double pf[PP][H];
  auto fetch = [&](uint64_t t) {
    const uint64_t row_base = t * kTileRows;
#pragma unroll
    for (int j = 0; j < PP; ++j) {
      const int pr = tid + j * kMfmaThreads;
      const uint64_t row = row_base + uint32_t(pr >> 1);
      const bool ok = row < n;
      const double* src = pos + (ok ? row : n - 1) * DP + H * (pr & 1);
#pragma unroll
      for (int d = 0; d < H; ++d) pf[j][d] = ok ? src[d] : INFINITY;
    }
  };
  // a wave whose 32 query slots all lie past the batch only helps staging the tiles
  const bool sweeping = by * kMfmaQueries + wave * 32 < B;
  if (t_count > 0) fetch(t_first);
  for (uint64_t it = 0; it < t_count; ++it) {
    // the staging indices are recomputed every tile (an opaque copy of the thread id): hoisted out of the loop they
    // stay live across the slab loop and push its registers into scratch
    int tl = tid;
    asm volatile("" : "+v"(tl));
    {
#pragma unroll
      for (int j = 0; j < PP; ++j) {
        const int pr = tl + j * kMfmaThreads;
        float x[H];
        float nrm = 0.0f;
#pragma unroll
        for (int d = 0; d < H; ++d) {
          // rows past the end of the tree and removed vertices: a large finite float (estimate ~1e37: never a candidate)
          x[d] = pf[j][d] < INFINITY ? float(pf[j][d]) : 1e18f;
          nrm = __builtin_fmaf(x[d], x[d], nrm);
        }
        uint4 frag[NI];
        Bf16Operand<H>::build_a(x, nrm, frag);
#pragma unroll
        for (int i = 0; i < NI; ++i) tileA[i][pr & 1][pr >> 1] = frag[i];
      }
    }
    if (it + 1 < t_count) fetch(t_first + (it + 1) * t_step);
    __syncthreads();
    if (sweeping) {
      // operands of slab g: the fragments of (row 32 g + col, this lane half); the chain starts from zero
      auto load_ops = [&](int g, uint4 (&aop)[NI], rkh_f16v& c) {
#pragma unroll
        for (int i = 0; i < NI; ++i) aop[i] = tileA[i][hi][32 * g + col];
#pragma unroll
        for (int k = 0; k < 16; ++k) c[k] = 0.0f;
      };
      auto chain = [&](const uint4 (&aop)[NI], rkh_f16v& c) {
#pragma unroll
        for (int i = 0; i < NI; ++i)
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(rkh_bf16x8, aop[i]),
                                                      __builtin_bit_cast(rkh_bf16x8, bop[i]), c, 0, 0, 0);
      };
      // the estimates of slab g are complete: running minimum, and (sweep proper) one entry if a row is within the band
      auto settle = [&](int g, const rkh_f16v& c) {
        float m = min3_f(min3_f(c[0], c[1], c[2]), min3_f(c[3], c[4], c[5]), min3_f(c[6], c[7], c[8]));
        m = min3_f(m, min3_f(c[9], c[10], c[11]), min3_f(c[12], c[13], c[14]));
        m = __builtin_fminf(m, c[15]);
        cmin = __builtin_fminf(cmin, m);
        // over both lane halves: the pair then meets a new minimum as often as ONE sequence of twice the length would
        cmin = min_over_halves(cmin);
        const float lim = cmin + band;
        if (m <= lim) {
          if (cnt == kBf16Cap) compact(lim);
          uint32_t mask = 0;
#pragma unroll
          for (int i = 0; i < 16; ++i) mask |= (c[i] <= lim) ? (1u << i) : 0u;
          cand_key[cnt][tid] = uint32_t(it) * uint32_t(kSlabs) + uint32_t(g);
          cand_mask[cnt][tid] = mask;
          cand_m[cnt][tid] = m;
          ++cnt;
        }
      };
#if RKH_BF16_PIPELINED
      rkh_f16v c0, c1;
      uint4 a0[NI], a1[NI];
      load_ops(0, a0, c0);
      chain(a0, c0);
#pragma unroll
      for (int g = 0; g < kSlabs; g += 2) {
        load_ops(g + 1, a1, c1);
        chain(a1, c1);
        settle(g, c0);
        if (g + 2 < kSlabs) {
          load_ops(g + 2, a0, c0);
          chain(a0, c0);
        }
        settle(g + 1, c1);
      }
#else
      // one accumulator: the bf16 matrix instructions overlap with the VALU work of the SIMD's other waves, and the
      // registers of a second accumulator are worth a fourth wave per SIMD
#pragma unroll
      for (int g = 0; g < kSlabs; ++g) {
        rkh_f16v c0;
        uint4 a0[NI];
        load_ops(g, a0, c0);
        chain(a0, c0);
        settle(g, c0);
      }
#endif
    }
    __syncthreads();
  }
  cmin = min_over_halves(cmin);
  // resolve what the final minimum (of both halves) leaves of the list
  {
    const float lim = cmin + band;
#pragma unroll 1
    for (int k = 0; k < cnt; ++k)
      if (cand_m[k][tid] <= lim) resolve_entry(k);
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

// The matrix-core pre-filter for FEW queries per sweep (at most 32) over a LARGE tree with known coordinate bounds: the
// HBM-bound regime.  The exact fp64 test costs 3 Dp - 1 VALU operations per (row, query) -- at 8 queries that is 3
// operations per byte of the tree, and the register-direct fp64 sweep above stops at ~4.5 TB/s -- whereas the estimate
// |x|^2 - 2 x.q of 32 rows x 32 query slots costs a wave Dp/2 + 1 matrix instructions and ~25 VALU instructions, so
// the sweep runs at the speed the rows arrive.  No LDS tile and no block barrier in the loop: a wave streams its own
// 32-row slabs, lane (r, h) = (l & 31, l >> 5) loads HALF of row r (coordinates [h Dp/2, (h + 1) Dp/2): Dp/4 16-byte
// loads; the wave's loads together cover the slab's cache lines exactly once) and these registers ARE the A operands:
// the k-index of instruction j is the lane half, so instruction j multiplies coordinate j (lanes 0-31) and Dp/2 + j
// (lanes 32-63) with B = -2 q^ of the same coordinates.  The order in which the coordinates enter the sum only matters
// to the rounding, which the band covers.  |x^|^2 enters through one more instruction (A = the lane's half of the sum of
// squares, B = 1), so no lane ever needs another lane's registers.  Slabs are dealt round-robin to all waves of the grid
// (at any time the grid reads one moving window of the tree), three slabs of a wave are in flight.
// Candidates, band and the exact fp64 resolution are those of nn1_sweep_mfma_kernel (per-lane entry list in LDS; the
// operation chain here has Dp + 2 fused steps and two Dp/2-step partial sums: 4 (Dp + 2) u Dp M'^2 bounds it).
// Query slots past the batch start from a running minimum of -inf and never record anything.
// The four waves of a block share their running minimum through one LDS word per query (ordered-integer ds_min when a
// wave's own minimum improves, read back every slab): a running minimum over ANY rows of the tree keeps the candidate
// argument intact, fewer entries are recorded, and after the sweep -- one barrier -- only the entries within the band of
// the BLOCK's minimum are resolved (all coordinates of a row loaded at once: one memory round trip, the waves' tails
// would otherwise be chains of five).  (Tried: the same word shared by the whole grid in HBM -- every wave polling one
// cache line made the sweep five times slower.)  Bit-identical results.
static constexpr int kFewQueries = 32;
static constexpr int kFewThreads = 256;
static constexpr int kFewDepth = 3;  // row buffers per wave (4 measured the same: the memory system is the limit)
template <int DP>
__global__ __launch_bounds__(kFewThreads, 4) void nn1_few_mfma_kernel(NnArgs single, const NnArgs* __restrict__ table,
                                                                      int D, uint32_t Bpad, double coord_bound) {
  constexpr int H = DP / 2;
  constexpr int PD = kFewDepth;
  constexpr int kWaves = kFewThreads / 64;
  __shared__ uint32_t cand_key[kCandCap][kFewThreads];
  __shared__ uint32_t cand_mask[kCandCap][kFewThreads];
  __shared__ float cand_m[kCandCap][kFewThreads];
  __shared__ double red_d[kWaves][kFewQueries];
  __shared__ uint32_t red_i[kWaves][kFewQueries];
  __shared__ uint32_t blk_min[kFewQueries];  // seed_encode of the block's running minimum per query slot

  auto uniform64 = [](uint64_t v) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane(uint32_t(v)), hi = __builtin_amdgcn_readfirstlane(uint32_t(v >> 32));
    return (uint64_t(hi) << 32) | lo;
  };
  const NnArgs a = table ? table[blockIdx.z] : single;
  typedef const __attribute__((address_space(1))) double* nn_gdouble_p;
  typedef const __attribute__((address_space(1))) nn_d2* nn_grow_p;
  const uint64_t pos = uniform64(reinterpret_cast<uint64_t>(a.pos));
  const uint64_t n = uniform64(a.d_n ? uint64_t(*a.d_n) : a.n);
  const uint32_t B = __builtin_amdgcn_readfirstlane(a.d_B ? *a.d_B : a.B);
  const double* __restrict__ q = reinterpret_cast<const double*>(
      uniform64(reinterpret_cast<uint64_t>(a.q + (a.d_qoff ? uint64_t(*a.d_qoff) : 0ull) * D)));
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int col = lane & 31, hi = lane >> 5;
  const bool live = uint32_t(col) < B;  // this lane's query slot holds a query
  const uint32_t qsrc = live ? uint32_t(col) : 0u;
  if (tid < kFewQueries) blk_min[tid] = 0xFFFFFFFFu;  // "none"

  float bop[H];  // B operand of instruction j: -2 q^[h Dp/2 + j] (filled below, behind the first row loads)
  const double u32 = 5.9604644775390625e-08;  // 2^-24
  const double Mb = coord_bound * (1.0 + u32);
  const double e_one = 2.0 * (4.0 * double(DP + 2) * u32 * double(DP) * Mb * Mb + 8.0 * u32 * double(DP) * Mb * Mb +
                              4.0 * u32 * u32 * Mb * Mb * double(DP));
  const float band = __double2float_ru(2.0 * e_one + 8.0 * u32 * 3.0 * double(DP) * Mb * Mb);

  // slab `it` of this wave is slab it * step + first of the tree
  const uint32_t GW = gridDim.x * kWaves, gw = blockIdx.x * kWaves + wave;
  const uint64_t slabs_total = (n + 31) / 32;
  const uint64_t first = gw, step = GW;
  const uint64_t my_count = slabs_total > gw ? (slabs_total - gw + GW - 1) / GW : 0;
  __syncthreads();  // blk_min is initialised

  double best_d = INFINITY;
  uint32_t best_i = 0xFFFFFFFFu;
  float cmin = live ? INFINITY : -INFINITY;
  int cnt = 0;

  // exact fp64 distance of a row: the operation sequence of nn1_sweep_kernel.  AT_ONCE: every coordinate is loaded
  // before the first use (one round trip; the end of the sweep), otherwise a few per round trip (inside the sweep the
  // registers belong to the slabs in flight)
  auto resolve = [&](uint64_t row, auto at_once) {
    if (row >= n) return;
    nn_gdouble_p p = (nn_gdouble_p)(pos + row * (DP * sizeof(double)));
    const double* qq = q + uint64_t(qsrc) * D;
    double s;
    if constexpr (decltype(at_once)::value) {
      double pv[DP], qv[DP];
#pragma unroll
      for (int d = 0; d < DP; ++d) {
        pv[d] = p[d];
        qv[d] = d < D ? qq[d] : 0.0;
      }
      {
        const double df = qv[0] - pv[0];
        s = df * df;
      }
#pragma unroll
      for (int d = 1; d < DP; ++d) {
        const double df = qv[d] - pv[d];
        s = s + df * df;
      }
    } else {
      {
        const double df = qq[0] - p[0];
        s = df * df;
      }
#pragma unroll 3
      for (int d = 1; d < DP; ++d) {
        const double df = (d < D ? qq[d] : 0.0) - p[d];
        s = s + df * df;
      }
    }
    const double dd = sqrt(s);
    if (lex_less(dd, uint32_t(row), best_d, best_i)) {
      best_d = dd;
      best_i = uint32_t(row);
    }
  };
  auto resolve_entry = [&](int k, auto at_once) {
    uint32_t mask = cand_mask[k][tid];
    const uint64_t base = (uint64_t(cand_key[k][tid]) * step + first) * 32u + 4u * uint32_t(hi);
#pragma unroll 1
    while (mask) {
      const uint32_t i = uint32_t(__builtin_ctz(mask));
      mask &= mask - 1;
      resolve(base + 8u * (i >> 2) + (i & 3u), at_once);
    }
  };
  auto compact = [&](float lim) {
    int w = 0;
#pragma unroll 1
    for (int k = 0; k < cnt; ++k) {
      const float mm = cand_m[k][tid];
      if (mm <= lim) {
        const uint32_t kk = cand_key[k][tid], mk = cand_mask[k][tid];
        cand_m[w][tid] = mm;
        cand_key[w][tid] = kk;
        cand_mask[w][tid] = mk;
        ++w;
      }
    }
    cnt = w;
    if (cnt == kCandCap) {
#pragma unroll 1
      for (int k = 0; k < cnt; ++k) resolve_entry(k, std::false_type{});
      cnt = 0;
    }
  };

  // half a row per lane; slabs past the wave's last one re-read it (the prefetches are unconditional, see
  // nn1_stream_kernel), rows past the tree re-read the last row and are replaced below
  auto fetch = [&](uint64_t it, double (&buf)[H]) {
    const uint64_t itc = it < my_count ? it : my_count - 1;
    uint64_t row = (itc * step + first) * 32u + uint32_t(col);
    if (row >= n) row = n - 1;
    const uint64_t addr = pos + (row * DP + uint32_t(H * hi)) * sizeof(double);
    if constexpr (H % 2 == 0) {
      nn_grow_p src = (nn_grow_p)addr;
#pragma unroll
      for (int j = 0; j < H / 2; ++j) {
        const nn_d2 v = src[j];
        buf[2 * j] = v.x;
        buf[2 * j + 1] = v.y;
      }
    } else {
      nn_gdouble_p src = (nn_gdouble_p)addr;
#pragma unroll
      for (int j = 0; j < H; ++j) buf[j] = src[j];
    }
  };
  auto process = [&](uint64_t it, const double (&buf)[H]) {
    const bool ok = (it * step + first) * 32u + uint32_t(col) < n;
    const float blk = seed_decode(blk_min[col]);  // "none" decodes to NaN
    float x[H];
    float nrm = 0.0f;
#pragma unroll
    for (int j = 0; j < H; ++j) {
      // rows past the tree and removed vertices (rows of +inf): a large finite float, estimates ~1e37, never candidates
      x[j] = (ok && buf[j] < INFINITY) ? float(buf[j]) : 1e18f;
      nrm = __builtin_fmaf(x[j], x[j], nrm);
    }
    rkh_f16v c = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(nrm, 1.0f, c, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < H; ++j) c = __builtin_amdgcn_mfma_f32_32x32x2f32(x[j], bop[j], c, 0, 0, 0);
    float m = min3_f(min3_f(c[0], c[1], c[2]), min3_f(c[3], c[4], c[5]), min3_f(c[6], c[7], c[8]));
    m = min3_f(m, min3_f(c[9], c[10], c[11]), min3_f(c[12], c[13], c[14]));
    m = __builtin_fminf(m, c[15]);
    const float seen = live ? fminf(cmin, blk) : cmin;  // what was known before this slab (fminf drops the NaN)
    cmin = min3_raw(seen, m, m);
    cmin = min_over_halves(cmin);
    const float lim = cmin + band;
    if (m <= lim) {
      if (live && hi == 0 && cmin < seen) atomicMin(&blk_min[col], seed_encode(cmin));
      if (cnt == kCandCap) compact(lim);
      uint32_t mask = 0;
#pragma unroll
      for (int i = 0; i < 16; ++i) mask |= (c[i] <= lim) ? (1u << i) : 0u;
      cand_key[cnt][tid] = uint32_t(it);
      cand_mask[cnt][tid] = mask;
      cand_m[cnt][tid] = m;
      ++cnt;
    }
  };

  if (my_count > 0 && B > 0) {
    // PD row buffers in turn: PD - 1 slabs are on their way while one is processed
    double r[PD][H];
#pragma unroll
    for (int k = 0; k < PD - 1; ++k) fetch(k, r[k]);
    {  // all query coordinates in one round trip (unconditional loads at a clamped coordinate, then the select)
      nn_gdouble_p qg = (nn_gdouble_p)(reinterpret_cast<uint64_t>(q) + uint64_t(qsrc) * D * sizeof(double));
      double qv[H];
#pragma unroll
      for (int j = 0; j < H; ++j) {
        const int d = H * hi + j;
        qv[j] = qg[d < D ? d : D - 1];
      }
#pragma unroll
      for (int j = 0; j < H; ++j) bop[j] = (H * hi + j < D) ? -2.0f * float(qv[j]) : 0.0f;
    }
    uint64_t it = 0;
    for (; it + PD <= my_count; it += PD) {
#pragma unroll
      for (int k = 0; k < PD; ++k) {
        fetch(it + k + PD - 1, r[(k + PD - 1) % PD]);
        process(it + k, r[k]);
      }
    }
#pragma unroll
    for (int k = 0; k < PD - 1; ++k)
      if (it + k < my_count) process(it + k, r[k]);
    if (live && hi == 0) atomicMin(&blk_min[col], seed_encode(cmin));
  }
  __syncthreads();
  if (live && B > 0) {  // resolve what the block's minimum leaves of the list
    const float lim = fminf(cmin, seed_decode(blk_min[col])) + band;
#pragma unroll 1
    for (int k = 0; k < cnt; ++k)
      if (cand_m[k][tid] <= lim) resolve_entry(k, std::true_type{});
  }
  {  // the two halves of the wave hold different rows of the same 32 query slots
    const double od = __shfl_xor(best_d, 32, 64);
    const uint32_t oi = __shfl_xor(best_i, 32, 64);
    if (lex_less(od, oi, best_d, best_i)) {
      best_d = od;
      best_i = oi;
    }
  }
  if (hi == 0) {
    red_d[wave][col] = best_d;
    red_i[wave][col] = best_i;
  }
  __syncthreads();
  if (tid < kFewQueries && uint32_t(tid) < B) {
    double bd = red_d[0][tid];
    uint32_t bi = red_i[0][tid];
#pragma unroll
    for (int w = 1; w < kWaves; ++w) {
      if (lex_less(red_d[w][tid], red_i[w][tid], bd, bi)) {
        bd = red_d[w][tid];
        bi = red_i[w][tid];
      }
    }
    a.part_dist[uint64_t(blockIdx.x) * Bpad + tid] = bd;
    a.part_idx[uint64_t(blockIdx.x) * Bpad + tid] = bi;
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
    if (a.seed) a.seed[qi] = 0xFFFFFFFFu;  // consumed: "no seed" again for the next sweep
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

// The split-bf16 band is ~100x the f32 one: in very few dimensions, where the nearest neighbours of a dense cloud sit
// close together, it lets more rows through to the exact test than the cheaper instructions save (unit cube, 1 Mi x 1024
// queries, tests/diag_nn_lowdim.py: 3-D 159 against 154 us, 65 536 x 384: 27 against 23; 4-D 30 against 33; 6-D 123
// against 199) -- from 6 padded dimensions on the bf16 form is used.
static int bf16_min_dims() {  // RKH_NN_BF16_MIN_DIMS overrides (diagnostics)
  static const int v = [] {
    const char* e = getenv("RKH_NN_BF16_MIN_DIMS");
    return e ? atoi(e) : 6;
  }();
  return v;
}
// RKH_NN_BF16=0 keeps the f32-input matrix instructions in the many-queries sweep (diagnostics: tests/prof_nn_variants.sh)
static bool bf16_enabled() {
  static const bool on = [] {
    const char* e = getenv("RKH_NN_BF16");
    return !(e && e[0] == '0');
  }();
  return on;
}

static uint32_t pick_qb(uint32_t B, bool bounded = false) {
  // queries per block: the smallest padded query count wins (a block computes all its QB slots); ties go to the larger
  // block (fewer re-reads of the tiles).  With a coordinate bound 33..64 queries already take the matrix-core block of
  // 128 (half of it padding, and still three times faster than the packed-fp32 pre-filter of a 64-query block: 51 against
  // 147 us at 1 Mi rows); the row-slice count does not depend on this choice (one query block either way).
  if (B <= 8) return 8;
  if (B <= 16) return 16;
  if (B <= 32) return 32;
  if (B <= 64 && !(bounded && mfma_enabled() && bf16_enabled())) return 64;
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

// the register-direct sweep: exactly one resident set of blocks (`resident` = what the occupancy of the instantiation
// allows on the whole device; a partial second round of blocks would run at a fraction of the machine); a thread's
// fixed costs (the merge of the lanes at the end) are worth at least 4 rows
static constexpr uint32_t kStreamBlocksMax = 2048;  // 8 blocks of 256 threads on each of 256 CUs
static uint32_t pick_gx_stream(uint64_t n_upper, uint32_t n_problems, uint32_t resident) {
  uint64_t tiles = (n_upper + kTileRows - 1) / kTileRows;
  if (tiles < 1) tiles = 1;
  if (resident > kStreamBlocksMax) resident = kStreamBlocksMax;
  uint64_t want = resident / (n_problems ? n_problems : 1);
  if (want < 1) want = 1;
  const uint64_t most = tiles >= 4 ? tiles / 4 : 1;
  if (want > most) want = most;
  return uint32_t(want);
}
template <int DP, int QB>
static uint32_t stream_resident_blocks() {
  static const uint32_t v = [] {
    int per_cu = 0, cus = 0, dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 1024u;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) return 1024u;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, nn1_stream_kernel<DP, QB>, kThreads, 0) != hipSuccess ||
        per_cu <= 0)
      return 1024u;
    return uint32_t(per_cu) * uint32_t(cus);
  }();
  return v;
}

// the few-queries matrix-core sweep: one resident set of blocks, like the register-direct sweep
template <int DP>
static uint32_t few_resident_blocks() {
  static const uint32_t v = [] {
    int per_cu = 0, cus = 0, dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 1024u;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) return 1024u;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, nn1_few_mfma_kernel<DP>, kFewThreads, 0) != hipSuccess ||
        per_cu <= 0)
      return 1024u;
    return uint32_t(per_cu) * uint32_t(cus);
  }();
  return v;
}
// Which few-queries sweeps take the matrix-core pre-filter (measured on 12-dimensional trees of 16 Ki .. 4 Mi rows,
// tests/diag_nn_few.sh): from 5 queries on it beats the fp64 sweeps at every size (8 queries: 5.2 against 4.4 TB/s at
// 4 Mi rows, 32 queries: 5.0 against 1.4); up to 4 queries the register-direct fp64 sweep is as fast or faster (5.8 TB/s
// at one query) and stays, where it applies (D equal to its padded width).  Trees of a few tiles keep the tiled sweeps.
static constexpr uint64_t kFewMinRows = 8192;
static bool few_applies(uint32_t B, int D, int DP, uint64_t n_upper, double coord_bound) {
  if (!(coord_bound > 0.0) || B > uint32_t(kFewQueries) || DP > 16 || n_upper < kFewMinRows || !mfma_enabled()) return false;
  return B > 4 || D != DP;
}

uint32_t nn1_partial_blocks(uint64_t n_upper, uint32_t B, uint32_t n_problems) {
  const uint32_t qb = pick_qb(B);
  const uint32_t gy = (B + qb - 1) / qb;
  const uint32_t np = n_problems ? n_problems : 1;
  const uint32_t a = pick_gx(n_upper, gy * np);
  const uint32_t b = B <= uint32_t(kFewQueries) ? pick_gx_stream(n_upper, np, kStreamBlocksMax) : 0;
  return a > b ? a : b;
}

uint32_t nn1_mfma_queries() { return uint32_t(kMfmaQueries); }

static const char* g_last_kernel = "";
const char* nn_last_kernel_name() { return g_last_kernel; }
void nn_set_last_kernel_name(const char* name) { g_last_kernel = name; }

template <int DP>
static rkh_status launch_nn1_dp(hipStream_t s, int D, const NnArgs& single, const NnArgs* d_table, uint32_t n_problems,
                                uint64_t n_upper, uint32_t B, uint32_t part_capacity_blocks, hipEvent_t ev0,
                                hipEvent_t ev1, double coord_bound, const uint32_t* d_yblock_base, bool table_has_seed) {
  const uint32_t qb = pick_qb(B, coord_bound > 0.0 && DP >= bf16_min_dims() && DP <= 16);
  const uint32_t gy = (B + qb - 1) / qb;
  // few queries over a large tree with known coordinate bounds: matrix-core pre-filter at the speed of HBM
  const bool few = few_applies(B, D, DP, n_upper, coord_bound);
  const bool stream = !few && B <= 8 && D == DP;  // HBM-bound regime: rows straight into registers
  uint32_t gx = pick_gx(n_upper, gy * n_problems);
  if (few) {
    if constexpr (DP <= 16) gx = pick_gx_stream(n_upper, n_problems, few_resident_blocks<DP>());
  } else if (stream) {
    const uint32_t resident = B <= 1 ? stream_resident_blocks<DP, 1>()
                                     : (B <= 2 ? stream_resident_blocks<DP, 2>()
                                               : (B <= 4 ? stream_resident_blocks<DP, 4>() : stream_resident_blocks<DP, 8>()));
    gx = pick_gx_stream(n_upper, n_problems, resident);
  }
  if (gx > part_capacity_blocks) gx = part_capacity_blocks;
  const uint32_t Bpad = B;
  dim3 grid(gx, gy, n_problems), block(kThreads);
#define RKH_NN1_LAUNCH(QB) hipLaunchKernelGGL((nn1_sweep_kernel<DP, QB>), grid, block, 0, s, single, d_table, D, Bpad)
#define RKH_NN1_LAUNCH_F32(QB) \
  hipLaunchKernelGGL((nn1_sweep_f32_kernel<DP, QB>), grid, block, 0, s, single, d_table, D, Bpad, coord_bound)
  if (ev0) (void)hipEventRecord(ev0, s);
  const bool f32 = coord_bound > 0.0 && qb >= 32;  // compute-bound regime with known coordinate bounds
  const bool mfma = f32 && qb == kMfmaQueries && mfma_enabled() && DP <= 16;
  g_last_kernel = few ? "nn1_few_mfma_kernel"
                      : (stream ? "nn1_stream_kernel"
                                : (mfma ? "nn1_sweep_mfma_kernel" : (f32 ? "nn1_sweep_f32_kernel" : "nn1_sweep_kernel")));
  if (few) {
    if constexpr (DP <= 16)
      hipLaunchKernelGGL((nn1_few_mfma_kernel<DP>), dim3(gx, 1, n_problems), dim3(kFewThreads), 0, s, single, d_table, D,
                         Bpad, coord_bound);
  } else if (stream) {
#define RKH_STREAM(QB) hipLaunchKernelGGL((nn1_stream_kernel<DP, QB>), grid, block, 0, s, single, d_table, Bpad)
    if (B <= 1) { RKH_STREAM(1); }
    else if (B <= 2) { RKH_STREAM(2); }
    else if (B <= 4) { RKH_STREAM(4); }
    else { RKH_STREAM(8); }
#undef RKH_STREAM
  } else if (mfma) {
    if constexpr (DP <= 16) {
      const uint32_t* yb = d_table ? d_yblock_base : nullptr;
      auto blocks_for = [&](uint32_t slices) { return dim3((slices * gy * n_problems + 7) / 8 * 8); };
      // the sampled-minimum pass (trees of at least kSeedMinTiles tiles; smaller ones leave "no seed" behind)
      const uint64_t tiles_upper = (n_upper + kTileRows - 1) / kTileRows;
      const bool seeded = (d_table ? table_has_seed : single.seed != nullptr) && tiles_upper >= uint64_t(kSeedMinTiles);
      if (seeded) {  // kSeedSample sample tiles, four per seed block
        const uint32_t gxs = kSeedSample / 4;
        hipLaunchKernelGGL((nn1_sweep_mfma_kernel<DP, true>), blocks_for(gxs), dim3(kMfmaThreads), 0, s, single, d_table, D,
                           Bpad, coord_bound, yb, n_problems, gxs, gy);
      }
      if (bf16_enabled() && DP >= bf16_min_dims()) {
        g_last_kernel = "nn1_sweep_bf16_kernel";
        hipLaunchKernelGGL((nn1_sweep_bf16_kernel<DP>), blocks_for(gx), dim3(kMfmaThreads), 0, s, single, d_table, D, Bpad,
                           coord_bound, yb, n_problems, gx, gy);
      } else
        hipLaunchKernelGGL((nn1_sweep_mfma_kernel<DP, false>), blocks_for(gx), dim3(kMfmaThreads), 0, s, single, d_table, D,
                           Bpad, coord_bound, yb, n_problems, gx, gy);
    }
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
                      double coord_bound, const uint32_t* d_yblock_base, bool table_has_seed) {
  if (B == 0 || n_problems == 0) return RKH_OK;
  // the error analysis of the pre-filters is relative to the bound: it assumes that neither the float products nor the
  // bf16 pieces leave the normal range (and that 1e18, the stand-in for rows that must never qualify, is far outside
  // the cloud).  Clouds scaled beyond that are swept by the exact kernels.
  if (!(coord_bound >= 1e-6 && coord_bound <= 1e6)) coord_bound = 0.0;
  switch (padded_dims(D)) {
#define RKH_CASE(DP) \
  case DP: return launch_nn1_dp<DP>(s, D, single, d_table, n_problems, n_upper, B, part_capacity_blocks, ev0, ev1, coord_bound, \
                                    d_yblock_base, table_has_seed)
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
