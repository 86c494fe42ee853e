// nn_mirror.hip -- the planner-regime 1-NN sweep over the half-precision mirror of the trees (nn_mirror.h).
//
// Replaces min_dist_linear_search (R/ctrl/path_planning/topological_search.hpp:95-118) with euclidean_distance_metric
// (R/ctrl/topologies/vect_distance_metrics.hpp:113-137) for the batches of a planner round: P trees, a few hundred
// queries each, every query against every vertex of its tree.  Exact results (index AND distance, first minimum wins):
// the matrix cores only produce a BOUNDED estimate that selects, per query, the few vertices whose fp64 distance is
// then evaluated with the reference's operation sequence.
//
// Why it is built this way (measured on the kernel it replaces, nn1_sweep_bf16_kernel): that sweep
// was bound by the VALU / LDS work AROUND its matrix instructions -- converting fp64 rows to split-bf16 operands for
// every query block that sweeps them (~15 VALU instructions per wave-slab), three matrix instructions and three LDS
// operand reads per 32 x 32 block of pairs, a per-lane candidate list in LDS -- and it re-read each tree once per 128
// queries (1.7 x the algorithmic bytes).  Here
//   * the operands are stored, not staged: one 16-byte load per lane and slab straight into the A operand (rows), one
//     per lane and query group into the B operands (nn1_mirror_prep_kernel builds those once per round);
//   * ONE instruction per 32 x 32 pairs: with half-precision inputs the error of c against |x - q|^2 - |q_h|^2 is
//     relative to the DISTANCE (below), so a single half-precision product has a narrower band than the split bf16 one;
//   * a wave keeps the B operands of up to kMirG x 32 = 384 queries in registers and runs every slab it loads against
//     all of them: a tree is read once per 384 queries, 32 bytes per vertex;
//   * no candidate bookkeeping in the sweep: PASS 1 only takes minima (8 v_min3_f32 per 16 estimates, branch-free) and
//     leaves min_r c per (row slice, query) in NnArgs::est (plain stores: the first version merged the slices with an
//     ordered-integer atomicMin per lane and spent most of its time there); nn1_mirror_thr_kernel turns the minimum over
//     the slices into the query's threshold; PASS 2 repeats the sweep -- same operands, same instruction, bit-identical
//     estimates -- against that FIXED threshold and appends the rows below it to the query's list;
//     nn1_mirror_resolve_kernel evaluates those exactly.
//
// Error bound.  x_h, q_h = half(float(x)), half(float(q)) per coordinate.  The rounding errors are MEASURED, not bounded:
//   dx = max over the rows of a tree of |x - x_h| (kept per tree: mirror_store_row's callers raise it, ordered-integer
//        atomicMax on the float's bits), dq = |q - q_h| of the query, delta = dx + dq >= | |x_h - q_h| - |x - q| |;
//   c_real = |x_h|^2 - 2 x_h.q_h = |x_h - q_h|^2 - |q_h|^2  =>  | c_real + |q_h|^2 - s | <= 2 d delta + delta^2
// with s = |x - q|^2, d = sqrt(s).  The instruction sums <= 16 exact products in float (order and intermediate roundings
// unspecified: 15 roundings of 2^-24 on partial sums below S = X^2 + 2 X |q|, X >= |x_h| for every row) and the float
// |x_h|^2 of the mirror carries 12 more: together below eps = 2^-19 S.  So s_r = c_r + |q_h|^2 + e_r with
// |e_r| <= E(d_r) := 2 d_r delta + delta^2 + eps.  Let m be the row of the smallest estimate and r* the true nearest row
// (any row tying with it included): d_r* <= d_m, hence
//   c_r* = s_r* - |q_h|^2 - e_r* <= s_m - |q_h|^2 + E(d_m) = c_m + e_m + E(d_m) <= c_m + 2 E(d_m),
// and d_m <= delta + sqrt(c_m + |q_h|^2 + 2 delta^2 + eps).  Rows with c <= c_min + band(c_min), band = 2 E(that bound)
// (evaluated in double, rounded up) are a superset of the candidates; distances that differ in the last bits of their
// sqrt lie far inside the band, so "first minimum wins" is decided by the exact pass among them.  At the planner's
// scale (12-D, |x| ~ 9, nearest neighbours at d ~ 2) delta ~ 3e-3 and the band ~ 0.03 in squared-distance units.
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "nn_mirror.h"
#include "rkh_internal.h"

namespace rkh {

void nn_set_last_kernel_name(const char* name);  // nn_sweep.hip (rkh_nn_kernel_name)

namespace {

typedef float mir_f16v __attribute__((ext_vector_type(16)));
typedef _Float16 mir_h8 __attribute__((ext_vector_type(8)));

constexpr int kMirG = 12;                  // query groups of 32 per block
constexpr int kMirQueries = 32 * kMirG;    // 384
constexpr int kMirThreads = 256;           // four waves: the same queries, every fourth slab of the block's slice each
constexpr uint32_t kMirCandCap = 32;       // candidate rows kept per query (more: the exact scan of the resolve kernel)
constexpr uint32_t kMirMaxSlices = 32;     // row slices per tree (rows of NnArgs::est)

__device__ __forceinline__ float fmin3(float a, float b, float c) { return __builtin_fminf(__builtin_fminf(a, b), c); }
__device__ __forceinline__ bool lex_less_m(double da, uint32_t ia, double db, uint32_t ib) {
  return (da < db) || (da == db && ia < ib);
}

// ---- once per round and query: the B operand and the query's share of the error bound
// qinfo[q] = {|q_h|^2 (rounded up), dq = |q - q_h| (rounded up), |q| (rounded up)}
__global__ __launch_bounds__(256) void nn1_mirror_prep_kernel(const NnArgs* __restrict__ table, int D) {
  const NnArgs a = table[blockIdx.y];
  const uint32_t B = a.d_B ? *a.d_B : a.B;
  const uint32_t qi = blockIdx.x * 256u + threadIdx.x;
  if (qi >= B) return;
  const double* __restrict__ qq = a.q + ((a.d_qoff ? uint64_t(*a.d_qoff) : 0ull) + qi) * D;
  uint32_t e[16];
  double qn2 = 0.0, qh2 = 0.0, dq2 = 0.0;
#pragma unroll
  for (int d = 0; d < kMirrorMaxDims; ++d) {
    const double qv = d < D ? qq[d < D ? d : 0] : 0.0;
    const uint32_t hb = mirror_half_bits(float(qv));
    const double qh = double(mirror_half_value(hb));
    qn2 += qv * qv;
    qh2 += qh * qh;
    dq2 += (qv - qh) * (qv - qh);
    e[d] = mirror_half_bits(-2.0f * mirror_half_value(hb));  // exact: a power-of-two multiple of a half
  }
  e[12] = e[13] = e[14] = 0x3C00u;  // 1.0 against the three pieces of |x_h|^2
  e[15] = 0u;
  uint4* __restrict__ qf = a.qfrag + uint64_t(qi) * 2;
  qf[0] = make_uint4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16));
  qf[1] = make_uint4(e[8] | (e[9] << 16), e[10] | (e[11] << 16), e[12] | (e[13] << 16), e[14] | (e[15] << 16));
  double* __restrict__ qi3 = a.qinfo + uint64_t(qi) * 3;
  qi3[0] = qh2 * (1.0 + 1e-12) + 1e-300;
  qi3[1] = sqrt(dq2) * (1.0 + 1e-12);
  qi3[2] = sqrt(qn2) * (1.0 + 1e-12);
}

// ---- between the passes: threshold of a query = (minimum over the row slices) + band
__global__ __launch_bounds__(256) void nn1_mirror_thr_kernel(const NnArgs* __restrict__ table, uint32_t gx,
                                                             double x_norm_bound) {
  const NnArgs a = table[blockIdx.y];
  const uint32_t B = a.d_B ? *a.d_B : a.B;
  const uint32_t qi = blockIdx.x * 256u + threadIdx.x;
  if (qi >= B) return;
  float cmin = INFINITY;
  for (uint32_t sl = 0; sl < gx; ++sl) cmin = __builtin_fminf(cmin, a.est[uint64_t(sl) * a.est_stride + qi]);
  const double* __restrict__ qi3 = a.qinfo + uint64_t(qi) * 3;
  const double qh2 = qi3[0], dq = qi3[1], qn = qi3[2];
  const double dx = double(__uint_as_float(*a.dx_max_bits)) * (1.0 + 1e-6);
  const double delta = dx + dq;
  const double X = x_norm_bound * (1.0 + 1e-3);  // |x_h| <= |x| (1 + 2^-11)
  const double eps = 1.9073486328125e-06 * (X * X + 2.0 * X * qn) + 1e-9;
  const double base = fmax(0.0, double(cmin) + qh2) + 2.0 * delta * delta + eps;
  const double d_up = delta + sqrt(base) * (1.0 + 1e-12);
  const double E = 2.0 * d_up * delta + delta * delta + eps;
  const double band = 2.0 * E * (1.0 + 1e-9);
  // (+inf, an empty tree, stays +inf: every row is a candidate and the resolve kernel sorts it out)
  a.thr[qi] = cmin < INFINITY ? __double2float_ru(double(cmin) + band) : INFINITY;
}

// Work items as in the bf16 sweep: 1-D grid of 8 * ceil(W / 8) blocks for W = gx * (query blocks of all problems) items
// (row slice, query block, problem); block L runs on XCD L % 8 as that XCD's (L / 8)-th block, XCD x takes the items
// [x Wc, (x + 1) Wc), numbered with the query block fastest.
__device__ __forceinline__ bool mirror_item(const uint32_t* __restrict__ yblock_base, uint32_t n_problems, uint32_t gx,
                                            uint32_t* bx, uint32_t* by, uint32_t* bz) {
  const uint32_t L = blockIdx.x;
  const uint32_t ytot = yblock_base[n_problems];
  const uint32_t W = ytot * gx, Wc = (W + 7) >> 3;
  const uint32_t slot = L >> 3, w = (L & 7) * Wc + slot;
  if (slot >= Wc || w >= W) return false;
  const uint32_t yy = w / gx;
  uint32_t p = 0, hi_p = n_problems;  // yblock_base[p] <= yy < yblock_base[hi_p]
  while (hi_p - p > 1) {
    const uint32_t mid = (p + hi_p) >> 1;
    if (yblock_base[mid] <= yy) p = mid;
    else hi_p = mid;
  }
  const uint32_t y0 = yblock_base[p], cnt = yblock_base[p + 1] - y0;
  const uint32_t r = w - y0 * gx;
  *bx = r / cnt;
  *by = r - (*bx) * cnt;
  *bz = p;
  return true;
}

// PASS 1: minimum of the estimates per (slice, query) -> NnArgs::est.  PASS 2: rows at or below NnArgs::thr -> lists.
template <int PASS>
__global__ __launch_bounds__(kMirThreads, PASS == 1 ? 4 : 3) void nn1_mirror_kernel(const NnArgs* __restrict__ table,
                                                                    const uint32_t* __restrict__ yblock_base,
                                                                    uint32_t n_problems, uint32_t gx) {
  __shared__ float wave_min[PASS == 1 ? kMirThreads / 64 : 1][PASS == 1 ? kMirQueries : 1];
  uint32_t bx, by, bz;
  if (!mirror_item(yblock_base, n_problems, gx, &bx, &by, &bz)) return;
  auto uniform64 = [](uint64_t v) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane(uint32_t(v)), hi = __builtin_amdgcn_readfirstlane(uint32_t(v >> 32));
    return (uint64_t(hi) << 32) | lo;
  };
  const NnArgs a = table[bz];
  const uint4* __restrict__ mirror = reinterpret_cast<const uint4*>(uniform64(reinterpret_cast<uint64_t>(a.mirror)));
  const uint4* __restrict__ qfrag = reinterpret_cast<const uint4*>(uniform64(reinterpret_cast<uint64_t>(a.qfrag)));
  const uint32_t n = __builtin_amdgcn_readfirstlane(a.d_n ? *a.d_n : uint32_t(a.n));
  const uint32_t B = __builtin_amdgcn_readfirstlane(a.d_B ? *a.d_B : a.B);
  const uint32_t q0 = by * uint32_t(kMirQueries);
  if (q0 >= B) return;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int col = lane & 31, hi = lane >> 5;
  const int ng = int(((B - q0 < uint32_t(kMirQueries) ? B - q0 : uint32_t(kMirQueries)) + 31u) >> 5);  // uniform

  // B operands: lane (col, hi) holds slots 8 hi .. 8 hi + 7 of query q0 + 32 g + col
  uint4 bop[kMirG];
  float thr[kMirG];  // PASS 1: running minimum; PASS 2: the fixed threshold
#pragma unroll
  for (int g = 0; g < kMirG; ++g) {
    bop[g] = make_uint4(0u, 0u, 0u, 0u);
    thr[g] = PASS == 1 ? INFINITY : -INFINITY;
    if (g < ng) {
      const uint32_t qi = q0 + 32u * uint32_t(g) + uint32_t(col);
      const uint32_t qsrc = qi < B ? qi : (B - 1u);
      bop[g] = qfrag[uint64_t(qsrc) * 2 + uint32_t(hi)];
      if (PASS == 2) thr[g] = qi < B ? a.thr[qsrc] : -INFINITY;  // a query slot past the batch never records
    }
  }

  // this wave's slabs: every fourth one of the block's contiguous slice of the tree
  const uint32_t slabs_total = (n + 31u) >> 5;
  const uint32_t per_block = (slabs_total + gx - 1u) / gx;
  const uint32_t s_first = bx * per_block;
  uint32_t s_end = s_first + per_block;
  if (s_end > slabs_total) s_end = slabs_total;
  const uint4* __restrict__ my = mirror + uint32_t(lane);
  constexpr uint32_t kStep = 4;
  auto load = [&](uint32_t s) -> uint4 { return my[uint64_t(s < s_end ? s : s_first) * 64u]; };
  uint32_t s = s_first + uint32_t(wave);
  uint4 a0 = make_uint4(0u, 0u, 0u, 0u), a1 = a0, a2 = a0;
  if (s_first < s_end) {
    a0 = load(s);
    a1 = load(s + kStep);
    a2 = load(s + 2 * kStep);
  }
  mir_f16v zero;
#pragma unroll
  for (int k = 0; k < 16; ++k) zero[k] = 0.0f;
  // one 32 x 32 block of estimates: minimum tree (PASS 1) / threshold test (PASS 2)
  auto settle = [&](int g, const mir_f16v& c, uint32_t slab) {
    const float t1 = fmin3(c[0], c[1], c[2]), t2 = fmin3(c[3], c[4], c[5]), t3 = fmin3(c[6], c[7], c[8]);
    const float t4 = fmin3(c[9], c[10], c[11]), t5 = fmin3(c[12], c[13], c[14]);
    const float t6 = fmin3(t1, t2, t3), t7 = fmin3(t4, t5, c[15]);
    if (PASS == 1) {
      thr[g] = fmin3(t6, t7, thr[g]);
    } else {
      const float m = __builtin_fminf(t6, t7);
      // rare (a few rows per query over the whole tree): a wave-uniform test first, so that the common case falls
      // through one untaken scalar branch (a per-lane `if` compiles to a TAKEN exec-mask branch around the inlined
      // slow path in every block: pass 2 then ran at 0.4 of pass 1's speed)
      if (__builtin_expect(__any(m <= thr[g]), 0) && m <= thr[g]) {
        const uint32_t qi = q0 + 32u * uint32_t(g) + uint32_t(col);
        const uint32_t row0 = slab * 32u + 4u * uint32_t(hi);
        uint32_t mask = 0u;
#pragma unroll
        for (int i = 0; i < 16; ++i) mask |= (c[i] <= thr[g]) ? (1u << i) : 0u;
        const uint32_t cnt = uint32_t(__builtin_popcount(mask));
        uint32_t pos = atomicAdd(&a.cand_cnt[qi], cnt);
#pragma unroll 1
        while (mask) {
          const uint32_t i = uint32_t(__builtin_ctz(mask));
          mask &= mask - 1u;
          const uint32_t row = row0 + 8u * (i >> 2) + (i & 3u);
          if (pos < kMirCandCap) a.cand_rows[uint64_t(qi) * kMirCandCap + pos] = row;
          ++pos;
        }
      }
    }
  };
  // Query groups go in pairs with two accumulators, software-pipelined: the matrix instruction of the next group is
  // issued before the minimum tree of the current one, so a wave's own VALU work covers its matrix latency (one
  // accumulator: 12 wait states behind every instruction, 71 cycles per block at four waves per SIMD).  An odd group
  // count runs one idle group (B operand zero: its estimates are |x_h|^2, its threshold never matches).
  const int npairs = (ng + 1) >> 1;  // uniform
  // (the pair count is a compile-time constant of the loop body: with a run-time bound the conditional re-assignment of
  // the rolling accumulator costs sixteen register copies per pair)
  auto sweep = [&](auto np_c) {
    constexpr int NP = decltype(np_c)::value;
#pragma unroll 1
    for (; s < s_end; s += kStep) {
      const uint4 an = load(s + 3 * kStep);
      const mir_h8 av = __builtin_bit_cast(mir_h8, a0);
      // rolling: the matrix instruction of group g + 1 is in flight while the VALU settles group g
      mir_f16v cA = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, __builtin_bit_cast(mir_h8, bop[0]), zero, 0, 0, 0);
#pragma unroll
      for (int pr = 0; pr < NP; ++pr) {
        const mir_f16v cB =
            __builtin_amdgcn_mfma_f32_32x32x16_f16(av, __builtin_bit_cast(mir_h8, bop[2 * pr + 1]), zero, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);  // (the scheduler would sink the instruction behind the settling it overlaps)
        settle(2 * pr, cA, s);
        __builtin_amdgcn_sched_barrier(0);
        if (pr + 1 < NP)
          cA = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, __builtin_bit_cast(mir_h8, bop[2 * pr + 2 < kMirG ? 2 * pr + 2 : 0]),
                                                      zero, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        settle(2 * pr + 1, cB, s);
        __builtin_amdgcn_sched_barrier(0);
      }
      a0 = a1;
      a1 = a2;
      a2 = an;
    }
  };
  switch (npairs) {
    case 1: sweep(std::integral_constant<int, 1>()); break;
    case 2: sweep(std::integral_constant<int, 2>()); break;
    case 3: sweep(std::integral_constant<int, 3>()); break;
    case 4: sweep(std::integral_constant<int, 4>()); break;
    case 5: sweep(std::integral_constant<int, 5>()); break;
    default: sweep(std::integral_constant<int, 6>()); break;
  }
  if (PASS == 1) {
    // the block's four waves hold different slabs of the same queries: merge through LDS, one store per query
#pragma unroll
    for (int g = 0; g < kMirG; ++g) {
      if (g < ng) {
        float m = thr[g];
        m = __builtin_fminf(m, __shfl_xor(m, 32, 64));  // the two lane halves hold different rows of the same query
        if (hi == 0) wave_min[wave][32 * g + col] = m;
      }
    }
    __syncthreads();
    for (uint32_t t = uint32_t(tid); t < uint32_t(32 * ng); t += uint32_t(kMirThreads)) {
      const float m = __builtin_fminf(__builtin_fminf(wave_min[0][t], wave_min[1][t]),
                                      __builtin_fminf(wave_min[2][t], wave_min[3][t]));
      if (q0 + t < B) a.est[uint64_t(bx) * a.est_stride + q0 + t] = m;
    }
  }
}

// Exact evaluation of the candidates: one 16-lane group per query, lane j takes the candidates j, j + 16, ...; a list
// that overflowed (more than kMirCandCap rows within the band: many coincident vertices) is replaced by the exact scan
// of the whole tree.  Pad rows (past the end of the tree; a removed vertex evaluates to +inf by itself) are skipped.
// The operation sequence is nn1_sweep_kernel's: left-to-right sum of squares, sqrt, lexicographic (distance, index).
template <int DP>
__global__ __launch_bounds__(256) void nn1_mirror_resolve_kernel(const NnArgs* __restrict__ table, int D) {
  const NnArgs a = table[blockIdx.y];
  const uint32_t B = a.d_B ? *a.d_B : a.B;
  const uint32_t qi = blockIdx.x * 16u + (threadIdx.x >> 4);
  const uint32_t gl = threadIdx.x & 15u;
  if (qi >= B) return;  // whole 16-lane groups leave together
  const uint32_t n = a.d_n ? *a.d_n : uint32_t(a.n);
  const double* __restrict__ qq = a.q + ((a.d_qoff ? uint64_t(*a.d_qoff) : 0ull) + qi) * D;
  double qv[DP];
#pragma unroll
  for (int d = 0; d < DP; ++d) qv[d] = d < D ? qq[d < D ? d : 0] : 0.0;
  const uint32_t cnt = a.cand_cnt[qi];
  double bd = INFINITY;
  uint32_t bi = 0xFFFFFFFFu;
  auto exact = [&](uint32_t row) {
    if (row >= n) return;
    const double* __restrict__ p = a.pos + uint64_t(row) * DP;
    double s;
    {
      const double df = qv[0] - p[0];
      s = df * df;
    }
#pragma unroll
    for (int d = 1; d < DP; ++d) {
      const double df = qv[d] - p[d];
      s = s + df * df;
    }
    const double dd = sqrt(s);
    if (lex_less_m(dd, row, bd, bi)) {
      bd = dd;
      bi = row;
    }
  };
  if (cnt <= kMirCandCap) {
    for (uint32_t j = gl; j < cnt; j += 16u) exact(a.cand_rows[uint64_t(qi) * kMirCandCap + j]);
  } else {
    for (uint32_t row = gl; row < n; row += 16u) exact(row);
  }
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) {
    const double od = __shfl_xor(bd, off, 64);
    const uint32_t oi = __shfl_xor(bi, off, 64);
    if (lex_less_m(od, oi, bd, bi)) {
      bd = od;
      bi = oi;
    }
  }
  if (gl == 0) {
    a.idx[qi] = bi;
    a.dist[qi] = bd;
    a.cand_cnt[qi] = 0u;  // consumed: ready for the next round
  }
}

__global__ __launch_bounds__(256) void mirror_fill_kernel(uint4* __restrict__ mirror, uint64_t slabs) {
  const uint64_t i = blockIdx.x * uint64_t(blockDim.x) + threadIdx.x;  // one 16-byte fragment each
  if (i >= slabs * 64u) return;
  const bool upper = ((i >> 5) & 1u) != 0u;  // lane half 1 holds slots 8..15: the pad norm sits in slot 12
  mirror[i] = upper ? make_uint4(0u, 0u, mirror_half_bits(kMirrorPadNorm), 0u) : make_uint4(0u, 0u, 0u, 0u);
}

__global__ __launch_bounds__(256) void mirror_build_kernel(uint4* __restrict__ mirror, const double* __restrict__ pos,
                                                           uint64_t n, int D, int DP, uint32_t* __restrict__ dx_max_bits) {
  const uint64_t row = blockIdx.x * uint64_t(blockDim.x) + threadIdx.x;
  if (row >= n) return;
  mirror_store_row(mirror, row, pos + row * DP, D, dx_max_bits);
}

}  // namespace

uint32_t nn1_mirror_queries() { return uint32_t(kMirQueries); }
uint32_t nn1_mirror_cand_cap() { return kMirCandCap; }
uint32_t nn1_mirror_max_slices() { return kMirMaxSlices; }
size_t nn1_mirror_bytes(uint64_t capacity_rows) { return size_t((capacity_rows + 31) / 32) * 1024; }
// per-query scratch of a tree's sweeps, in bytes per query slot: B operand (32) + qinfo (24) + threshold (4) +
// candidate count (4) + candidate rows + the per-slice minima
size_t nn1_mirror_query_bytes() { return 32 + 24 + 4 + 4 + 4 * size_t(kMirCandCap) + 4 * size_t(kMirMaxSlices); }
// carve a block of nn1_mirror_query_bytes() * b_max bytes (b_max a multiple of 8, base 256-byte aligned) into the
// per-query arrays of `a`
void nn1_mirror_carve(void* base, uint32_t b_max, NnArgs* a) {
  char* p = static_cast<char*>(base);
  a->qfrag = reinterpret_cast<uint4*>(p);
  p += size_t(b_max) * 32;
  a->qinfo = reinterpret_cast<double*>(p);
  p += size_t(b_max) * 24;
  a->thr = reinterpret_cast<float*>(p);
  p += size_t(b_max) * 4;
  a->cand_cnt = reinterpret_cast<uint32_t*>(p);
  p += size_t(b_max) * 4;
  a->cand_rows = reinterpret_cast<uint32_t*>(p);
  p += size_t(b_max) * 4 * kMirCandCap;
  a->est = reinterpret_cast<float*>(p);
  a->est_stride = b_max;
}

// does the mirror sweep take this problem shape?  (RKH_NN_MIRROR=0 keeps the bf16 sweep: diagnostics / A-B runs)
bool nn1_mirror_applies(int D, double coord_bound) {
  static const bool on = [] {
    const char* e = getenv("RKH_NN_MIRROR");
    return !(e && e[0] == '0');
  }();
  return on && D >= 1 && D <= kMirrorMaxDims && coord_bound >= 1e-3 && coord_bound <= kMirrorMaxBound;
}

rkh_status launch_mirror_fill(hipStream_t s, void* d_mirror, uint64_t capacity_rows) {
  const uint64_t slabs = (capacity_rows + 31) / 32;
  hipLaunchKernelGGL(mirror_fill_kernel, dim3(uint32_t((slabs * 64 + 255) / 256)), dim3(256), 0, s,
                     static_cast<uint4*>(d_mirror), slabs);
  RKH_HIP(hipGetLastError());
  return RKH_OK;
}

rkh_status launch_mirror_build(hipStream_t s, void* d_mirror, const double* d_pos, uint64_t n, int D, int DP,
                               uint32_t* d_dx_max_bits) {
  if (n == 0) return RKH_OK;
  hipLaunchKernelGGL(mirror_build_kernel, dim3(uint32_t((n + 255) / 256)), dim3(256), 0, s, static_cast<uint4*>(d_mirror),
                     d_pos, n, D, DP, d_dx_max_bits);
  RKH_HIP(hipGetLastError());
  return RKH_OK;
}

// The launches of a round's NN search over the mirrors: B operands, pass 1, thresholds, pass 2, exact resolution.
// d_yblock_base: [n_problems + 1] exclusive prefix of ceil(B_p / nn1_mirror_queries()); n_upper / B_upper (host bounds)
// only size the grids; x_norm_bound >= |x| for every vertex.  ev0 / ev1 bracket all five launches.
rkh_status launch_nn1_mirror(hipStream_t s, int D, const NnArgs* d_table, uint32_t n_problems, uint64_t n_upper,
                             uint32_t B_upper, double x_norm_bound, const uint32_t* d_yblock_base, hipEvent_t ev0,
                             hipEvent_t ev1) {
  if (B_upper == 0 || n_problems == 0) return RKH_OK;
  const uint32_t gy = (B_upper + kMirQueries - 1) / kMirQueries;
  const uint64_t slabs = (n_upper + 31) / 32;
  // row slices per tree: ~8 k blocks over the whole grid, at least 32 slabs (8 per wave) per block
  static const long forced = [] {
    const char* e = getenv("RKH_NN_MIRROR_SLICES");
    return e ? atol(e) : 0L;
  }();
  uint64_t gx = forced > 0 ? uint64_t(forced) : 8192 / (uint64_t(gy) * n_problems);
  if (gx > slabs / 32) gx = slabs / 32;
  if (gx > kMirMaxSlices) gx = kMirMaxSlices;
  if (gx < 1) gx = 1;
  const dim3 grid(uint32_t((gx * gy * n_problems + 7) / 8 * 8));
  const dim3 qgrid((B_upper + 255) / 256, n_problems);
  nn_set_last_kernel_name("nn1_mirror_kernel");
  if (ev0) (void)hipEventRecord(ev0, s);
  hipLaunchKernelGGL(nn1_mirror_prep_kernel, qgrid, dim3(256), 0, s, d_table, D);
  hipLaunchKernelGGL((nn1_mirror_kernel<1>), grid, dim3(kMirThreads), 0, s, d_table, d_yblock_base, n_problems, uint32_t(gx));
  hipLaunchKernelGGL(nn1_mirror_thr_kernel, qgrid, dim3(256), 0, s, d_table, uint32_t(gx), x_norm_bound);
  hipLaunchKernelGGL((nn1_mirror_kernel<2>), grid, dim3(kMirThreads), 0, s, d_table, d_yblock_base, n_problems, uint32_t(gx));
  const dim3 rgrid((B_upper + 15) / 16, n_problems);
  switch (nn_padded_dims(D)) {
    case 2: hipLaunchKernelGGL((nn1_mirror_resolve_kernel<2>), rgrid, dim3(256), 0, s, d_table, D); break;
    case 4: hipLaunchKernelGGL((nn1_mirror_resolve_kernel<4>), rgrid, dim3(256), 0, s, d_table, D); break;
    case 6: hipLaunchKernelGGL((nn1_mirror_resolve_kernel<6>), rgrid, dim3(256), 0, s, d_table, D); break;
    case 8: hipLaunchKernelGGL((nn1_mirror_resolve_kernel<8>), rgrid, dim3(256), 0, s, d_table, D); break;
    case 12: hipLaunchKernelGGL((nn1_mirror_resolve_kernel<12>), rgrid, dim3(256), 0, s, d_table, D); break;
    default: set_error("nn mirror: unsupported dimension"); return RKH_ERR_BAD_ARG;
  }
  if (ev1) (void)hipEventRecord(ev1, s);
  RKH_HIP(hipGetLastError());
  return RKH_OK;
}

}  // namespace rkh

// Diagnostics (include/rkh_diag.h): the mirror sweep on a caller's point cloud, outside a planner -- the parity tests
// drive it with adversarial clouds (duplicates, one-ulp neighbours, more coincident vertices than a candidate list holds).
extern "C" rkh_status rkh_diag_nn_mirror_query(rkh_ctx* ctx, const double* pts, uint64_t n, int D, const double* q,
                                               uint32_t B, double coord_bound, uint32_t* idx, double* dist) {
  using namespace rkh;
  if (!ctx || !pts || !q || !idx || !dist || n == 0 || B == 0) return RKH_ERR_BAD_ARG;
  if (!nn1_mirror_applies(D, coord_bound)) {
    set_error("nn mirror: needs 1 <= D <= 12 and a coordinate bound in [1e-3, 32]");
    return RKH_ERR_UNSUPPORTED;
  }
  RKH_HIP(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const int DP = nn_padded_dims(D);
  std::vector<double> rows(size_t(n) * DP, 0.0);
  for (uint64_t i = 0; i < n; ++i)
    for (int d = 0; d < D; ++d) rows[i * DP + d] = pts[i * D + d];
  const uint32_t Bp = (B + 7u) / 8u * 8u;
  double *d_pos = nullptr, *d_q = nullptr, *d_dist = nullptr;
  void *d_mirror = nullptr, *d_scratch = nullptr;
  uint32_t *d_idx = nullptr, *d_yb = nullptr, *d_dx = nullptr;
  NnArgs* d_tab = nullptr;
  auto cleanup = [&]() {
    void* bufs[] = {d_pos, d_q, d_dist, d_mirror, d_idx, d_scratch, d_yb, d_tab, d_dx};
    for (void* b : bufs) (void)hipFree(b);
  };
#define RKH_TRY(expr)                                               \
  do {                                                              \
    hipError_t _e = (expr);                                         \
    if (_e != hipSuccess) {                                         \
      set_error(std::string(#expr) + ": " + hipGetErrorString(_e)); \
      cleanup();                                                    \
      return RKH_ERR_DEVICE;                                        \
    }                                                               \
  } while (0)
  const size_t scratch_bytes = nn1_mirror_query_bytes() * Bp;
  RKH_TRY(hipMalloc(&d_pos, rows.size() * sizeof(double)));
  RKH_TRY(hipMalloc(&d_q, size_t(B) * D * sizeof(double)));
  RKH_TRY(hipMalloc(&d_dist, size_t(B) * sizeof(double)));
  RKH_TRY(hipMalloc(&d_idx, size_t(B) * sizeof(uint32_t)));
  RKH_TRY(hipMalloc(&d_scratch, scratch_bytes));
  RKH_TRY(hipMalloc(&d_yb, 2 * sizeof(uint32_t)));
  RKH_TRY(hipMalloc(&d_dx, sizeof(uint32_t)));
  RKH_TRY(hipMalloc(&d_tab, sizeof(NnArgs)));
  RKH_TRY(hipMalloc(&d_mirror, nn1_mirror_bytes(n)));
  RKH_TRY(hipMemcpy(d_pos, rows.data(), rows.size() * sizeof(double), hipMemcpyHostToDevice));
  RKH_TRY(hipMemcpy(d_q, q, size_t(B) * D * sizeof(double), hipMemcpyHostToDevice));
  RKH_TRY(hipMemset(d_scratch, 0, scratch_bytes));
  RKH_TRY(hipMemset(d_dx, 0, sizeof(uint32_t)));
  const uint32_t yb[2] = {0u, (B + nn1_mirror_queries() - 1) / nn1_mirror_queries()};
  RKH_TRY(hipMemcpy(d_yb, yb, sizeof(yb), hipMemcpyHostToDevice));
  NnArgs a;
  a.pos = d_pos;
  a.n = n;
  a.q = d_q;
  a.B = B;
  a.idx = d_idx;
  a.dist = d_dist;
  a.mirror = d_mirror;
  a.dx_max_bits = d_dx;
  nn1_mirror_carve(d_scratch, Bp, &a);
  RKH_TRY(hipMemcpy(d_tab, &a, sizeof(a), hipMemcpyHostToDevice));
  rkh_status st = launch_mirror_fill(s, d_mirror, n);
  if (st == RKH_OK) st = launch_mirror_build(s, d_mirror, d_pos, n, D, DP, d_dx);
  if (st == RKH_OK)
    st = launch_nn1_mirror(s, D, d_tab, 1, n, B, std::sqrt(double(D)) * coord_bound, d_yb, nullptr, nullptr);
  if (st != RKH_OK) {
    cleanup();
    return st;
  }
  RKH_TRY(hipStreamSynchronize(s));
  RKH_TRY(hipMemcpy(idx, d_idx, size_t(B) * sizeof(uint32_t), hipMemcpyDeviceToHost));
  RKH_TRY(hipMemcpy(dist, d_dist, size_t(B) * sizeof(double), hipMemcpyDeviceToHost));
#undef RKH_TRY
  cleanup();
  return RKH_OK;
}
