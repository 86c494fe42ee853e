// nn_mirror.h -- the half-precision MIRROR of a tree that the planner-regime 1-NN sweep reads (nn_mirror.hip).
//
// Beside its fp64 rows [n][Dp] a tree keeps, per vertex, one ready-made A operand of v_mfma_f32_32x32x16_f16: the 16
// k-slots  [ x_h(0) .. x_h(11) | n0 n1 n2 | 0 ]  with x_h(d) = half(float(x(d))) and n0 + n1 + n2 = |x_h|^2 exactly (a
// float split in three halves).  Against a query's B operand  [ -2 q_h(0) .. -2 q_h(11) | 1 1 1 | 0 ]  ONE matrix
// instruction gives c = |x_h|^2 - 2 x_h.q_h = |x_h - q_h|^2 - |q_h|^2 for 32 rows x 32 queries.  Rows are stored by
// 32-row slab in the operand's own lane order -- slab s, lane half h, row r: 16 bytes at ((s * 2 + h) * 32 + r) * 16,
// lane (r, h) of a wave loads its fragment with one coalesced 16-byte load, no LDS, no conversion in the sweep -- so
// a sweep reads 32 bytes per vertex where the fp64 rows are 8 Dp.  The mirror is written where rows are written
// (commit_kernel, planner.hip); rows past the end of a tree and removed vertices are PAD rows (coordinates 0, n0 =
// 60000): their estimate is 60000, above every real one while Dp (2 M)^2 < 60000.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace rkh {

constexpr int kMirrorMaxDims = 12;       // coordinates a fragment holds (slots 0..11)
constexpr float kMirrorPadNorm = 60000.0f;
constexpr double kMirrorMaxBound = 32.0;  // |coordinate| bound up to which the pad rows stay out of reach (Dp = 12)

__device__ __forceinline__ uint32_t mirror_half_bits(float v) {  // round to nearest even (v_cvt_f16_f32)
  return uint32_t(__builtin_bit_cast(unsigned short, _Float16(v)));
}
__device__ __forceinline__ float mirror_half_value(uint32_t bits) {
  return float(__builtin_bit_cast(_Float16, (unsigned short)(bits)));
}

// the two 16-byte fragments (lane half 0: slots 0..7, lane half 1: slots 8..15) of one vertex row; *err = |x - x_h|
// rounded up (0 for a removed vertex)
__device__ __forceinline__ void mirror_row_fragments(const double* __restrict__ row, int D, uint4* f0, uint4* f1, float* err) {
  uint32_t e[16];
  float nrm = 0.0f;
  double e2 = 0.0;
  bool finite = true;
#pragma unroll
  for (int d = 0; d < kMirrorMaxDims; ++d) {
    const double xv = d < D ? row[d < D ? d : 0] : 0.0;
    finite = finite && (xv - xv == 0.0);
    const uint32_t hb = mirror_half_bits(float(xv));
    const float xh = mirror_half_value(hb);
    nrm = __builtin_fmaf(xh, xh, nrm);  // exact products (22 bits), float sum
    e2 += (xv - double(xh)) * (xv - double(xh));
    e[d] = hb;
  }
  *err = finite ? __double2float_ru(sqrt(e2) * (1.0 + 1e-12)) : 0.0f;
  const uint32_t n0 = mirror_half_bits(nrm);
  const float r1 = nrm - mirror_half_value(n0);  // exact
  const uint32_t n1 = mirror_half_bits(r1);
  const float r2 = r1 - mirror_half_value(n1);   // exact
  e[12] = n0;
  e[13] = n1;
  e[14] = mirror_half_bits(r2);
  e[15] = 0u;
  if (!finite) {  // a removed vertex (+inf row): a pad row
#pragma unroll
    for (int d = 0; d < 16; ++d) e[d] = 0u;
    e[12] = mirror_half_bits(kMirrorPadNorm);
  }
  *f0 = make_uint4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16));
  *f1 = make_uint4(e[8] | (e[9] << 16), e[10] | (e[11] << 16), e[12] | (e[13] << 16), e[14] | (e[15] << 16));
}

// dx_max_bits: the tree's running maximum of |x - x_h| over its rows (bits of a non-negative float: ordered as integers)
__device__ __forceinline__ void mirror_store_row(uint4* __restrict__ mirror, uint64_t row, const double* __restrict__ src,
                                                 int D, uint32_t* __restrict__ dx_max_bits) {
  uint4 f0, f1;
  float err;
  mirror_row_fragments(src, D, &f0, &f1, &err);
  const uint64_t slab = row >> 5, r = row & 31u;
  mirror[(slab * 2 + 0) * 32 + r] = f0;
  mirror[(slab * 2 + 1) * 32 + r] = f1;
  if (err > 0.0f) atomicMax(dx_max_bits, __float_as_uint(err));
}

}  // namespace rkh
