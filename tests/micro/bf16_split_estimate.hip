// Split-bf16 estimate |x|^2 - 2 x.q of 32 rows x 32 queries on v_mfma_f32_32x32x16_bf16 against fp64, for several half-row
// widths H (the operand construction of nn1_sweep_bf16_kernel, reak_amd/csrc/nn_sweep.hip).
// hipcc --offload-arch=gfx950 -O3 -o bf16_split_estimate bf16_split_estimate.hip && ./bf16_split_estimate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f2v __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));
__device__ __forceinline__ uint32_t bf16_bits(float a) {
  const f2v v = {a, 0.0f};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2)) & 0xFFFFu;
}
template <int H>
struct Op {
  static constexpr int K = 3 * H + 3, NI = (K + 7) / 8;
  __device__ static void pack(const uint32_t (&e)[8 * NI], uint4 (&f)[NI]) {
    for (int i = 0; i < NI; ++i) {
      f[i].x = e[8 * i] | (e[8 * i + 1] << 16); f[i].y = e[8 * i + 2] | (e[8 * i + 3] << 16);
      f[i].z = e[8 * i + 4] | (e[8 * i + 5] << 16); f[i].w = e[8 * i + 6] | (e[8 * i + 7] << 16);
    }
  }
  __device__ static void split(float v, uint32_t& hi, uint32_t& lo) { hi = bf16_bits(v); lo = bf16_bits(v - __uint_as_float(hi << 16)); }
  __device__ static void build_a(const float (&x)[H], float nrm, uint4 (&f)[NI]) {
    uint32_t e[8 * NI]; for (int k = 0; k < 8 * NI; ++k) e[k] = 0;
    for (int j = 0; j < H; ++j) { uint32_t hi, lo; split(x[j], hi, lo); e[j] = hi; e[H + j] = lo; e[2 * H + j] = hi; }
    const uint32_t n0 = bf16_bits(nrm); const float r1 = nrm - __uint_as_float(n0 << 16);
    const uint32_t n1 = bf16_bits(r1); const float r2 = r1 - __uint_as_float(n1 << 16);
    e[3 * H] = n0; e[3 * H + 1] = n1; e[3 * H + 2] = bf16_bits(r2);
    pack(e, f);
  }
  __device__ static void build_b(const float (&q)[H], uint4 (&f)[NI]) {
    uint32_t e[8 * NI]; for (int k = 0; k < 8 * NI; ++k) e[k] = 0;
    for (int j = 0; j < H; ++j) { uint32_t hi, lo; split(q[j], hi, lo); e[j] = hi; e[H + j] = hi; e[2 * H + j] = lo; }
    e[3 * H] = e[3 * H + 1] = e[3 * H + 2] = 0x3F80u;
    pack(e, f);
  }
};
template <int H>
__global__ void k(const double* x, const double* q, float* out) {  // x: 32 rows x 2H, q: 32 x 2H; out[row][query]
  const int lane = threadIdx.x, col = lane & 31, hi = lane >> 5;
  float xf[H], qf[H], nrm = 0.f;
  for (int j = 0; j < H; ++j) { xf[j] = float(x[col * 2 * H + H * hi + j]); nrm = fmaf(xf[j], xf[j], nrm); qf[j] = -2.0f * float(q[col * 2 * H + H * hi + j]); }
  uint4 a[Op<H>::NI], b[Op<H>::NI];
  Op<H>::build_a(xf, nrm, a); Op<H>::build_b(qf, b);
  f16v c = {0};
  for (int i = 0; i < Op<H>::NI; ++i) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[i]), c, 0, 0, 0);
  for (int i = 0; i < 16; ++i) out[((i & 3) + 8 * (i >> 2) + 4 * hi) * 32 + col] = c[i];
}
template <int H>
void run() {
  std::mt19937_64 rng(H);
  std::uniform_real_distribution<double> u(-3.14159, 3.14159);
  std::vector<double> x(32 * 2 * H), q(32 * 2 * H);
  for (auto& v : x) v = u(rng);
  for (auto& v : q) v = u(rng);
  double *dx, *dq; float* dout;
  hipMalloc(&dx, x.size() * 8); hipMalloc(&dq, q.size() * 8); hipMalloc(&dout, 1024 * 4);
  hipMemcpy(dx, x.data(), x.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dq, q.data(), q.size() * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k<H>, dim3(1), dim3(64), 0, 0, dx, dq, dout);
  std::vector<float> out(1024);
  hipMemcpy(out.data(), dout, 4096, hipMemcpyDeviceToHost);
  double worst = 0, scale = 2.0 * H * 3.14159 * 3.14159;
  for (int r = 0; r < 32; ++r) for (int c = 0; c < 32; ++c) {
    double t = 0;
    for (int d = 0; d < 2 * H; ++d) t += x[r * 2 * H + d] * x[r * 2 * H + d] - 2.0 * x[r * 2 * H + d] * q[c * 2 * H + d];
    worst = std::fmax(worst, std::fabs(double(out[r * 32 + c]) - t));
  }
  printf("H=%d (Dp=%d, %d instructions): max |estimate - exact| = %.3e = %.2f * 2^-16 Dp M^2\n", H, 2 * H, Op<H>::NI, worst, worst / (scale * 1.52587890625e-05));
}
int main() { run<1>(); run<2>(); run<3>(); run<4>(); run<6>(); run<8>(); return 0; }
