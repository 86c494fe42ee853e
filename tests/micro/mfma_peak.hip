// Achievable rate of the fp32 MFMA shapes on this GPU (no memory traffic): waves x independent accumulator chains.
// hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_peak.hip && ./mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
template <int CHAINS, int SHAPE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f16v c[CHAINS];
  f4v d[CHAINS];
  for (int i = 0; i < CHAINS; ++i) { c[i] = f16v(float(threadIdx.x + i)); d[i] = f4v(float(threadIdx.x + i)); }
  float a = float(threadIdx.x) * 1e-3f, b = 1.0f + float(blockIdx.x) * 1e-6f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) {
      if (SHAPE == 0) c[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c[i], 0, 0, 0);
      else d[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, d[i], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < CHAINS; ++i) { for (int j = 0; j < 16; ++j) s += c[i][j]; for (int j = 0; j < 4; ++j) s += d[i][j]; }
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int CHAINS, int SHAPE>
void run(int blocks_per_cu) {
  const int blocks = 256 * blocks_per_cu, iters = 20000;
  float* out; hipMalloc(&out, size_t(blocks) * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<CHAINS, SHAPE>), dim3(blocks), dim3(256), 0, 0, out, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<CHAINS, SHAPE>), dim3(blocks), dim3(256), 0, 0, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = double(blocks) * 4 /*waves*/ * iters * CHAINS * (SHAPE == 0 ? 4096.0 : 2048.0);
  printf("%s chains=%d waves/SIMD=%d: %.1f TFLOP/s\n", SHAPE == 0 ? "32x32x2f32" : "16x16x4f32", CHAINS, blocks_per_cu, flops / ms / 1e9);
  hipFree(out);
}
int main() {
  run<1, 0>(1); run<2, 0>(1); run<1, 0>(2); run<2, 0>(2); run<1, 0>(4); run<2, 0>(4); run<4, 0>(4);
  run<1, 1>(1); run<2, 1>(1); run<4, 1>(1); run<2, 1>(4); run<4, 1>(4);
  return 0;
}
