// Does independent VALU work of the same wave (and of other waves of the SIMD) run under the fp32 MFMA's 64 cycles?
// One 32x32x2 f32 MFMA followed by NV independent v_fma_f32 per iteration, 1 or 2 waves per SIMD; cycles per iteration.
// hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap mfma_valu_overlap.hip && ./mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
template <int NV, bool MF>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f16v c = f16v(float(threadIdx.x));
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = float(threadIdx.x + i);
  float a = float(threadIdx.x) * 1e-3f, b = 1.0f + float(blockIdx.x) * 1e-6f;
  for (int it = 0; it < iters; ++it) {
    if (MF) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i & 7] = __builtin_fmaf(v[i & 7], b, a);   // 8 independent chains
    __builtin_amdgcn_sched_barrier(0);
  }
  float s = 0.f;
  for (int j = 0; j < 16; ++j) s += c[j];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NV, bool MF>
void run(int blocks_per_cu) {
  const int blocks = 256 * blocks_per_cu, iters = 20000;
  float* out; hipMalloc(&out, size_t(blocks) * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NV, MF>), dim3(blocks), dim3(256), 0, 0, out, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NV, MF>), dim3(blocks), dim3(256), 0, 0, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // per SIMD: blocks_per_cu waves, each `iters` iterations
  printf("mfma=%d valu/iter=%2d waves/SIMD=%d: %.1f ns per iteration and wave, %.1f ns per iteration of the SIMD\n", int(MF), NV,
         blocks_per_cu, ms * 1e6 / iters, ms * 1e6 / iters / blocks_per_cu);
  hipFree(out);
}
int main() {
  run<0, true>(1); run<8, true>(1); run<16, true>(1); run<32, true>(1);
  run<8, false>(1); run<16, false>(1); run<32, false>(1);
  run<0, true>(2); run<8, true>(2); run<16, true>(2); run<32, true>(2);
  run<16, false>(2); run<32, false>(2);
  return 0;
}
