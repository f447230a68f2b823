// MFMA-only ceiling sweep for v_mfma_f32_16x16x4_f32: number of independent accumulators x waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f32x4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0, 0, 0, 0};
  float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
void run(int blocks_per_cu) {
  float* d; (void)hipMalloc(&d, 256 * 8 * 256 * 4);
  const int iters = 400000 / NACC;
  hipEvent_t s, e; (void)hipEventCreate(&s); (void)hipEventCreate(&e);
  int blocks = 256 * blocks_per_cu;
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, d, iters);   // warm-up / clock ramp
  (void)hipEventRecord(s);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, d, iters);
  (void)hipEventRecord(e); (void)hipEventSynchronize(e);
  float ms; (void)hipEventElapsedTime(&ms, s, e);
  double flops = 2048.0 * NACC * iters * 4.0 * blocks;
  printf("NACC=%2d waves/SIMD=%d: %6.1f TFLOP/s (%.2f ms)\n", NACC, blocks_per_cu, flops / ms / 1e9, ms);
  (void)hipFree(d);
}
int main() {
  for (int b = 1; b <= 3; ++b) { run<1>(b); run<2>(b); run<3>(b); run<4>(b); run<6>(b); run<8>(b); run<12>(b); run<16>(b); }
  return 0;
}
