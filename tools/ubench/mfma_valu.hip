// Does vector-ALU work hide in the shadow of the f32 MFMAs?  N_V independent v_fma_f32 (or v_exp_f32) are placed behind every
// v_mfma_f32_16x16x4_f32 / v_mfma_f32_32x32x2_f32 of an MFMA-bound loop (4 independent accumulators), at 1 and 2 waves per SIMD.
// If the matrix pipe ran beside the vector ALU the time would stay at the MFMA-only time until the shadow (32 / 64 cycles) is full.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NV, int BIG, int EXP>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f32x4 acc[4];
  f32x16 big[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) big[i][j] = 0.f;
  float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = a + j;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (BIG) big[i & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, big[i & 1], 0, 0, 0);
      else acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        if (EXP) v[j % 8] = __builtin_amdgcn_exp2f(v[j % 8]);
        else v[j % 8] = __builtin_fmaf(v[j % 8], 0.999f, 0.001f);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if (NV) __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
  for (int j = 0; j < 16; ++j) s += big[0][j] + big[1][j];
#pragma unroll
  for (int j = 0; j < 8; ++j) s += v[j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NV, int BIG, int EXP>
void run(int blocks_per_cu) {
  float* d; (void)hipMalloc(&d, 256 * 8 * 256 * 4);
  const int iters = 50000;
  hipEvent_t s, e; (void)hipEventCreate(&s); (void)hipEventCreate(&e);
  int blocks = 256 * blocks_per_cu;
  hipLaunchKernelGGL((k<NV, BIG, EXP>), dim3(blocks), dim3(256), 0, 0, d, iters);
  (void)hipEventRecord(s);
  hipLaunchKernelGGL((k<NV, BIG, EXP>), dim3(blocks), dim3(256), 0, 0, d, iters);
  (void)hipEventRecord(e); (void)hipEventSynchronize(e);
  float ms; (void)hipEventElapsedTime(&ms, s, e);
  double flops = (BIG ? 4096.0 : 2048.0) * 4 * iters * 4.0 * blocks;
  printf("%s %s per MFMA = %2d, waves/SIMD=%d: %6.1f MFMA-TFLOP/s (%.2f ms)\n", BIG ? "32x32x2 " : "16x16x4 ", EXP ? "v_exp" : "v_fma", NV, blocks_per_cu,
         flops / ms / 1e9, ms);
  (void)hipFree(d);
}
int main() {
  for (int b = 1; b <= 2; ++b) {
    run<0, 0, 0>(b); run<1, 0, 0>(b); run<2, 0, 0>(b); run<4, 0, 0>(b); run<6, 0, 0>(b); run<8, 0, 0>(b); run<4, 0, 1>(b);
    run<0, 1, 0>(b); run<2, 1, 0>(b); run<4, 1, 0>(b); run<8, 1, 0>(b); run<12, 1, 0>(b); run<16, 1, 0>(b); run<8, 1, 1>(b);
  }
  return 0;
}
