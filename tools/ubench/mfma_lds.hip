// LDS-fed MFMA loop (no global traffic, no barriers): isolates the ds_read_b128 -> v_mfma_f32_16x16x4_f32 feed rate
// for the register-tile shapes used by the kernels: NT (A-operand tiles) x MT (B-operand tiles) per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int LDK = 36;
template <int NT, int MT>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) float A[64 * LDK], B[128 * LDK];
  for (int i = threadIdx.x; i < 64 * LDK; i += 256) A[i] = i * 1e-4f;
  for (int i = threadIdx.x; i < 128 * LDK; i += 256) B[i] = i * 2e-4f;
  __syncthreads();
  const int lane = threadIdx.x & 63, lr = lane & 15, kq = lane >> 4, wave = threadIdx.x >> 6;
  f32x4 acc[NT][MT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};
  const float* ap = A + lr * LDK + kq * 4;
  const float* bp = B + (wave * 16 + lr) * LDK + kq * 4;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int kc = 0; kc < 32; kc += 16) {
      f32x4 wf[NT], xf[MT];
#pragma unroll
      for (int i = 0; i < NT; ++i) wf[i] = *reinterpret_cast<const f32x4*>(ap + (i * 16 % 64) * LDK + kc);
#pragma unroll
      for (int j = 0; j < MT; ++j) xf[j] = *reinterpret_cast<const f32x4*>(bp + (j * 16 % 64) * LDK + kc);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
          for (int j = 0; j < MT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[i][s], xf[j][s], acc[i][j], 0, 0, 0);
    }
    asm volatile("" ::: "memory");
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NT, int MT>
void run(int bpc) {
  float* d; (void)hipMalloc(&d, 256 * 8 * 256 * 4);
  const int iters = 40000 / (NT * MT);
  hipEvent_t s, e; (void)hipEventCreate(&s); (void)hipEventCreate(&e);
  hipLaunchKernelGGL((k<NT, MT>), dim3(256 * bpc), dim3(256), 0, 0, d, iters);
  (void)hipEventRecord(s);
  hipLaunchKernelGGL((k<NT, MT>), dim3(256 * bpc), dim3(256), 0, 0, d, iters);
  (void)hipEventRecord(e); (void)hipEventSynchronize(e);
  float ms; (void)hipEventElapsedTime(&ms, s, e);
  double flops = 2048.0 * NT * MT * 8 * iters * 4.0 * 256 * bpc;
  printf("NT=%d MT=%d waves/SIMD=%d: %6.1f TFLOP/s\n", NT, MT, bpc, flops / ms / 1e9);
  (void)hipFree(d);
}
int main() {
  for (int b = 1; b <= 3; ++b) { run<3, 1>(b); run<3, 2>(b); run<4, 2>(b); run<4, 4>(b); run<2, 2>(b); run<1, 2>(b); }
  return 0;
}
