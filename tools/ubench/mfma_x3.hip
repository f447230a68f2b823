// What rate does the bf16x3 inner loop reach on its own?  v_mfma_f32_16x16x32_bf16 in the six-term pattern of the x3 kernels (wave tile
// MT x NT accumulators, three planes per operand): (a) operands constant in registers -- the pure MFMA ceiling of the pattern;
// (b) operand fragments re-read from LDS every chunk as the kernels do (ds_read_b128, 96-byte rows), no barrier; (c) as (b) with the
// two barriers per chunk of the single-buffer kernels.  1 and 2 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_x3.hip -o /tmp/mfma_x3 && /tmp/mfma_x3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int LDB = 48;      // halves per row (96 bytes)

template <int MT, int NT, int MODE>
__global__ __launch_bounds__(256, 2) void k(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned short X[3 * 128 * LDB], W[3 * 128 * LDB];
  for (int i = threadIdx.x; i < 3 * 128 * LDB; i += 256) { X[i] = (unsigned short)(0x3f80 + (i & 7)); W[i] = (unsigned short)(0x3f80 + (i & 3)); }
  __syncthreads();
  const int lane = threadIdx.x & 63, lr = lane & 15, kq = lane >> 4, wave = threadIdx.x >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  f32x4 acc[NT][MT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};
  const unsigned short* xa = X + ((wm * MT * 16) % 128 + lr) * LDB + kq * 8;
  const unsigned short* wa = W + ((wn * NT * 16) % 128 + lr) * LDB + kq * 8;
  bf16x8 w0[NT], w1[NT], w2[NT], x0[MT], x1[MT], x2[MT];
  auto frags = [&]() {
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      w0[i] = *reinterpret_cast<const bf16x8*>(wa + (i * 16 % 64) * LDB);
      w1[i] = *reinterpret_cast<const bf16x8*>(wa + 128 * LDB + (i * 16 % 64) * LDB);
      w2[i] = *reinterpret_cast<const bf16x8*>(wa + 2 * 128 * LDB + (i * 16 % 64) * LDB);
    }
#pragma unroll
    for (int j = 0; j < MT; ++j) {
      x0[j] = *reinterpret_cast<const bf16x8*>(xa + (j * 16 % 64) * LDB);
      x1[j] = *reinterpret_cast<const bf16x8*>(xa + 128 * LDB + (j * 16 % 64) * LDB);
      x2[j] = *reinterpret_cast<const bf16x8*>(xa + 2 * 128 * LDB + (j * 16 % 64) * LDB);
    }
  };
  frags();
  for (int it = 0; it < iters; ++it) {
    if (MODE >= 1) { asm volatile("" ::: "memory"); frags(); }
#define TERM(XF, WF)                                                                          \
    _Pragma("unroll") for (int i = 0; i < NT; ++i)                                            \
      _Pragma("unroll") for (int j = 0; j < MT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WF[i], XF[j], acc[i][j], 0, 0, 0);
    TERM(x0, w2) TERM(x0, w1) TERM(x0, w0) TERM(x1, w1) TERM(x1, w0) TERM(x2, w0)
#undef TERM
    if (MODE == 2) { __syncthreads(); __syncthreads(); }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MT, int NT, int MODE>
void run(int bpc) {
  float* d; (void)hipMalloc(&d, 256 * 4 * 256 * 4);
  const int iters = 20000;
  hipEvent_t s, e; (void)hipEventCreate(&s); (void)hipEventCreate(&e);
  const int blocks = 256 * bpc;
  hipLaunchKernelGGL((k<MT, NT, MODE>), dim3(blocks), dim3(256), 0, 0, d, iters);
  (void)hipEventRecord(s);
  hipLaunchKernelGGL((k<MT, NT, MODE>), dim3(blocks), dim3(256), 0, 0, d, iters);
  (void)hipEventRecord(e); (void)hipEventSynchronize(e);
  float ms; (void)hipEventElapsedTime(&ms, s, e);
  const double mf = 6.0 * MT * NT * iters * 4.0 * blocks;            // MFMAs
  printf("wave tile %dx%d, %s, %d wave(s)/SIMD: %7.1f bf16 TFLOP/s = %6.1f fp32-equivalent (x3) TFLOP/s, %.1f cycles per MFMA per SIMD at 2.4 GHz\n",
         MT * 16, NT * 16, MODE == 0 ? "operands in registers" : MODE == 1 ? "fragments from LDS per chunk" : "fragments from LDS + 2 barriers per chunk",
         bpc, mf * 16384 / ms / 1e9, mf * 16384 / 6 / ms / 1e9, ms * 1e-3 * 2.4e9 / (mf / (256.0 * 4)));
  (void)hipFree(d);
}

int main() {
  for (int b = 1; b <= 2; ++b) {
    run<4, 4, 0>(b); run<4, 4, 1>(b); run<4, 4, 2>(b);
    run<2, 4, 0>(b); run<2, 4, 1>(b); run<2, 4, 2>(b);
    run<4, 3, 1>(b);
  }
  return 0;
}
