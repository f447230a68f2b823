// ds_read_b128 / ds_write_b64 rate as a function of the LDS row stride and the lane -> address map of the MFMA operand reads
// (lane (lr = l & 15, kq = l >> 4) reads 16 bytes at row lr, chunk kq): cycles per wave-instruction, one wave per CU alone and four
// waves per CU together.  hipcc --offload-arch=gfx950 -O3 tools/ubench/lds_stride.hip -o /tmp/lds_stride && /tmp/lds_stride
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// MODE 0: b128 reads, address = lr * stride + kq * 16        (the operand-fragment read of every MFMA kernel)
// MODE 1: b128 reads, chunk swizzled: (kq ^ (lr >> 1 & 3)) * 16, stride as given
// MODE 2: b64 writes of the staging pattern: thread t writes 8 bytes at row t >> 3, byte (t & 7) * 8   (x3 plane staging)
// MODE 3: b128 reads, 16 rows x 4 chunks but rows = 4 * kq + (lr & 3), chunk = lr >> 2   (transposed lane map)
template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned* out, long long* cyc, int stride, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  for (int i = threadIdx.x; i < 65536 / 4; i += blockDim.x) reinterpret_cast<unsigned*>(lds)[i] = i;
  __syncthreads();
  const int lane = threadIdx.x & 63, lr = lane & 15, kq = lane >> 4, wave = threadIdx.x >> 6;
  int addr;
  if (MODE == 0) addr = lr * stride + kq * 16;
  else if (MODE == 1) addr = lr * stride + ((kq ^ ((lr >> 1) & 3)) * 16);
  else if (MODE == 2) addr = (lane >> 3) * stride + (lane & 7) * 8;
  else addr = (4 * kq + (lr & 3)) * stride + (lr >> 2) * 16;
  addr += wave * 16 * stride;
  u32x4 acc = {0, 0, 0, 0};
  __builtin_amdgcn_s_barrier();
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 2) {
#pragma unroll
      for (int u = 0; u < 8; ++u) *reinterpret_cast<u32x2*>(lds + addr + (u & 1) * 8 * stride) = (u32x2){(unsigned)it, (unsigned)u};
    } else {
      u32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const u32x4*>(lds + addr + (u & 1) * 16 * stride);      // eight reads in flight
#pragma unroll
      for (int u = 0; u < 8; ++u) asm volatile("" :: "v"(v[u]));      // consumed without vector ALU work
      acc[0] += v[0][0];
    }
    asm volatile("" ::: "memory");
  }
  const long long t1 = clock64();
  if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
  out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

template <int MODE>
void run(const char* what, int threads) {
  unsigned* d; long long* c;
  (void)hipMalloc(&d, 256 * 256 * 4); (void)hipMalloc(&c, 256 * 4 * 8);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  const int iters = 2000;
  printf("%s, %d wave(s) per CU:", what, threads / 64);
  for (int stride : {64, 80, 96, 112, 128, 144, 160, 192, 208, 224, 400, 416}) {
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 65536, 0, d, c, stride, iters);
    (void)hipDeviceSynchronize();
    long long h[1024];
    (void)hipMemcpy(h, c, 256 * 4 * 8, hipMemcpyDeviceToHost);
    double s = 0; int n = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < threads / 64; ++w) { s += (double)h[b * 4 + w]; ++n; }
    printf("  %d:%.1f", stride, s / n / (iters * 8.0));
  }
  printf("   (clock64 ticks per wave instruction)\n");
  (void)hipFree(d); (void)hipFree(c);
}

int main() {
  for (int th : {64, 256}) {
    run<0>("b128 read, row lr, chunk kq", th);
    run<1>("b128 read, chunk kq ^ (lr >> 1 & 3)", th);
    run<3>("b128 read, row 4 kq + (lr & 3), chunk lr >> 2", th);
    run<2>("b64 write, 8 lanes per row", th);
  }
  return 0;
}
