// Image-space helpers of the SR path: branch-2 mask prior, eval blend, PSNR / SSIM metrics.
//   toMask             utils/util.py:27-35     (per image: uint8 cast, PIL 'L', mean threshold)
//   alpha blend        interfaces/super_resolution.py:449
//   calculate_psnr     utils/ssim_psnr.py:9-13 ; SSIM._ssim utils/ssim_psnr.py:28-48
// All tensors NCHW fp32 with an explicit per-image stride so a (B,4,H,W) tensor can be read as its first 3 channels.
#include "common.h"

namespace {

// one workgroup per image; L values kept in LDS; exact integer mean test  L*HW <= sum(L)
__global__ __launch_bounds__(256) void k_to_mask(const float* __restrict__ img, long img_stride, float* __restrict__ out, int HW) {
  extern __shared__ int Ls[];
  __shared__ long long wsum[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* p = img + (size_t)b * img_stride;
  long long s = 0;
  auto lum = [](float rf, float gf, float bf) {
    // ToPILImage: mul(255).byte() -> truncate toward zero, wrap modulo 256 (quirk Q13)
    const int r = ((int)(rf * 255.0f)) & 255, g = ((int)(gf * 255.0f)) & 255, bl = ((int)(bf * 255.0f)) & 255;
    return (r * 19595 + g * 38470 + bl * 7471 + 0x8000) >> 16;   // PIL ImagingConvert RGB -> L
  };
  int i = tid;
  for (; i + 768 < HW; i += 1024) {      // four pixels' twelve loads in flight (one block per image: the kernel is its load latency)
    float v[4][3];
#pragma unroll
    for (int u = 0; u < 4; ++u) { v[u][0] = p[i + 256 * u]; v[u][1] = p[HW + i + 256 * u]; v[u][2] = p[2 * HW + i + 256 * u]; }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int L = lum(v[u][0], v[u][1], v[u][2]);
      Ls[i + 256 * u] = L;
      s += L;
    }
  }
  for (; i < HW; i += 256) {
    const int L = lum(p[i], p[HW + i], p[2 * HW + i]);
    Ls[i] = L;
    s += L;
  }
  for (int o = 32; o > 0; o >>= 1) s += xshfl_v(s, o);
  if ((tid & 63) == 0) wsum[tid >> 6] = s;
  __syncthreads();
  const long long total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  float* o = out + (size_t)b * 3 * HW;
  for (int i = tid; i < HW; i += 256) {
    const float m = ((long long)Ls[i] * HW <= total) ? 1.0f : 0.0f;   // 255 where L <= mean, then ToTensor /255
    o[i] = m; o[HW + i] = m; o[2 * HW + i] = m;
  }
}

// GPU half of the TextZoom collate (dataset/dataset.py:1266-1319 resizeNormalize, 2007-2013 alignCollate_realWTLAMask): the host
// decodes and resizes with PIL (the reference's own library) and uploads the uint8 HWC pixels -- 3 bytes per pixel instead of the
// 16 of a float CHW + mask tensor -- and this kernel does ToTensor (/255, HWC -> CHW) and the mask channel:
// L = PIL's RGB -> L, threshold = mean(L) of the image (`np.array(mask).mean()`), channel 3 = (L > mean ? 0 : 255) / 255.
// One workgroup per image, L (0..255) kept in LDS as bytes (16 KB at the 64 x 256 HR size), exact integer mean test L * HW <= sum(L).
__global__ __launch_bounds__(256) void k_collate_u8(const unsigned char* __restrict__ img, float* __restrict__ out, int HW, int with_mask) {
  extern __shared__ unsigned char Lb[];
  __shared__ long long wsum[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  const unsigned char* p = img + (size_t)b * HW * 3;
  const int cout = with_mask ? 4 : 3;
  float* o = out + (size_t)b * cout * HW;
  long long s = 0;
  for (int i = tid; i < HW; i += 256) {
    const int r = p[3 * i], g = p[3 * i + 1], bl = p[3 * i + 2];
    o[i] = (float)r / 255.0f; o[HW + i] = (float)g / 255.0f; o[2 * HW + i] = (float)bl / 255.0f;   // ToTensor: .div(255)
    if (with_mask) {
      const int L = (r * 19595 + g * 38470 + bl * 7471 + 0x8000) >> 16;   // PIL ImagingConvert RGB -> L
      Lb[i] = (unsigned char)L;
      s += L;
    }
  }
  if (!with_mask) return;
  for (int o_ = 32; o_ > 0; o_ >>= 1) s += xshfl_v(s, o_);
  if ((tid & 63) == 0) wsum[tid >> 6] = s;
  __syncthreads();
  const long long total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  for (int i = tid; i < HW; i += 256) o[3 * HW + i] = ((long long)Lb[i] * HW <= total) ? 1.0f : 0.0f;
}

// torch_rotate_img (utils/util.py:37-58): theta = [[cos, sin*r, 0], [-sin/r, cos, 0]] with r = H/W + (2*rand-1)*off_range,
// F.affine_grid (align_corners=False: base x_j = (2j+1)/W - 1) and F.grid_sample (bilinear, zeros padding).
// One thread per output pixel, all channels (the sampling position is channel-independent).
__global__ void k_rotate_img(const float* __restrict__ img, const float* __restrict__ arc, const float* __restrict__ rand_off,
                             float off_range, float* __restrict__ out, int N, int Cc, int H, int W) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)N * H * W) return;
  const int x = idx % W, y = (idx / W) % H, n = idx / ((long)W * H);
  const float ratio = (float)H / (float)W + rand_off[n] * off_range * 2.0f - off_range;
  const float c = cosf(arc[n]), s = sinf(arc[n]);
  const float bx = (2.0f * x + 1.0f) / (float)W - 1.0f, by = (2.0f * y + 1.0f) / (float)H - 1.0f;
  const float gx = c * bx + (s * ratio) * by, gy = (-s / ratio) * bx + c * by;
  const float ix = ((gx + 1.0f) * (float)W - 1.0f) * 0.5f, iy = ((gy + 1.0f) * (float)H - 1.0f) * 0.5f;
  const float fx = floorf(ix), fy = floorf(iy);
  const int x0 = (int)fx, y0 = (int)fy;
  const float wx1 = ix - fx, wy1 = iy - fy, wx0 = 1.0f - wx1, wy0 = 1.0f - wy1;
  const bool vx0 = x0 >= 0 && x0 < W, vx1 = x0 + 1 >= 0 && x0 + 1 < W, vy0 = y0 >= 0 && y0 < H, vy1 = y0 + 1 >= 0 && y0 + 1 < H;
  for (int ch = 0; ch < Cc; ++ch) {
    const float* p = img + ((size_t)n * Cc + ch) * H * W;
    float v = 0.f;
    if (vy0 && vx0) v += p[y0 * W + x0] * wy0 * wx0;
    if (vy0 && vx1) v += p[y0 * W + x0 + 1] * wy0 * wx1;
    if (vy1 && vx0) v += p[(y0 + 1) * W + x0] * wy1 * wx0;
    if (vy1 && vx1) v += p[(y0 + 1) * W + x0 + 1] * wy1 * wx1;
    out[((size_t)n * Cc + ch) * H * W + y * W + x] = v;
  }
}

__global__ void k_blend(const float* __restrict__ a, long a_stride, const float* __restrict__ b, long b_stride,
                        float* __restrict__ out, float alpha, int chw, long total) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const long n = idx / chw, r = idx % chw;
  out[idx] = alpha * a[n * a_stride + r] + (1.0f - alpha) * b[n * b_stride + r];
}

// per-block partial sums of [squared error * 255^2, ssim map]; 11x11 Gaussian sigma 1.5 window, zero padding
__global__ __launch_bounds__(256) void k_psnr_ssim_partial(const float* __restrict__ x, long x_stride, const float* __restrict__ y,
                                                            long y_stride, float* __restrict__ partial, int C, int H, int W,
                                                            long total) {
  __shared__ float g[11];
  __shared__ float red[2][4];
  if (threadIdx.x < 11) {
    float s = 0.f;
    for (int i = 0; i < 11; ++i) s += expf(-(float)((i - 5) * (i - 5)) / 4.5f);
    g[threadIdx.x] = expf(-(float)((threadIdx.x - 5) * (threadIdx.x - 5)) / 4.5f) / s;
  }
  __syncthreads();
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  float se = 0.f, sm = 0.f;
  if (idx < total) {
    const int px = idx % W, py = (idx / W) % H, c = (idx / ((long)W * H)) % C;
    const long n = idx / ((long)W * H * C);
    const float* xp = x + n * x_stride + (size_t)c * H * W;
    const float* yp = y + n * y_stride + (size_t)c * H * W;
    const float d = xp[py * W + px] * 255.0f - yp[py * W + px] * 255.0f;
    se = d * d;
    float mu1 = 0.f, mu2 = 0.f, s11 = 0.f, s22 = 0.f, s12 = 0.f;
    for (int ky = 0; ky < 11; ++ky) {
      const int yy = py + ky - 5;
      if (yy < 0 || yy >= H) continue;
      for (int kx = 0; kx < 11; ++kx) {
        const int xx = px + kx - 5;
        if (xx < 0 || xx >= W) continue;
        const float wgt = g[ky] * g[kx];
        const float a = xp[yy * W + xx], b = yp[yy * W + xx];
        mu1 += wgt * a; mu2 += wgt * b; s11 += wgt * a * a; s22 += wgt * b * b; s12 += wgt * a * b;
      }
    }
    const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
    const float m11 = mu1 * mu1, m22 = mu2 * mu2, m12 = mu1 * mu2;
    sm = ((2.f * m12 + C1) * (2.f * (s12 - m12) + C2)) / ((m11 + m22 + C1) * ((s11 - m11) + (s22 - m22) + C2));
  }
  se = wave_sum(se); sm = wave_sum(sm);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = se; red[1][threadIdx.x >> 6] = sm; }
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[2 * blockIdx.x] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    partial[2 * blockIdx.x + 1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  }
}

__global__ void k_psnr_ssim_final(const float* __restrict__ partial, int nblocks, float inv_count, float* __restrict__ out2) {
  double se = 0.0, sm = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += 64) { se += partial[2 * i]; sm += partial[2 * i + 1]; }
  for (int o = 32; o > 0; o >>= 1) { se += xshfl_v(se, o); sm += xshfl_v(sm, o); }
  if (threadIdx.x == 0) {
    const double mse = se * inv_count;
    out2[0] = (float)(20.0 * log10(255.0 / sqrt(mse)));
    out2[1] = (float)(sm * inv_count);
  }
}

// device self-test of common.h xshfl<O>: every offset against __shfl_xor on a 2-D block (lane != threadIdx.x & 63 there)
__global__ void k_selftest_xshfl(unsigned* mismatches) {
  const unsigned tid = threadIdx.y * blockDim.x + threadIdx.x;
  const float v = __uint_as_float(0x3f800000u + 977u * tid + 13u * blockIdx.x);
  unsigned bad = 0;
  bad += xshfl<1>(v) != __shfl_xor(v, 1, 64);
  bad += xshfl<2>(v) != __shfl_xor(v, 2, 64);
  bad += xshfl<4>(v) != __shfl_xor(v, 4, 64);
  bad += xshfl<8>(v) != __shfl_xor(v, 8, 64);
  bad += xshfl<16>(v) != __shfl_xor(v, 16, 64);
  bad += xshfl<32>(v) != __shfl_xor(v, 32, 64);
  float s = v, s2 = v;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s2 += __shfl_xor(s2, o, 64);
  s = wave_sum(s);
  bad += s != s2;
  const double dv = (double)v * 1.000000119 + 1e-9 * tid;      // 64-bit payload: both words must travel
  bad += xshfl<4>(dv) != __shfl_xor(dv, 4, 64);
  bad += xshfl<16>(dv) != __shfl_xor(dv, 16, 64);
  bad += xshfl<32>(dv) != __shfl_xor(dv, 32, 64);
  if (bad) atomicAdd(mismatches, bad);
}

}  // namespace

namespace {
__global__ __launch_bounds__(256) void k_lds_poison(unsigned pattern, int words) {
  extern __shared__ unsigned lds_all[];
  for (int i = threadIdx.x; i < words; i += 256) lds_all[i] = pattern;
  __syncthreads();
  if (lds_all[(threadIdx.x * 97) % words] != pattern) __builtin_trap();      // (keeps the stores alive)
}
}  // namespace

extern "C" {

int dpmn_selftest_lds_poison(unsigned pattern, int blocks, dpmn_stream_t stream) {
  DPMN_REQUIRE(blocks > 0, "selftest_lds_poison: blocks > 0");
  constexpr int bytes = 160 * 1024;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lds_poison), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    attr_set = true;
  }
  hipLaunchKernelGGL(k_lds_poison, dim3(blocks), dim3(256), bytes, as_stream(stream), pattern, bytes / 4);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

int dpmn_selftest_xshfl(unsigned* mismatches_out) {
  DPMN_REQUIRE(mismatches_out, "selftest_xshfl: null pointer");
  unsigned* d = nullptr;
  if (hipMalloc(&d, sizeof(unsigned)) != hipSuccess) return dpmn_set_error(DPMN_ERR_LAUNCH, "selftest_xshfl: hipMalloc failed");
  (void)hipMemset(d, 0, sizeof(unsigned));
  hipLaunchKernelGGL(k_selftest_xshfl, dim3(4), dim3(32, 8), 0, nullptr, d);
  hipLaunchKernelGGL(k_selftest_xshfl, dim3(4), dim3(256, 1), 0, nullptr, d);
  const hipError_t e = hipMemcpy(mismatches_out, d, sizeof(unsigned), hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (e != hipSuccess) return dpmn_set_error(DPMN_ERR_LAUNCH, hipGetErrorString(e));
  return DPMN_OK;
}

int dpmn_to_mask_f32(const float* img, long img_stride, float* out, int B, int H, int W, dpmn_stream_t stream) {
  DPMN_REQUIRE(img && out && B > 0 && H * W * 4 <= 64 * 1024, "to_mask: bad arguments (image must fit 64 KB of LDS as ints)");
  hipLaunchKernelGGL(k_to_mask, dim3(B), dim3(256), (size_t)H * W * 4, as_stream(stream), img, img_stride, out, H * W);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

int dpmn_collate_u8_f32(const unsigned char* img, float* out, int B, int H, int W, int with_mask, dpmn_stream_t stream) {
  DPMN_REQUIRE(img && out && B > 0 && H > 0 && W > 0 && (size_t)H * W <= 60 * 1024, "collate_u8: bad arguments (one byte of luma per pixel must fit 60 KB of LDS)");
  hipLaunchKernelGGL(k_collate_u8, dim3(B), dim3(256), with_mask ? (((size_t)H * W + 15) & ~(size_t)15) : 0, as_stream(stream), img, out, H * W, with_mask);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

int dpmn_rotate_img_f32(const float* img, const float* arc, const float* rand_offs, float off_range, float* out, int N, int C,
                        int H, int W, dpmn_stream_t stream) {
  DPMN_REQUIRE(img && arc && rand_offs && out && N > 0 && C > 0 && H > 0 && W > 0, "rotate_img: bad arguments");
  const long total = (long)N * H * W;
  hipLaunchKernelGGL(k_rotate_img, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), img, arc, rand_offs,
                     off_range, out, N, C, H, W);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

int dpmn_blend_f32(const float* a, long a_stride, const float* b, long b_stride, float* out, float alpha, int B, int chw,
                   dpmn_stream_t stream) {
  DPMN_REQUIRE(a && b && out && B > 0 && chw > 0, "blend: bad arguments");
  const long total = (long)B * chw;
  hipLaunchKernelGGL(k_blend, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), a, a_stride, b, b_stride,
                     out, alpha, chw, total);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

size_t dpmn_psnr_ssim_workspace_bytes(int B, int C, int H, int W) {
  const long total = (long)B * C * H * W;
  return (size_t)((total + 255) / 256) * 2 * sizeof(float);
}

int dpmn_psnr_ssim_f32(const float* x, long x_stride, const float* y, long y_stride, float* out2, void* workspace, int B,
                       int C, int H, int W, dpmn_stream_t stream) {
  DPMN_REQUIRE(x && y && out2 && workspace && B > 0, "psnr_ssim: bad arguments");
  const long total = (long)B * C * H * W;
  const int nb = (int)((total + 255) / 256);
  hipLaunchKernelGGL(k_psnr_ssim_partial, dim3(nb), dim3(256), 0, as_stream(stream), x, x_stride, y, y_stride,
                     static_cast<float*>(workspace), C, H, W, total);
  DPMN_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_psnr_ssim_final, dim3(1), dim3(64), 0, as_stream(stream), static_cast<const float*>(workspace), nb,
                     1.0f / (float)total, out2);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

}  // extern "C"
