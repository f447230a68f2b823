// Spatial-transformer front end of the PSNs (row a15): STNHead.forward (stn_head.py:92-106) and
// TPSSpatialTransformer.forward (tps_spatial_transformer.py:97-112).  Only reached in PSN train mode in the reference
// (tatt.py / tbsrn.py `if self.stn and self.training`); DPMN keeps the PSN in eval, so these are latency-size kernels
// (B x 16 x 64 images, 20 control points) written for correctness and few launches, not for a roofline.
//   k_maxpool_nhwc   MaxPool2d(k = stride) over NHWC with the producer's BatchNorm affine + ReLU applied on load
//   k_stn_fc         Linear(512,512) + BatchNorm1d (batch or running statistics) + ReLU, then Linear(512, 2*N)(0.1 * feat)
//   k_tps_sample     control points -> TPS mapping -> sampling grid (clamped to the image) -> bilinear grid_sample
#include "common.h"

namespace {

__global__ void k_maxpool_nhwc(const float* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
                               float* __restrict__ y, int B, int H, int W, int C, int kh, int kw) {
  const int Ho = H / kh, Wo = W / kw, C4 = C / 4;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)B * Ho * Wo * C4) return;
  const int c = (idx % C4) * 4;
  const long p = idx / C4;
  const int ox = p % Wo, oy = (p / Wo) % Ho, b = p / ((long)Wo * Ho);
  float4 s = make_float4(1.f, 1.f, 1.f, 1.f), t = make_float4(0.f, 0.f, 0.f, 0.f);
  if (scale) { s = *reinterpret_cast<const float4*>(scale + c); t = *reinterpret_cast<const float4*>(shift + c); }
  float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
  for (int dy = 0; dy < kh; ++dy)
    for (int dx = 0; dx < kw; ++dx) {
      float4 v = *reinterpret_cast<const float4*>(x + (((size_t)b * H + oy * kh + dy) * W + ox * kw + dx) * C + c);
      if (scale) {     // BatchNorm affine, then ReLU (conv3x3_block, stn_head.py:13-22)
        v.x = fmaxf(v.x * s.x + t.x, 0.f); v.y = fmaxf(v.y * s.y + t.y, 0.f);
        v.z = fmaxf(v.z * s.z + t.z, 0.f); v.w = fmaxf(v.w * s.w + t.w, 0.f);
      }
      m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
    }
  *reinterpret_cast<float4*>(y + (((size_t)b * Ho + oy) * Wo + ox) * C + c) = m;
}

// One block of 512 threads; thread j owns column j of fc1.  x: last conv block's raw NHWC output (B, 1, W2, C) with
// F = W2*C = 512 features; the reference flattens NCHW, feature f = c*W2 + w (stn_head.py:95).
constexpr int STN_F = 512, STN_ROWS = 16;
__global__ __launch_bounds__(512) void k_stn_fc(const float* __restrict__ x, const float* __restrict__ in_scale,
                                                 const float* __restrict__ in_shift, int W2, const float* __restrict__ w1t,
                                                 const float* __restrict__ b1, const float* __restrict__ gamma,
                                                 const float* __restrict__ beta, float* __restrict__ running_mean,
                                                 float* __restrict__ running_var, int training, float momentum, float eps,
                                                 const float* __restrict__ w2, const float* __restrict__ b2,
                                                 float* __restrict__ feat, float* __restrict__ ctrl, int B, int n_out) {
  __shared__ float fs[STN_ROWS * STN_F];
  const int j = threadIdx.x;
  const int C = STN_F / W2;
  float s1 = 0.f, s2 = 0.f;
  for (int r0 = 0; r0 < B; r0 += STN_ROWS) {
    const int rows = min(STN_ROWS, B - r0);
    __syncthreads();
    for (int i = j; i < rows * STN_F; i += 512) {
      const int r = i / STN_F, f = i % STN_F, c = f / W2, w = f % W2;
      float v = x[((size_t)(r0 + r) * W2 + w) * C + c];
      if (in_scale) v = fmaxf(v * in_scale[c] + in_shift[c], 0.f);
      fs[i] = v;
    }
    __syncthreads();
    float acc[STN_ROWS];
#pragma unroll
    for (int r = 0; r < STN_ROWS; ++r) acc[r] = 0.f;
    for (int k = 0; k < STN_F; ++k) {
      const float w = w1t[(size_t)k * STN_F + j];
#pragma unroll
      for (int r = 0; r < STN_ROWS; ++r) acc[r] += w * fs[r * STN_F + k];     // rows >= `rows` read stale LDS, never stored
    }
    for (int r = 0; r < rows; ++r) {
      const float pre = acc[r] + b1[j];
      feat[(size_t)(r0 + r) * STN_F + j] = pre;
      s1 += pre; s2 += pre * pre;
    }
  }
  float mean, var;
  if (training) {       // BatchNorm1d over the batch (biased variance normalises, unbiased feeds the running estimate)
    mean = s1 / (float)B;
    var = fmaxf(s2 / (float)B - mean * mean, 0.f);
    running_mean[j] = (1.f - momentum) * running_mean[j] + momentum * mean;
    running_var[j] = (1.f - momentum) * running_var[j] + momentum * var * ((float)B / (float)(B - 1));
  } else {
    mean = running_mean[j];
    var = running_var[j];
  }
  const float sc = gamma[j] / sqrtf(var + eps), sh = beta[j] - mean * sc;
  for (int b = 0; b < B; ++b) feat[(size_t)b * STN_F + j] = fmaxf(feat[(size_t)b * STN_F + j] * sc + sh, 0.f);
  __threadfence_block();
  __syncthreads();
  for (int i = j; i < B * n_out; i += 512) {     // stn_fc2(0.1 * img_feat), stn_head.py:100
    const int b = i / n_out, o = i % n_out;
    float a = 0.f;
    for (int k = 0; k < STN_F; ++k) a += (0.1f * feat[(size_t)b * STN_F + k]) * w2[(size_t)o * STN_F + k];
    ctrl[i] = a + b2[o];
  }
}

// grid (pixel tiles, B).  mapping = inverse_kernel[:, :N] . ctrl  (the 3 padding rows of Y are zero, tps:103-104);
// source = coord_repr . mapping (105); grid = 2*clamp(source, 0, 1) - 1 (108-110); F.grid_sample bilinear / zeros /
// align_corners=False (the torch >= 1.3 default the reference runs under).
constexpr int TPS_MAXN = 64;
__global__ __launch_bounds__(256) void k_tps_sample(const float* __restrict__ img, const float* __restrict__ ctrl,
                                                     const float* __restrict__ inv_kernel, const float* __restrict__ coord_repr,
                                                     float* __restrict__ out, float* __restrict__ src_coord, int Cc, int Hin,
                                                     int Win, int Hout, int Wout, int N) {
  __shared__ float map_s[(TPS_MAXN + 3) * 2];
  const int b = blockIdx.y, N3 = N + 3;
  for (int i = threadIdx.x; i < N3 * 2; i += 256) {
    const int r = i / 2, d = i % 2;
    float a = 0.f;
    for (int k = 0; k < N; ++k) a += inv_kernel[r * N3 + k] * ctrl[((size_t)b * N + k) * 2 + d];
    map_s[i] = a;
  }
  __syncthreads();
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= Hout * Wout) return;
  float sx = 0.f, sy = 0.f;
  for (int k = 0; k < N3; ++k) {
    const float r = coord_repr[(size_t)p * N3 + k];
    sx += r * map_s[2 * k];
    sy += r * map_s[2 * k + 1];
  }
  src_coord[((size_t)b * Hout * Wout + p) * 2] = sx;
  src_coord[((size_t)b * Hout * Wout + p) * 2 + 1] = sy;
  const float gx = 2.0f * fminf(fmaxf(sx, 0.f), 1.f) - 1.0f, gy = 2.0f * fminf(fmaxf(sy, 0.f), 1.f) - 1.0f;
  const float ix = ((gx + 1.0f) * (float)Win - 1.0f) * 0.5f, iy = ((gy + 1.0f) * (float)Hin - 1.0f) * 0.5f;
  const float fx = floorf(ix), fy = floorf(iy);
  const int x0 = (int)fx, y0 = (int)fy;
  const float wx1 = ix - fx, wy1 = iy - fy, wx0 = 1.0f - wx1, wy0 = 1.0f - wy1;
  const bool vx0 = x0 >= 0 && x0 < Win, vx1 = x0 + 1 >= 0 && x0 + 1 < Win, vy0 = y0 >= 0 && y0 < Hin, vy1 = y0 + 1 >= 0 && y0 + 1 < Hin;
  for (int ch = 0; ch < Cc; ++ch) {
    const float* q = img + ((size_t)b * Cc + ch) * Hin * Win;
    float v = 0.f;
    if (vy0 && vx0) v += q[y0 * Win + x0] * wy0 * wx0;
    if (vy0 && vx1) v += q[y0 * Win + x0 + 1] * wy0 * wx1;
    if (vy1 && vx0) v += q[(y0 + 1) * Win + x0] * wy1 * wx0;
    if (vy1 && vx1) v += q[(y0 + 1) * Win + x0 + 1] * wy1 * wx1;
    out[((size_t)b * Cc + ch) * Hout * Wout + p] = v;
  }
}

}  // namespace

extern "C" {

int dpmn_maxpool_f32(const float* x, const float* scale, const float* shift, float* y, int B, int H, int W, int C, int kh, int kw,
                     dpmn_stream_t stream) {
  DPMN_REQUIRE(x && y && B > 0 && C % 4 == 0 && kh > 0 && kw > 0 && H % kh == 0 && W % kw == 0, "maxpool: NHWC, C % 4 == 0, window must tile the plane");
  DPMN_REQUIRE((scale == nullptr) == (shift == nullptr), "maxpool: scale and shift come together");
  const long total = (long)B * (H / kh) * (W / kw) * (C / 4);
  hipLaunchKernelGGL(k_maxpool_nhwc, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), x, scale, shift, y, B, H, W, C, kh, kw);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

int dpmn_stn_fc_f32(const float* x, const float* in_scale, const float* in_shift, int W2, const float* w1t, const float* b1,
                    const float* bn_gamma, const float* bn_beta, float* running_mean, float* running_var, int training,
                    float momentum, float eps, const float* w2, const float* b2, float* img_feat, float* ctrl, int B, int n_out,
                    dpmn_stream_t stream) {
  DPMN_REQUIRE(x && w1t && b1 && bn_gamma && bn_beta && running_mean && running_var && w2 && b2 && img_feat && ctrl, "stn_fc: null pointer");
  DPMN_REQUIRE(W2 > 0 && STN_F % W2 == 0 && B > 0 && n_out > 0, "stn_fc: 512 input features as (W2, 512/W2)");
  DPMN_REQUIRE(!training || B > 1, "stn_fc: BatchNorm1d in training needs more than one sample (torch raises the same)");
  hipLaunchKernelGGL(k_stn_fc, dim3(1), dim3(512), 0, as_stream(stream), x, in_scale, in_shift, W2, w1t, b1, bn_gamma, bn_beta,
                     running_mean, running_var, training, momentum, eps, w2, b2, img_feat, ctrl, B, n_out);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

int dpmn_tps_sample_f32(const float* img, const float* ctrl, const float* inverse_kernel, const float* coord_repr, float* out,
                        float* src_coord, int B, int C, int Hin, int Win, int Hout, int Wout, int N, dpmn_stream_t stream) {
  DPMN_REQUIRE(img && ctrl && inverse_kernel && coord_repr && out && src_coord, "tps_sample: null pointer");
  DPMN_REQUIRE(B > 0 && C > 0 && N > 0 && N <= TPS_MAXN, "tps_sample: at most 64 control points");
  hipLaunchKernelGGL(k_tps_sample, dim3(cdiv(Hout * Wout, 256), B), dim3(256), 0, as_stream(stream), img, ctrl, inverse_kernel,
                     coord_repr, out, src_coord, C, Hin, Win, Hout, Wout, N);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

}  // extern "C"
