// Kernels of the in-loop text-prior path: the pieces of the VisionLAN recogniser
// (model/VisionLAN/VisionLAN.py, eval path) that are not a conv / GEMM / attention core, and the glyph-atlas composer
// that replaces utils/render_standard_text.py.  The reference runs this path per image on the host (recogniser at batch 1,
// pygame rasteriser, cv2.resize; interfaces/super_resolution.py:174-199); here the whole batch stays on the GPU.
//   k_vl_resize        parse_visionlan_data (base.py:473-478): uint8 quantise, bilinear 32x128 -> 64x256, /255, NHWC(4)
//   k_vl_tokens        MLM_VRM.forward 70-75 + PositionalEncoding (modules.py:19-20): NHWC feature map -> tokens (w*8+h) + table
//   k_vl_pp_pool       PP_layer.forward 168-171 + Prediction w_vrm (199-202): softmax over positions, pooled feature, logits
//   k_vl_decode        MLM_VRM.forward 107-126: argmax per step, length = first EOS step + 1 (else 25)
//   k_text_prior       glyph-atlas layout + bilinear stretch to the prior image (specified by oracle/visionlan.py)
#include "common.h"

namespace {

__global__ void k_vl_resize(const float* __restrict__ img, long img_stride, float* __restrict__ out, int B, int H, int W, int Ho,
                            int Wo) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)B * Ho * Wo) return;
  const int x = idx % Wo, y = (idx / Wo) % Ho, b = idx / ((long)Wo * Ho);
  const float sy = (y + 0.5f) * ((float)H / Ho) - 0.5f, sx = (x + 0.5f) * ((float)W / Wo) - 0.5f;
  const float y0f = floorf(sy), x0f = floorf(sx);
  const float fy = sy - y0f, fx = sx - x0f;
  const int y0 = min(max((int)y0f, 0), H - 1), y1 = min(max((int)y0f + 1, 0), H - 1);
  const int x0 = min(max((int)x0f, 0), W - 1), x1 = min(max((int)x0f + 1, 0), W - 1);
  float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
  float* op = &o.x;
  for (int c = 0; c < 3; ++c) {
    const float* p = img + (size_t)b * img_stride + (size_t)c * H * W;
    auto q = [&](int yy, int xx) { return floorf(fminf(fmaxf(p[yy * W + xx], 0.f), 1.f) * 255.0f); };
    const float v = (q(y0, x0) * (1.f - fx) + q(y0, x1) * fx) * (1.f - fy) + (q(y1, x0) * (1.f - fx) + q(y1, x1) * fx) * fy;
    op[c] = fminf(fmaxf(floorf(v + 0.5f), 0.f), 255.f) / 255.0f;
  }
  *reinterpret_cast<float4*>(out + idx * 4) = o;
}

// feat (B, Hf, Wf, C) NHWC -> tok (B, Wf*Hf, C) with token = w * Hf + h, + pos[token][c]
__global__ void k_vl_tokens(const float* __restrict__ feat, const float* __restrict__ pos, float* __restrict__ tok, int B, int Hf,
                            int Wf, int C) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;       // one float4
  const int C4 = C / 4;
  if (idx >= (long)B * Hf * Wf * C4) return;
  const int c4 = idx % C4;
  const long t = idx / C4;
  const int token = t % (Hf * Wf), b = t / (Hf * Wf);
  const int w = token / Hf, h = token % Hf;
  const float4 v = *reinterpret_cast<const float4*>(feat + (((size_t)b * Hf + h) * Wf + w) * C + c4 * 4);
  const float4 p = *reinterpret_cast<const float4*>(pos + (size_t)token * C + c4 * 4);
  *reinterpret_cast<float4*>(tok + (size_t)t * C + c4 * 4) = make_float4(v.x + p.x, v.y + p.y, v.z + p.z, v.w + p.w);
}

// one block per (image b, step n): p = softmax_l scores[b, l, n]; g = sum_l p_l enc[b, l, :]; logits[b, n, :] = g . Wvrm^T + bvrm
template <int L, int C>
__global__ __launch_bounds__(256) void k_vl_pp_pool(const float* __restrict__ scores, int ld_scores, const float* __restrict__ enc,
                                                     const float* __restrict__ w_vrm, const float* __restrict__ b_vrm,
                                                     float* __restrict__ logits, int n_steps, int n_class) {
  static_assert(L == 256 && C == 512, "VisionLAN: 256 positions x 512 channels");
  __shared__ float p[L];
  __shared__ float g[C];
  __shared__ float red[4];
  const int n = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const float s = scores[((size_t)b * L + tid) * ld_scores + n];
  float m = wave_max(s);
  if ((tid & 63) == 0) red[tid >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  const float e = expf(s - m);
  float t = wave_sum(e);
  if ((tid & 63) == 0) red[tid >> 6] = t;
  __syncthreads();
  t = (red[0] + red[1]) + (red[2] + red[3]);
  p[tid] = e / t;
  __syncthreads();
  const float* eb = enc + (size_t)b * L * C;
  float a0 = 0.f, a1 = 0.f;
  for (int l = 0; l < L; ++l) {
    const float pl = p[l];
    a0 += pl * eb[(size_t)l * C + tid];
    a1 += pl * eb[(size_t)l * C + tid + 256];
  }
  g[tid] = a0;
  g[tid + 256] = a1;
  __syncthreads();
  if (tid < n_class) {
    float acc = b_vrm[tid];
    const float* wr = w_vrm + (size_t)tid * C;
    for (int c = 0; c < C; ++c) acc += wr[c] * g[c];
    logits[((size_t)b * n_steps + n) * n_class + tid] = acc;
  }
}

// per image: cls[b][s] = argmax_c logits[b][s][c] (first maximum, like torch.topk(1)), s < 25; length = first s with class 0, + 1
__global__ void k_vl_decode(const float* __restrict__ logits, int* __restrict__ cls, int* __restrict__ length, int B, int n_steps,
                            int n_class, int max_len) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  int len = max_len;
  for (int s = max_len - 1; s >= 0; --s) {
    const float* row = logits + ((size_t)b * n_steps + s) * n_class;
    int best = 0;
    float bv = row[0];
    for (int c = 1; c < n_class; ++c)
      if (row[c] > bv) { bv = row[c]; best = c; }
    cls[b * max_len + s] = best;
    if (best == 0) len = s + 1;
  }
  length[b] = len;
}

// prior[b][case][y][x]: the string's glyph cells concatenated at their advances into a virtual GH x Wc canvas, sampled
// bilinearly at half-pixel centres with edge clamp, rounded to integers (oracle/visionlan.py compose_text_prior)
__global__ void k_text_prior(const int* __restrict__ cls, const int* __restrict__ length, const float* __restrict__ atlas,
                             const int* __restrict__ advance, float* __restrict__ out, int B, int max_len, int n_glyph, int GH, int GW,
                             int Ho, int Wo) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)B * 2 * Ho * Wo) return;
  const int x = idx % Wo, y = (idx / Wo) % Ho, cs = (idx / ((long)Wo * Ho)) % 2, b = idx / ((long)Wo * Ho * 2);
  const int* adv = advance + cs * n_glyph;
  const int len = length[b];
  int chars[32], nch = 0, Wc = 0;
  for (int s = 0; s < len && s < max_len; ++s) {
    const int c = cls[b * max_len + s];
    if (c > 0 && c < n_glyph) { chars[nch++] = c; Wc += adv[c]; }
  }
  if (nch == 0) { chars[0] = 0; nch = 1; Wc = adv[0]; }
  const float sy = (y + 0.5f) * ((float)GH / Ho) - 0.5f, sx = (x + 0.5f) * ((float)Wc / Wo) - 0.5f;
  const float y0f = floorf(sy), x0f = floorf(sx);
  const float fy = sy - y0f, fx = sx - x0f;
  const int y0 = min(max((int)y0f, 0), GH - 1), y1 = min(max((int)y0f + 1, 0), GH - 1);
  const int x0 = min(max((int)x0f, 0), Wc - 1), x1 = min(max((int)x0f + 1, 0), Wc - 1);
  auto sample = [&](int yy, int xx) {
    int k = 0, base = 0;
    while (k + 1 < nch && xx >= base + adv[chars[k]]) { base += adv[chars[k]]; ++k; }
    return atlas[(((size_t)cs * n_glyph + chars[k]) * GH + yy) * GW + (xx - base)];
  };
  const float v = (sample(y0, x0) * (1.f - fx) + sample(y0, x1) * fx) * (1.f - fy) + (sample(y1, x0) * (1.f - fx) + sample(y1, x1) * fx) * fy;
  out[idx] = fminf(fmaxf(floorf(v + 0.5f), 0.f), 255.f);
}

}  // namespace

extern "C" {

int dpmn_vl_resize_f32(const float* img, long img_stride, float* out_nhwc4, int B, int H, int W, int Ho, int Wo, dpmn_stream_t stream) {
  DPMN_REQUIRE(img && out_nhwc4 && B > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0, "vl_resize: bad arguments");
  const long n = (long)B * Ho * Wo;
  hipLaunchKernelGGL(k_vl_resize, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), img, img_stride, out_nhwc4, B, H, W, Ho, Wo);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

int dpmn_vl_tokens_f32(const float* feat_nhwc, const float* pos_table, float* tokens, int B, int Hf, int Wf, int C, dpmn_stream_t stream) {
  DPMN_REQUIRE(feat_nhwc && pos_table && tokens && B > 0 && C % 4 == 0, "vl_tokens: bad arguments");
  const long n = (long)B * Hf * Wf * (C / 4);
  hipLaunchKernelGGL(k_vl_tokens, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), feat_nhwc, pos_table, tokens, B, Hf, Wf, C);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

int dpmn_vl_pp_pool_f32(const float* scores, int ld_scores, const float* enc, const float* w_vrm, const float* b_vrm, float* logits,
                        int B, int L, int C, int n_steps, int n_class, dpmn_stream_t stream) {
  DPMN_REQUIRE(scores && enc && w_vrm && b_vrm && logits && B > 0, "vl_pp_pool: bad arguments");
  DPMN_REQUIRE(L == 256 && C == 512 && n_steps <= ld_scores && n_class <= 256, "vl_pp_pool: built for 256 positions x 512 channels (VisionLAN cfgs)");
  hipLaunchKernelGGL((k_vl_pp_pool<256, 512>), dim3(n_steps, B), dim3(256), 0, as_stream(stream), scores, ld_scores, enc, w_vrm, b_vrm, logits,
                     n_steps, n_class);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

int dpmn_vl_decode_i32(const float* logits, int* cls, int* length, int B, int n_steps, int n_class, int max_len, dpmn_stream_t stream) {
  DPMN_REQUIRE(logits && cls && length && B > 0 && max_len <= n_steps && max_len <= 32, "vl_decode: bad arguments");
  hipLaunchKernelGGL(k_vl_decode, dim3((B + 63) / 64), dim3(64), 0, as_stream(stream), logits, cls, length, B, n_steps, n_class, max_len);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

int dpmn_text_prior_compose_f32(const int* cls, const int* length, const float* atlas, const int* advance, float* out, int B, int max_len,
                                int n_glyph, int GH, int GW, int Ho, int Wo, dpmn_stream_t stream) {
  DPMN_REQUIRE(cls && length && atlas && advance && out && B > 0 && max_len <= 32 && n_glyph > 0, "text_prior_compose: bad arguments");
  const long n = (long)B * 2 * Ho * Wo;
  hipLaunchKernelGGL(k_text_prior, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), cls, length, atlas, advance, out, B,
                     max_len, n_glyph, GH, GW, Ho, Wo);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

}  // extern "C"
