// TBSRN FeatureEnhancer kernels (model/tbsrn.py:76-92, the PSN of config 3):
//   k_mha32      MultiHeadedAttention core (tbsrn.py:110-150): softmax(Q K^T / sqrt(32)) V over all L = 1024 positions of an
//                image, 4 heads x 32 dims, flash style (scores never leave registers), fp32 MFMA 16x16x4
//   k_ln_std     the file's own LayerNorm (tbsrn.py:23-36): a * (x - mean) / (std_unbiased + eps) + b
// The linears around them are the whole-K GEMMs of gemm.hip (K = 128), the convs are conv.hip.
#include "common.h"

namespace {

// qkv: (B*L, 3*H*32) rows = [q | k | v], head h at columns h*32 of each third.  out: (B*L, H*32).
// Block = 64 queries of one (image, head); wave = 16 queries.  Per 64-key tile (K as [key][d], V transposed [d][key] in LDS):
//   S^T  = K Q^T    : A = K rows (key), B = Q rows (query)  -> lane (lr,kq) holds S[query lr][keys 16 kt + 4 kq + r]
//   O^T += V^T P^T  : A = V^T rows (d), B = P -- the D layout of S^T is exactly the B-operand layout (k = key), no shuffle
// Online softmax per query column: running max / sum live replicated in the 4 kq lanes of a query.
template <int D>      // head dim: 32 (TBSRN FeatureEnhancer) or 64 (VisionLAN encoder, modules.py:43-81)
__global__ __launch_bounds__(256) void k_mha(const float* __restrict__ qkv, float* __restrict__ out, int L, int H, float scale) {
  constexpr int LDKs = D + 4, LDV = 68, DC = D / 16;
  __shared__ __attribute__((aligned(16))) float Ks[64 * LDKs];
  __shared__ __attribute__((aligned(16))) float Vt[D * LDV];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, kq = lane >> 4;
  const int h = blockIdx.y, b = blockIdx.z;
  const int ld = 3 * H * D;
  const float* base = qkv + (size_t)b * L * ld;
  const int qrow = blockIdx.x * 64 + wave * 16 + lr;
  f32x4 qf[DC];
#pragma unroll
  for (int c = 0; c < DC; ++c) {
    qf[c] = *reinterpret_cast<const f32x4*>(base + (size_t)qrow * ld + h * D + c * 16 + kq * 4);
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[c][s] *= scale;
  }
  float m_run = -1e30f, l_run = 0.f;
  f32x4 o[DC];
#pragma unroll
  for (int c = 0; c < DC; ++c) o[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < L; k0 += 64) {
    __syncthreads();
#pragma unroll
    for (int p = 0; p < D / 16; ++p) {
      const int idx = tid + p * 256;
      const int key = idx / (D / 4), c4 = (idx % (D / 4)) * 4;
      const float* row = base + (size_t)(k0 + key) * ld + h * D + c4;
      const float4 kv = *reinterpret_cast<const float4*>(row + H * D);
      const float4 vv = *reinterpret_cast<const float4*>(row + 2 * H * D);
      *reinterpret_cast<float4*>(&Ks[key * LDKs + c4]) = kv;
      Vt[(c4 + 0) * LDV + key] = vv.x; Vt[(c4 + 1) * LDV + key] = vv.y;
      Vt[(c4 + 2) * LDV + key] = vv.z; Vt[(c4 + 3) * LDV + key] = vv.w;
    }
    __syncthreads();
    f32x4 s[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      s[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < DC; ++c) {
        const f32x4 kf = *reinterpret_cast<const f32x4*>(&Ks[(kt * 16 + lr) * LDKs + c * 16 + kq * 4]);
#pragma unroll
        for (int st = 0; st < 4; ++st) s[kt] = mfma16(kf[st], qf[c][st], s[kt]);
      }
    }
    float mx = -1e30f;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kt][r]);
    mx = fmaxf(mx, xshfl<16>(mx));
    mx = fmaxf(mx, xshfl<32>(mx));
    const float m_new = fmaxf(m_run, mx);
    const float corr = __expf(m_run - m_new);
    float psum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float pv = __expf(s[kt][r] - m_new); s[kt][r] = pv; psum += pv; }
    psum += xshfl<16>(psum);
    psum += xshfl<32>(psum);
    l_run = l_run * corr + psum;
    m_run = m_new;
#pragma unroll
    for (int dt = 0; dt < DC; ++dt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) o[dt][r] *= corr;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        const f32x4 vf = *reinterpret_cast<const f32x4*>(&Vt[(dt * 16 + lr) * LDV + kt * 16 + kq * 4]);
#pragma unroll
        for (int st = 0; st < 4; ++st) o[dt] = mfma16(vf[st], s[kt][st], o[dt]);
      }
    }
  }
  const float inv = 1.0f / l_run;
  float* orow = out + ((size_t)b * L + qrow) * (H * D) + h * D + kq * 4;
#pragma unroll
  for (int dt = 0; dt < DC; ++dt)
    *reinterpret_cast<float4*>(orow + dt * 16) = make_float4(o[dt][0] * inv, o[dt][1] * inv, o[dt][2] * inv, o[dt][3] * inv);
}

// one wave per row of C = 64 * V floats
template <int V>
__global__ __launch_bounds__(256) void k_ln_std(const float* __restrict__ x, const float* __restrict__ a2, const float* __restrict__ b2,
                                                float eps, float* __restrict__ y, long M) {
  constexpr int C = 64 * V;
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  float v[V];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < V; ++i) { v[i] = x[row * C + lane * V + i]; s += v[i]; }
  const float mean = wave_sum(s) * (1.0f / C);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < V; ++i) { v[i] -= mean; q += v[i] * v[i]; }
  const float stdv = sqrtf(wave_sum(q) * (1.0f / (C - 1)));     // torch.std: unbiased
  const float inv = 1.0f / (stdv + eps);
#pragma unroll
  for (int i = 0; i < V; ++i) y[row * C + lane * V + i] = a2[lane * V + i] * v[i] * inv + b2[lane * V + i];
}

}  // namespace

extern "C" {

int dpmn_mha64_f32(const float* qkv, float* out, int B, int L, int heads, float scale, dpmn_stream_t stream) {
  DPMN_REQUIRE(qkv && out && B > 0 && heads > 0 && L > 0 && L % 64 == 0, "mha64: L must be a multiple of 64 (d_k is 64)");
  ProfScope prof(PT_MHA32, as_stream(stream), 4.0 * L * (double)L * 64 * heads * B, 4.0 * 4 * 64 * heads * (double)L * B);
  hipLaunchKernelGGL((k_mha<64>), dim3(L / 64, heads, B), dim3(256), 0, as_stream(stream), qkv, out, L, heads, scale);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

int dpmn_mha32_f32(const float* qkv, float* out, int B, int L, int heads, float scale, dpmn_stream_t stream) {
  DPMN_REQUIRE(qkv && out && B > 0 && heads > 0 && L > 0 && L % 64 == 0, "mha32: L must be a multiple of 64 (d_k is 32)");
  ProfScope prof(PT_MHA32, as_stream(stream), 4.0 * L * (double)L * 32 * heads * B, 4.0 * 4 * 32 * heads * (double)L * B);
  hipLaunchKernelGGL((k_mha<32>), dim3(L / 64, heads, B), dim3(256), 0, as_stream(stream), qkv, out, L, heads, scale);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

int dpmn_layernorm_std_f32(const float* x, const float* a2, const float* b2, float eps, float* y, long M, int C,
                           dpmn_stream_t stream) {
  DPMN_REQUIRE(x && a2 && b2 && y && M > 0, "layernorm_std: bad arguments");
  const unsigned blocks = (unsigned)((M + 3) / 4);
  if (C == 128) hipLaunchKernelGGL((k_ln_std<2>), dim3(blocks), dim3(256), 0, as_stream(stream), x, a2, b2, eps, y, M);
  else if (C == 64) hipLaunchKernelGGL((k_ln_std<1>), dim3(blocks), dim3(256), 0, as_stream(stream), x, a2, b2, eps, y, M);
  else if (C == 256) hipLaunchKernelGGL((k_ln_std<4>), dim3(blocks), dim3(256), 0, as_stream(stream), x, a2, b2, eps, y, M);
  else return dpmn_set_error(DPMN_ERR_ARG, "layernorm_std: C must be 64, 128 or 256");
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

}  // extern "C"
