// Convolution weight gradients and train-mode BatchNorm plumbing for the NHWC implicit-GEMM convs (conv.hip).
//   k_conv_wgrad      dWp[co][k] += sum_pixels dY[pix][co] * pro(in)[pix @ tap(k)][ci(k)]   (packed (Cout, Kp) layout)
//   k_bn_finalize     per-channel (sum, sumsq) -> (scale, shift) for the consumers' affine-on-load + running stats
//   k_affine_act_bwd  G (+)= dA * act'(scale*r + shift)          (consumer-side activation backward)
//   k_bn_bwd_*        BatchNorm backward through batch statistics: dgamma, dbeta, d(raw conv output)
//   k_se_gate_bwd     backward of the CMM channel gate (cmm.py:135-147)
// Data gradients are ordinary convolutions of dY with re-packed weights and run through k_conv_igemm / k_conv_halo.
#include "common.h"

namespace {

struct WgArgs {
  const float* in[3];
  const float* in_scale[3];
  const float* in_shift[3];
  int cseg[3];
  int cin;
  int B, Hin, Win, KH, KW, stride, dil_y, dil_x, pad_y, pad_x, Hp, Wp;
  int Hout, Wout, ostep, ooy, oox;
  int pro_act;
  const float* dy;      // NHWC (B, Hout, Wout, Cout)
  int Cout, Kp;
  float* dwp;           // (Cout, Kp), accumulated with atomics
  int pix_per_block;
};

// Block 256 threads: 64 (co) x 64 (k) tile of dWp, loops over its pixel range in chunks of 32.
__global__ __launch_bounds__(256) void k_conv_wgrad(WgArgs a) {
  constexpr int BT = 64, BMc = 32, LD = BT + 4;
  __shared__ __attribute__((aligned(16))) float Ys[2][BMc * LD];
  __shared__ __attribute__((aligned(16))) float Xs[2][BMc * LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n_blk = blockIdx.x * BT, k_blk = blockIdx.y * BT;
  const int M = a.B * a.Hp * a.Wp;
  const int m_lo = blockIdx.z * a.pix_per_block;
  const int m_hi = min(M, m_lo + a.pix_per_block);
  const int c01 = a.cseg[0] + a.cseg[1];
  const int ktaps = a.KH * a.KW;
  // loader: 32 pixels x 16 float4 per operand -> 2 per thread; thread -> (row = tid>>4 (+16), col4 = tid & 15)
  const int lrow = tid >> 4, lc4 = (tid & 15) * 4;
  // k column of this thread is fixed: decode tap / channel / segment once
  const int kcol = k_blk + lc4;
  const int tap = kcol / a.cin, cch = kcol - tap * a.cin;
  const int ky = tap / a.KW, kx = tap - ky * a.KW;
  int seg = 0, cl = cch;
  if (cch >= c01) { seg = 2; cl = cch - c01; }
  else if (cch >= a.cseg[0]) { seg = 1; cl = cch - a.cseg[0]; }
  const float* src = a.in[seg];
  const int cs = a.cseg[seg];
  const bool kvalid = tap < ktaps;
  float4 s4 = make_float4(1.f, 1.f, 1.f, 1.f), h4 = make_float4(0.f, 0.f, 0.f, 0.f);
  const bool aff = a.in_scale[seg] != nullptr;
  if (aff && kvalid) { s4 = *reinterpret_cast<const float4*>(a.in_scale[seg] + cl); h4 = *reinterpret_cast<const float4*>(a.in_shift[seg] + cl); }
  float4 yr[2], xr[2];
  auto gload = [&](int m0) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int m = m0 + lrow + p * 16;
      float4 yv = make_float4(0.f, 0.f, 0.f, 0.f), xv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m < m_hi) {
        const int b = m / (a.Hp * a.Wp), rr = m % (a.Hp * a.Wp);
        const int py = rr / a.Wp, px = rr % a.Wp;
        const int oy = py * a.ostep + a.ooy, ox = px * a.ostep + a.oox;
        const int n = n_blk + lc4;
        if (n < a.Cout) {
          const float* yp = a.dy + (((size_t)b * a.Hout + oy) * a.Wout + ox) * a.Cout + n;
          if (n + 3 < a.Cout) yv = *reinterpret_cast<const float4*>(yp);
          else { float t4[4] = {0, 0, 0, 0}; for (int r = 0; r < 4; ++r) if (n + r < a.Cout) t4[r] = yp[r]; yv = make_float4(t4[0], t4[1], t4[2], t4[3]); }
        }
        const int iy = py * a.stride - a.pad_y + ky * a.dil_y, ix = px * a.stride - a.pad_x + kx * a.dil_x;
        if (kvalid && iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win) {
          xv = *reinterpret_cast<const float4*>(src + (((size_t)b * a.Hin + iy) * a.Win + ix) * cs + cl);
          if (aff) { xv.x = xv.x * s4.x + h4.x; xv.y = xv.y * s4.y + h4.y; xv.z = xv.z * s4.z + h4.z; xv.w = xv.w * s4.w + h4.w; }
          float v4[4] = {xv.x, xv.y, xv.z, xv.w};
          apply_act4(v4, a.pro_act, 0.f);
          xv = make_float4(v4[0], v4[1], v4[2], v4[3]);
        }
      }
      yr[p] = yv; xr[p] = xv;
    }
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      *reinterpret_cast<float4*>(&Ys[buf][(lrow + p * 16) * LD + lc4]) = yr[p];
      *reinterpret_cast<float4*>(&Xs[buf][(lrow + p * 16) * LD + lc4]) = xr[p];
    }
  };
  const int wn = wave & 1, wk = wave >> 1;
  const int lr = lane & 15, kq = lane >> 4;
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (m_lo < m_hi) { gload(m_lo); sstore(0); }
  __syncthreads();
  int buf = 0;
  for (int m0 = m_lo; m0 < m_hi; m0 += BMc) {
    if (m0 + BMc < m_hi) gload(m0 + BMc);
#pragma unroll
    for (int mc = 0; mc < BMc; mc += 16)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int row = mc + kq * 4 + s;
        float av[2], bv[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) av[i] = Ys[buf][row * LD + wn * 32 + i * 16 + lr];
#pragma unroll
        for (int j = 0; j < 2; ++j) bv[j] = Xs[buf][row * LD + wk * 32 + j * 16 + lr];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = mfma16(av[i], bv[j], acc[i][j]);
      }
    if (m0 + BMc < m_hi) sstore(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int k = k_blk + wk * 32 + j * 16 + lr;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n_blk + wn * 32 + i * 16 + kq * 4 + r;
        if (n < a.Cout && k < a.Kp) atomicAdd(a.dwp + (size_t)n * a.Kp + k, acc[i][j][r]);
      }
    }
}

// ---------------------------------------------------------------------------------- train-mode BatchNorm
// stats (32,2,C) = slotted (sum, sumsq) over `count` values per channel (accumulated by the conv epilogue)
__global__ void k_bn_finalize(const float* __restrict__ stats, const float* __restrict__ gamma, const float* __restrict__ beta,
                              float count, float eps, float momentum, float* __restrict__ scale, float* __restrict__ shift,
                              float* __restrict__ mean_out, float* __restrict__ rstd_out, float* __restrict__ running_mean,
                              float* __restrict__ running_var, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float s1 = 0.f, s2 = 0.f;
  for (int slot = 0; slot < 32; ++slot) { s1 += stats[(size_t)slot * 2 * C + c]; s2 += stats[(size_t)slot * 2 * C + C + c]; }   // STAT_SLOTS
  const float mean = s1 / count;
  float var = s2 / count - mean * mean;   // biased variance used for normalisation
  var = var > 0.f ? var : 0.f;
  const float rstd = 1.0f / sqrtf(var + eps);
  const float s = gamma[c] * rstd;
  scale[c] = s;
  shift[c] = beta[c] - mean * s;
  mean_out[c] = mean;
  rstd_out[c] = rstd;
  if (running_mean) {
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * var * (count / (count - 1.f));   // unbiased
  }
}
// G (+)= dA * act'(scale*r + shift) ; r, dA, G: (pixels, C) NHWC with optional strides for G
__global__ void k_affine_act_bwd(const float* __restrict__ dA, const float* __restrict__ r, const float* __restrict__ scale,
                                 const float* __restrict__ shift, int act, float* __restrict__ G, int accumulate, long pixels, int C) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= pixels * C) return;
  const int c = idx % C;
  const float z = scale ? r[idx] * scale[c] + shift[c] : r[idx];
  float g = dA[idx];
  switch (act) {
    case ACT_RELU: g = z > 0.f ? g : 0.f; break;
    case ACT_LEAKY02: g = z > 0.f ? g : 0.2f * g; break;
    case ACT_LEAKY001: g = z > 0.f ? g : 0.01f * g; break;
    default: break;
  }
  G[idx] = accumulate ? G[idx] + g : g;
}
// sums (2,C): sum G, sum G * xhat
__global__ __launch_bounds__(256) void k_bn_bwd_reduce(const float* __restrict__ G, const float* __restrict__ r,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd,
                                                        float* __restrict__ sums, long pixels, int C, int pix_per_block) {
  const long p0 = (long)blockIdx.x * pix_per_block;
  const long p1 = p0 + pix_per_block < pixels ? p0 + pix_per_block : pixels;
  for (int c = threadIdx.x; c < C; c += 256) {
    const float mu = mean[c], rs = rstd[c];
    float s1 = 0.f, s2 = 0.f;
    for (long p = p0; p < p1; ++p) {
      const float g = G[p * C + c];
      s1 += g;
      s2 += g * (r[p * C + c] - mu) * rs;
    }
    atomicAdd(sums + c, s1);
    atomicAdd(sums + C + c, s2);
  }
}
// dr = gamma*rstd*(G - sums0/count - xhat*sums1/count) ; dgamma += sums1 ; dbeta += sums0 (done once by block 0)
__global__ void k_bn_bwd_apply(const float* __restrict__ G, const float* __restrict__ r, const float* __restrict__ gamma,
                               const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ sums,
                               float count, float* __restrict__ dr, float* __restrict__ dgamma, float* __restrict__ dbeta,
                               long pixels, int C) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < C && dgamma) { atomicAdd(dgamma + idx, sums[C + idx]); atomicAdd(dbeta + idx, sums[idx]); }
  if (idx >= pixels * C) return;
  const int c = idx % C;
  const float xh = (r[idx] - mean[c]) * rstd[c];
  dr[idx] = gamma[c] * rstd[c] * (G[idx] - sums[c] / count - xh * sums[C + c] / count);
}

// ---------------------------------------------------------------------------------- CMM channel gate backward
// g = x * (1 + w), w = sigmoid(fc2(relu(fc1(mean_p x)))) ; one workgroup per image
__global__ __launch_bounds__(256) void k_se_gate_bwd(const float* __restrict__ x, const float* __restrict__ dg,
                                                      const float* __restrict__ fc1_w, const float* __restrict__ fc1_b,
                                                      const float* __restrict__ fc2_w, const float* __restrict__ fc2_b,
                                                      float* __restrict__ dx, float* __restrict__ dfc1_w, float* __restrict__ dfc1_b,
                                                      float* __restrict__ dfc2_w, float* __restrict__ dfc2_b, int P, int C, int Cm) {
  extern __shared__ float sm[];
  float* S = sm;            // [C] mean
  float* Hp = S + C;        // [Cm] pre-relu
  float* Wg = Hp + Cm;      // [C] gate
  float* dlog = Wg + C;     // [C] grad wrt fc2 output (pre-sigmoid)
  float* dh = dlog + C;     // [Cm] grad wrt fc1 output (pre-relu)
  float* dSm = dh + Cm;     // [C] grad wrt S
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* xb = x + (size_t)b * P * C;
  const float* gb = dg + (size_t)b * P * C;
  for (int c = tid; c < C; c += 256) { float s = 0.f; for (int p = 0; p < P; ++p) s += xb[p * C + c]; S[c] = s / (float)P; }
  __syncthreads();
  for (int j = wave; j < Cm; j += 4) {
    float a = 0.f;
    for (int k = lane; k < C; k += 64) a += fc1_w[(size_t)j * C + k] * S[k];
    a = wave_sum(a);
    if (lane == 0) Hp[j] = a + fc1_b[j];
  }
  __syncthreads();
  for (int c = wave; c < C; c += 4) {
    float a = 0.f;
    for (int k = lane; k < Cm; k += 64) a += fc2_w[(size_t)c * Cm + k] * fmaxf(Hp[k], 0.f);
    a = wave_sum(a);
    if (lane == 0) {
      const float w = sigmoid_f(a + fc2_b[c]);
      Wg[c] = w;
      float dwsum = 0.f;
      for (int p = 0; p < P; ++p) dwsum += gb[p * C + c] * xb[p * C + c];
      dlog[c] = dwsum * w * (1.f - w);
    }
  }
  __syncthreads();
  for (int c = tid; c < C; c += 256) atomicAdd(dfc2_b + c, dlog[c]);
  for (int i = tid; i < C * Cm; i += 256) atomicAdd(dfc2_w + i, dlog[i / Cm] * fmaxf(Hp[i % Cm], 0.f));
  for (int j = wave; j < Cm; j += 4) {
    float a = 0.f;
    for (int c = lane; c < C; c += 64) a += dlog[c] * fc2_w[(size_t)c * Cm + j];
    a = wave_sum(a);
    if (lane == 0) dh[j] = Hp[j] > 0.f ? a : 0.f;
  }
  __syncthreads();
  for (int j = tid; j < Cm; j += 256) atomicAdd(dfc1_b + j, dh[j]);
  for (int i = tid; i < Cm * C; i += 256) atomicAdd(dfc1_w + i, dh[i / C] * S[i % C]);
  for (int c = tid; c < C; c += 256) {
    float a = 0.f;
    for (int j = 0; j < Cm; ++j) a += dh[j] * fc1_w[(size_t)j * C + c];
    dSm[c] = a / (float)P;
  }
  __syncthreads();
  float* db_ = dx + (size_t)b * P * C;
  for (int i = tid; i < P * C; i += 256) db_[i] = gb[i] * (1.f + Wg[i % C]) + dSm[i % C];
}

// ---------------------------------------------------------------------------------- DistillModule tail + optimizer
// f = relu(scale*r + shift) on NHWC (pixels, C) ; distill_module.py:21-27
__global__ void k_affine_act_fwd(const float* __restrict__ r, const float* __restrict__ scale, const float* __restrict__ shift,
                                 int act, float* __restrict__ y, long pixels, int C) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= pixels * C) return;
  const int c = idx % C;
  const float z = scale ? r[idx] * scale[c] + shift[c] : r[idx];
  y[idx] = apply_act(z, act, 0.f);
}
// L1 between two feature maps: part[blk] = sum |a-b| ; optional gradient kernel below
__global__ __launch_bounds__(256) void k_l1_partial(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ part,
                                                     long n) {
  __shared__ float red[4];
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  float v = i < n ? fabsf(a[i] - b[i]) : 0.f;
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void k_sum_final(const float* __restrict__ part, int n, float mul, float* __restrict__ out) {
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 64) s += part[i];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (threadIdx.x == 0) out[0] = (float)(s * mul);
}
// da = gs*c*sign(a-b) (+ extra) ; db = -gs*c*sign(a-b)
__global__ void k_l1_bwd(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ gscale, float c,
                         const float* __restrict__ extra_a, float* __restrict__ da, float* __restrict__ db, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float d = a[i] - b[i];
  const float g = gscale[0] * c * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
  da[i] = g + (extra_a ? extra_a[i] : 0.f);
  db[i] = -g;
}
// sum of squares of a flat buffer -> part[blk]; final: out[0] = sum
__global__ __launch_bounds__(256) void k_sumsq_partial(const float* __restrict__ x, float* __restrict__ part, long n) {
  __shared__ float red[4];
  float s = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) { const float v = x[i]; s += v * v; }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
// clip_grad_norm_(max_norm) + Adam (torch.optim.Adam semantics, no weight decay / amsgrad); normsq[0] = ||g||^2
__global__ void k_adam_clip(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            const float* __restrict__ normsq, float max_norm, float lr, float b1, float b2, float eps, float bc1,
                            float bc2_sqrt, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float coef = 1.f;
  if (max_norm > 0.f) { coef = max_norm / (sqrtf(normsq[0]) + 1e-6f); coef = coef < 1.f ? coef : 1.f; }
  const float gi = g[i] * coef;
  const float mi = b1 * m[i] + (1.f - b1) * gi;
  const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
  m[i] = mi; v[i] = vi;
  p[i] -= (lr / bc1) * mi / (sqrtf(vi) / bc2_sqrt + eps);
}

}  // namespace

extern "C" {

int dpmn_conv2d_wgrad_f32(const dpmn_conv_desc* d, const float* dy, float* dwp, dpmn_stream_t stream) {
  DPMN_REQUIRE(d && d->in[0] && dy && dwp, "conv2d_wgrad: null pointer");
  WgArgs a{};
  int cin = 0;
  for (int s = 0; s < 3; ++s) {
    a.in[s] = d->in[s]; a.in_scale[s] = d->in_scale[s]; a.in_shift[s] = d->in_shift[s]; a.cseg[s] = d->in[s] ? d->cseg[s] : 0;
    DPMN_REQUIRE(a.cseg[s] % 4 == 0, "conv2d_wgrad: segment channel counts must be multiples of 4");
    cin += a.cseg[s];
  }
  a.cin = cin; a.B = d->B; a.Hin = d->Hin; a.Win = d->Win; a.KH = d->KH; a.KW = d->KW; a.stride = d->stride;
  a.dil_y = d->dil_y; a.dil_x = d->dil_x; a.pad_y = d->pad_y; a.pad_x = d->pad_x; a.Hp = d->Hp; a.Wp = d->Wp;
  a.Hout = d->Hout; a.Wout = d->Wout; a.ostep = d->ostep; a.ooy = d->ooy; a.oox = d->oox; a.pro_act = d->pro_act;
  a.dy = dy; a.Cout = d->Cout; a.dwp = dwp;
  a.Kp = ((d->KH * d->KW * cin + 31) / 32) * 32;
  const int M = a.B * a.Hp * a.Wp;
  const int tiles = cdiv(a.Cout, 64) * cdiv(a.Kp, 64);
  int splits = cdiv(1024, tiles);
  int ppb = cdiv(cdiv(M, splits), 32) * 32;
  if (ppb < 32) ppb = 32;
  splits = cdiv(M, ppb);
  a.pix_per_block = ppb;
  dim3 grid(cdiv(a.Cout, 64), cdiv(a.Kp, 64), splits);
  hipLaunchKernelGGL(k_conv_wgrad, grid, dim3(256), 0, as_stream(stream), a);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

int dpmn_bn_finalize_f32(const float* stats, const float* gamma, const float* beta, float count, float eps, float momentum,
                         float* scale, float* shift, float* mean, float* rstd, float* running_mean, float* running_var, int C,
                         dpmn_stream_t stream) {
  DPMN_REQUIRE(stats && gamma && beta && scale && shift && mean && rstd && C > 0 && count > 1.f, "bn_finalize: bad arguments");
  hipLaunchKernelGGL(k_bn_finalize, dim3(cdiv(C, 128)), dim3(128), 0, as_stream(stream), stats, gamma, beta, count, eps, momentum,
                     scale, shift, mean, rstd, running_mean, running_var, C);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

int dpmn_affine_act_bwd_f32(const float* dA, const float* r, const float* scale, const float* shift, int act, float* G,
                            int accumulate, long pixels, int C, dpmn_stream_t stream) {
  DPMN_REQUIRE(dA && r && G && pixels > 0 && C > 0, "affine_act_bwd: bad arguments");
  const long total = pixels * C;
  hipLaunchKernelGGL(k_affine_act_bwd, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), dA, r, scale, shift,
                     act, G, accumulate, pixels, C);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

int dpmn_bn_bwd_f32(const float* G, const float* r, const float* gamma, const float* mean, const float* rstd, float* sums_ws,
                    float* dr, float* dgamma, float* dbeta, long pixels, int C, dpmn_stream_t stream) {
  DPMN_REQUIRE(G && r && gamma && mean && rstd && sums_ws && dr && dgamma && dbeta && pixels > 1, "bn_bwd: bad arguments");
  (void)hipMemsetAsync(sums_ws, 0, (size_t)2 * C * sizeof(float), as_stream(stream));
  const int ppb = 64;
  hipLaunchKernelGGL(k_bn_bwd_reduce, dim3((unsigned)((pixels + ppb - 1) / ppb)), dim3(256), 0, as_stream(stream), G, r, mean, rstd,
                     sums_ws, pixels, C, ppb);
  DPMN_CHECK_LAUNCH();
  const long total = pixels * C;
  hipLaunchKernelGGL(k_bn_bwd_apply, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), G, r, gamma, mean, rstd,
                     sums_ws, (float)pixels, dr, dgamma, dbeta, pixels, C);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

int dpmn_se_gate_bwd_f32(const float* x, const float* dg, const float* fc1_w, const float* fc1_b, const float* fc2_w,
                         const float* fc2_b, float* dx, float* dfc1_w, float* dfc1_b, float* dfc2_w, float* dfc2_b, int B, int P,
                         int C, int Cmid, dpmn_stream_t stream) {
  DPMN_REQUIRE(x && dg && fc1_w && fc1_b && fc2_w && fc2_b && dx && dfc1_w && dfc1_b && dfc2_w && dfc2_b && B > 0, "se_gate_bwd: bad arguments");
  hipLaunchKernelGGL(k_se_gate_bwd, dim3(B), dim3(256), (size_t)(4 * C + 2 * Cmid) * 4, as_stream(stream), x, dg, fc1_w, fc1_b, fc2_w,
                     fc2_b, dx, dfc1_w, dfc1_b, dfc2_w, dfc2_b, P, C, Cmid);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

int dpmn_affine_act_fwd_f32(const float* r, const float* scale, const float* shift, int act, float* y, long pixels, int C,
                            dpmn_stream_t stream) {
  DPMN_REQUIRE(r && y && pixels > 0 && C > 0, "affine_act_fwd: bad arguments");
  const long total = pixels * C;
  hipLaunchKernelGGL(k_affine_act_fwd, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), r, scale, shift, act, y, pixels, C);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

int dpmn_l1_loss_fwd_f32(const float* a, const float* b, float inv_count, float* loss, float* part_ws, long n, dpmn_stream_t stream) {
  DPMN_REQUIRE(a && b && loss && part_ws && n > 0, "l1_loss_fwd: bad arguments (part_ws: ceil(n/256) floats)");
  const int nb = (int)((n + 255) / 256);
  hipLaunchKernelGGL(k_l1_partial, dim3(nb), dim3(256), 0, as_stream(stream), a, b, part_ws, n);
  DPMN_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_sum_final, dim3(1), dim3(64), 0, as_stream(stream), part_ws, nb, inv_count, loss);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

int dpmn_l1_loss_bwd_f32(const float* a, const float* b, const float* grad_scale, float inv_count, const float* extra_a, float* da,
                         float* db, long n, dpmn_stream_t stream) {
  DPMN_REQUIRE(a && b && grad_scale && da && db && n > 0, "l1_loss_bwd: bad arguments");
  hipLaunchKernelGGL(k_l1_bwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), a, b, grad_scale, inv_count, extra_a, da, db, n);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

int dpmn_sumsq_f32(const float* x, float* out, float* part_ws, long n, dpmn_stream_t stream) {
  DPMN_REQUIRE(x && out && part_ws && n > 0, "sumsq: bad arguments (part_ws: 1024 floats)");
  const int nb = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
  hipLaunchKernelGGL(k_sumsq_partial, dim3(nb), dim3(256), 0, as_stream(stream), x, part_ws, n);
  DPMN_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_sum_final, dim3(1), dim3(64), 0, as_stream(stream), part_ws, nb, 1.0f, out);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

int dpmn_adam_clip_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, const float* grad_normsq, float max_norm,
                       float lr, float beta1, float beta2, float eps, int step, long n, dpmn_stream_t stream) {
  DPMN_REQUIRE(param && grad && exp_avg && exp_avg_sq && n > 0 && step >= 1, "adam_clip: bad arguments");
  DPMN_REQUIRE(max_norm <= 0.f || grad_normsq, "adam_clip: grad_normsq required when clipping");
  const float bc1 = 1.f - powf(beta1, (float)step);
  const float bc2s = sqrtf(1.f - powf(beta2, (float)step));
  hipLaunchKernelGGL(k_adam_clip, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), param, grad, exp_avg, exp_avg_sq,
                     grad_normsq, max_norm, lr, beta1, beta2, eps, bc1, bc2s, n);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

}  // extern "C"
