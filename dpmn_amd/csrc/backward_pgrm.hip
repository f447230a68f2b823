// PGRM-specific backward kernels (autograd of model/pgrm.py in the reference): window attention, SKConv gate,
// depthwise conv, elementwise helpers.  Linear / conv data- and weight-gradients use gemm.hip / conv*.hip / backward.hip.
#include <cstdlib>
#include "common.h"

namespace {

// ---------------------------------------------------------------------------------- window attention backward
// Same slab decomposition as k_window_attn (pgrm.hip): block = 2 slabs x 2 heads, one wave per (slab, head).
// Pass 1 (lane = query row): recompute the softmax statistics (max, 1/sum), O = P.V, delta = dO.O, and dQ.
// Pass 2 (lane = key row):  dK, dV and the relative-position-bias table gradient (LDS atomics, then global atomics).
// All gathers/scatters use the forward's roll + window-major token map (quirk Q1); every token belongs to exactly
// one window per group, so dq / dkv are plain stores.
// DROP: attn_drop (pgrm.py:248) -- P is multiplied by the regenerated mask M (0 or 1/(1-p)) before P.V, so dV = (P o M)^T dO,
// dP = (dO V^T) o M, and delta = dO . O is unchanged in form (O is the dropped output).
template <int WS, int D, bool DROP>
__global__ __launch_bounds__(256) void k_window_attn_bwd(const float* __restrict__ q, const float* __restrict__ kv,
                                                          const float* __restrict__ bias_table, const float* __restrict__ dout,
                                                          float* __restrict__ dq, float* __restrict__ dkv,
                                                          float* __restrict__ dtable, int B, int H, int W, int C, int g, int shift,
                                                          float p_drop, unsigned long long seed, int part_mode) {
  // part_mode 1: dtable is a (gridDim.x, TBL * 2) buffer -- every block STORES its table-gradient partial row (the caller adds
  // the rows in block order).  The block's two slabs accumulate into their own LDS copies (the two heads of a slab write
  // disjoint entries), added slab 0 + slab 1 at the end: no cross-wave LDS atomics on one word, bitwise reproducible.
  constexpr int N = WS * WS, CG = 2 * D, ROWS = 64, LDR = CG + 4, TBL = (2 * WS - 1) * (2 * WS - 1);
  static_assert(N <= 64, "windows larger than 64 tokens are not built yet");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int TB4 = (TBL * 2 + 3) & ~3;
  float* tbl = smem;                                 // [TBL*2]
  float* dtb0 = smem + TB4;                          // [2 slabs][TBL*2] gradient accumulators
  float* base = dtb0 + 2 * TB4;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int slab_in_blk = wave >> 1, head = wave & 1;
  float* dtb = dtb0 + slab_in_blk * TB4;
  float* Qs = base + slab_in_blk * (4 * ROWS * LDR + 6 * ROWS);
  float* Ks = Qs + ROWS * LDR;
  float* Vs = Ks + ROWS * LDR;
  float* Gs = Vs + ROWS * LDR;                       // dO rows
  float* stat = Gs + ROWS * LDR;                     // [2 heads][3][ROWS]: max, 1/sum, delta
  int* reg_s = reinterpret_cast<int*>(base + 2 * (4 * ROWS * LDR + 6 * ROWS)) + slab_in_blk * ROWS;

  const int L = H * W, slabs_per_img = L / ROWS;
  const long slab = (long)blockIdx.x * 2 + slab_in_blk;
  const int b = slab / slabs_per_img;
  const int t0 = (slab % slabs_per_img) * ROWS;
  const int nWc = W / WS;
  const bool active = b < B;
  for (int i = threadIdx.x; i < TBL * 2; i += 256) { tbl[i] = bias_table[i]; dtb0[i] = 0.f; dtb0[TB4 + i] = 0.f; }
  size_t src_tok = 0;
  if (active) {
    const int tl = threadIdx.x & 127;
    constexpr int V4 = CG / 4;
    for (int i = tl; i < ROWS * V4; i += 128) {
      const int r = i / V4, c4 = (i % V4) * 4;
      const int t = t0 + r, win = t / N, n = t % N;
      const int hr = (win / nWc) * WS + n / WS, wcol = (win % nWc) * WS + n % WS;
      const size_t src = (size_t)b * L + ((hr + shift) % H) * W + (wcol + shift) % W;
      *reinterpret_cast<float4*>(Qs + r * LDR + c4) = *reinterpret_cast<const float4*>(q + src * C + g * CG + c4);
      *reinterpret_cast<float4*>(Ks + r * LDR + c4) = *reinterpret_cast<const float4*>(kv + src * 2 * C + g * CG + c4);
      *reinterpret_cast<float4*>(Vs + r * LDR + c4) = *reinterpret_cast<const float4*>(kv + src * 2 * C + C + g * CG + c4);
      *reinterpret_cast<float4*>(Gs + r * LDR + c4) = *reinterpret_cast<const float4*>(dout + ((size_t)b * L + t) * C + g * CG + c4);
      if (c4 == 0) {
        const int rh = hr < H - WS ? 0 : (hr < H - shift ? 1 : 2), rw = wcol < W - WS ? 0 : (wcol < W - shift ? 1 : 2);
        reg_s[r] = 3 * rh + rw;
      }
    }
    const int t = t0 + lane, win = t / N, n = t % N;
    const int hr = (win / nWc) * WS + n / WS, wcol = (win % nWc) * WS + n % WS;
    src_tok = (size_t)b * L + ((hr + shift) % H) * W + (wcol + shift) % W;
  }
  __syncthreads();
  const float scale = 1.0f / sqrtf((float)D);
  const int nl = (t0 + lane) % N;                       // row index inside its window
  const int krow0 = (lane / N) * N;
  const int il = nl / WS, jl = nl % WS;
  const int my_reg = active ? reg_s[lane] : 0;
  float* smax = stat + head * 3 * ROWS, *sinv = smax + ROWS, *sdel = sinv + ROWS;
  // mask element index = ((((b*G + g)*2 + head)*L + window-major query token)*N + key row in window)   (G = C / CG)
  const unsigned long long mrow0 = ((unsigned long long)((size_t)b * (C / CG) + g) * 2 + head) * L;
  const float inv_keep = DROP ? 1.0f / (1.0f - p_drop) : 1.0f;

  if (active) {
    // ---------------- pass 1: lane = query
    float qv[D], go[D];
#pragma unroll
    for (int d = 0; d < D; ++d) { qv[d] = Qs[lane * LDR + head * D + d] * scale; go[d] = Gs[lane * LDR + head * D + d]; }
    float mx = -INFINITY;
    for (int m = 0; m < N; ++m) {
      const float* kr = Ks + (krow0 + m) * LDR + head * D;
      float a = 0.f;
#pragma unroll
      for (int d = 0; d < D; ++d) a += qv[d] * kr[d];
      a += tbl[((il - m / WS + WS - 1) * (2 * WS - 1) + (jl - m % WS + WS - 1)) * 2 + head];
      if (shift > 0 && reg_s[krow0 + m] != my_reg) a += -100.0f;
      mx = fmaxf(mx, a);
    }
    float den = 0.f, dlt = 0.f;
    for (int m = 0; m < N; ++m) {
      const float* kr = Ks + (krow0 + m) * LDR + head * D;
      const float* vr = Vs + (krow0 + m) * LDR + head * D;
      float a = 0.f, dp = 0.f;
#pragma unroll
      for (int d = 0; d < D; ++d) { a += qv[d] * kr[d]; dp += go[d] * vr[d]; }
      a += tbl[((il - m / WS + WS - 1) * (2 * WS - 1) + (jl - m % WS + WS - 1)) * 2 + head];
      if (shift > 0 && reg_s[krow0 + m] != my_reg) a += -100.0f;
      const float p = __expf(a - mx);
      den += p;
      if (DROP) dp *= drop_scale(seed, (mrow0 + t0 + lane) * N + m, p_drop, inv_keep);
      dlt += p * dp;                                    // sum_m P dP = dO . O
    }
    const float inv = 1.0f / den;
    dlt *= inv;
    smax[lane] = mx; sinv[lane] = inv; sdel[lane] = dlt;
    float dqa[D];
#pragma unroll
    for (int d = 0; d < D; ++d) dqa[d] = 0.f;
    for (int m = 0; m < N; ++m) {
      const float* kr = Ks + (krow0 + m) * LDR + head * D;
      const float* vr = Vs + (krow0 + m) * LDR + head * D;
      float a = 0.f, dp = 0.f;
#pragma unroll
      for (int d = 0; d < D; ++d) { a += qv[d] * kr[d]; dp += go[d] * vr[d]; }
      a += tbl[((il - m / WS + WS - 1) * (2 * WS - 1) + (jl - m % WS + WS - 1)) * 2 + head];
      if (shift > 0 && reg_s[krow0 + m] != my_reg) a += -100.0f;
      if (DROP) dp *= drop_scale(seed, (mrow0 + t0 + lane) * N + m, p_drop, inv_keep);
      const float ds = __expf(a - mx) * inv * (dp - dlt);
#pragma unroll
      for (int d = 0; d < D; ++d) dqa[d] += ds * kr[d];
    }
    float* dst = dq + src_tok * C + g * CG + head * D;
#pragma unroll
    for (int d = 0; d < D; d += 4)
      *reinterpret_cast<float4*>(dst + d) = make_float4(dqa[d] * scale, dqa[d + 1] * scale, dqa[d + 2] * scale, dqa[d + 3] * scale);
  }
  __syncthreads();
  if (active) {
    // ---------------- pass 2: lane = key
    float kvv[D], vv[D], dk[D], dv[D];
#pragma unroll
    for (int d = 0; d < D; ++d) { kvv[d] = Ks[lane * LDR + head * D + d]; vv[d] = Vs[lane * LDR + head * D + d]; dk[d] = 0.f; dv[d] = 0.f; }
    for (int n = 0; n < N; ++n) {
      const float* qr = Qs + (krow0 + n) * LDR + head * D;
      const float* gr = Gs + (krow0 + n) * LDR + head * D;
      float a = 0.f, dp = 0.f;
#pragma unroll
      for (int d = 0; d < D; ++d) { a += qr[d] * kvv[d]; dp += gr[d] * vv[d]; }
      a *= scale;
      const int tix = ((n / WS - il + WS - 1) * (2 * WS - 1) + (n % WS - jl + WS - 1)) * 2 + head;
      a += tbl[tix];
      if (shift > 0 && reg_s[krow0 + n] != my_reg) a += -100.0f;
      const float p = __expf(a - smax[krow0 + n]) * sinv[krow0 + n];
      const float mk = DROP ? drop_scale(seed, (mrow0 + t0 + krow0 + n) * N + nl, p_drop, inv_keep) : 1.0f;
      const float ds = p * (dp * mk - sdel[krow0 + n]);
      const float pv = p * mk;
#pragma unroll
      for (int d = 0; d < D; ++d) { dv[d] += pv * gr[d]; dk[d] += ds * scale * qr[d]; }
      atomicAdd(dtb + tix, ds);
    }
    float* dkp = dkv + src_tok * 2 * C + g * CG + head * D;
    float* dvp = dkp + C;
#pragma unroll
    for (int d = 0; d < D; d += 4) {
      *reinterpret_cast<float4*>(dkp + d) = make_float4(dk[d], dk[d + 1], dk[d + 2], dk[d + 3]);
      *reinterpret_cast<float4*>(dvp + d) = make_float4(dv[d], dv[d + 1], dv[d + 2], dv[d + 3]);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < TBL * 2; i += 256) {
    const float v = dtb0[i] + dtb0[TB4 + i];
    if (part_mode) dtable[(size_t)blockIdx.x * (TBL * 2) + i] = v;
    else atomicAdd(dtable + i, v);
  }
}

// ---------------------------------------------------------------------------------- 8x8 window attention backward on MFMA
// One block = one window (64 tokens) x 2 heads, one wave per head.  Everything is a 64 x 64 x 16 product on
// v_mfma_f32_16x16x4_f32; the trick of k_window_attn8_mfma (pgrm.hip) is used four times: a 16x16 accumulator tile holds
// element [row = 4kq + r][col = lr] in lane (lr, kq), which is exactly the B-operand slot of reduction step (tile, r) --
// so P^T / dS^T (pass A) and P / dS (pass B) feed the next product straight from their accumulator registers.
//   pass A, keys x queries:  S^T = K Q^T, dP^T = V dO^T  ->  softmax statistics per query column, delta = sum_k P dP,
//                            dS^T = P^T o (dP^T - delta);  dQ^T = K^T dS^T;  bias-table gradient (LDS atomics)
//   pass B, queries x keys:  S = Q K^T, dP = dO V^T (recomputed in the transposed layout; the per-query statistics come
//                            back through LDS)  ->  P, dS;  dV^T = dO^T P;  dK^T = Q^T dS
// K^T / Q^T / dO^T operands (rows = head dims) are ds_read_b32 of the staged rows, shared by the 4 tiles of the other axis.
template <bool DROP>
__global__ __launch_bounds__(128) void k_window_attn8_bwd_mfma(const float* __restrict__ q, const float* __restrict__ kv,
                                                                const float* __restrict__ bias_table, const float* __restrict__ dout,
                                                                float* __restrict__ dq, float* __restrict__ dkv,
                                                                float* __restrict__ dtable, int H, int W, int C, int g, int shift,
                                                                float p_drop, unsigned long long seed, int part_mode) {
  // part_mode 1: dtable is (gridDim.x, TBL * 2): the block stores its partial row (each wave = head owns its half of the LDS table)
  constexpr int WS = 8, D = 16, N = 64, CG = 2 * D, LDR = CG + 4, TBL = (2 * WS - 1) * (2 * WS - 1), TB4 = (TBL * 2 + 3) & ~3;
  __shared__ __attribute__((aligned(16))) float tbl[TB4];
  __shared__ __attribute__((aligned(16))) float dtb[TB4];
  __shared__ __attribute__((aligned(16))) float Qs[N * LDR], Ks[N * LDR], Vs[N * LDR], Gs[N * LDR];
  __shared__ float stat[2][3][N];          // [head][max, 1/sum, delta][query]
  __shared__ int reg_s[N];
  __shared__ int src_s[N];                 // source token of slab row r (roll by -shift + window-major map, quirk Q1)
  const int tid = threadIdx.x, lane = tid & 63, head = tid >> 6;
  const int L = H * W, slabs_per_img = L / N;
  const int b = blockIdx.x / slabs_per_img;
  const int t0 = (blockIdx.x % slabs_per_img) * N;
  const int nWc = W / WS;
  for (int i = tid; i < TBL * 2; i += 128) { tbl[i] = bias_table[i]; dtb[i] = 0.f; }
  {
    constexpr int V4 = CG / 4;
    for (int i = tid; i < N * V4; i += 128) {
      const int r = i / V4, c4 = (i % V4) * 4;
      const int t = t0 + r, win = t / N, n = t % N;
      const int hr = (win / nWc) * WS + n / WS, wcol = (win % nWc) * WS + n % WS;
      const int src = b * L + ((hr + shift) % H) * W + (wcol + shift) % W;
      *reinterpret_cast<float4*>(Qs + r * LDR + c4) = *reinterpret_cast<const float4*>(q + (size_t)src * C + g * CG + c4);
      *reinterpret_cast<float4*>(Ks + r * LDR + c4) = *reinterpret_cast<const float4*>(kv + (size_t)src * 2 * C + g * CG + c4);
      *reinterpret_cast<float4*>(Vs + r * LDR + c4) = *reinterpret_cast<const float4*>(kv + (size_t)src * 2 * C + C + g * CG + c4);
      *reinterpret_cast<float4*>(Gs + r * LDR + c4) = *reinterpret_cast<const float4*>(dout + ((size_t)b * L + t) * C + g * CG + c4);
      if (c4 == 0) {
        const int rh = hr < H - WS ? 0 : (hr < H - shift ? 1 : 2), rw = wcol < W - WS ? 0 : (wcol < W - shift ? 1 : 2);
        reg_s[r] = 3 * rh + rw;
        src_s[r] = src;
      }
    }
  }
  __syncthreads();
  const int lr = lane & 15, kq = lane >> 4;
  const float scale = 0.25f;
  const unsigned long long mrow0 = ((unsigned long long)((size_t)b * (C / CG) + g) * 2 + head) * L;   // mask index base (include/dpmn_hip.h)
  const float inv_keep = DROP ? 1.0f / (1.0f - p_drop) : 1.0f;
  f32x4 kf[4], qf[4], vf[4], gf[4];        // row fragments: [row = 16t + lr][d = 4kq .. 4kq+3]
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int o = (16 * t + lr) * LDR + head * D + 4 * kq;
    kf[t] = *reinterpret_cast<const f32x4*>(Ks + o);
    qf[t] = *reinterpret_cast<const f32x4*>(Qs + o);
    qf[t] *= scale;
    vf[t] = *reinterpret_cast<const f32x4*>(Vs + o);
    gf[t] = *reinterpret_cast<const f32x4*>(Gs + o);
  }
  float* smax = stat[head][0];
  float* sinv = stat[head][1];
  float* sdel = stat[head][2];
  // ================= pass A: rows = keys (16t + 4kq + r), columns = queries (16qt + lr)
  {
    f32x4 ps[4][4], dp[4][4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int qt = 0; qt < 4; ++qt) {
        f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f}, c = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) { a = mfma16(kf[t][s4], qf[qt][s4], a); c = mfma16(vf[t][s4], gf[qt][s4], c); }
        ps[t][qt] = a; dp[t][qt] = c;
      }
    float tacc[7][4];
#pragma unroll
    for (int dl = 0; dl < 7; ++dl)
#pragma unroll
      for (int r = 0; r < 4; ++r) tacc[dl][r] = 0.f;
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
      const int nq = 16 * qt + lr, iq = nq / WS, jq = nq % WS;
      const int my_reg = reg_s[nq];
      float mx = -INFINITY;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = 16 * t + 4 * kq + r, im = m / WS, jm = m % WS;
          float a = ps[t][qt][r] + tbl[((iq - im + WS - 1) * (2 * WS - 1) + (jq - jm + WS - 1)) * 2 + head];
          if (shift > 0 && reg_s[m] != my_reg) a += -100.0f;
          ps[t][qt][r] = a;
          mx = fmaxf(mx, a);
        }
      mx = fmaxf(mx, xshfl<16>(mx));
      mx = fmaxf(mx, xshfl<32>(mx));
      float den = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float p = __expf(ps[t][qt][r] - mx); ps[t][qt][r] = p; den += p; }
      den += xshfl<16>(den);
      den += xshfl<32>(den);
      const float inv = 1.0f / den;
      if (DROP) {    // dP = (dO V^T) o M with the forward's mask (0 or 1/(1-p)); delta = sum_k P dP is unchanged in form
        const unsigned long long mrow = (mrow0 + t0 + nq) * N;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) dp[t][qt][r] *= drop_scale(seed, mrow + 16 * t + 4 * kq + r, p_drop, inv_keep);
      }
      float dlt = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) { ps[t][qt][r] *= inv; dlt += ps[t][qt][r] * dp[t][qt][r]; }
      dlt += xshfl<16>(dlt);
      dlt += xshfl<32>(dlt);
      if (kq == 0) { smax[nq] = mx; sinv[nq] = inv; sdel[nq] = dlt; }
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float ds = ps[t][qt][r] * (dp[t][qt][r] - dlt);
          dp[t][qt][r] = ds;                                     // dS^T
          tacc[qt - t + 3][r] += ds;      // bias-table gradient: the entry depends on (qt - t, r) only for a given lane
        }
    }
    {
      // query (iq, jq) = (2qt + lr/8, lr%8), key (im, jm) = (2t + kq/2, 4(kq%2) + r): di = 2(qt - t) + lr/8 - kq/2,
      // dj = lr%8 - 4(kq%2) - r -- 28 pre-summed LDS atomics per lane instead of 64
      const int ci = (lr >> 3) - (kq >> 1) + WS - 1, cj = (lr & 7) - 4 * (kq & 1) + WS - 1;
#pragma unroll
      for (int dl = 0; dl < 7; ++dl)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          atomicAdd(dtb + head * TBL + (2 * (dl - 3) + ci) * (2 * WS - 1) + (cj - r), tacc[dl][r]);
    }
    // dQ^T (16 d x 16 queries per tile) = K^T . dS^T
    f32x4 dqa[4];
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) dqa[qt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float kt = Ks[(16 * t + 4 * kq + r) * LDR + head * D + lr];     // K^T[d = lr][key]
#pragma unroll
        for (int qt = 0; qt < 4; ++qt) dqa[qt] = mfma16(kt, dp[t][qt][r], dqa[qt]);
      }
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
      float* dst = dq + (size_t)src_s[16 * qt + lr] * C + g * CG + head * D + 4 * kq;
      *reinterpret_cast<float4*>(dst) = make_float4(dqa[qt][0] * scale, dqa[qt][1] * scale, dqa[qt][2] * scale, dqa[qt][3] * scale);
    }
  }
  __syncthreads();          // statistics of both heads visible (each wave only needs its own, but keep the waves together)
  // ================= pass B: rows = queries (16qt + 4kq + r), columns = keys (16t + lr)
  {
    f32x4 ps[4][4], dp[4][4];           // [qt][t]
#pragma unroll
    for (int qt = 0; qt < 4; ++qt)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f}, c = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) { a = mfma16(qf[qt][s4], kf[t][s4], a); c = mfma16(gf[qt][s4], vf[t][s4], c); }
        ps[qt][t] = a; dp[qt][t] = c;
      }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int m = 16 * t + lr, im = m / WS, jm = m % WS;
      const int key_reg = reg_s[m];
#pragma unroll
      for (int qt = 0; qt < 4; ++qt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int nq = 16 * qt + 4 * kq + r, iq = nq / WS, jq = nq % WS;
          float a = ps[qt][t][r] + tbl[((iq - im + WS - 1) * (2 * WS - 1) + (jq - jm + WS - 1)) * 2 + head];
          if (shift > 0 && reg_s[nq] != key_reg) a += -100.0f;
          const float p = __expf(a - smax[nq]) * sinv[nq];
          const float mk = DROP ? drop_scale(seed, (mrow0 + t0 + nq) * N + m, p_drop, inv_keep) : 1.0f;
          ps[qt][t][r] = p * mk;                                 // P o M (what multiplies V in the forward)
          dp[qt][t][r] = p * (dp[qt][t][r] * mk - sdel[nq]);      // dS
        }
    }
    // dV^T = dO^T . P ; dK^T = Q^T . dS   (16 d x 16 keys per tile)
    f32x4 dva[4], dka[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) { dva[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; dka[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int qt = 0; qt < 4; ++qt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = (16 * qt + 4 * kq + r) * LDR + head * D + lr;
        const float gt = Gs[row], qt_ = Qs[row];                 // dO^T[d = lr][query], Q^T[d = lr][query]
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          dva[t] = mfma16(gt, ps[qt][t][r], dva[t]);
          dka[t] = mfma16(qt_, dp[qt][t][r], dka[t]);
        }
      }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      float* dkp = dkv + (size_t)src_s[16 * t + lr] * 2 * C + g * CG + head * D + 4 * kq;
      *reinterpret_cast<float4*>(dkp) = make_float4(dka[t][0] * scale, dka[t][1] * scale, dka[t][2] * scale, dka[t][3] * scale);
      *reinterpret_cast<float4*>(dkp + C) = make_float4(dva[t][0], dva[t][1], dva[t][2], dva[t][3]);
    }
  }
  __syncthreads();
  for (int i = tid; i < TBL * 2; i += 128) {     // table layout is [entry][head]
    if (part_mode) dtable[(size_t)blockIdx.x * (TBL * 2) + i] = dtb[(i & 1) * TBL + (i >> 1)];
    else atomicAdd(dtable + i, dtb[(i & 1) * TBL + (i >> 1)]);
  }
}

template <int WS, int D, bool DROP>
int launch_wattn_bwd(const float* q, const float* kv, const float* tbl, const float* dout, float* dq, float* dkv, float* dtable,
                     int B, int H, int W, int C, int g, int shift, float p_drop, unsigned long long seed, hipStream_t st, int part_mode = 0) {
  constexpr int CG = 2 * D, LDR = CG + 4, TBL = (2 * WS - 1) * (2 * WS - 1);
  const size_t smem = (size_t)(3 * ((TBL * 2 + 3) & ~3) + 2 * (4 * 64 * LDR + 6 * 64)) * 4 + 2 * 64 * 4;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_window_attn_bwd<WS, D, DROP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_set = true;
  }
  const long slabs = (long)B * (H * W / 64);
  hipLaunchKernelGGL((k_window_attn_bwd<WS, D, DROP>), dim3((unsigned)((slabs + 1) / 2)), dim3(256), smem, st, q, kv, tbl, dout, dq, dkv,
                     dtable, B, H, W, C, g, shift, p_drop, seed, part_mode);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

// ---------------------------------------------------------------------------------- SKConv backward pieces
// V[m][c] = sum_g A[b][g][c] * cat[m][g*cg + c]
__global__ void k_sk_select(const float* __restrict__ cat, const float* __restrict__ A, float* __restrict__ V, long M, int L,
                            int C, int G) {
  const int cg = C / G;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= M * cg) return;
  const long m = idx / cg;
  const int c = idx % cg, b = m / L;
  float v = 0.f;
  for (int g = 0; g < G; ++g) v += A[((size_t)b * G + g) * cg + c] * cat[m * C + g * cg + c];
  V[idx] = v;
}
// dcat[m][g*cg+c] (+)= A[b][g][c] * dV[m][c] ; dA[b][g][c] += sum over the block's tokens of cat * dV
__global__ __launch_bounds__(256) void k_sk_select_bwd(const float* __restrict__ cat, const float* __restrict__ A,
                                                        const float* __restrict__ dV, float* __restrict__ dcat,
                                                        float* __restrict__ dA, int L, int C, int G, int rows_per_block, int part_mode,
                                                        int overwrite) {
  // part_mode 1: dA is (gridDim.x, B, C) -- every block STORES its partial row, the gate backward adds them in block order (no atomics)
  // overwrite 1: dcat = A dV (the caller's buffer is fresh: no zero fill of it before, no read-modify-write here)
  const int cg = C / G;
  const int b = blockIdx.y;
  const int r0 = blockIdx.x * rows_per_block;
  const int r1 = min(L, r0 + rows_per_block);
  for (int col = threadIdx.x; col < C; col += blockDim.x) {
    const int g = col / cg, c = col % cg;
    const float a = A[((size_t)b * G + g) * cg + c];
    float acc = 0.f;
    int r = r0;
    for (; r + 4 <= r1; r += 4) {        // four rows' loads in flight, sums in row order
      float dv[4], cv[4], ov[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const size_t m = (size_t)b * L + r + u;
        dv[u] = dV[m * cg + c];
        cv[u] = cat[m * C + col];
        ov[u] = overwrite ? 0.f : dcat[m * C + col];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        acc += cv[u] * dv[u];
        dcat[((size_t)b * L + r + u) * C + col] = overwrite ? a * dv[u] : ov[u] + a * dv[u];
      }
    }
    for (; r < r1; ++r) {
      const size_t m = (size_t)b * L + r;
      const float dv = dV[m * cg + c];
      acc += cat[m * C + col] * dv;
      dcat[m * C + col] = overwrite ? a * dv : dcat[m * C + col] + a * dv;
    }
    if (part_mode) dA[((size_t)blockIdx.x * gridDim.y + b) * C + col] = acc;
    else atomicAdd(dA + ((size_t)b * G + g) * cg + c, acc);
  }
}
// one workgroup per image: gate MLP backward (pgrm.py:86-91).  S = mean_t GELU(feats) from the forward partials.
__global__ void k_sk_gate_bwd(const float* __restrict__ partial, int parts_per_image, int L, const float* __restrict__ fc1_w,
                              const float* __restrict__ fc1_b, const float* __restrict__ fc2_w, const float* __restrict__ A,
                              const float* __restrict__ dA, float* __restrict__ dS, float* __restrict__ dfc1_w,
                              float* __restrict__ dfc1_b, float* __restrict__ dfc2_w, float* __restrict__ dfc2_b, int C, int G,
                              int dmid, int nparts, float* __restrict__ wpart2, float* __restrict__ wpart1) {
  // nparts > 0: dA is the (nparts, B, C) partial-row buffer of k_sk_select_bwd (part_mode), added here in block order;
  // wpart2 / wpart1 != null: this image's weight-gradient contributions are STORED as row b of (B, C dmid + C) / (B, dmid C + dmid)
  // buffers (fc2 | bias, fc1 | bias) that the caller adds in image order (dpmn_rows_reduce_f32 / the deferred multi reduce)
  extern __shared__ float sm[];
  float* S = sm;             // [C]
  float* zp = S + C;         // [dmid] pre-GELU
  float* Z = zp + dmid;      // [dmid]
  float* dl = Z + dmid;      // [C] dlogit
  float* dz = dl + C;        // [dmid] grad wrt pre-GELU
  float* dAs = dz + dmid;    // [C] (nparts > 0)
  const int b = blockIdx.x, cg = C / G;
  // (this kernel is one latency chain per image on 48 blocks: the two partial-row sums share a phase, and the two matrix-vector
  //  products of the gate MLP are spread over eight lanes per output instead of one lane walking all C inputs: 32 -> see DESIGN)
  // partial rows: eight loads in flight, added in row order (a one-load-per-iteration loop waits out a cache latency per row)
  auto row_sum = [&](const float* base, size_t stride, int n) {
    float s = 0.f;
    int p = 0;
    for (; p + 8 <= n; p += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = base[(size_t)(p + u) * stride];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; p < n; ++p) s += base[(size_t)p * stride];
    return s;
  };
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    if (nparts > 0) dAs[c] = row_sum(dA + (size_t)b * C + c, (size_t)gridDim.x * C, nparts);
    S[c] = row_sum(partial + (size_t)b * parts_per_image * C + c, C, parts_per_image) / (float)L;
  }
  __syncthreads();
  auto dA_at = [&](int g, int c) { return nparts > 0 ? dAs[g * cg + c] : dA[((size_t)b * G + g) * cg + c]; };
  const int sub = threadIdx.x & 7;
  for (int j = threadIdx.x >> 3; j < dmid; j += blockDim.x >> 3) {      // 8 lanes per output (blockDim is a multiple of 64)
    float a = 0.f;
    for (int c = sub; c < C; c += 8) a += fc1_w[j * C + c] * S[c];
    a += xshfl<1>(a); a += xshfl<2>(a); a += xshfl<4>(a);
    if (sub == 0) {
      a += fc1_b[j];
      zp[j] = a;
      Z[j] = gelu_erf(a);
    }
  }
  for (int c = threadIdx.x; c < cg; c += blockDim.x) {
    float dot = 0.f;
    for (int g = 0; g < G; ++g) dot += A[((size_t)b * G + g) * cg + c] * dA_at(g, c);
    for (int g = 0; g < G; ++g) {
      const float a = A[((size_t)b * G + g) * cg + c];
      dl[g * cg + c] = a * (dA_at(g, c) - dot);
    }
  }
  __syncthreads();
  if (wpart2) {
    float* row = wpart2 + (size_t)b * (C * dmid + C);
    for (int i = threadIdx.x; i < C * dmid; i += blockDim.x) row[i] = dl[i / dmid] * Z[i % dmid];
    for (int c = threadIdx.x; c < C; c += blockDim.x) row[C * dmid + c] = dl[c];
  } else {
  for (int i = threadIdx.x; i < C * dmid; i += blockDim.x) atomicAdd(dfc2_w + i, dl[i / dmid] * Z[i % dmid]);
  for (int c = threadIdx.x; c < C; c += blockDim.x) atomicAdd(dfc2_b + c, dl[c]);
  }
  for (int j = threadIdx.x >> 3; j < dmid; j += blockDim.x >> 3) {
    float a = 0.f;
    for (int c = sub; c < C; c += 8) a += dl[c] * fc2_w[c * dmid + j];
    a += xshfl<1>(a); a += xshfl<2>(a); a += xshfl<4>(a);
    if (sub == 0) {
      const float x = zp[j];
      const float cdf = 0.5f * (1.0f + erf_as(x * 0.70710678118654752440f));
      dz[j] = a * (cdf + x * 0.3989422804014327f * __expf(-0.5f * x * x));
    }
  }
  __syncthreads();
  if (wpart1) {
    float* row = wpart1 + (size_t)b * (dmid * C + dmid);
    for (int i = threadIdx.x; i < dmid * C; i += blockDim.x) row[i] = dz[i / C] * S[i % C];
    for (int j = threadIdx.x; j < dmid; j += blockDim.x) row[dmid * C + j] = dz[j];
  } else {
  for (int i = threadIdx.x; i < dmid * C; i += blockDim.x) atomicAdd(dfc1_w + i, dz[i / C] * S[i % C]);
  for (int j = threadIdx.x; j < dmid; j += blockDim.x) atomicAdd(dfc1_b + j, dz[j]);
  }
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float a = 0.f;
    for (int j = 0; j < dmid; ++j) a += dz[j] * fc1_w[j * C + c];
    dS[(size_t)b * C + c] = a;
  }
}
// dfeats[m][c] = dout[m][c] + gelu'(feats[m][c]) * dS[b][c] / L
__global__ void k_sk_feats_grad(const float* __restrict__ dout, const float* __restrict__ feats, const float* __restrict__ dS,
                                float* __restrict__ dfeats, long M, int L, int C) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= M * C) return;
  const long m = idx / C;
  const int c = idx % C, b = m / L;
  const float x = feats[idx];
  const float cdf = 0.5f * (1.0f + erf_as(x * 0.70710678118654752440f));
  dfeats[idx] = dout[idx] + (cdf + x * 0.3989422804014327f * __expf(-0.5f * x * x)) * dS[(size_t)b * C + c] / (float)L;
}

// ---------------------------------------------------------------------------------- depthwise 3x3 backward
// per plane (b, c'): dP = full-correlation of dg with the kernel, dW[c'] += sum dg * shifted P, db[c'] += sum dg
// Fused activation backward / forward around it (the Mlp chain fc1 -> GELU -> dwconv -> GELU -> pointwise, pgrm.py:31-37):
//   gpre != NULL : dg is the gradient of GELU(conv output); it is multiplied by GELU'(gpre) on the way into the LDS tile
//   in_gelu      : P is fc1's pre-activation, GELU is applied on load (the forward input of the conv)
//   out_gelu_bwd : dP is multiplied by GELU'(P raw) before the store (the gradient of fc1's pre-activation); needs in_gelu
__device__ __forceinline__ float gelu_grad(float x) {
  const float cdf = 0.5f * (1.0f + erf_as(x * 0.70710678118654752440f));
  return cdf + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}
// GELU(x) and GELU'(x) from ONE erf evaluation: the Abramowitz-Stegun erf of common.h already holds exp(-x^2 / 2), which is the
// Gaussian of the derivative too (gelu_erf + gelu_grad cost 3 v_exp + 2 v_rcp; this is 1 + 1).  g is bitwise gelu_erf(x).
// all-lanes wave sum without the LDS pipe: four DPP adds inside a row of 16 lanes (quad swaps, half mirror, mirror), then the
// gfx950 row / half-wave swaps.  (__shfl_xor = ds_bpermute + its address arithmetic: ~90 instructions per reduced value in this
// kernel's epilogue, 45 % of its vector instructions.)
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
  typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
  v += dpp_mov<0xB1>(v);       // quad_perm [1, 0, 3, 2]
  v += dpp_mov<0x4E>(v);       // quad_perm [2, 3, 0, 1]
  v += dpp_mov<0x141>(v);      // row_half_mirror
  v += dpp_mov<0x140>(v);      // row_mirror
  u32x2_ r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
#ifndef DWB_SKIP
#define DWB_SKIP 0        // timing ablations only: 1 no GELU arithmetic, 2 no stencil arithmetic
#endif
__device__ __forceinline__ void gelu_both(float x, float& g, float& d) {
  if (DWB_SKIP & 1) { g = x; d = 1.0f; return; }
  const float ax = fabsf(x * 0.70710678118654752440f);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
  const float e = __expf(-ax * ax);
  const float er = copysignf(fmaf(-poly, e, 1.0f), x);
  g = 0.5f * x * (1.0f + er);
  d = 0.5f * (1.0f + er) + x * 0.3989422804014327f * e;
}
__device__ __forceinline__ float gelu_grad1(float x) {      // GELU'(x) alone, one v_exp
  float g, d;
  gelu_both(x, g, d);
  return d;
}
// KEEP: in_gelu && out_gelu_bwd && gpre && 32 x 32 planes (the Mlp of the 16 x 64 -> 32 x 128 stacks) -- GELU'(P) of the lane's 16 pixels stays in registers from the load
// loop to the store loop (same pixel -> lane map in both) instead of a second read of P and a second erf + exp per pixel
template <bool KEEP>
__global__ __launch_bounds__(256, KEEP ? 2 : 1) void k_dwconv_bwd(const float* __restrict__ P, const float* __restrict__ dg,
                                                     const float* __restrict__ w, float* __restrict__ dP, float* __restrict__ dw,
                                                     float* __restrict__ db, int Ch, int r, long planes,
                                                     const float* __restrict__ gpre = nullptr, int in_gelu = 0, int out_gelu_bwd = 0,
                                                     float p_drop = 0.f, unsigned long long seed = 0ull, float* __restrict__ part = nullptr, int rb = 0) {
  // !KEEP: a wave takes a BAND of rb rows of a plane (rb = r: the whole plane), the halo rows above / below re-read and re-activated
  // from the neighbouring bands (64 x 64 planes as whole-plane tiles: 38 KB of LDS per wave, 4 waves per CU, 0.20 of the HBM rate)
  // p_drop > 0: the forward input was dropout(GELU(P)) -- the same mask on load, and again on dP before GELU'
  const float inv_keep = p_drop > 0.f ? 1.0f / (1.0f - p_drop) : 1.0f;
  // per wave: P tile and dg tile, (r+2) rows x LD = r+8 floats, plane starting at column 4 (16-byte aligned rows, r % 4 == 0).
  // The tiles are private to the wave (LDS operations of one wave execute in order): no block barrier anywhere.
  // KEEP: persistent waves, plane = 4 blockIdx + wave, + 4 gridDim, ...; the next plane's 12 x 16 bytes per lane are in flight
  // while this plane's GELUs and stencils run.  (One plane per wave, load -> compute -> store, ran the whole chip in lockstep --
  // every block loading, then every block computing: the kernel took memory time PLUS arithmetic time, 47 + 21 + 26 us.)
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (KEEP || rb <= 0) rb = r;
  const int nb = r / rb;
  const int LD = r + 8, r4 = r >> 2, TS = (rb + 2) * LD, npix4 = rb * r4;
  float* tp = sm + wave * 2 * TS;
  float* tg = tp + TS;
  long plane = (long)blockIdx.x * 4 + wave;      // (!KEEP: virtual plane = plane * nb + band)
  const long nvp = planes * nb;
  if (plane >= nvp) return;
  const long pstep = KEEP ? (long)gridDim.x * 4 : nvp;
  // zero halo: KEEP -- rows 0 and r + 1 and the two halo columns, written once (the interior stores never touch them); bands -- the
  // halo columns only (the halo rows are loaded, zero outside the plane)
  if (KEEP)
    for (int i = lane; i < LD; i += 64) { tp[i] = 0.f; tp[(r + 1) * LD + i] = 0.f; tg[i] = 0.f; tg[(r + 1) * LD + i] = 0.f; }
  for (int i = lane; i < rb + 2; i += 64) {
    tp[i * LD + 3] = 0.f; tp[i * LD + 4 + r] = 0.f;
    tg[i * LD + 3] = 0.f; tg[i * LD + 4 + r] = 0.f;
  }
  float4 pre_p[KEEP ? 4 : 1], pre_g[KEEP ? 4 : 1], pre_q[KEEP ? 4 : 1], gpk[KEEP ? 4 : 1];
  auto prefetch = [&](long pl) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      // KEEP <=> r == 32 and gpre given: 256 float4 per plane = exactly 4 per lane, no predicate and no branch around a load
      // (hipcc drains vmcnt(0) at the join of a branch that contains one)
      const long o = pl * 1024 + 4 * (lane + 64 * it);
      pre_p[KEEP ? it : 0] = *reinterpret_cast<const float4*>(P + o);
      pre_g[KEEP ? it : 0] = *reinterpret_cast<const float4*>(dg + o);
      pre_q[KEEP ? it : 0] = *reinterpret_cast<const float4*>(gpre + o);
    }
  };
  if (KEEP) prefetch(plane);
  for (; plane < nvp; plane += pstep) {
    const long pl_ = KEEP ? plane : plane / nb;                 // the real plane
    const int y0 = KEEP ? 0 : (int)(plane % nb) * rb;          // first row of the band
    const int c = (int)(pl_ % Ch);
    const float* ps = P + pl_ * r * r;
    const float* gs = dg + pl_ * r * r;
    // the channel's 9 weights BEFORE the next plane's prefetch is issued: vmcnt retires in order, a load behind the prefetch
    // would wait for all of it
    float k[9], aw[9], ab = 0.f;
#pragma unroll
    for (int i = 0; i < 9; ++i) { k[i] = w[c * 9 + i]; aw[i] = 0.f; }
    const int nld = KEEP ? 256 : (rb + 2) * r4;                // bands: tile row j holds plane row y0 - 1 + j
#pragma unroll 4
    for (int it = 0; it < (KEEP ? 4 : (nld + 63) / 64); ++it) {
      const int i = lane + 64 * it;
      if (!KEEP && i >= nld) break;
      const int j = i / r4, x4 = (i - j * r4) * 4;
      const int yy = KEEP ? j : y0 - 1 + j;
      float4 pv = make_float4(0.f, 0.f, 0.f, 0.f), gv = pv;
      if (KEEP) { pv = pre_p[it]; gv = pre_g[it]; }
      else if (yy >= 0 && yy < r) { pv = *reinterpret_cast<const float4*>(ps + yy * r + x4); gv = *reinterpret_cast<const float4*>(gs + yy * r + x4); }
      if (KEEP || (yy >= 0 && yy < r)) {
        if (KEEP) {
          float4 d;
          gelu_both(pv.x, pv.x, d.x); gelu_both(pv.y, pv.y, d.y); gelu_both(pv.z, pv.z, d.z); gelu_both(pv.w, pv.w, d.w);
          gpk[it] = d;
        } else if (in_gelu) { pv.x = gelu_erf(pv.x); pv.y = gelu_erf(pv.y); pv.z = gelu_erf(pv.z); pv.w = gelu_erf(pv.w); }
        if (p_drop > 0.f) {
          const unsigned long long e0 = (unsigned long long)(pl_ * r * r + yy * r + x4);
          const unsigned long long z0 = drop_z0(seed, e0);      // one 64-bit multiply per 4 elements, constant adds for the rest
          pv.x *= drop_scale_z(z0, p_drop, inv_keep); pv.y *= drop_scale_z(z0 + DROP_PHI, p_drop, inv_keep);
          pv.z *= drop_scale_z(z0 + 2 * DROP_PHI, p_drop, inv_keep); pv.w *= drop_scale_z(z0 + 3 * DROP_PHI, p_drop, inv_keep);
        }
        if (KEEP || gpre) {
          const float4 q = KEEP ? pre_q[it] : *reinterpret_cast<const float4*>(gpre + pl_ * r * r + yy * r + x4);
          gv.x *= gelu_grad1(q.x); gv.y *= gelu_grad1(q.y); gv.z *= gelu_grad1(q.z); gv.w *= gelu_grad1(q.w);
        }
      }
      *reinterpret_cast<float4*>(tp + (KEEP ? j + 1 : j) * LD + 4 + x4) = pv;
      *reinterpret_cast<float4*>(tg + (KEEP ? j + 1 : j) * LD + 4 + x4) = gv;
    }
    if (KEEP) prefetch(plane + pstep < planes ? plane + pstep : plane);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    float* dst = dP + pl_ * r * r + (size_t)y0 * r;
#pragma unroll 4
    for (int it = 0; it < (KEEP ? 4 : (npix4 + 63) / 64); ++it) {
      const int i = lane + 64 * it;
      if (!KEEP && i >= npix4) break;
      const int yy = i / r4, x4 = (i - yy * r4) * 4;
      float a[4] = {0.f, 0.f, 0.f, 0.f};
      const float4 gc = *reinterpret_cast<const float4*>(tg + (yy + 1) * LD + 4 + x4);     // dg at the 4 output pixels
      if (DWB_SKIP & 2) { a[0] = gc.x; a[1] = gc.y; a[2] = gc.z; a[3] = gc.w; aw[0] += gc.x; }
      else
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        // dP[y][x] = sum_k dg[y - ky + 1][x - kx + 1] w[ky][kx] : dg row (yy + 2 - ky) of the halo tile, columns x4 - 1 .. x4 + 4
        const float* pg = tg + (yy + 2 - ky) * LD + 4 + x4;
        const float4 gm = *reinterpret_cast<const float4*>(pg);
        const float gl = pg[-1], gr = pg[4];
        const float k0 = k[ky * 3], k1 = k[ky * 3 + 1], k2 = k[ky * 3 + 2];
        a[0] += k0 * gm.y + k1 * gm.x + k2 * gl;
        a[1] += k0 * gm.z + k1 * gm.y + k2 * gm.x;
        a[2] += k0 * gm.w + k1 * gm.z + k2 * gm.y;
        a[3] += k0 * gr + k1 * gm.w + k2 * gm.z;
        // dW[ky][kx] += dg[y][x] P[y + ky - 1][x + kx - 1] : P row (yy + ky) of the halo tile
        const float* pp = tp + (yy + ky) * LD + 4 + x4;
        const float4 pm = *reinterpret_cast<const float4*>(pp);
        const float pl = pp[-1], pr = pp[4];
        aw[ky * 3] += gc.x * pl + gc.y * pm.x + gc.z * pm.y + gc.w * pm.z;
        aw[ky * 3 + 1] += gc.x * pm.x + gc.y * pm.y + gc.z * pm.z + gc.w * pm.w;
        aw[ky * 3 + 2] += gc.x * pm.y + gc.y * pm.z + gc.z * pm.w + gc.w * pr;
      }
      ab += (gc.x + gc.y) + (gc.z + gc.w);
      if (p_drop > 0.f) {
        const unsigned long long e0 = (unsigned long long)(pl_ * r * r + (y0 + yy) * r + x4);
#pragma unroll
        for (int q = 0; q < 4; ++q) a[q] *= drop_scale_z(drop_z0(seed, e0) + (unsigned long long)q * DROP_PHI, p_drop, inv_keep);
      }
      if (KEEP) {
        const float4 d = gpk[it];
        a[0] *= d.x; a[1] *= d.y; a[2] *= d.z; a[3] *= d.w;
      } else if (out_gelu_bwd) {
        const float4 q = *reinterpret_cast<const float4*>(P + pl_ * r * r + (y0 + yy) * r + x4);
        a[0] *= gelu_grad1(q.x); a[1] *= gelu_grad1(q.y); a[2] *= gelu_grad1(q.z); a[3] *= gelu_grad1(q.w);
      }
      *reinterpret_cast<float4*>(dst + yy * r + x4) = make_float4(a[0], a[1], a[2], a[3]);
    }
    // part: row (image b, band) = [Ch * 9 weight sums | Ch bias sums], added over the rows in order by dpmn_rows_reduce_f32 (no atomics)
    float* prow = part ? part + ((pl_ / Ch) * nb + (KEEP ? 0 : plane % nb)) * (long)(Ch * 10) : nullptr;
#pragma unroll
    for (int i = 0; i < 9; ++i) aw[i] = wave_sum_dpp(aw[i]);
    ab = wave_sum_dpp(ab);
    if (lane < 10) {             // lane i < 9 stores weight sum i, lane 9 the bias sum (every lane holds all ten totals)
      float v = ab;
#pragma unroll
      for (int i = 0; i < 9; ++i) v = lane == i ? aw[i] : v;
      if (prow) prow[lane < 9 ? c * 9 + lane : Ch * 9 + c] = v;
      else atomicAdd(lane < 9 ? dw + c * 9 + lane : db + c, v);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // the next plane's tile stores come after this plane's reads
  }
}

// ---------------------------------------------------------------------------------- small elementwise helpers
__global__ void k_act_fwd(const float* __restrict__ x, float* __restrict__ y, int act, float slope, long n4) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const float4 p = reinterpret_cast<const float4*>(x)[i];
  float v[4] = {p.x, p.y, p.z, p.w};
  apply_act4(v, act, slope);
  reinterpret_cast<float4*>(y)[i] = make_float4(v[0], v[1], v[2], v[3]);
}
// y = res + x * m_elem(i) * m_row(i / row_len): nn.Dropout (p_elem) and/or timm DropPath (p_row, one draw per batch sample,
// row_len = elements per sample) with the residual add of SwinTransformerBlock.forward (pgrm.py:329-330) folded in.
// The backward is the same kernel on the gradient with res = NULL.  In place (y == x) is fine.
__global__ void k_dropout(const float* __restrict__ x, const float* __restrict__ res, float* __restrict__ y, long n4, long row_len,
                          float p_elem, unsigned long long seed_elem, float p_row, unsigned long long seed_row) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const float4 v = reinterpret_cast<const float4*>(x)[i];
  float o[4] = {v.x, v.y, v.z, v.w};
  if (p_elem > 0.f) {
    const float ik = 1.0f / (1.0f - p_elem);
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] *= drop_scale_z(drop_z0(seed_elem, (unsigned long long)(i * 4)) + (unsigned long long)r * DROP_PHI, p_elem, ik);
  }
  if (p_row > 0.f) {
    const float m = drop_scale(seed_row, (unsigned long long)((i * 4) / row_len), p_row, 1.0f / (1.0f - p_row));
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] *= m;
  }
  if (res) {
    const float4 rr = reinterpret_cast<const float4*>(res)[i];
    o[0] += rr.x; o[1] += rr.y; o[2] += rr.z; o[3] += rr.w;
  }
  reinterpret_cast<float4*>(y)[i] = make_float4(o[0], o[1], o[2], o[3]);
}
// y = LayerNorm(x) (E = C), one row per 32 threads
template <int C>
__global__ __launch_bounds__(256) void k_ln_fwd(const float* __restrict__ x, const float* __restrict__ gamma,
                                                 const float* __restrict__ beta, float eps, float* __restrict__ y, long M) {
  constexpr int PER = C / 32;
  const long row = (long)blockIdx.x * 8 + (threadIdx.x >> 5);
  const int t = threadIdx.x & 31;
  if (row >= M) return;
  float xv[PER], s = 0.f;
#pragma unroll
  for (int i = 0; i < PER; ++i) { xv[i] = x[row * C + t + 32 * i]; s += xv[i]; }
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) s += xshfl_v(s, o);
  const float mean = s * (1.0f / C);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < PER; ++i) { const float d = xv[i] - mean; q += d * d; }
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) q += xshfl_v(q, o);
  const float rstd = 1.0f / sqrtf(q * (1.0f / C) + eps);
#pragma unroll
  for (int i = 0; i < PER; ++i) y[row * C + t + 32 * i] = (xv[i] - mean) * rstd * gamma[t + 32 * i] + beta[t + 32 * i];
}
// y (+)= a*x (+ b*z)
__global__ void k_axpby(const float* __restrict__ x, const float* __restrict__ z, float* __restrict__ y, float a, float b,
                        int accumulate, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float v = a * x[i] + (z ? b * z[i] : 0.f);
  y[i] = accumulate ? y[i] + v : v;
}
// rowsum: out[r] += sum_c x[r*cols + c]  (bias gradient of the pointwise conv over the raw (B*Ch, L) view folded per channel)
__global__ __launch_bounds__(256) void k_rowsum_mod(const float* __restrict__ x, float* __restrict__ out, long rows, int cols,
                                                     int mod, float* __restrict__ part = nullptr) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  float s = 0.f;
  if ((cols & 3) == 0) {          // float4 loads, all of a row's loads of a lane issued before the adds
    const float4* xr = reinterpret_cast<const float4*>(x + row * cols);
    float4 a4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
    for (int c = lane; c < (cols >> 2); c += 64) { const float4 v = xr[c]; a4.x += v.x; a4.y += v.y; a4.z += v.z; a4.w += v.w; }
    s = (a4.x + a4.y) + (a4.z + a4.w);
  } else {
    for (int c = lane; c < cols; c += 64) s += x[row * cols + c];
  }
  s = wave_sum(s);
  if (lane == 0) {
    if (part) part[row] = s;      // (rows / mod, mod): added over the leading axis in order by dpmn_rows_reduce_f32
    else atomicAdd(out + row % mod, s);
  }
}

// ---------------------------------------------------------------------------------- PGRM tail (training variant)
// out[b,c,Y,X] = lrelu(c1[b, Y/2, X/2, 4c + 2(Y%2) + X%2]) * wl0[c,Y,X] + sum_i res_i[b,c,Y,X] * wl_i[c,Y,X]   (pgrm.py:560-565)
struct TailElem {
  const float* res[8];
  const float* wl[8];
  float* dres[8];
  float* dwl[8];
  int n;
};
__global__ void k_tail_elem_fwd(const float* __restrict__ c1, const float* __restrict__ wl0, TailElem t, float* __restrict__ out,
                                int B, int H, int W) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int Ho = 2 * H, Wo = 2 * W;
  if (idx >= (long)B * 3 * Ho * Wo) return;
  const int X = idx % Wo, Y = (idx / Wo) % Ho, c = (idx / ((long)Wo * Ho)) % 3, b = idx / ((long)Wo * Ho * 3);
  const long pos = ((long)c * Ho + Y) * Wo + X;
  float v = c1[(((size_t)b * H + Y / 2) * W + X / 2) * 12 + 4 * c + 2 * (Y & 1) + (X & 1)];
  v = (v > 0.f ? v : 0.01f * v) * wl0[pos];
  for (int i = 0; i < t.n; ++i) v += t.res[i][idx] * t.wl[i][pos];
  out[idx] = v;
}
// thread = one (c,Y,X) position, loops over the batch: no atomics for the weight_list gradients
__global__ void k_tail_elem_bwd(const float* __restrict__ dout, const float* __restrict__ c1, const float* __restrict__ wl0,
                                TailElem t, float* __restrict__ dc1, float* __restrict__ dwl0, int B, int H, int W) {
  const long pos = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int Ho = 2 * H, Wo = 2 * W;
  if (pos >= (long)3 * Ho * Wo) return;
  const int X = pos % Wo, Y = (pos / Wo) % Ho, c = pos / ((long)Wo * Ho);
  const int ch = 4 * c + 2 * (Y & 1) + (X & 1);
  const float w0 = wl0[pos];
  float a0 = 0.f, ai[8];
  float wli[8];
  for (int i = 0; i < t.n; ++i) { ai[i] = 0.f; wli[i] = t.wl[i][pos]; }
  // four images per iteration, every load of the four issued before the first store (the pointers in `t` are not restrict: a
  // one-image-per-iteration loop was a chain of ~5 cache latencies per image, 48 images deep); sums in image order as before
  const long istride = (long)3 * Ho * Wo;
  const size_t cstride = (size_t)H * W * 12;
  const size_t cbase = ((size_t)(Y / 2) * W + X / 2) * 12 + ch;
  int b = 0;
  for (; b + 4 <= B; b += 4) {
    float g[4], pre[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      g[u] = dout[(long)(b + u) * istride + pos];
      pre[u] = c1[(size_t)(b + u) * cstride + cbase];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      a0 += g[u] * (pre[u] > 0.f ? pre[u] : 0.01f * pre[u]);
      dc1[(size_t)(b + u) * cstride + cbase] = g[u] * w0 * (pre[u] > 0.f ? 1.f : 0.01f);
    }
    for (int i = 0; i < t.n; ++i) {        // one residual at a time: its four images' loads, then its four stores
      float rv[4], dv[4];
      const float* rp = t.res[i];
      float* dp = t.dres[i];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        rv[u] = rp[(long)(b + u) * istride + pos];
        dv[u] = dp ? dp[(long)(b + u) * istride + pos] : 0.f;
      }
      float acc = ai[i];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        acc += g[u] * rv[u];
        if (dp) dp[(long)(b + u) * istride + pos] = dv[u] + g[u] * wli[i];
      }
      ai[i] = acc;
    }
  }
  for (; b < B; ++b) {
    const long idx = (long)b * istride + pos;
    const float g = dout[idx];
    const size_t ci = (size_t)b * cstride + cbase;
    const float pre = c1[ci];
    a0 += g * (pre > 0.f ? pre : 0.01f * pre);
    dc1[ci] = g * w0 * (pre > 0.f ? 1.f : 0.01f);
    for (int i = 0; i < t.n; ++i) {
      ai[i] += g * t.res[i][idx];
      if (t.dres[i]) t.dres[i][idx] += g * wli[i];
    }
  }
  dwl0[pos] += a0;
  for (int i = 0; i < t.n; ++i) t.dwl[i][pos] += ai[i];
}

// ---------------------------------------------------------------------------------- patch embed backward
// Recomputes the embedding (optionally through prior_fusion) per token, applies the LayerNorm backward and emits
//   dconv (M, C)   gradient wrt the conv output (for dW = dconv^T . patches, db = colsum(dconv))
//   patches (M, 16) the 12 (fused) input values of the token, zero padded
// dgamma / dbeta are accumulated with atomics.
template <int C, bool FUSE>
__global__ __launch_bounds__(256) void k_patch_embed_bwd(const float* __restrict__ img, int cin, const float* __restrict__ pf_w,
                                                          const float* __restrict__ pf_b, const float* __restrict__ pe_w,
                                                          const float* __restrict__ pe_b, const float* __restrict__ ln_w,
                                                          const float* __restrict__ dtok, float* __restrict__ dconv,
                                                          float* __restrict__ patches, float* __restrict__ dgamma,
                                                          float* __restrict__ dbeta, int B, int Hi, int Wi, float* __restrict__ lnpart,
                                                          float p_drop, unsigned long long seed, float* __restrict__ wpart,
                                                          float* __restrict__ dimg) {
  // dimg != null (with wpart, no prior_fusion): the gradient of the 3-channel input image (B, 3, Hi, Wi), written directly -- every
  // pixel belongs to exactly one 2 x 2 patch: din[k] = sum_c dconv[c] W[c][k], four lanes per token reduced by shuffles (instead of
  // a (M, 96) x (96, 16) Linear launch on a freshly transposed weight and a scatter launch into a zero-filled image)
  // wpart != null (C = 96): the block also STORES its partial of the conv's weight and bias gradient -- row blockIdx.x of (blocks,
  // 12 C + C): sum over the block's 64 tokens of dconv[c] * patch[k] | dconv[c] -- and writes no `patches`: the caller adds the rows
  // in block order (dpmn_rows_reduce_f32) instead of running dW = dconv^T . patches as a separate skinny GEMM + reduce + add
  // p_drop > 0: dtok is the gradient BEHIND pos_drop (pgrm.py:550-551): the forward's mask (dpmn_dropout_f32's, regenerated from
  // the seed) is applied as it is loaded -- the same product as a dropout launch over dtok in front of this kernel
  // lnpart != null: the block STORES its [dgamma (C) | dbeta (C)] partial as row blockIdx.x (added in block order by the caller);
  // the 16 tokens of a wave are summed by shuffles, the four waves' sums in wave order -- no atomics, bitwise reproducible
  constexpr int CQ = C / 4, KP = 12;
  __shared__ __attribute__((aligned(16))) float wt[KP * C];
  __shared__ float pfw[57];
  __shared__ float rg[C], rb[C];
  __shared__ float wsum[4][2 * C];
  for (int i = threadIdx.x; i < KP * C; i += 256) wt[(i % KP) * C + i / KP] = pe_w[i];
  if (FUSE && threadIdx.x < 57) pfw[threadIdx.x] = threadIdx.x < 54 ? pf_w[threadIdx.x] : pf_b[threadIdx.x - 54];
  for (int i = threadIdx.x; i < C; i += 256) { rg[i] = 0.f; rb[i] = 0.f; }
  __syncthreads();
  const int Ht = Hi / 2, Wt = Wi / 2;
  const int lane = threadIdx.x & 63;
  long token = (long)blockIdx.x * 64 + (threadIdx.x >> 6) * 16 + (lane >> 2);
  const long ntok = (long)B * Ht * Wt;
  const bool valid = token < ntok;
  if (!valid) token = ntok - 1;
  const int part = lane & 3;
  const int b = token / (Ht * Wt), t = token % (Ht * Wt);
  const int th = t / Wt, tw = t % Wt;
  float in[KP];
  if (FUSE) {
    const int dy = part >> 1, dx = part & 1;
    const int y = th * 2 + dy, x = tw * 2 + dx;
    float a[3] = {pfw[54], pfw[55], pfw[56]};
#pragma unroll
    for (int ci = 0; ci < 2; ++ci)
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int yy = y + ky - 1;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int xx = x + kx - 1;
          const bool inb = yy >= 0 && yy < Hi && xx >= 0 && xx < Wi;
          const float v = inb ? img[(((size_t)b * 2 + ci) * Hi + yy) * Wi + xx] : 0.f;
#pragma unroll
          for (int c = 0; c < 3; ++c) a[c] += pfw[((c * 2 + ci) * 3 + ky) * 3 + kx] * v;
        }
      }
    const int base = lane & ~3;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int pix = 0; pix < 4; ++pix) in[c * 4 + pix] = __shfl(a[c], base + pix, 64);
  } else {
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) in[(c * 2 + dy) * 2 + dx] = img[(((size_t)b * cin + c) * Hi + th * 2 + dy) * Wi + tw * 2 + dx];
  }
  // k outermost: a lane's CQ channels are contiguous in wt[k][:], so the weights arrive as 16-byte LDS reads (CQ / 4 per k) instead
  // of one ds_read_b32 per multiply; every output still adds its products in the order k = 0 .. KP - 1 (same bits)
  float o[CQ], s = 0.f;
#pragma unroll
  for (int i = 0; i < CQ; i += 4) {
    const float4 b4 = *reinterpret_cast<const float4*>(pe_b + part * CQ + i);
    o[i] = b4.x; o[i + 1] = b4.y; o[i + 2] = b4.z; o[i + 3] = b4.w;
  }
#pragma unroll
  for (int k = 0; k < KP; ++k) {
#pragma unroll
    for (int i = 0; i < CQ; i += 4) {
      const float4 w4 = *reinterpret_cast<const float4*>(wt + k * C + part * CQ + i);
      o[i] += w4.x * in[k]; o[i + 1] += w4.y * in[k]; o[i + 2] += w4.z * in[k]; o[i + 3] += w4.w * in[k];
    }
  }
#pragma unroll
  for (int i = 0; i < CQ; ++i) s += o[i];
  s += xshfl<1>(s); s += xshfl<2>(s);
  const float mean = s * (1.0f / C);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < CQ; ++i) { const float d = o[i] - mean; q += d * d; }
  q += xshfl<1>(q); q += xshfl<2>(q);
  const float rstd = 1.0f / sqrtf(q * (1.0f / C) + 1e-5f);
  float dg[CQ], xh[CQ], s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < CQ; ++i) {
    const int c = part * CQ + i;
    float d = valid ? dtok[(size_t)token * C + c] : 0.f;
    if (p_drop > 0.f) d *= drop_scale(seed, (unsigned long long)token * C + c, p_drop, 1.0f / (1.0f - p_drop));
    xh[i] = (o[i] - mean) * rstd;
    dg[i] = d * ln_w[c];
    s1 += dg[i];
    s2 += dg[i] * xh[i];
    if (lnpart) {
      float a = d * xh[i], bsum = d;            // sum over the wave's 16 tokens (lanes with the same channel quarter)
      a += xshfl<4>(a); a += xshfl<8>(a); a += xshfl<16>(a); a += xshfl<32>(a);
      bsum += xshfl<4>(bsum); bsum += xshfl<8>(bsum); bsum += xshfl<16>(bsum); bsum += xshfl<32>(bsum);
      if (lane < 4) { wsum[threadIdx.x >> 6][c] = a; wsum[threadIdx.x >> 6][C + c] = bsum; }
    } else {
      atomicAdd(&rg[c], d * xh[i]);
      atomicAdd(&rb[c], d);
    }
  }
  s1 += xshfl<1>(s1); s1 += xshfl<2>(s1);
  s2 += xshfl<1>(s2); s2 += xshfl<2>(s2);
  s1 *= (1.0f / C); s2 *= (1.0f / C);
  float dcv[CQ];
#pragma unroll
  for (int i = 0; i < CQ; ++i) dcv[i] = valid ? rstd * (dg[i] - s1 - xh[i] * s2) : 0.f;
  if (valid) {
#pragma unroll
    for (int i = 0; i < CQ; ++i) dconv[(size_t)token * C + part * CQ + i] = dcv[i];
    if (part == 0 && !wpart) {
#pragma unroll
      for (int k = 0; k < 16; ++k) patches[(size_t)token * 16 + k] = k < KP ? in[k] : 0.f;
    }
  }
  if (dimg) {
    float dk[KP];
#pragma unroll
    for (int k = 0; k < KP; ++k) {
      float a = 0.f;
#pragma unroll
      for (int i = 0; i < CQ; ++i) a = fmaf(dcv[i], wt[k * C + part * CQ + i], a);
      a += xshfl<1>(a); a += xshfl<2>(a);
      dk[k] = a;
    }
    if (valid && part == 0) {
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
          float2 v2 = make_float2(dk[(c * 2 + dy) * 2], dk[(c * 2 + dy) * 2 + 1]);
          *reinterpret_cast<float2*>(dimg + (((size_t)b * 3 + c) * Hi + th * 2 + dy) * Wi + tw * 2) = v2;
        }
    }
  }
  if constexpr (C == 96) {
    if (wpart) {
      __shared__ float sdc[64][C + 1];
      __shared__ float sin_[64][KP + 1];
      const int tl = (threadIdx.x >> 6) * 16 + (lane >> 2);      // token slot of the block
#pragma unroll
      for (int i = 0; i < CQ; ++i) sdc[tl][part * CQ + i] = dcv[i];
      if (part == 0) {
#pragma unroll
        for (int k = 0; k < KP; ++k) sin_[tl][k] = valid ? in[k] : 0.f;
      }
      __syncthreads();
      if (threadIdx.x < 2 * C) {          // thread -> (channel c, half of the 12 patch inputs)
        const int c = threadIdx.x % C, kg = threadIdx.x / C;
        float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, bsum = 0.f;
        for (int tk = 0; tk < 64; ++tk) {      // tokens in slot order: a fixed summation order
          const float dv_ = sdc[tk][c];
          bsum += dv_;
#pragma unroll
          for (int k = 0; k < 6; ++k) acc[k] = fmaf(dv_, sin_[tk][kg * 6 + k], acc[k]);
        }
        float* row = wpart + (size_t)blockIdx.x * (KP * C + C);
#pragma unroll
        for (int k = 0; k < 6; ++k) row[c * KP + kg * 6 + k] = acc[k];
        if (kg == 0) row[KP * C + c] = bsum;
      }
    }
  }
  __syncthreads();
  if (lnpart) {
    for (int i = threadIdx.x; i < 2 * C; i += 256)
      lnpart[(size_t)blockIdx.x * 2 * C + i] = ((wsum[0][i] + wsum[1][i]) + wsum[2][i]) + wsum[3][i];
  } else {
    for (int i = threadIdx.x; i < C; i += 256) { atomicAdd(dgamma + i, rg[i]); atomicAdd(dbeta + i, rb[i]); }
  }
}
// din (M,16): gradient wrt the 12 patch inputs of each token.  Without prior fusion: scatter (+=) into the NCHW image
// gradient.  With prior fusion: accumulate the 3x3 conv weight / bias gradients (the text prior itself needs no gradient).
__global__ void k_patch_scatter(const float* __restrict__ din, float* __restrict__ dimg, int cimg, int B, int Hi, int Wi) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)B * 3 * Hi * Wi) return;
  const int x = idx % Wi, y = (idx / Wi) % Hi, c = (idx / ((long)Wi * Hi)) % 3, b = idx / ((long)Wi * Hi * 3);
  const long token = ((long)b * (Hi / 2) + y / 2) * (Wi / 2) + x / 2;
  dimg[(((size_t)b * cimg + c) * Hi + y) * Wi + x] += din[token * 16 + (c * 2 + (y & 1)) * 2 + (x & 1)];
}
__global__ __launch_bounds__(256) void k_prior_fusion_wgrad(const float* __restrict__ din, const float* __restrict__ prior,
                                                             float* __restrict__ dpf_w, float* __restrict__ dpf_b, int B, int Hi,
                                                             int Wi, float* __restrict__ part) {
  // part != null: the block stores its [dw (54) | db (3)] partial as row blockIdx.x; the waves' sums are added in wave order
  __shared__ float acc[57];
  __shared__ float wacc[4][57];
  if (threadIdx.x < 57) acc[threadIdx.x] = 0.f;
  __syncthreads();
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  float loc[57];
#pragma unroll
  for (int i = 0; i < 57; ++i) loc[i] = 0.f;
  if (idx < (long)B * Hi * Wi) {
    const int x = idx % Wi, y = (idx / Wi) % Hi, b = idx / ((long)Wi * Hi);
    const long token = ((long)b * (Hi / 2) + y / 2) * (Wi / 2) + x / 2;
    float d[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { d[c] = din[token * 16 + (c * 2 + (y & 1)) * 2 + (x & 1)]; loc[54 + c] = d[c]; }
#pragma unroll
    for (int ci = 0; ci < 2; ++ci)
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int yy = y + ky - 1, xx = x + kx - 1;
          const float v = (yy >= 0 && yy < Hi && xx >= 0 && xx < Wi) ? prior[(((size_t)b * 2 + ci) * Hi + yy) * Wi + xx] : 0.f;
#pragma unroll
          for (int c = 0; c < 3; ++c) loc[((c * 2 + ci) * 3 + ky) * 3 + kx] = d[c] * v;
        }
  }
#pragma unroll
  for (int i = 0; i < 57; ++i) {
    const float s = wave_sum(loc[i]);
    if ((threadIdx.x & 63) == 0) {
      if (part) wacc[threadIdx.x >> 6][i] = s;
      else atomicAdd(&acc[i], s);
    }
  }
  __syncthreads();
  if (part) {
    if (threadIdx.x < 57) part[(size_t)blockIdx.x * 57 + threadIdx.x] = ((wacc[0][threadIdx.x] + wacc[1][threadIdx.x]) + wacc[2][threadIdx.x]) + wacc[3][threadIdx.x];
  } else if (threadIdx.x < 54) atomicAdd(dpf_w + threadIdx.x, acc[threadIdx.x]);
  else if (threadIdx.x < 57) atomicAdd(dpf_b + threadIdx.x - 54, acc[threadIdx.x]);
}

}  // namespace

int dpmn_wattn_bwd_mfma(int ws, int D, const float* q, const float* kv, const float* tbl, const float* dout, float* dq, float* dkv,
                        float* dtable, int B, int H, int W, int C, int g, int shift, float p_drop, unsigned long long seed,
                        hipStream_t st, int part_mode, int* rows);      // wattn_bwd_mfma.hip

extern "C" {

int dpmn_window_attn_bwd_f32(const float* q, const float* kv, const float* const* bias_tables, const int* windows,
                             const int* shifts, int n_groups, int heads_per_group, const float* dout, float* dq, float* dkv,
                             float* const* dtables, int B, int H, int W, int C, dpmn_stream_t stream) {
  return dpmn_window_attn_drop_bwd_f32(q, kv, bias_tables, windows, shifts, n_groups, heads_per_group, dout, dq, dkv, dtables, B,
                                       H, W, C, 0.f, 0ull, stream);
}

static int window_attn_bwd_impl(const float* q, const float* kv, const float* const* bias_tables, const int* windows,
                                const int* shifts, int n_groups, int heads_per_group, const float* dout, float* dq, float* dkv,
                                float* const* dtables, int B, int H, int W, int C, float p_drop, unsigned long long seed,
                                dpmn_stream_t stream, int part_mode, int* rows_out) {
  DPMN_REQUIRE(q && kv && bias_tables && windows && shifts && dout && dq && dkv && dtables, "window_attn_bwd: null pointer");
  DPMN_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "window_attn_bwd: attn_drop must be in [0, 1)");
  DPMN_REQUIRE(heads_per_group == 2 && C % n_groups == 0 && (H * W) % 64 == 0, "window_attn_bwd: unsupported geometry");
  const int D = C / n_groups / heads_per_group;
  hipStream_t st = as_stream(stream);
  for (int g = 0; g < n_groups; ++g) {
    const int ws = windows[g], sh = shifts[g];
    DPMN_REQUIRE(H % ws == 0 && W % ws == 0 && sh >= 0 && sh < ws, "window_attn_bwd: bad window / shift");
    int rc = DPMN_ERR_ARG;
    static const int wb_mfma = getenv("DPMN_WATTN_MFMA") ? atoi(getenv("DPMN_WATTN_MFMA")) : 1;
    const long slabs = (long)B * (H * W / 64);
    if (ws == 8 && D == 16 && wb_mfma) {
      if (rows_out) rows_out[g] = (int)slabs;
      if (p_drop > 0.f)
        hipLaunchKernelGGL((k_window_attn8_bwd_mfma<true>), dim3((unsigned)slabs), dim3(128), 0, st, q, kv, bias_tables[g],
                           dout, dq, dkv, dtables[g], H, W, C, g, sh, p_drop, seed, part_mode);
      else
        hipLaunchKernelGGL((k_window_attn8_bwd_mfma<false>), dim3((unsigned)slabs), dim3(128), 0, st, q, kv, bias_tables[g],
                           dout, dq, dkv, dtables[g], H, W, C, g, sh, 0.f, 0ull, part_mode);
      DPMN_CHECK_LAUNCH();
      continue;
    }
    if (wb_mfma && ((ws == 16 && (D == 32 || D == 16)) || (ws == 8 && D == 32))) {
      // 256- / 64-token windows on the matrix cores (wattn_bwd_mfma.hip): one block per (window, head)
      rc = dpmn_wattn_bwd_mfma(ws, D, q, kv, bias_tables[g], dout, dq, dkv, dtables[g], B, H, W, C, g, sh, p_drop, seed, st, part_mode,
                               rows_out ? rows_out + g : nullptr);
      if (rc != DPMN_OK) return rc;
      continue;
    }
    if (rows_out) rows_out[g] = (int)((slabs + 1) / 2);
#define WB_CASE(WSV, DV) if (ws == WSV && D == DV) rc = p_drop > 0.f \
      ? launch_wattn_bwd<WSV, DV, true>(q, kv, bias_tables[g], dout, dq, dkv, dtables[g], B, H, W, C, g, sh, p_drop, seed, st, part_mode) \
      : launch_wattn_bwd<WSV, DV, false>(q, kv, bias_tables[g], dout, dq, dkv, dtables[g], B, H, W, C, g, sh, 0.f, 0ull, st, part_mode); else
    WB_CASE(2, 16) WB_CASE(4, 16) WB_CASE(8, 16) WB_CASE(4, 32) WB_CASE(8, 32)
    return dpmn_set_error(DPMN_ERR_ARG, "window_attn_bwd: unsupported (window, head_dim)");
#undef WB_CASE
    if (rc != DPMN_OK) return rc;
  }
  return DPMN_OK;
}

int dpmn_window_attn_drop_bwd_f32(const float* q, const float* kv, const float* const* bias_tables, const int* windows,
                                  const int* shifts, int n_groups, int heads_per_group, const float* dout, float* dq, float* dkv,
                                  float* const* dtables, int B, int H, int W, int C, float p_drop, unsigned long long seed,
                                  dpmn_stream_t stream) {
  return window_attn_bwd_impl(q, kv, bias_tables, windows, shifts, n_groups, heads_per_group, dout, dq, dkv, dtables, B, H, W, C, p_drop, seed,
                              stream, 0, nullptr);
}

// the same without atomics on the bias-table gradients: dtable_parts[g] is a (rows, (2 ws_g - 1)^2 * 2) buffer with
// rows = dpmn_window_attn_bwd_part_rows(B, H, W, ws_g, D) >= rows_out[g]: every block stores its partial row, the caller adds the rows
// in order (dpmn_rows_reduce_f32)
int dpmn_window_attn_drop_bwd_det_f32(const float* q, const float* kv, const float* const* bias_tables, const int* windows,
                                      const int* shifts, int n_groups, int heads_per_group, const float* dout, float* dq, float* dkv,
                                      float* const* dtable_parts, int* rows_out, int B, int H, int W, int C, float p_drop,
                                      unsigned long long seed, dpmn_stream_t stream) {
  DPMN_REQUIRE(rows_out, "window_attn_bwd_det: null pointer");
  return window_attn_bwd_impl(q, kv, bias_tables, windows, shifts, n_groups, heads_per_group, dout, dq, dkv, dtable_parts, B, H, W, C, p_drop,
                              seed, stream, 1, rows_out);
}

int dpmn_window_attn_bwd_part_rows(int B, int H, int W) {
  return B * (H * W / 64);          // the largest row count any (window, head_dim) kernel uses: one block per 64-token slab
}

int dpmn_sk_select_only_f32(const float* cat, const float* attn_vec, float* V, long M, int L, int C, int G, dpmn_stream_t stream) {
  DPMN_REQUIRE(cat && attn_vec && V && M > 0 && C % G == 0, "sk_select_only: bad arguments");
  const long total = M * (C / G);
  hipLaunchKernelGGL(k_sk_select, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), cat, attn_vec, V, M, L, C, G);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

int dpmn_sk_select_bwd_f32(const float* cat, const float* attn_vec, const float* dV, float* dcat, float* dA, int B, int L, int C,
                           int G, dpmn_stream_t stream) {
  DPMN_REQUIRE(cat && attn_vec && dV && dcat && dA && B > 0, "sk_select_bwd: bad arguments");
  const int rows = 32;
  hipLaunchKernelGGL(k_sk_select_bwd, dim3(cdiv(L, rows), B), dim3(128), 0, as_stream(stream), cat, attn_vec, dV, dcat, dA, L, C, G, rows, 0, 0);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

// the same without atomics: dA_part is (ceil(L / 32), B, C); dpmn_sk_gate_bwd_det_f32 adds the rows in order
int dpmn_sk_select_bwd_det_f32(const float* cat, const float* attn_vec, const float* dV, float* dcat, float* dA_part, int B, int L, int C,
                               int G, dpmn_stream_t stream) {
  DPMN_REQUIRE(cat && attn_vec && dV && dcat && dA_part && B > 0, "sk_select_bwd_det: bad arguments");
  const int rows = 32;
  hipLaunchKernelGGL(k_sk_select_bwd, dim3(cdiv(L, rows), B), dim3(128), 0, as_stream(stream), cat, attn_vec, dV, dcat, dA_part, L, C, G, rows, 1, 0);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

// the same with dcat WRITTEN (= A dV) instead of accumulated into: for a caller whose buffer is fresh (no zero fill, no read)
int dpmn_sk_select_bwd_det_set_f32(const float* cat, const float* attn_vec, const float* dV, float* dcat, float* dA_part, int B, int L, int C,
                                   int G, dpmn_stream_t stream) {
  DPMN_REQUIRE(cat && attn_vec && dV && dcat && dA_part && B > 0, "sk_select_bwd_det_set: bad arguments");
  const int rows = 32;
  hipLaunchKernelGGL(k_sk_select_bwd, dim3(cdiv(L, rows), B), dim3(128), 0, as_stream(stream), cat, attn_vec, dV, dcat, dA_part, L, C, G, rows, 1, 1);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

int dpmn_sk_gate_bwd_f32(const float* colsum_partials, int parts_per_image, int L, const float* fc1_w, const float* fc1_b,
                         const float* fc2_w, const float* attn_vec, const float* dA, float* dS, float* dfc1_w, float* dfc1_b,
                         float* dfc2_w, float* dfc2_b, int B, int C, int G, int dmid, dpmn_stream_t stream) {
  DPMN_REQUIRE(colsum_partials && fc1_w && fc1_b && fc2_w && attn_vec && dA && dS && dfc1_w && dfc1_b && dfc2_w && dfc2_b,
               "sk_gate_bwd: null pointer");
  hipLaunchKernelGGL(k_sk_gate_bwd, dim3(B), dim3(128), (size_t)(3 * C + 3 * dmid) * 4, as_stream(stream), colsum_partials,
                     parts_per_image, L, fc1_w, fc1_b, fc2_w, attn_vec, dA, dS, dfc1_w, dfc1_b, dfc2_w, dfc2_b, C, G, dmid, 0,
                     static_cast<float*>(nullptr), static_cast<float*>(nullptr));
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

// the same without atomics: dA_part (nparts, B, C) from dpmn_sk_select_bwd_det_f32; the weight gradients as per-image rows
// wpart2 (B, C dmid + C) = [dfc2_w | dfc2_b], wpart1 (B, dmid C + dmid) = [dfc1_w | dfc1_b] for dpmn_rows_reduce_f32
int dpmn_sk_gate_bwd_det_f32(const float* colsum_partials, int parts_per_image, int L, const float* fc1_w, const float* fc1_b,
                             const float* fc2_w, const float* attn_vec, const float* dA_part, int nparts, float* dS, float* wpart2,
                             float* wpart1, int B, int C, int G, int dmid, dpmn_stream_t stream) {
  DPMN_REQUIRE(colsum_partials && fc1_w && fc1_b && fc2_w && attn_vec && dA_part && nparts > 0 && dS && wpart2 && wpart1,
               "sk_gate_bwd_det: null pointer");
  hipLaunchKernelGGL(k_sk_gate_bwd, dim3(B), dim3(128), (size_t)(3 * C + 3 * dmid) * 4, as_stream(stream), colsum_partials,
                     parts_per_image, L, fc1_w, fc1_b, fc2_w, attn_vec, dA_part, dS, static_cast<float*>(nullptr), static_cast<float*>(nullptr),
                     static_cast<float*>(nullptr), static_cast<float*>(nullptr), C, G, dmid, nparts, wpart2, wpart1);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

int dpmn_sk_feats_grad_f32(const float* dout, const float* feats, const float* dS, float* dfeats, long M, int L, int C,
                           dpmn_stream_t stream) {
  DPMN_REQUIRE(dout && feats && dS && dfeats && M > 0, "sk_feats_grad: bad arguments");
  const long total = M * C;
  hipLaunchKernelGGL(k_sk_feats_grad, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), dout, feats, dS, dfeats, M, L, C);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

// KEEP variant: persistent, 2 blocks per CU (256 registers per wave for the prefetch), never more blocks than planes / 4
// rows per band of the generic (non-KEEP) kernel: whole planes up to 32 x 32, 16-row bands above; DPMN_DW_BAND overrides (pgrm.hip has the twin)
static int dwconv_bwd_band_rows(int r) {
  static const int env = getenv("DPMN_DW_BAND") ? atoi(getenv("DPMN_DW_BAND")) : -1;
  int rb = env >= 0 ? env : (r > 32 && r % 16 == 0 ? 16 : r);
  if (rb <= 0 || rb > r || r % rb != 0) rb = r;
  return rb;
}

static long dwconv_bwd_grid(long planes) {
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    hipDeviceProp_t prop;
    n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
  }
  static const int bpc = getenv("DPMN_DWB_BPC") ? atoi(getenv("DPMN_DWB_BPC")) : 2;
  const long want = (long)bpc * n_cu, need = (planes + 3) / 4;
  return want < need ? want : need;
}

int dpmn_dwconv3x3_bwd_f32(const float* P, const float* dg, const float* w, float* dP, float* dw, float* db, int B, int Ch, int r,
                           dpmn_stream_t stream) {
  DPMN_REQUIRE(P && dg && w && dP && dw && db && r >= 4 && r <= 64 && r % 4 == 0, "dwconv_bwd: plane side must be a multiple of 4 in [4, 64]");
  const long planes = (long)B * Ch;
  const int rb = dwconv_bwd_band_rows(r);
  const size_t smem = (size_t)8 * (rb + 2) * (r + 8) * 4;
  if (smem > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dwconv_bwd<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipLaunchKernelGGL(k_dwconv_bwd<false>, dim3((unsigned)((planes * (r / rb) + 3) / 4)), dim3(256), smem, as_stream(stream), P, dg, w, dP, dw, db, Ch, r,
                     planes, static_cast<const float*>(nullptr), 0, 0, 0.f, 0ull, static_cast<float*>(nullptr), rb);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

int dpmn_dwconv3x3_bwd_fused_f32(const float* P, const float* dg, const float* gpre, const float* w, float* dP, float* dw, float* db,
                                 int in_gelu, int out_gelu_bwd, float p_drop, unsigned long long seed, int B, int Ch, int r,
                                 dpmn_stream_t stream) {
  DPMN_REQUIRE(P && dg && w && dP && dw && db && r >= 4 && r <= 64 && r % 4 == 0, "dwconv_bwd_fused: plane side must be a multiple of 4 in [4, 64]");
  DPMN_REQUIRE(!out_gelu_bwd || in_gelu, "dwconv_bwd_fused: out_gelu_bwd needs P to be the pre-activation (in_gelu)");
  const long planes = (long)B * Ch;
  const bool keep = in_gelu && out_gelu_bwd && gpre && r == 32;
  const int rb = keep ? r : dwconv_bwd_band_rows(r);
  const size_t smem = (size_t)8 * (rb + 2) * (r + 8) * 4;
  // dP (9 taps) + dw (9 taps) = 36 FLOPs per element; P, dg (, gpre) read, dP written
  ProfScope prof(PT_DWCONV_BWD, as_stream(stream), 36.0 * planes * r * r, 4.0 * (gpre ? 4 : 3) * (double)planes * r * r);
  if (smem > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dwconv_bwd<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  if (keep) hipLaunchKernelGGL(k_dwconv_bwd<true>, dim3((unsigned)dwconv_bwd_grid(planes)), dim3(256), smem, as_stream(stream), P, dg, w, dP, dw, db, Ch, r,
                     planes, gpre, in_gelu, out_gelu_bwd, p_drop, seed);
  else hipLaunchKernelGGL(k_dwconv_bwd<false>, dim3((unsigned)((planes * (r / rb) + 3) / 4)), dim3(256), smem, as_stream(stream), P, dg, w, dP, dw, db, Ch, r,
                     planes, gpre, in_gelu, out_gelu_bwd, p_drop, seed, static_cast<float*>(nullptr), rb);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

// workspace of the atomics-free variant: one [Ch * 9 | Ch] partial row per (image, band)
size_t dpmn_dwconv3x3_bwd_det_bytes(int B, int Ch, int r) {
  if (B <= 0 || Ch <= 0 || r <= 0) return 0;
  const int rb = dwconv_bwd_band_rows(r);      // (the whole-plane KEEP kernel of the 32 x 32 planes needs B rows: never more than this)
  return (size_t)B * (r / rb) * Ch * 10 * sizeof(float);
}

// the same without atomics: per-(image, band) [Ch * 9 | Ch] partial rows in ws (dpmn_dwconv3x3_bwd_det_bytes), added in row order -- bitwise reproducible
int dpmn_dwconv3x3_bwd_fused_det_f32(const float* P, const float* dg, const float* gpre, const float* w, float* dP, float* dw, float* db,
                                     int in_gelu, int out_gelu_bwd, float p_drop, unsigned long long seed, int B, int Ch, int r,
                                     float* ws, size_t ws_bytes, dpmn_stream_t stream) {
  DPMN_REQUIRE(P && dg && w && dP && dw && db && ws && r >= 4 && r <= 64 && r % 4 == 0, "dwconv_bwd_fused_det: bad arguments");
  DPMN_REQUIRE(!out_gelu_bwd || in_gelu, "dwconv_bwd_fused_det: out_gelu_bwd needs P to be the pre-activation (in_gelu)");
  if (dpmn_dwconv3x3_bwd_det_bytes(B, Ch, r) > ws_bytes) return dpmn_set_error(DPMN_ERR_WORKSPACE, "dwconv_bwd_fused_det: workspace too small (dpmn_dwconv3x3_bwd_det_bytes)");
  const long planes = (long)B * Ch;
  const bool keep = in_gelu && out_gelu_bwd && gpre && r == 32;
  const int rb = keep ? r : dwconv_bwd_band_rows(r);
  const size_t smem = (size_t)8 * (rb + 2) * (r + 8) * 4;
  // dP (9 taps) + dw (9 taps) = 36 FLOPs per element; P, dg (, gpre) read, dP written
  ProfScope prof(PT_DWCONV_BWD, as_stream(stream), 36.0 * planes * r * r, 4.0 * (gpre ? 4 : 3) * (double)planes * r * r);
  if (smem > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dwconv_bwd<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  if (keep) hipLaunchKernelGGL(k_dwconv_bwd<true>, dim3((unsigned)dwconv_bwd_grid(planes)), dim3(256), smem, as_stream(stream), P, dg, w, dP, dw, db, Ch, r,
                     planes, gpre, in_gelu, out_gelu_bwd, p_drop, seed, ws);
  else hipLaunchKernelGGL(k_dwconv_bwd<false>, dim3((unsigned)((planes * (r / rb) + 3) / 4)), dim3(256), smem, as_stream(stream), P, dg, w, dP, dw, db, Ch, r,
                     planes, gpre, in_gelu, out_gelu_bwd, p_drop, seed, ws, rb);
  DPMN_CHECK_LAUNCH();
  prof.close();      // (the row reduction below is not part of the family)
  return dpmn_rows_reduce_f32(ws, dw, db, Ch * 9, Ch, B * (r / rb), stream);
}

int dpmn_rowsum_mod_det_f32(const float* x, float* out, long rows, int cols, int mod, float* ws, size_t ws_bytes, dpmn_stream_t stream) {
  DPMN_REQUIRE(x && out && ws && rows > 0 && cols > 0 && mod > 0 && rows % mod == 0, "rowsum_mod_det: bad arguments (rows a multiple of mod)");
  if ((size_t)rows * sizeof(float) > ws_bytes) return dpmn_set_error(DPMN_ERR_WORKSPACE, "rowsum_mod_det: workspace too small");
  hipLaunchKernelGGL(k_rowsum_mod, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, as_stream(stream), x, out, rows, cols, mod, ws);
  DPMN_CHECK_LAUNCH();
  return dpmn_rows_reduce_f32(ws, out, nullptr, mod, 0, (int)(rows / mod), stream);
}

int dpmn_act_fwd_f32(const float* x, float* y, int act, float slope, long n, dpmn_stream_t stream) {
  DPMN_REQUIRE(x && y && n > 0 && n % 4 == 0, "act_fwd: bad arguments");
  hipLaunchKernelGGL(k_act_fwd, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, as_stream(stream), x, y, act, slope, n / 4);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

int dpmn_dropout_f32(const float* x, const float* res, float* y, long n, long row_len, float p_elem,
                     unsigned long long seed_elem, float p_row, unsigned long long seed_row, dpmn_stream_t stream) {
  DPMN_REQUIRE(x && y && n > 0 && n % 4 == 0, "dropout: n must be a positive multiple of 4");
  DPMN_REQUIRE(p_elem >= 0.f && p_elem < 1.f && p_row >= 0.f && p_row < 1.f, "dropout: probabilities must be in [0, 1)");
  DPMN_REQUIRE(p_row == 0.f || (row_len > 0 && row_len % 4 == 0 && n % row_len == 0), "dropout: row_len must divide n (multiple of 4)");
  hipLaunchKernelGGL(k_dropout, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, as_stream(stream), x, res, y, n / 4,
                     row_len > 0 ? row_len : n, p_elem, seed_elem, p_row, seed_row);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

int dpmn_layernorm_f32(const float* x, const float* gamma, const float* beta, float eps, float* y, long M, int C,
                       dpmn_stream_t stream) {
  DPMN_REQUIRE(x && gamma && beta && y && M > 0, "layernorm: bad arguments");
  const unsigned blocks = (unsigned)((M + 7) / 8);
  if (C == 96) hipLaunchKernelGGL((k_ln_fwd<96>), dim3(blocks), dim3(256), 0, as_stream(stream), x, gamma, beta, eps, y, M);
  else if (C == 192) hipLaunchKernelGGL((k_ln_fwd<192>), dim3(blocks), dim3(256), 0, as_stream(stream), x, gamma, beta, eps, y, M);
  else if (C == 64) hipLaunchKernelGGL((k_ln_fwd<64>), dim3(blocks), dim3(256), 0, as_stream(stream), x, gamma, beta, eps, y, M);
  else if (C == 512) hipLaunchKernelGGL((k_ln_fwd<512>), dim3(blocks), dim3(256), 0, as_stream(stream), x, gamma, beta, eps, y, M);
  else return dpmn_set_error(DPMN_ERR_ARG, "layernorm: C must be 64, 96, 192 or 512");
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

int dpmn_axpby_f32(const float* x, const float* z, float* y, float a, float b, int accumulate, long n, dpmn_stream_t stream) {
  DPMN_REQUIRE(x && y && n > 0, "axpby: bad arguments");
  hipLaunchKernelGGL(k_axpby, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), x, z, y, a, b, accumulate, n);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

int dpmn_rowsum_mod_f32(const float* x, float* out, long rows, int cols, int mod, dpmn_stream_t stream) {
  DPMN_REQUIRE(x && out && rows > 0 && cols > 0 && mod > 0, "rowsum_mod: bad arguments");
  hipLaunchKernelGGL(k_rowsum_mod, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, as_stream(stream), x, out, rows, cols, mod);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

int dpmn_pgrm_tail_elem_f32(const float* c1, const float* const* weight_list, const float* const* residuals, int n_residuals,
                            float* out, int B, int H, int W, dpmn_stream_t stream) {
  DPMN_REQUIRE(c1 && weight_list && out && n_residuals >= 0 && n_residuals <= 8, "tail_elem: bad arguments");
  TailElem t{};
  for (int i = 1; i < n_residuals; ++i) { t.res[t.n] = residuals[i]; t.wl[t.n] = weight_list[i]; ++t.n; }   // quirk Q11
  const long total = (long)B * 3 * 4 * H * W;
  hipLaunchKernelGGL(k_tail_elem_fwd, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), c1, weight_list[0], t, out, B, H, W);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

int dpmn_pgrm_tail_elem_bwd_f32(const float* dout, const float* c1, const float* const* weight_list,
                                const float* const* residuals, float* const* dresiduals, float* const* dweight_list,
                                int n_residuals, float* dc1, int B, int H, int W, dpmn_stream_t stream) {
  DPMN_REQUIRE(dout && c1 && weight_list && dweight_list && dc1 && n_residuals >= 0 && n_residuals <= 8, "tail_elem_bwd: bad arguments");
  TailElem t{};
  for (int i = 1; i < n_residuals; ++i) {
    t.res[t.n] = residuals[i]; t.wl[t.n] = weight_list[i]; t.dres[t.n] = dresiduals ? dresiduals[i] : nullptr; t.dwl[t.n] = dweight_list[i];
    ++t.n;
  }
  const long total = (long)3 * 4 * H * W;
  hipLaunchKernelGGL(k_tail_elem_bwd, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), dout, c1, weight_list[0],
                     t, dc1, dweight_list[0], B, H, W);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

static int patch_embed_bwd_impl(const float* img, int cin, const float* pf_w, const float* pf_b, const float* pe_w,
                                const float* pe_b, const float* ln_w, const float* dtok, float* dconv, float* patches,
                                float* dgamma, float* dbeta, float* part, int B, int Hi, int Wi, int C, dpmn_stream_t stream,
                                float p_drop = 0.f, unsigned long long seed = 0ull, float* wpart = nullptr, float* dimg = nullptr) {
  DPMN_REQUIRE(!wpart || C == 96, "patch_embed_bwd: the in-kernel weight-gradient partials exist for embed_dim 96");
  DPMN_REQUIRE(!dimg || (cin == 3 && !pf_w && Wi % 2 == 0), "patch_embed_bwd: the direct image gradient is for a 3-channel input without prior_fusion");
  DPMN_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "patch_embed_bwd: drop probability must be in [0, 1)");
  DPMN_REQUIRE(img && pe_w && pe_b && ln_w && dtok && dconv && patches && ((dgamma && dbeta) || part), "patch_embed_bwd: null pointer");
  const long tokens_n = (long)B * (Hi / 2) * (Wi / 2);
  dim3 grid((unsigned)((tokens_n + 63) / 64));
  hipStream_t st = as_stream(stream);
#define PB_LAUNCH(CV, FV) hipLaunchKernelGGL((k_patch_embed_bwd<CV, FV>), grid, dim3(256), 0, st, img, cin, pf_w, pf_b, pe_w, pe_b, ln_w, dtok, dconv, patches, dgamma, dbeta, B, Hi, Wi, part, p_drop, seed, wpart, dimg)
  if (C == 96 && pf_w) PB_LAUNCH(96, true);
  else if (C == 96) PB_LAUNCH(96, false);
  else if (C == 192 && pf_w) PB_LAUNCH(192, true);
  else if (C == 192) PB_LAUNCH(192, false);
  else return dpmn_set_error(DPMN_ERR_ARG, "patch_embed_bwd: embed dim must be 96 or 192");
#undef PB_LAUNCH
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

int dpmn_patch_embed_bwd_f32(const float* img, int cin, const float* pf_w, const float* pf_b, const float* pe_w,
                             const float* pe_b, const float* ln_w, const float* dtok, float* dconv, float* patches,
                             float* dgamma, float* dbeta, int B, int Hi, int Wi, int C, dpmn_stream_t stream) {
  return patch_embed_bwd_impl(img, cin, pf_w, pf_b, pe_w, pe_b, ln_w, dtok, dconv, patches, dgamma, dbeta, nullptr, B, Hi, Wi, C, stream);
}

// the same without atomics: ln_part is (ceil(tokens / 64), 2 C) rows of [dgamma | dbeta] partial sums for dpmn_rows_reduce_f32
int dpmn_patch_embed_bwd_det_f32(const float* img, int cin, const float* pf_w, const float* pf_b, const float* pe_w,
                                 const float* pe_b, const float* ln_w, const float* dtok, float* dconv, float* patches,
                                 float* ln_part, int B, int Hi, int Wi, int C, dpmn_stream_t stream) {
  DPMN_REQUIRE(ln_part, "patch_embed_bwd_det: null pointer");
  return patch_embed_bwd_impl(img, cin, pf_w, pf_b, pe_w, pe_b, ln_w, dtok, dconv, patches, nullptr, nullptr, ln_part, B, Hi, Wi, C, stream);
}

int dpmn_patch_embed_bwd_det_drop_f32(const float* img, int cin, const float* pf_w, const float* pf_b, const float* pe_w,
                                      const float* pe_b, const float* ln_w, const float* dtok, float* dconv, float* patches,
                                      float* ln_part, int B, int Hi, int Wi, int C, float p_drop, unsigned long long seed,
                                      dpmn_stream_t stream) {
  DPMN_REQUIRE(ln_part, "patch_embed_bwd_det: null pointer");
  return patch_embed_bwd_impl(img, cin, pf_w, pf_b, pe_w, pe_b, ln_w, dtok, dconv, patches, nullptr, nullptr, ln_part, B, Hi, Wi, C, stream,
                              p_drop, seed);
}

int dpmn_patch_embed_bwd_det_wgrad_f32(const float* img, int cin, const float* pf_w, const float* pf_b, const float* pe_w,
                                       const float* pe_b, const float* ln_w, const float* dtok, float* dconv, float* ln_part,
                                       float* w_part, float* dimg, int B, int Hi, int Wi, int C, float p_drop, unsigned long long seed,
                                       dpmn_stream_t stream) {
  DPMN_REQUIRE(ln_part && w_part, "patch_embed_bwd_det_wgrad: null pointer");
  return patch_embed_bwd_impl(img, cin, pf_w, pf_b, pe_w, pe_b, ln_w, dtok, dconv, /* patches: not written */ dconv, nullptr, nullptr, ln_part, B,
                              Hi, Wi, C, stream, p_drop, seed, w_part, dimg);
}

int dpmn_patch_scatter_f32(const float* din, float* dimg, int cimg, int B, int Hi, int Wi, dpmn_stream_t stream) {
  DPMN_REQUIRE(din && dimg && cimg >= 3, "patch_scatter: bad arguments");
  const long total = (long)B * 3 * Hi * Wi;
  hipLaunchKernelGGL(k_patch_scatter, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), din, dimg, cimg, B, Hi, Wi);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

int dpmn_prior_fusion_wgrad_f32(const float* din, const float* prior, float* dpf_w, float* dpf_b, int B, int Hi, int Wi,
                                dpmn_stream_t stream) {
  DPMN_REQUIRE(din && prior && dpf_w && dpf_b, "prior_fusion_wgrad: bad arguments");
  const long total = (long)B * Hi * Wi;
  hipLaunchKernelGGL(k_prior_fusion_wgrad, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), din, prior, dpf_w, dpf_b, B, Hi, Wi,
                     static_cast<float*>(nullptr));
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

// the same without atomics: part is (ceil(B Hi Wi / 256), 57) rows of [dw (54) | db (3)] partial sums for dpmn_rows_reduce_f32
int dpmn_prior_fusion_wgrad_det_f32(const float* din, const float* prior, float* part, int B, int Hi, int Wi, dpmn_stream_t stream) {
  DPMN_REQUIRE(din && prior && part, "prior_fusion_wgrad_det: bad arguments");
  const long total = (long)B * Hi * Wi;
  hipLaunchKernelGGL(k_prior_fusion_wgrad, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), din, prior,
                     static_cast<float*>(nullptr), static_cast<float*>(nullptr), B, Hi, Wi, part);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

}  // extern "C"
