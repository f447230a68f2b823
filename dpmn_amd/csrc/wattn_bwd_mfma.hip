// Window-attention backward on the matrix cores for windows of 64 and 256 tokens (8x8, 16x16) at any head dim that is a multiple
// of 16 -- the stress configuration's (window 16, head dim 32) group (BASELINE.json configs[4]; autograd of pgrm.py:184-271) had no
// backward at all, and its (8, 32) group ran lane-per-row on the vector ALU.
//
// One block = one (window, head); a wave owns two 16-token tiles of the window.  Every product is v_mfma_f32_16x16x4_f32 and, as in
// k_window_attn8_bwd_mfma (backward_pgrm.hip), the logits are built twice so that P / dS always sit in the accumulator layout the
// next product wants as its B operand (lane (lr, kq) of a 16x16 tile holds [row 4kq + r][column lr] = the B slot of k-step r):
//   pass A, keys x queries (wave = two QUERY tiles, all key tiles):  S^T = K Q^T, dP^T = V dO^T  ->  max / 1/sum / delta = sum_k P dP
//           per query column (16 accumulator tiles + two xor exchanges), dS^T = P^T o (dP^T - delta), dQ^T = K^T dS^T.
//           LDS holds K and V of the window (row reads for the A operands, column reads for K^T); Q / dO rows come straight from
//           global memory into the B-operand registers.
//   pass B, queries x keys (wave = two KEY tiles, all query tiles):  S = Q K^T, dP = dO V^T again, P and dS from the statistics
//           pass A left in LDS, dV^T = dO^T P, dK^T = Q^T dS, and the relative-position-bias table gradient.
//           LDS now holds Q (pre-scaled) and dO in the SAME region (row reads + the column reads of Q^T / dO^T); K / V rows of the
//           wave's two key tiles sit in registers.
// So a 256-token window needs 2 x 256 x 36 floats = 72 KB of LDS instead of the 144 KB all four operands would take.
// Table gradient: every wave adds its dS entries into its OWN LDS copy of the head's table (LDS atomics inside one wave only), the
// copies are summed in wave order at the end -- bitwise reproducible; part_mode stores the block's row (the two heads of a window
// share row blockIdx.x / 2, disjoint entries), else the sum is added to dtable with global atomics.
// attn_drop (DROP): P is multiplied by the regenerated mask M before P.V (pgrm.py:248), so dV = (P o M)^T dO, dP = (dO V^T) o M.
#include "common.h"

namespace {

template <int WS, int D, bool DROP>
__global__ __launch_bounds__(WS == 16 ? 512 : 128) void k_window_attn_bwd_mfma(
    const float* __restrict__ q, const float* __restrict__ kv, const float* __restrict__ bias_table, const float* __restrict__ dout,
    float* __restrict__ dq, float* __restrict__ dkv, float* __restrict__ dtable, int H, int W, int C, int g, int shift, float p_drop,
    unsigned long long seed, int part_mode) {
  static_assert(WS == 8 || WS == 16, "64- and 256-token windows");
  static_assert(D % 16 == 0, "head dim: a multiple of 16");
  constexpr int N = WS * WS, NT = N / 16, NW = NT / 2, TH = 64 * NW, DC = D / 16, CG = 2 * D, LDR = D + 4;
  constexpr int T1 = 2 * WS - 1, TBL = T1 * T1, TB4 = (TBL + 3) & ~3;
  constexpr int KSTEP = (16 / WS) * T1;       // table-index step of one 16-token tile along the key / query axis
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* tbl = smem;                          // [TB4] this head's column of the bias table
  float* R1 = tbl + TB4;                      // pass A: K [N][LDR] | V [N][LDR];  pass B: Q * scale | dO
  float* R2 = R1 + 2 * N * LDR;               // [NW][TB4] per-wave table gradient copies
  float* stat = R2 + NW * TB4;                // [3][N] max, 1/sum, delta per query
  int* reg_s = reinterpret_cast<int*>(stat + 3 * N);   // [N] shift-mask region (pgrm.py:153-176)
  int* src_s = reg_s + N;                     // [N] source token through the roll + window partition (quirk Q1: never undone)
  float* As = R1;
  float* Bs = R1 + N * LDR;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, kq = lane >> 4;
  const int head = blockIdx.x & 1;
  const int L = H * W, wins = L / N;
  const int b = (int)((blockIdx.x >> 1) / wins), win = (int)((blockIdx.x >> 1) % wins);
  const int t0 = win * N;                     // first window-major token of the window
  const int nWc = W / WS;
  for (int i = tid; i < TBL; i += TH) tbl[i] = bias_table[2 * i + head];
  for (int r = tid; r < N; r += TH) {
    const int hr = (win / nWc) * WS + r / WS, wcol = (win % nWc) * WS + r % WS;
    src_s[r] = b * L + ((hr + shift) % H) * W + (wcol + shift) % W;
    const int rh = hr < H - WS ? 0 : (hr < H - shift ? 1 : 2), rw = wcol < W - WS ? 0 : (wcol < W - shift ? 1 : 2);
    reg_s[r] = 3 * rh + rw;
  }
  __syncthreads();
  constexpr int V4 = D / 4;
  for (int i = tid; i < N * V4; i += TH) {
    const int r = i / V4, c4 = (i % V4) * 4;
    const float* kp = kv + (size_t)src_s[r] * 2 * C + g * CG + head * D + c4;
    *reinterpret_cast<float4*>(As + r * LDR + c4) = *reinterpret_cast<const float4*>(kp);
    *reinterpret_cast<float4*>(Bs + r * LDR + c4) = *reinterpret_cast<const float4*>(kp + C);
  }
  __syncthreads();
  const float scale = D == 32 ? 0.17677669529663687f : (D == 16 ? 0.25f : 1.0f / sqrtf((float)D));      // head_dim ** -0.5
  const unsigned long long mrow0 = ((unsigned long long)((size_t)b * (C / CG) + g) * 2 + head) * L + t0;   // mask index base (include/dpmn_hip.h)
  const float inv_keep = DROP ? 1.0f / (1.0f - p_drop) : 1.0f;
  const size_t cofs = (size_t)g * CG + head * D;
  float* smax = stat;
  float* sinv = stat + N;
  float* sdel = stat + 2 * N;
  // table index of (query, key): (iq - im + WS - 1) T1 + (jq - jm + WS - 1); 16-token tiles are whole window rows (or pairs of
  // them), so the index is a per-lane base plus compile-time steps in the tile indices and r
  const int lane_k = ((4 * kq) / WS) * T1 + (4 * kq) % WS;      // the 4kq part of a row index 16 t + 4kq + r
  const int lane_c = (lr / WS) * T1 + lr % WS;                  // a column index 16 t + lr
  const int tcen = (WS - 1) * T1 + (WS - 1);

  // ================= pass A: rows = keys (16 kt + 4kq + r), columns = queries (16 qt + lr)
#pragma unroll 1
  for (int qi = 0; qi < 2; ++qi) {
    const int qt = wave + NW * qi, rq = 16 * qt + lr;
    const int qsrc = src_s[rq];
    f32x4 qf[DC], gf[DC];
#pragma unroll
    for (int dc = 0; dc < DC; ++dc) {
      qf[dc] = *reinterpret_cast<const f32x4*>(q + (size_t)qsrc * C + cofs + 16 * dc + 4 * kq);
      qf[dc] *= scale;
      gf[dc] = *reinterpret_cast<const f32x4*>(dout + ((size_t)b * L + t0 + rq) * C + cofs + 16 * dc + 4 * kq);
    }
    f32x4 ps[NT], dp[NT];
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
      f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f}, c = a;
#pragma unroll
      for (int dc = 0; dc < DC; ++dc) {
        const f32x4 kf = *reinterpret_cast<const f32x4*>(As + (16 * kt + lr) * LDR + 16 * dc + 4 * kq);
        const f32x4 vf = *reinterpret_cast<const f32x4*>(Bs + (16 * kt + lr) * LDR + 16 * dc + 4 * kq);
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) { a = mfma16(kf[s4], qf[dc][s4], a); c = mfma16(vf[s4], gf[dc][s4], c); }
      }
      ps[kt] = a; dp[kt] = c;
      if (NT > 4) __builtin_amdgcn_sched_barrier(0);      // keep hipcc from hoisting all 16 tiles' operand reads to the top
    }
    const float* tb = tbl + tcen + lane_c + KSTEP * qt - lane_k;     // - KSTEP kt - r
    const int my_reg = reg_s[rq];
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
      int4 kr = make_int4(my_reg, my_reg, my_reg, my_reg);
      if (shift > 0) kr = *reinterpret_cast<const int4*>(reg_s + 16 * kt + 4 * kq);
      const int krr[4] = {kr.x, kr.y, kr.z, kr.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float a = ps[kt][r] + tb[-KSTEP * kt - r];
        if (krr[r] != my_reg) a += -100.0f;
        ps[kt][r] = a;
        mx = fmaxf(mx, a);
      }
      if (NT > 4) __builtin_amdgcn_sched_barrier(0);
    }
    mx = fmaxf(mx, xshfl<16>(mx));
    mx = fmaxf(mx, xshfl<32>(mx));
    float den = 0.f;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float p = __expf(ps[kt][r] - mx); ps[kt][r] = p; den += p; }
    den += xshfl<16>(den);
    den += xshfl<32>(den);
    const float inv = 1.0f / den;
    if (DROP) {      // dP = (dO V^T) o M with the forward's mask (0 or 1/(1-p)); delta = sum_k P dP keeps its form
      const unsigned long long mrow = (mrow0 + rq) * N;
#pragma unroll
      for (int kt = 0; kt < NT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) dp[kt][r] *= drop_scale(seed, mrow + 16 * kt + 4 * kq + r, p_drop, inv_keep);
    }
    float dlt = 0.f;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) { ps[kt][r] *= inv; dlt = fmaf(ps[kt][r], dp[kt][r], dlt); }
    dlt += xshfl<16>(dlt);
    dlt += xshfl<32>(dlt);
    if (kq == 0) { smax[rq] = mx; sinv[rq] = inv; sdel[rq] = dlt; }
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) dp[kt][r] = ps[kt][r] * (dp[kt][r] - dlt);      // dS^T
    // dQ^T (16 d x 16 queries per tile) = K^T . dS^T
    f32x4 dqa[DC];
#pragma unroll
    for (int dt = 0; dt < DC; ++dt) dqa[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float* kcol = As + (16 * kt + 4 * kq + r) * LDR + lr;          // K^T[d = 16 dt + lr][key]
#pragma unroll
        for (int dt = 0; dt < DC; ++dt) dqa[dt] = mfma16(kcol[16 * dt], dp[kt][r], dqa[dt]);
      }
      if (NT > 4) __builtin_amdgcn_sched_barrier(0);
    }
    float* dst = dq + (size_t)qsrc * C + cofs + 4 * kq;
#pragma unroll
    for (int dt = 0; dt < DC; ++dt)
      *reinterpret_cast<float4*>(dst + 16 * dt) = make_float4(dqa[dt][0] * scale, dqa[dt][1] * scale, dqa[dt][2] * scale, dqa[dt][3] * scale);
  }
  __syncthreads();          // K / V are dead, the statistics of every query are in LDS
  for (int i = tid; i < N * V4; i += TH) {
    const int r = i / V4, c4 = (i % V4) * 4;
    f32x4 qv = *reinterpret_cast<const f32x4*>(q + (size_t)src_s[r] * C + cofs + c4);
    qv *= scale;
    *reinterpret_cast<f32x4*>(As + r * LDR + c4) = qv;
    *reinterpret_cast<float4*>(Bs + r * LDR + c4) = *reinterpret_cast<const float4*>(dout + ((size_t)b * L + t0 + r) * C + cofs + c4);
  }
  for (int i = tid; i < NW * TB4; i += TH) R2[i] = 0.f;
  // the wave's two key tiles: K / V rows in operand order, straight from global memory
  f32x4 kf[2][DC], vf[2][DC];
  int ksrc[2], key_reg[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int rk = 16 * (wave + NW * j) + lr;
    ksrc[j] = src_s[rk];
    key_reg[j] = reg_s[rk];
    const float* kp = kv + (size_t)ksrc[j] * 2 * C + cofs + 4 * kq;
#pragma unroll
    for (int dc = 0; dc < DC; ++dc) {
      kf[j][dc] = *reinterpret_cast<const f32x4*>(kp + 16 * dc);
      vf[j][dc] = *reinterpret_cast<const f32x4*>(kp + C + 16 * dc);
    }
  }
  __syncthreads();
  // ================= pass B: rows = queries (16 qt + 4kq + r), columns = keys (16 kt + lr)
  {
    float* mytb = R2 + wave * TB4;
    f32x4 dva[2][DC], dka[2][DC];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int dt = 0; dt < DC; ++dt) { dva[j][dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; dka[j][dt] = dva[j][dt]; }
    int tbase[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) tbase[j] = tcen + lane_k - lane_c - KSTEP * (wave + NW * j);      // + KSTEP qt + r
#pragma unroll 1
    for (int qt = 0; qt < NT; ++qt) {
      f32x4 qrow[DC], grow[DC];
#pragma unroll
      for (int dc = 0; dc < DC; ++dc) {
        qrow[dc] = *reinterpret_cast<const f32x4*>(As + (16 * qt + lr) * LDR + 16 * dc + 4 * kq);
        grow[dc] = *reinterpret_cast<const f32x4*>(Bs + (16 * qt + lr) * LDR + 16 * dc + 4 * kq);
      }
      const f32x4 qmax = *reinterpret_cast<const f32x4*>(smax + 16 * qt + 4 * kq);
      const f32x4 qinv = *reinterpret_cast<const f32x4*>(sinv + 16 * qt + 4 * kq);
      const f32x4 qdel = *reinterpret_cast<const f32x4*>(sdel + 16 * qt + 4 * kq);
      int4 qr4 = make_int4(0, 0, 0, 0);
      if (shift > 0) qr4 = *reinterpret_cast<const int4*>(reg_s + 16 * qt + 4 * kq);
      const int qreg[4] = {qr4.x, qr4.y, qr4.z, qr4.w};
      f32x4 pm[2], ds[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f}, c = a;
#pragma unroll
        for (int dc = 0; dc < DC; ++dc)
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) { a = mfma16(qrow[dc][s4], kf[j][dc][s4], a); c = mfma16(grow[dc][s4], vf[j][dc][s4], c); }
        const int ti = tbase[j] + KSTEP * qt;
        const int mkey = 16 * (wave + NW * j) + lr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float lg = a[r] + tbl[ti + r];
          if (shift > 0 && qreg[r] != key_reg[j]) lg += -100.0f;
          const float p = __expf(lg - qmax[r]) * qinv[r];
          const float mk = DROP ? drop_scale(seed, (mrow0 + 16 * qt + 4 * kq + r) * N + mkey, p_drop, inv_keep) : 1.0f;
          const float dsv = p * (c[r] * mk - qdel[r]);
          pm[j][r] = p * mk;               // P o M: what multiplied V in the forward
          ds[j][r] = dsv;
          atomicAdd(mytb + ti + r, dsv);   // this wave's copy only
        }
      }
      // dV^T += dO^T . (P o M) ; dK^T += (scale Q)^T . dS   (16 d x 16 keys per tile)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = (16 * qt + 4 * kq + r) * LDR + lr;
#pragma unroll
        for (int dt = 0; dt < DC; ++dt) {
          const float gt = Bs[row + 16 * dt], qv = As[row + 16 * dt];
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            dva[j][dt] = mfma16(gt, pm[j][r], dva[j][dt]);
            dka[j][dt] = mfma16(qv, ds[j][r], dka[j][dt]);
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float* dkp = dkv + (size_t)ksrc[j] * 2 * C + cofs + 4 * kq;
#pragma unroll
      for (int dt = 0; dt < DC; ++dt) {
        *reinterpret_cast<float4*>(dkp + 16 * dt) = make_float4(dka[j][dt][0], dka[j][dt][1], dka[j][dt][2], dka[j][dt][3]);
        *reinterpret_cast<float4*>(dkp + C + 16 * dt) = make_float4(dva[j][dt][0], dva[j][dt][1], dva[j][dt][2], dva[j][dt][3]);
      }
    }
  }
  __syncthreads();
  for (int e = tid; e < TBL; e += TH) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) v += R2[w * TB4 + e];       // wave order: reproducible
    if (part_mode) dtable[(size_t)(blockIdx.x >> 1) * (TBL * 2) + 2 * e + head] = v;
    else atomicAdd(dtable + 2 * e + head, v);
  }
}

template <int WS, int D, bool DROP>
int launch(const float* q, const float* kv, const float* tbl, const float* dout, float* dq, float* dkv, float* dtable, int B, int H,
           int W, int C, int g, int shift, float p_drop, unsigned long long seed, hipStream_t st, int part_mode) {
  constexpr int N = WS * WS, NW = N / 32, LDR = D + 4, TBL = (2 * WS - 1) * (2 * WS - 1), TB4 = (TBL + 3) & ~3;
  const size_t smem = (size_t)(TB4 + 2 * N * LDR + NW * TB4 + 3 * N) * 4 + 2 * N * 4;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_window_attn_bwd_mfma<WS, D, DROP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_set = true;
  }
  const long blocks = (long)B * (H * W / N) * 2;
  // per (window, head): S and dP twice, dQ, dK, dV = 7 products of 2 N^2 D FLOPs; q, k, v, dO in, dq, dk, dv out
  ProfScope prof(PT_WATTN_BWD, st, 7.0 * 2.0 * N * D * 2 * (double)B * H * W, 4.0 * 7 * 2 * D * (double)B * H * W);
  hipLaunchKernelGGL((k_window_attn_bwd_mfma<WS, D, DROP>), dim3((unsigned)blocks), dim3(64 * NW), smem, st, q, kv, tbl, dout, dq, dkv,
                     dtable, H, W, C, g, shift, p_drop, seed, part_mode);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

}  // namespace

// rows of the bias-table gradient partial buffer (part_mode) = windows; returns DPMN_ERR_ARG for shapes this file does not build
int dpmn_wattn_bwd_mfma(int ws, int D, const float* q, const float* kv, const float* tbl, const float* dout, float* dq, float* dkv,
                        float* dtable, int B, int H, int W, int C, int g, int shift, float p_drop, unsigned long long seed,
                        hipStream_t st, int part_mode, int* rows) {
  if (rows) *rows = B * (H * W / (ws * ws));
#define WBM_CASE(WSV, DV) if (ws == WSV && D == DV) return p_drop > 0.f \
      ? launch<WSV, DV, true>(q, kv, tbl, dout, dq, dkv, dtable, B, H, W, C, g, shift, p_drop, seed, st, part_mode) \
      : launch<WSV, DV, false>(q, kv, tbl, dout, dq, dkv, dtable, B, H, W, C, g, shift, 0.f, 0ull, st, part_mode);
  WBM_CASE(16, 32) WBM_CASE(8, 32) WBM_CASE(16, 16)
#undef WBM_CASE
  return DPMN_ERR_ARG;
}
