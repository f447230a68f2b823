// Fused LayerNorm + q/kv projection + multi-size window attention at embed_dim 192 = 3 groups x 2 heads x 32 (BASELINE.json
// configs[4]: windows 4 / 8 / 16 on a 32 x 128 token grid; pgrm.py:322-323, 188-194, 197-266) -- the head-dim-32 sibling of
// attn_fused.hip.  The stress stack ran 2 x ln_linear + 3 x k_window_attn_mfma per Swin block with q (B L 192) and kv (B L 384)
// round-tripping through HBM (226 MB written + read per block at B = 96); here they never leave the CU.
//
// Work unit = (image, group g, 256 consecutive window-major tokens of that group's partition): one 16 x 16 window, four 8 x 8
// windows or sixteen 4 x 4 windows.  One persistent 512-thread block per CU walks the units of "its" XCD (images b = xcd mod 8:
// the three groups' gathers of an image's token rows meet in one L2), expensive windows first.  Wave w owns the token tiles 2w, 2w + 1.
//   1. the group's folded weights W' = [q 64 | k 64 | v 64 rows] x 192 (k_attn192_fold: LayerNorm gamma folded in, beta into the
//      bias, q rows times head_dim^-0.5 log2 e) go to LDS -- 147 KB, the same bytes K / V use afterwards: a CU cannot hold both,
//      so they are restaged per unit from L2 (150 KB against ~2300 MFMAs per wave);
//   2. per token tile: the raw rows go straight from global memory into the MFMA B-operand registers (the roll + window gather is
//      an address computation), LayerNorm statistics in registers (two-pass), 576 x v_mfma_f32_16x16x4_f32 against the LDS
//      weights, normalisation behind the MFMAs: y = rstd (W' x - mean rowsum(W')) + b';
//   3. barrier; k, v -> LDS over the dead weights; q stays in registers (accumulator layout = B-operand layout of S^T = K Q^T);
//   4. per (query tile, head): S^T over the key tiles of the query's window (16 / 4 / 1), + bias table + shift mask, softmax in
//      exp2, O^T = V^T P with P fed from the accumulator registers, window-major write without un-roll (quirk Q1).
#include <cstdlib>
#include "common.h"

namespace {

constexpr float LOG2E = 1.44269504088896340736f;
constexpr float QS32 = 0.17677669529663687f * LOG2E;      // head_dim ** -0.5 * log2(e)
constexpr int FC = 192, FCG = 64, NF = 192, LDW = FC + 4, LDR = FCG + 4, SET = 256;
constexpr int TBLPAD = 1924;                               // (2 * 16 - 1)^2 * 2 = 1922, padded to a multiple of 4
constexpr int WREG = NF * LDW;                             // floats of the weight region (>= 2 * SET * LDR = 34816: K / V alias it)
constexpr int FOLD = WREG + 2 * NF + TBLPAD;               // floats per group in the folded-weight workspace
static_assert(WREG >= 2 * SET * LDR, "K / V must fit over the weights");

struct Args192 {
  const float *tq, *tkv, *lnq_w, *lnq_b, *lnkv_w, *lnkv_b, *wq, *bq, *wkv, *bkv;
  const float* table[3];      // by slot
  float* folded;              // [3 groups][FOLD], indexed by GROUP
  float* out;
  int ws[3], shift[3], gid[3];      // processing slot (largest windows first) -> window, shift, group
  int B, H, W, lgW;
  float eps;
};

// Folded weights of one call: grid (NF + 1, 3) x 64 threads; block (r, g) = row r of group g, block NF = the bias table.
__global__ __launch_bounds__(64) void k_attn192_fold(Args192 a) {
  const int g = blockIdx.y, r = blockIdx.x, tid = threadIdx.x;
  float* dst = a.folded + (size_t)g * FOLD;
  if (r == NF) {
    int slot = 0;
    for (int s_ = 1; s_ < 3; ++s_) if (a.gid[s_] == g) slot = s_;
    const int n = (2 * a.ws[slot] - 1) * (2 * a.ws[slot] - 1) * 2;
    for (int i = tid; i < TBLPAD; i += 64) dst[WREG + 2 * NF + i] = i < n ? a.table[slot][i] * LOG2E : 0.f;
    return;
  }
  __shared__ float part[48][2];
  const bool isq = r < 64;
  const float* srcw = isq ? a.wq + (size_t)(FCG * g + r) * FC
                          : (r < 128 ? a.wkv + (size_t)(FCG * g + r - 64) * FC : a.wkv + (size_t)(FC + FCG * g + r - 128) * FC);
  const float* gam = isq ? a.lnq_w : a.lnkv_w;
  const float* bet = isq ? a.lnq_b : a.lnkv_b;
  if (tid < 48) {
    const int c4 = 4 * tid;
    const f32x4 wv = *reinterpret_cast<const f32x4*>(srcw + c4);
    const f32x4 wg = wv * *reinterpret_cast<const f32x4*>(gam + c4), wb = wv * *reinterpret_cast<const f32x4*>(bet + c4);
    *reinterpret_cast<f32x4*>(dst + r * LDW + c4) = wg;
    part[tid][0] = (wg[0] + wg[1]) + (wg[2] + wg[3]);
    part[tid][1] = (wb[0] + wb[1]) + (wb[2] + wb[3]);
  } else if (tid == 48) {
    *reinterpret_cast<f32x4*>(dst + r * LDW + FC) = (f32x4){0.f, 0.f, 0.f, 0.f};      // row padding
  }
  __syncthreads();
  if (tid == 0) {
    float cw = 0.f, bb = isq ? a.bq[FCG * g + r] : (r < 128 ? a.bkv[FCG * g + r - 64] : a.bkv[FC + FCG * g + r - 128]);
    for (int k = 0; k < 48; ++k) { cw += part[k][0]; bb += part[k][1]; }      // fixed order
    dst[WREG + r] = isq ? bb * QS32 : bb;             // b' = b + W beta
    dst[WREG + NF + r] = isq ? cw * QS32 : cw;        // rowsum(W diag gamma)
  }
}

// mean and 1 / sqrt(var + eps) of the 192-value row a lane shares with its 3 kq partners (two-pass, like nn.LayerNorm)
__device__ __forceinline__ void row_stats192(const f32x4 (&x)[12], float eps, float& mean, float& rstd) {
  float s0 = 0.f, s1 = 0.f;
#pragma unroll
  for (int c = 0; c < 12; ++c) { s0 += x[c][0] + x[c][1]; s1 += x[c][2] + x[c][3]; }
  float s = s0 + s1;
  s += xshfl<16>(s); s += xshfl<32>(s);
  mean = s * (1.0f / FC);
  float q0 = 0.f, q1 = 0.f, q2 = 0.f, q3 = 0.f;
#pragma unroll
  for (int c = 0; c < 12; ++c) {
    const float d0 = x[c][0] - mean, d1 = x[c][1] - mean, d2 = x[c][2] - mean, d3 = x[c][3] - mean;
    q0 = fmaf(d0, d0, q0); q1 = fmaf(d1, d1, q1); q2 = fmaf(d2, d2, q2); q3 = fmaf(d3, d3, q3);
  }
  float q = (q0 + q1) + (q2 + q3);
  q += xshfl<16>(q); q += xshfl<32>(q);
  rstd = 1.0f / sqrtf(q * (1.0f / FC) + eps);
}

// source token (inside the image) of window-major token t through the roll + window partition (pgrm.py:209-213)
template <int WS>
__device__ __forceinline__ int src_token(int t, int H, int W, int lgW, int shift, int& hr, int& wc) {
  constexpr int N = WS * WS, LG = WS == 16 ? 4 : (WS == 8 ? 3 : 2);
  const int lgn = lgW - LG;                    // log2(windows per row)
  const int win = t / N, n = t % N;
  hr = ((win >> lgn) << LG) + n / WS;
  wc = ((win & ((1 << lgn) - 1)) << LG) + n % WS;
  return (((hr + shift) & (H - 1)) << lgW) + ((wc + shift) & (W - 1));
}

template <int WS>
__device__ __forceinline__ void load_rows192(const Args192& a, int b, int t, int shift, int kq, f32x4 (&xq)[12], f32x4 (&xkv)[12]) {
  int hr, wc;
  const size_t src = (size_t)b * a.H * a.W + src_token<WS>(t, a.H, a.W, a.lgW, shift, hr, wc);
  const float* pq = a.tq + src * FC + 4 * kq;
  const float* pk = a.tkv + src * FC + 4 * kq;
#pragma unroll
  for (int c = 0; c < 12; ++c) xq[c] = *reinterpret_cast<const f32x4*>(pq + 16 * c);
#pragma unroll
  for (int c = 0; c < 12; ++c) xkv[c] = *reinterpret_cast<const f32x4*>(pk + 16 * c);
}

// One unit.  smem: [ W' [192][LDW] (aliased by K [256][LDR] | V [256][LDR]) | b' [192] | rowsum [192] | table [TBLPAD] | region [256] ]
template <int WS>
__device__ __forceinline__ void run_unit(const Args192& a, int slot, int b, int set, bool restage_tables, float* smem) {
  constexpr int N = WS * WS, KT = WS == 16 ? 16 : (WS == 8 ? 4 : 1);
  float* Wsm = smem;
  float* Ks = smem;
  float* Vs = smem + SET * LDR;
  float* pb = smem + WREG;                  // b' | rowsum
  float* tbl = pb + 2 * NF;
  int* reg_s = reinterpret_cast<int*>(tbl + TBLPAD);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, kq = lane >> 4;
  const int H = a.H, W = a.W, L = H * W;
  const int g = a.gid[slot], shift = a.shift[slot];
  const int t0 = set * SET;
  // ---- row pointers of the wave's two token tiles (the roll + window gather is an address computation)
  const float *pq[2], *pk[2];
#pragma unroll
  for (int ti = 0; ti < 2; ++ti) {
    int hr, wc;
    const size_t src = (size_t)b * L + src_token<WS>(t0 + 32 * wave + 16 * ti + lr, H, W, a.lgW, shift, hr, wc);
    pq[ti] = a.tq + src * FC + 4 * kq;
    pk[ti] = a.tkv + src * FC + 4 * kq;
  }
  // The rows stream through a 3-deep ring of 16-column chunks (step k = tile k / 12, chunk k % 12; loads run 3 steps = ~150 MFMAs
  // ahead): holding whole rows (96 registers) next to the accumulators and the first tile's q / k / v spilled.  LayerNorm statistics
  // are therefore one-pass, on values shifted by the row's first element (no cancellation: |mean - x0| is of the order of the
  // row's spread): mean = x0 + S1 / n, var = S2 / n - (S1 / n)^2.
  f32x4 rq_[3], rk_[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    rq_[k] = *reinterpret_cast<const f32x4*>(pq[0] + 16 * k);
    rk_[k] = *reinterpret_cast<const f32x4*>(pk[0] + 16 * k);
  }
  __syncthreads();                           // the previous unit's attention is done with K / V
  {
    const f32x4* src = reinterpret_cast<const f32x4*>(a.folded + (size_t)g * FOLD);
    f32x4* dst = reinterpret_cast<f32x4*>(smem);
    const int nv = (restage_tables ? FOLD : WREG) / 4;
    for (int i = tid; i < nv; i += 512) dst[i] = src[i];
    if (shift > 0 && tid < SET) {
      int hr, wc;
      (void)src_token<WS>(t0 + tid, H, W, a.lgW, shift, hr, wc);
      const int rh = hr < H - WS ? 0 : (hr < H - shift ? 1 : 2), rw = wc < W - WS ? 0 : (wc < W - shift ? 1 : 2);
      reg_s[tid] = 3 * rh + rw;
    }
  }
  __syncthreads();
  // ---- projections of the wave's two token tiles; q / k / v of both stay in registers until the weights are dead
  f32x4 qa[2][4], ka[2][4], va[2][4];       // [tile][feature tile: head * 2 + dc]
  f32x4 acc[12];
  float shq = 0.f, shk = 0.f, s1q = 0.f, s2q = 0.f, s1k = 0.f, s2k = 0.f;
#pragma unroll
  for (int k = 0; k < 24; ++k) {
    const int ti = k / 12, c = k % 12;
    const f32x4 xq = rq_[k % 3], xkv = rk_[k % 3];
    if (c == 0) {
#pragma unroll
      for (int j = 0; j < 12; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      shq = __shfl(xq[0], lr, 64);            // element 0 of the row (lane kq = 0 holds it)
      shk = __shfl(xkv[0], lr, 64);
      s1q = s2q = s1k = s2k = 0.f;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float dq_ = xq[e] - shq, dk_ = xkv[e] - shk;
      s1q += dq_; s2q = fmaf(dq_, dq_, s2q);
      s1k += dk_; s2k = fmaf(dk_, dk_, s2k);
    }
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {           // two halves of six feature tiles: 24 weight registers live at a time
      f32x4 wf[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) wf[j] = *reinterpret_cast<const f32x4*>(Wsm + (16 * (6 * hf + j) + lr) * LDW + 16 * c + 4 * kq);
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
        for (int j = 0; j < 6; ++j) acc[6 * hf + j] = mfma16(wf[j][s_], (6 * hf + j) < 4 ? xq[s_] : xkv[s_], acc[6 * hf + j]);
    }
    if (k + 3 < 24) {                          // refill the ring slot just consumed
      const int t2 = (k + 3) / 12, c2 = (k + 3) % 12;
      rq_[k % 3] = *reinterpret_cast<const f32x4*>(pq[t2] + 16 * c2);
      rk_[k % 3] = *reinterpret_cast<const f32x4*>(pk[t2] + 16 * c2);
    }
    __builtin_amdgcn_sched_barrier(0);         // one step at a time: hipcc otherwise hoists later steps' operand reads (spills)
    if (c == 11) {
      // row statistics (the 4 kq lanes of a row hold disjoint quarters), then y = rstd * acc - (rstd * mean) * rowsum(W') + b'
      s1q += xshfl<16>(s1q); s1q += xshfl<32>(s1q); s2q += xshfl<16>(s2q); s2q += xshfl<32>(s2q);
      s1k += xshfl<16>(s1k); s1k += xshfl<32>(s1k); s2k += xshfl<16>(s2k); s2k += xshfl<32>(s2k);
      const float m1q = s1q * (1.0f / FC), m1k = s1k * (1.0f / FC);
      const float mq = shq + m1q, mk = shk + m1k;
      const float rq = 1.0f / sqrtf(fmaxf(s2q * (1.0f / FC) - m1q * m1q, 0.f) + a.eps);
      const float rk = 1.0f / sqrtf(fmaxf(s2k * (1.0f / FC) - m1k * m1k, 0.f) + a.eps);
      const float rqs = rq * QS32, nmq = -mq * rq, nmk = -mk * rk;      // (b' / rowsum of the q rows carry head_dim ** -0.5 * log2 e)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 bq4 = *reinterpret_cast<const f32x4*>(pb + 16 * j + 4 * kq), cq = *reinterpret_cast<const f32x4*>(pb + NF + 16 * j + 4 * kq);
        const f32x4 bk4 = *reinterpret_cast<const f32x4*>(pb + 64 + 16 * j + 4 * kq), ck = *reinterpret_cast<const f32x4*>(pb + NF + 64 + 16 * j + 4 * kq);
        const f32x4 bv4 = *reinterpret_cast<const f32x4*>(pb + 128 + 16 * j + 4 * kq), cv = *reinterpret_cast<const f32x4*>(pb + NF + 128 + 16 * j + 4 * kq);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          qa[ti][j][e] = fmaf(acc[j][e], rqs, fmaf(nmq, cq[e], bq4[e]));
          ka[ti][j][e] = fmaf(acc[4 + j][e], rk, fmaf(nmk, ck[e], bk4[e]));
          va[ti][j][e] = fmaf(acc[8 + j][e], rk, fmaf(nmk, cv[e], bv4[e]));
        }
      }
    }
  }
  __syncthreads();                           // every wave is done with the weights
#pragma unroll
  for (int ti = 0; ti < 2; ++ti)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = (32 * wave + 16 * ti + lr) * LDR + 16 * j + 4 * kq;
      *reinterpret_cast<f32x4*>(Ks + row) = ka[ti][j];
      *reinterpret_cast<f32x4*>(Vs + row) = va[ti][j];
    }
  __syncthreads();
  // ---- attention of the wave's two query tiles
#pragma unroll
  for (int ti = 0; ti < 2; ++ti) {
    const int rq_ = 32 * wave + 16 * ti + lr;                 // my query's row inside the set
    const int kbase = WS == 16 ? 0 : (WS == 8 ? (rq_ >> 6) << 6 : (rq_ >> 4) << 4);      // first key row of the query tile's window(s)
    const int nq = rq_ % N, iq = nq / WS, jq = nq % WS;
    const int my_reg = shift > 0 ? reg_s[rq_] : 0;
#pragma unroll
    for (int head = 0; head < 2; ++head) {
      __builtin_amdgcn_sched_barrier(0);      // one (tile, head) at a time: interleaving them multiplies the live score registers
      const f32x4 qf[2] = {qa[ti][2 * head], qa[ti][2 * head + 1]};
      f32x4 sacc[KT];
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dc = 0; dc < 2; ++dc) {
          const f32x4 kf = *reinterpret_cast<const f32x4*>(Ks + (kbase + 16 * kt + lr) * LDR + head * 32 + 16 * dc + 4 * kq);
#pragma unroll
          for (int s = 0; s < 4; ++s) acc = mfma16(kf[s], qf[dc][s], acc);
        }
        sacc[kt] = acc;
        if (KT > 4) __builtin_amdgcn_sched_barrier(0);
      }
      // relative position bias (pgrm.py:234-238): 16-token tiles are whole window rows (or pairs / quarters of them), so the table
      // index of (my query, key 16 kt + 4 kq + r) is a per-lane base minus compile-time steps in kt and r
      constexpr int T1 = 2 * WS - 1, KSTEP = (16 / WS) * T1;
      const float* tb = tbl + 2 * ((iq + WS - 1) * T1 + (jq + WS - 1) - ((4 * kq) / WS) * T1 - (4 * kq) % WS) + head;
      float mx = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        int4 kr = make_int4(my_reg, my_reg, my_reg, my_reg);
        if (shift > 0) kr = *reinterpret_cast<const int4*>(reg_s + kbase + 16 * kt + 4 * kq);
        const int krr[4] = {kr.x, kr.y, kr.z, kr.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = sacc[kt][r] + tb[-2 * (KSTEP * kt + r)];
          if (krr[r] != my_reg) v += -100.0f * LOG2E;
          sacc[kt][r] = v;
          mx = fmaxf(mx, v);
        }
        if (KT > 4) __builtin_amdgcn_sched_barrier(0);
      }
      mx = fmaxf(mx, xshfl<16>(mx));
      mx = fmaxf(mx, xshfl<32>(mx));
      float den = 0.f;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = __builtin_amdgcn_exp2f(sacc[kt][r] - mx);
          sacc[kt][r] = p;
          den += p;
        }
      den += xshfl<16>(den);
      den += xshfl<32>(den);
      const float inv = 1.0f / den;
      f32x4 oacc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float* vrow = Vs + (kbase + 16 * kt + 4 * kq + r) * LDR + head * 32 + lr;
          oacc[0] = mfma16(vrow[0], sacc[kt][r], oacc[0]);
          oacc[1] = mfma16(vrow[16], sacc[kt][r], oacc[1]);
        }
        if (KT > 4) __builtin_amdgcn_sched_barrier(0);
      }
      float* dst = a.out + ((size_t)b * L + t0 + rq_) * FC + FCG * g + head * 32 + 4 * kq;
      *reinterpret_cast<float4*>(dst) = make_float4(oacc[0][0] * inv, oacc[0][1] * inv, oacc[0][2] * inv, oacc[0][3] * inv);
      *reinterpret_cast<float4*>(dst + 16) = make_float4(oacc[1][0] * inv, oacc[1][1] * inv, oacc[1][2] * inv, oacc[1][3] * inv);
    }
  }
}

__global__ __launch_bounds__(512, 1) void k_ln_qkv_window_attn192(Args192 a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3, nbx = gridDim.x >> 3;
  const int S = a.H * a.W / SET;
  const int ni = xcd < a.B ? (a.B - xcd + 7) / 8 : 0;      // images b = xcd, xcd + 8, ...
  const int per = ni * S;                                  // units per slot on this XCD
  int last_slot = -1;
  for (int u = jj; u < 3 * per; u += nbx) {                // slot-major list: the 16 x 16 units first
    const int slot = u / per, v = u - slot * per;
    const int b = xcd + 8 * (v / S), set = v % S;
    const int ws = a.ws[slot];
    const bool tables = slot != last_slot;
    last_slot = slot;
    if (ws == 16) run_unit<16>(a, slot, b, set, tables, smem);
    else if (ws == 8) run_unit<8>(a, slot, b, set, tables, smem);
    else run_unit<4>(a, slot, b, set, tables, smem);
  }
}

}  // namespace

extern "C" {

int dpmn_ln_qkv_window_attn_d32_supported(int C, int n_groups, int heads_per_group, const int* windows, int H, int W) {
  if (C != FC || n_groups != 3 || heads_per_group != 2 || !windows || (H * W) % SET != 0) return 0;
  if ((H & (H - 1)) || (W & (W - 1)) || H < 16 || W < 16) return 0;
  static const int off = getenv("DPMN_ATTN_FUSED") && atoi(getenv("DPMN_ATTN_FUSED")) == 0;
  if (off) return 0;
  for (int g = 0; g < 3; ++g) {
    const int ws = windows[g];
    if (!(ws == 4 || ws == 8 || ws == 16) || H % ws || W % ws) return 0;
  }
  return 1;
}

size_t dpmn_ln_qkv_window_attn_d32_workspace_bytes(void) { return sizeof(float) * 3 * FOLD; }

int dpmn_ln_qkv_window_attn_d32_f32(const float* tq, const float* tkv, const float* lnq_w, const float* lnq_b, const float* lnkv_w,
                                    const float* lnkv_b, float eps, const float* wq, const float* bq, const float* wkv,
                                    const float* bkv, const float* const* bias_tables, const int* windows, const int* shifts,
                                    int n_groups, int heads_per_group, float* out, void* workspace, int refold, int B, int H, int W,
                                    int C, dpmn_stream_t stream) {
  DPMN_REQUIRE(tq && tkv && lnq_w && lnq_b && lnkv_w && lnkv_b && wq && bq && wkv && bkv && bias_tables && windows && shifts && out,
               "ln_qkv_window_attn_d32: null pointer");
  DPMN_REQUIRE(B > 0 && dpmn_ln_qkv_window_attn_d32_supported(C, n_groups, heads_per_group, windows, H, W),
               "ln_qkv_window_attn_d32: built for dim 192 = 3 groups x 2 heads x 32, windows in {4, 8, 16}, power-of-two token grid with H W a multiple of 256");
  DPMN_REQUIRE(workspace && ((uintptr_t)workspace & 15) == 0, "ln_qkv_window_attn_d32: workspace (dpmn_ln_qkv_window_attn_d32_workspace_bytes, 16-byte aligned) missing");
  Args192 a{};
  a.tq = tq; a.tkv = tkv; a.lnq_w = lnq_w; a.lnq_b = lnq_b; a.lnkv_w = lnkv_w; a.lnkv_b = lnkv_b; a.wq = wq; a.bq = bq; a.wkv = wkv; a.bkv = bkv;
  a.B = B; a.H = H; a.W = W; a.eps = eps; a.out = out; a.folded = static_cast<float*>(workspace);
  for (a.lgW = 0; (1 << a.lgW) < W; ++a.lgW) {}
  int order[3] = {0, 1, 2};
  for (int i = 0; i < 3; ++i)
    for (int j = i + 1; j < 3; ++j)
      if (windows[order[j]] > windows[order[i]]) { const int t_ = order[i]; order[i] = order[j]; order[j] = t_; }
  for (int s = 0; s < 3; ++s) {
    const int g = order[s];
    DPMN_REQUIRE(shifts[g] >= 0 && shifts[g] < windows[g] && bias_tables[g], "ln_qkv_window_attn_d32: bad shift / null bias table");
    a.gid[s] = g; a.ws[s] = windows[g]; a.shift[s] = shifts[g]; a.table[s] = bias_tables[g];
  }
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return dpmn_set_error(DPMN_ERR_LAUNCH, "ln_qkv_window_attn_d32: device query failed");
    n_cu = prop.multiProcessorCount > 8 ? prop.multiProcessorCount / 8 * 8 : 8;
  }
  const size_t smem = (size_t)(FOLD + SET) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ln_qkv_window_attn192), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_set = true;
  }
  hipStream_t st = as_stream(stream);
  if (refold) hipLaunchKernelGGL(k_attn192_fold, dim3(NF + 1, 3), dim3(64), 0, st, a);
  const double tokens = (double)B * H * W;
  double attn = 0.0;
  for (int g = 0; g < 3; ++g) attn += 4.0 * windows[g] * windows[g] * 32 * 2 * tokens;
  ProfScope prof(PT_ATTN_FUSED, st, 2.0 * tokens * FC * (3 * FC) + attn, 4.0 * (3.0 * tokens * FC + 3.0 * FC * FC));
  hipLaunchKernelGGL(k_ln_qkv_window_attn192, dim3((unsigned)n_cu), dim3(512), smem, st, a);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

}  // extern "C"
