// Shared pieces of the fused LayerNorm + q/kv projection + window attention kernels: the forward (attn_fused.hip) and the
// recomputing backward (attn_fused_bwd.hip) walk the same unit lists, read the same folded weights and use the same operand layouts.
#pragma once
#include "common.h"

namespace dpmn_fa {

constexpr float QSCALE = 0.25f * 1.44269504088896340736f;      // head_dim ** -0.5 * log2(e)
constexpr float LOG2E = 1.44269504088896340736f;
constexpr int FC = 96, FCG = 32, FD = 16, LDW = FC + 4, LDK = FCG + 4, TBLMAX = 15 * 15 * 2, TBLPAD = 452;
constexpr int FOLD_STRIDE = FC * LDW + 2 * FC + TBLPAD;     // floats per group in the folded-weight workspace (a multiple of 4)

struct FusedAttnArgs {
  const float *tq, *tkv, *lnq_w, *lnq_b, *lnkv_w, *lnkv_b, *wq, *bq, *wkv, *bkv;
  const float* table[3];
  float *q_out, *kv_out;            // TRAIN: the projections, (B L, 96) and (B L, 192) in raster token order, saved for the backward
  float p_drop, inv_keep;           // TRAIN: attn_drop (pgrm.py:248), counter-based masks (common.h drop_scale)
  unsigned long long seed;
  float* folded;                    // [3 groups][FOLD_STRIDE]: k_attn_fold's output, indexed by GROUP (not slot)
  int ws[3], shift[3], gid[3];      // processing slot s (expensive windows first) -> group gid[s]
  int cost[3];                      // measured cost of one unit of slot s (hundreds of cycles), for the static load balance
  int nblk[2][3];                   // blocks per slot on an XCD holding ceil(B/8) ([0]) / floor(B/8) ([1]) images; nblk[.][0] = 0: contiguous ranges
  float* out;
  // backward (attn_fused_bwd.hip): gradient of out (window-major, like out), gradients of the q / kv projections in raster token
  // order, and per slot a (gridDim.x, (2 ws - 1)^2 * 2) buffer of bias-table gradient partial rows (one row per block)
  const float* dout;
  float *dq, *dkv;
  float* tpart[3];
  int B, H, W;
  int lgW, lgS;                     // H, W (hence S = H*W/64 and every W / ws) are powers of two: index math is shifts and masks
  float eps;
};

// source row of window-major token t of image b through the roll (pgrm.py:209-213); also the token's rolled-frame coordinates
template <int WS>
__device__ __forceinline__ unsigned source_row(int t, int H, int W, int lgW, int shift, int& hr, int& wcol) {
  constexpr int N = WS * WS, LGWS = WS == 8 ? 3 : (WS == 4 ? 2 : 1);
  const int lgnWc = lgW - LGWS;            // windows per row = W / WS
  const int win = t / N, n = t % N;
  hr = ((win >> lgnWc) << LGWS) + n / WS;
  wcol = ((win & ((1 << lgnWc) - 1)) << LGWS) + n % WS;
  return (unsigned)((((hr + shift) & (H - 1)) << lgW) + ((wcol + shift) & (W - 1)));   // token index inside the image
}


// xor-16 / xor-32 butterflies of the softmax / LayerNorm row reductions.  FA_PERMLANE: v_permlane16_swap / v_permlane32_swap (gfx950,
// one VALU instruction) instead of ds_bpermute (a round trip through the LDS pipe); the sums and maxima are bitwise the same.
#ifndef FA_PERMLANE
#define FA_PERMLANE 1
#endif
typedef unsigned fa_u32x2 __attribute__((ext_vector_type(2)));
// v (op) v[lane ^ D]: after the swap of a register with itself the two results hold (own, partner) in one half of every lane pair
// and (partner, own) in the other -- the operation is commutative, so no select is needed
template <int D>
__device__ __forceinline__ float fa_xor_sum(float v) {
#if FA_PERMLANE
  const unsigned u = __float_as_uint(v);
  const fa_u32x2 r = D == 16 ? __builtin_amdgcn_permlane16_swap(u, u, false, false) : __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
#else
  return v + __shfl_xor(v, D, 64);
#endif
}
template <int D>
__device__ __forceinline__ float fa_xor_max(float v) {
#if FA_PERMLANE
  const unsigned u = __float_as_uint(v);
  const fa_u32x2 r = D == 16 ? __builtin_amdgcn_permlane16_swap(u, u, false, false) : __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
#else
  return fmaxf(v, __shfl_xor(v, D, 64));
#endif
}

// mean and 1/sqrt(var + eps) of the 96-value row this lane shares with its 3 kq partners (two-pass, like nn.LayerNorm)
__device__ __forceinline__ void row_stats(const f32x4 (&x)[6], float eps, float& mean, float& rstd) {
  float s0 = 0.f, s1 = 0.f;
#pragma unroll
  for (int c = 0; c < 6; ++c) { s0 += x[c][0] + x[c][1]; s1 += x[c][2] + x[c][3]; }
  float s = s0 + s1;
  s = fa_xor_sum<16>(s); s = fa_xor_sum<32>(s);
  mean = s * (1.0f / FC);
  float q0 = 0.f, q1 = 0.f, q2 = 0.f, q3 = 0.f;      // four independent chains
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    const float d0 = x[c][0] - mean, d1 = x[c][1] - mean, d2 = x[c][2] - mean, d3 = x[c][3] - mean;
    q0 = fmaf(d0, d0, q0); q1 = fmaf(d1, d1, q1); q2 = fmaf(d2, d2, q2); q3 = fmaf(d3, d3, q3);
  }
  float q = (q0 + q1) + (q2 + q3);
  q = fa_xor_sum<16>(q); q = fa_xor_sum<32>(q);
  rstd = 1.0f / sqrtf(q * (1.0f / FC) + eps);
}

template <int WS>
__device__ __forceinline__ void load_rows(const FusedAttnArgs& a, int xcd, int i, int shift, int wave, int lr, int kq, f32x4 (&xq)[6],
                                          f32x4 (&xkv)[6]) {
  const int b = xcd + 8 * (i >> a.lgS), t = ((i & ((1 << a.lgS) - 1)) << 6) + 16 * wave + lr;
  int hr, wc;
  const size_t src = (size_t)b * a.H * a.W + source_row<WS>(t, a.H, a.W, a.lgW, shift, hr, wc);
  const float* pq = a.tq + src * FC + 4 * kq;
  const float* pk = a.tkv + src * FC + 4 * kq;
#pragma unroll
  for (int c = 0; c < 6; ++c) xq[c] = *reinterpret_cast<const f32x4*>(pq + 16 * c);
#pragma unroll
  for (int c = 0; c < 6; ++c) xkv[c] = *reinterpret_cast<const f32x4*>(pk + 16 * c);
}

// units of this XCD's list whose cumulative start cost is < c
__device__ __forceinline__ int units_before(long c, int per, const int (&cs)[3]) {
  long base = 0;
  int n = 0;
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    const long rem = c - base;
    long k = rem <= 0 ? 0 : (rem + cs[s] - 1) / cs[s];
    if (k > per) k = per;
    n += (int)k;
    base += (long)per * cs[s];
  }
  return n;
}

// host side (attn_fused.hip): argument checks, slot order (largest windows first), per-XCD block counts per slot for the given
// unit costs (cost_ws[0..2] = cost of one unit of window 8 / 4 / 2, hundreds of cycles); *blocks = grid size (a multiple of 8)
int fa_prepare(FusedAttnArgs& a, const float* tq, const float* tkv, const float* lnq_w, const float* lnq_b, const float* lnkv_w,
               const float* lnkv_b, float eps, const float* wq, const float* bq, const float* wkv, const float* bkv,
               const float* const* bias_tables, const int* windows, const int* shifts, int n_groups, int heads_per_group, int B, int H,
               int W, int C, void* workspace, const int* cost_ws, int blocks_per_cu, long* blocks);
void fa_fold(const FusedAttnArgs& a, hipStream_t st);      // k_attn_fold: the folded weights of this call into a.folded

}  // namespace dpmn_fa
