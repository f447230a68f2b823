// Backward of the fused LayerNorm + q/kv projection + multi-size window attention (attn_fused.hip), recomputing q / k / v:
//   d(cat) (window-major, the layout WindowAttention writes without un-roll, pgrm.py:262-266, quirk Q1)
//     -> dq (B L, 96) and dkv (B L, 192) = the gradients of the q / kv Linear outputs (pgrm.py:188,194) in raster token order,
//        which the LayerNorm backward and the weight-gradient GEMMs consume, and
//     -> the relative-position-bias table gradients (pgrm.py:234-238) as per-block partial rows (added in block order by the
//        caller: no atomics reach HBM, the result is bitwise reproducible).
// The training forward no longer writes q / kv (2 x 19 MB + 2 x 38 MB per launch at B = 48): this kernel reads the SAME inputs as
// the forward (raw token rows + the folded weights of k_attn_fold) and repeats its projection bit for bit.
//
// Same unit list, XCD mapping and operand layouts as the forward: unit = (group, 64 window-major tokens), wave w = token tile
// [16w, 16w+16).  Per unit, after the projection (q' = q * scale * log2e, k, v in accumulator layout: lane (j, kq) holds
// X[token j][d = 4kq + r]) the tiles go to LDS once (Q', K, V, dO: 4 x 9 KB) and two passes run on them:
//   pass A (my tile = queries; the forward's orientation, rows = keys):  S^T = K Q'^T, dP^T = V dO^T, softmax statistics
//     (max, 1/sum) and delta = sum_k P dP per query -> LDS;  dS^T = P o (dP - delta) feeds the table gradient (in registers
//     across ALL units of the block: a lane's (query, key) pairs sit at the same window offsets in every unit) and
//     dQ^T = K^T dS^T (A operand = K column-wise from LDS).
//   pass B (my tile = keys; rows = queries):  S = Q' K^T, dP = dO V^T with the other tiles' Q' / dO rows from LDS, P from the
//     statistics of pass A;  dV^T = dO^T (P o M),  dK^T = Q'^T dS (A operands column-wise from LDS).
// 8x8 windows: the unit is one window, 4 key / query tiles per pass (368 MFMAs per wave and unit with the projection);
// 4x4 / 2x2: the wave's tile is self-contained (200 MFMAs), no block barrier is needed at all.
#include <cstdlib>
#include "attn_fused.h"

using namespace dpmn_fa;

namespace {

#ifndef FAB_SCHED
#define FAB_SCHED 1       // 1: the forward's issue pattern for the projection (row statistics in the MFMA shadows)
#endif
#ifndef FAB_HB
#define FAB_HB 1          // scheduling barrier between the two heads of a pass
#endif
#ifndef FAB_HBB
#define FAB_HBB 1         // the same in pass B
#endif
#ifndef FAB_QB
#define FAB_QB 0          // 8x8: scheduling barrier between the query tiles of pass B (measured: 75.7 us with, 73.3 us without)
#endif
#ifndef FAB_SKIP
#define FAB_SKIP 0        // timing ablations only (tools/build_variants.sh): 1 no pass A, 2 no pass B, 4 no projection MFMAs
#endif

#ifndef FAB_TIMING
#define FAB_TIMING 0      // tools/fab_timeline.py: s_memtime stamps of the loop phases of wave 0 of every block
#endif
#if FAB_TIMING
__device__ unsigned long long g_fab_t[512][9][8];
#define FAB_STAMP(u, k) do { if (blockIdx.x < 512 && threadIdx.x == 0 && (u) < 9) g_fab_t[blockIdx.x][(u)][(k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define FAB_STAMP(u, k) do {} while (0)
#endif

constexpr int STAT = 2 * 3 * 64;           // [head][max, 1/sum, delta][token of the unit]

template <int WS>
__device__ __forceinline__ void load_dout(const FusedAttnArgs& a, int xcd, int i, int g, int wave, int lr, int kq, f32x4 (&gv)[2]) {
  const int b = xcd + 8 * (i >> a.lgS), t = ((i & ((1 << a.lgS) - 1)) << 6) + 16 * wave + lr;
  const float* p = a.dout + ((size_t)b * a.H * a.W + t) * FC + FCG * g + 4 * kq;
  gv[0] = *reinterpret_cast<const f32x4*>(p);
  gv[1] = *reinterpret_cast<const f32x4*>(p + 16);
}

template <int WS>
__device__ __forceinline__ void unit_sync() {
  // 8x8: the four waves exchange tiles -> block barrier.  4x4 / 2x2: every LDS word a wave reads was written by itself (LDS
  // operations of one wave execute in order), only the compiler must not move accesses across
  if (WS == 8) __syncthreads();
  else __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

template <int WS, bool DROP>
__device__ __forceinline__ void run_units_bwd(const FusedAttnArgs& a, int slot, int xcd, int first, int last, float* smem) {
  constexpr int N = WS * WS, TBL = (2 * WS - 1) * (2 * WS - 1);
  constexpr int KT = (WS == 8) ? 4 : 1;    // key tiles per query tile (pass A) = query tiles per key tile (pass B)
  float* Wsm = smem;                       // [96][LDW] folded weights | pbias [2][96] | tbl [TBLPAD]: as in the forward
  float* pbias = Wsm + FC * LDW;
  float* tbl = pbias + 2 * FC;
  float* Qs = tbl + TBLPAD;                // [64][LDK] q' = q * scale * log2e
  float* Ks = Qs + 64 * LDK;
  float* Vs = Ks + 64 * LDK;
  float* Gs = Vs + 64 * LDK;               // dO
  float* stat = Gs + 64 * LDK;             // [2][3][64]
  int* reg_s = reinterpret_cast<int*>(stat + STAT);      // [64] shift-mask region of each token of the unit
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, kq = lane >> 4;
  const int H = a.H, W = a.W, L = H * W, S = L / 64;
  const int g = a.gid[slot], shift = a.shift[slot];

  FAB_STAMP(8, 0);
  f32x4 xq[6], xkv[6], gv[2];
  {
    const f32x4* src = reinterpret_cast<const f32x4*>(a.folded + (size_t)g * FOLD_STRIDE);
    f32x4* dst = reinterpret_cast<f32x4*>(smem);
    constexpr int NV4 = FOLD_STRIDE / 4, NIT = (NV4 + 255) / 256;
    f32x4 wv[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int i = tid + 256 * k;
      wv[k] = src[i < NV4 ? i : NV4 - 1];
    }
    load_rows<WS>(a, xcd, first, shift, wave, lr, kq, xq, xkv);
    load_dout<WS>(a, xcd, first, g, wave, lr, kq, gv);
    __syncthreads();                       // the previous slot's readers of the staged tables are done
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int i = tid + 256 * k;
      if (i < NV4) dst[i] = wv[k];
    }
  }
  __syncthreads();
  FAB_STAMP(8, 1);

  // ---- relative position bias of this lane's (query, key) pairs: the same in every unit
  // pass A (query = my token 16w + lr, keys 16kt + 4kq + r): exactly the forward's lookup
  constexpr int RBN = (WS == 8) ? 1 : 4;
  float rbA[RBN][2], rbB[RBN][2];
  const float* tbA[4];
  const float* tbB;                        // 8x8, pass B (key = my token, queries 16qt + 4kq + r): index linear in (qt, r)
  {
    const int n = (16 * wave + lr) % N, iq = n / WS, jq = n % WS;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int krow = (WS == 8 ? 48 : 16 * wave) + 4 * kq + r;
      const int nk = krow % N, ik = nk / WS, jk = nk % WS;
      const int idx = ((iq - ik + WS - 1) * (2 * WS - 1) + (jq - jk + WS - 1)) * 2;
      tbA[r] = tbl + idx;
      if (WS != 8) {
        rbA[r % RBN][0] = tbl[idx];
        rbA[r % RBN][1] = tbl[idx + 1];
        if (WS == 2 && kq != (lr >> 2)) { rbA[r % RBN][0] = -INFINITY; rbA[r % RBN][1] = -INFINITY; }
        // pass B: query = token 4kq + r of my tile, key = my token lr
        const int nq = (16 * wave + 4 * kq + r) % N, iq2 = nq / WS, jq2 = nq % WS;
        const int idb = ((iq2 - iq + WS - 1) * (2 * WS - 1) + (jq2 - jq + WS - 1)) * 2;
        rbB[r % RBN][0] = tbl[idb];
        rbB[r % RBN][1] = tbl[idb + 1];
        if (WS == 2 && kq != (lr >> 2)) { rbB[r % RBN][0] = -INFINITY; rbB[r % RBN][1] = -INFINITY; }
      }
    }
    // 8x8: query (2qt + kq/2, 4(kq%2) + r), key (iq, jq):  idx = base + 30 qt + r
    tbB = tbl + (((kq >> 1) - iq + WS - 1) * (2 * WS - 1) + (4 * (kq & 1) - jq + WS - 1)) * 2;
  }
  // table gradient of this lane's pairs, summed over all units of the block: [kt][r][head]
  float tacc[KT][4][2];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) { tacc[kt][r][0] = 0.f; tacc[kt][r][1] = 0.f; }

  for (int i = first; i < last; ++i) {
    const int b = xcd + 8 * (i >> a.lgS), t0 = (i & (S - 1)) << 6, t = t0 + 16 * wave + lr;
    int hr_, wc_;
    const size_t src = (size_t)b * L + source_row<WS>(t, H, W, a.lgW, shift, hr_, wc_);
    FAB_STAMP(i - first, 0);
    float mq, rq, mk, rk;
    row_stats(xq, a.eps, mq, rq);
    row_stats(xkv, a.eps, mk, rk);
    const float rqs = rq * QSCALE, nmq = -mq * rq, nmk = -mk * rk;

    // ---- projections: identical instruction sequence to the forward (q / k / v come out bit for bit)
    f32x4 qa[2], ka[2], va[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) { qa[h] = (f32x4){0.f, 0.f, 0.f, 0.f}; ka[h] = qa[h]; va[h] = qa[h]; }
    if (FAB_SKIP & 4) { qa[0] = xq[0] + xq[2]; qa[1] = xq[1] + xq[3]; ka[0] = xkv[0] + xq[4]; ka[1] = xkv[1] + xq[5]; va[0] = xkv[2] + xkv[4]; va[1] = xkv[3] + xkv[5]; }
    else
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      f32x4 wf[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) wf[j] = *reinterpret_cast<const f32x4*>(Wsm + (16 * j + lr) * LDW + 16 * c + 4 * kq);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        qa[0] = mfma16(wf[0][s], xq[c][s], qa[0]);
        qa[1] = mfma16(wf[1][s], xq[c][s], qa[1]);
        ka[0] = mfma16(wf[2][s], xkv[c][s], ka[0]);
        ka[1] = mfma16(wf[3][s], xkv[c][s], ka[1]);
        va[0] = mfma16(wf[4][s], xkv[c][s], va[0]);
        va[1] = mfma16(wf[5][s], xkv[c][s], va[1]);
      }
    }
#if FAB_SCHED
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
      for (int m = 0; m < 24; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
      }
    }
#endif
    FAB_STAMP(i - first, 1);
    unit_sync<WS>();                       // the previous unit's pass B is done with the tiles
    FAB_STAMP(i - first, 2);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int f = 16 * h + 4 * kq;
      const f32x4 cv = *reinterpret_cast<const f32x4*>(pbias + FC + 64 + f), bv4 = *reinterpret_cast<const f32x4*>(pbias + 64 + f);
#pragma unroll
      for (int e = 0; e < 4; ++e) va[h][e] = fmaf(va[h][e], rk, fmaf(nmk, cv[e], bv4[e]));
      *reinterpret_cast<f32x4*>(Vs + (16 * wave + lr) * LDK + f) = va[h];
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int f = 16 * h + 4 * kq;
      const f32x4 ck = *reinterpret_cast<const f32x4*>(pbias + FC + 32 + f), bk4 = *reinterpret_cast<const f32x4*>(pbias + 32 + f);
#pragma unroll
      for (int e = 0; e < 4; ++e) ka[h][e] = fmaf(ka[h][e], rk, fmaf(nmk, ck[e], bk4[e]));
      *reinterpret_cast<f32x4*>(Ks + (16 * wave + lr) * LDK + f) = ka[h];
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int f = 16 * h + 4 * kq;
      const f32x4 cq = *reinterpret_cast<const f32x4*>(pbias + FC + f), bq4 = *reinterpret_cast<const f32x4*>(pbias + f);
#pragma unroll
      for (int e = 0; e < 4; ++e) qa[h][e] = fmaf(qa[h][e], rqs, fmaf(nmq, cq[e], bq4[e]));
      *reinterpret_cast<f32x4*>(Qs + (16 * wave + lr) * LDK + f) = qa[h];
      *reinterpret_cast<f32x4*>(Gs + (16 * wave + lr) * LDK + f) = gv[h];
    }
    if (shift > 0 && kq == 0) {
      const int rh = hr_ < H - WS ? 0 : (hr_ < H - shift ? 1 : 2), rw = wc_ < W - WS ? 0 : (wc_ < W - shift ? 1 : 2);
      reg_s[16 * wave + lr] = 3 * rh + rw;
    }
    f32x4 g_own[2] = {gv[0], gv[1]};
    // ---- the row registers are dead: the loads of unit i+1 fly during the passes (index clamped, not predicated).  4x4 / 2x2:
    // sent here; 8x8: after pass A (its 64 score / gradient registers and the 56 prefetch registers do not fit together, and
    // pass B alone -- 128 MFMAs -- covers the latency)
    const int nx = i + 1 < last ? i + 1 : i;
    if (WS != 8) {
      load_rows<WS>(a, xcd, nx, shift, wave, lr, kq, xq, xkv);
      load_dout<WS>(a, xcd, nx, g, wave, lr, kq, gv);
    }
    unit_sync<WS>();
    FAB_STAMP(i - first, 3);

    // ================= pass A: my tile = queries
    unsigned maskedA = 0u;
    int my_reg = 0;
    if (shift > 0) {
      my_reg = reg_s[16 * wave + lr];
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          maskedA |= (reg_s[(WS == 8 ? 16 * kt : 16 * wave) + 4 * kq + r] != my_reg ? 1u : 0u) << (4 * kt + r);
    }
    if (!(FAB_SKIP & 1))
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (FAB_HB) __builtin_amdgcn_sched_barrier(0);   // one head at a time: interleaving the heads doubles the live score registers
      f32x4 sacc[KT], dpt[KT];
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        f32x4 kf = ka[h], vf = va[h];
        if (WS == 8) {
          kf = *reinterpret_cast<const f32x4*>(Ks + (16 * kt + lr) * LDK + 16 * h + 4 * kq);
          vf = *reinterpret_cast<const f32x4*>(Vs + (16 * kt + lr) * LDK + 16 * h + 4 * kq);
        }
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f}, acc2 = acc;
#pragma unroll
        for (int s = 0; s < 4; ++s) { acc = mfma16(kf[s], qa[h][s], acc); acc2 = mfma16(vf[s], g_own[h][s], acc2); }
        sacc[kt] = acc;
        dpt[kt] = acc2;
      }
      float mx = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = sacc[kt][r] + (WS == 8 ? tbA[r][60 * (3 - kt) + h] : rbA[r % RBN][h]);
          if ((maskedA >> (4 * kt + r)) & 1u) v += -100.0f * LOG2E;
          sacc[kt][r] = v;
          mx = fmaxf(mx, v);
        }
      mx = fa_xor_max<16>(mx);
      mx = fa_xor_max<32>(mx);
      float den = 0.f;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = __builtin_amdgcn_exp2f(sacc[kt][r] - mx);
          sacc[kt][r] = p;
          den += p;
        }
      den = fa_xor_sum<16>(den);
      den = fa_xor_sum<32>(den);
      const float inv = 1.0f / den;
      if (DROP) {                          // dP = (V dO^T) o M with the forward's masks (0 or 1 / (1 - p))
        const unsigned long long e0 = ((((unsigned long long)b * 3 + g) * 2 + h) * L + t) * N;
        const unsigned long long z0 = drop_z0(a.seed, e0 + (WS == 2 ? 0 : 4 * kq));      // + (16 kt + r) * PHI: constant adds
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            dpt[kt][r] *= drop_scale_z(z0 + (unsigned long long)((WS == 8 ? 16 * kt : 0) + r) * DROP_PHI, a.p_drop, a.inv_keep);
      }
      float dlt = 0.f;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) { sacc[kt][r] *= inv; dlt = fmaf(sacc[kt][r], dpt[kt][r], dlt); }
      dlt = fa_xor_sum<16>(dlt);
      dlt = fa_xor_sum<32>(dlt);
      if (kq == 0) {
        stat[(3 * h + 0) * 64 + 16 * wave + lr] = mx;
        stat[(3 * h + 1) * 64 + 16 * wave + lr] = inv;
        stat[(3 * h + 2) * 64 + 16 * wave + lr] = dlt;
      }
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float ds = sacc[kt][r] * (dpt[kt][r] - dlt);      // dS^T[key][query]
          sacc[kt][r] = ds;
          tacc[kt][r][h] += ds;
        }
      // dQ^T = K^T . dS^T: two accumulator chains
      f32x4 d0 = (f32x4){0.f, 0.f, 0.f, 0.f}, d1 = d0;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int s = 0; s < 4; s += 2) {
          const int krow = (WS == 8 ? 16 * kt : 16 * wave) + 4 * kq + s;
          d0 = mfma16(Ks[krow * LDK + 16 * h + lr], sacc[kt][s], d0);
          d1 = mfma16(Ks[(krow + 1) * LDK + 16 * h + lr], sacc[kt][s + 1], d1);
        }
      d0 += d1;
      // q = Linear output, S = scale * q k + bias  =>  dq = scale * dS K
      *reinterpret_cast<f32x4*>(a.dq + src * FC + FCG * g + 16 * h + 4 * kq) = d0 * 0.25f;
    }
    FAB_STAMP(i - first, 4);
    if (WS == 8) {
      load_rows<WS>(a, xcd, nx, shift, wave, lr, kq, xq, xkv);
      load_dout<WS>(a, xcd, nx, g, wave, lr, kq, gv);
    }
    unit_sync<WS>();                       // the statistics of all four tiles are in LDS
    FAB_STAMP(i - first, 5);

    // ================= pass B: my tile = keys
    unsigned maskedB = 0u;                 // bit (4 qt + r): query 16 qt + 4 kq + r sits in another shift-mask region than my key
    if (shift > 0) {
#pragma unroll
      for (int qt = 0; qt < KT; ++qt) {
        const int4 qr = *reinterpret_cast<const int4*>(reg_s + (WS == 8 ? 16 * qt : 16 * wave) + 4 * kq);
        maskedB |= ((qr.x != my_reg ? 1u : 0u) | (qr.y != my_reg ? 2u : 0u) | (qr.z != my_reg ? 4u : 0u) | (qr.w != my_reg ? 8u : 0u)) << (4 * qt);
      }
    }
    // attn_drop mask index of (head h, query t0 + q0 + 4 kq + r, my key): linear in (qt, r) -> one 64-bit multiply per head here, constant
    // adds per element below
    unsigned long long zB[2] = {0ull, 0ull};
    if (DROP) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
        zB[h] = drop_z0(a.seed, ((((unsigned long long)b * 3 + g) * 2 + h) * L + (t0 + (WS == 8 ? 0 : 16 * wave) + 4 * kq)) * N + (16 * wave + lr) % N);
    }
    if (!(FAB_SKIP & 2))
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (FAB_HBB) __builtin_amdgcn_sched_barrier(0);
      f32x4 dva0 = (f32x4){0.f, 0.f, 0.f, 0.f}, dka0 = dva0, dva1 = dva0, dka1 = dva0;
#pragma unroll
      for (int qt = 0; qt < KT; ++qt) {
        if (WS == 8 && FAB_QB) __builtin_amdgcn_sched_barrier(0);
        const int q0 = WS == 8 ? 16 * qt : 16 * wave;
        const f32x4 qf = *reinterpret_cast<const f32x4*>(Qs + (q0 + lr) * LDK + 16 * h + 4 * kq);
        const f32x4 gf = *reinterpret_cast<const f32x4*>(Gs + (q0 + lr) * LDK + 16 * h + 4 * kq);
        f32x4 sa = (f32x4){0.f, 0.f, 0.f, 0.f}, dp = sa;
#pragma unroll
        for (int s = 0; s < 4; ++s) { sa = mfma16(qf[s], ka[h][s], sa); dp = mfma16(gf[s], va[h][s], dp); }
        const f32x4 mx4 = *reinterpret_cast<const f32x4*>(stat + (3 * h + 0) * 64 + q0 + 4 * kq);
        const f32x4 iv4 = *reinterpret_cast<const f32x4*>(stat + (3 * h + 1) * 64 + q0 + 4 * kq);
        const f32x4 dl4 = *reinterpret_cast<const f32x4*>(stat + (3 * h + 2) * 64 + q0 + 4 * kq);
        f32x4 pm, ds;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = sa[r] + (WS == 8 ? tbB[(30 * qt + r) * 2 + h] : rbB[r % RBN][h]);
          if ((maskedB >> (4 * qt + r)) & 1u) v += -100.0f * LOG2E;
          const float p = __builtin_amdgcn_exp2f(v - mx4[r]) * iv4[r];
          float m = 1.0f;
          if (DROP) m = drop_scale_z(zB[h] + (unsigned long long)(((WS == 8 ? 16 * qt : 0) + r) * N) * DROP_PHI, a.p_drop, a.inv_keep);
          pm[r] = p * m;                                  // P o M: what multiplies V in the forward
          ds[r] = p * (dp[r] * m - dl4[r]);               // dS[query][key]
        }
        // dV^T += dO^T . (P o M),  dK^T += Q'^T . dS   (A operands column-wise from LDS)
#pragma unroll
        for (int s = 0; s < 4; s += 2) {
          const int row = (q0 + 4 * kq + s) * LDK + 16 * h + lr;
          dva0 = mfma16(Gs[row], pm[s], dva0);
          dka0 = mfma16(Qs[row], ds[s], dka0);
          dva1 = mfma16(Gs[row + LDK], pm[s + 1], dva1);
          dka1 = mfma16(Qs[row + LDK], ds[s + 1], dka1);
        }
      }
      dva0 += dva1;
      dka0 += dka1;
      // dk = scale * dS^T q = dS^T q' / log2e
      float* dkp = a.dkv + src * (2 * FC) + FCG * g + 16 * h + 4 * kq;
      *reinterpret_cast<f32x4*>(dkp) = dka0 * (1.0f / LOG2E);
      *reinterpret_cast<f32x4*>(dkp + FC) = dva0;
    }
    FAB_STAMP(i - first, 6);
  }
  FAB_STAMP(8, 2);

  // ---- this block's table-gradient partial row.  Every lane parks its sums in LDS (aliasing the tiles, [value][thread]: no bank
  // conflicts), then one thread per table entry adds the 64 (query, key) pairs of its offset in a fixed order -- no atomics, and
  // nothing like the 64-way serialised LDS atomics of colliding lanes (measured: 40 k cycles per 8x8 block)
  __syncthreads();
  float* park = Qs;                        // [KT * 8][256]
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      park[((4 * kt + r) * 2 + 0) * 256 + tid] = tacc[kt][r][0];
      park[((4 * kt + r) * 2 + 1) * 256 + tid] = tacc[kt][r][1];
    }
  __syncthreads();
  {
    // pair (query n_q, key n_k) of a window lives in:  8x8: wave n_q / 16, lane (lr = n_q % 16, kq = (n_k / 4) % 4), value (kt = n_k / 16,
    // r = n_k % 4);  4x4: every wave (its tile = one window), lane (lr = n_q, kq = n_k / 4), r = n_k % 4;  2x2: every wave, each of
    // the tile's 4 windows c: lane (lr = 4c + n_q, kq = c), r = n_k
    constexpr int REP = WS == 8 ? 1 : (WS == 4 ? 4 : 16);
    float* row = a.tpart[slot] + (size_t)blockIdx.x * (TBL * 2);
    for (int e = tid; e < TBL * 2; e += 256) {
      const int h = e & 1, ent = e >> 1, di = ent / (2 * WS - 1) - (WS - 1), dj = ent % (2 * WS - 1) - (WS - 1);
      float sum = 0.f;
      for (int nq = 0; nq < N; ++nq) {
        const int iq = nq / WS, jq = nq % WS, ik = iq - di, jk = jq - dj;
        if (ik < 0 || ik >= WS || jk < 0 || jk >= WS) continue;
        const int nk = ik * WS + jk;
#pragma unroll
        for (int rep = 0; rep < REP; ++rep) {
          int thr, val;
          if (WS == 8) { thr = 64 * (nq >> 4) + 16 * ((nk >> 2) & 3) + (nq & 15); val = 4 * (nk >> 4) + (nk & 3); }
          else if (WS == 4) { thr = 64 * rep + 16 * (nk >> 2) + nq; val = nk & 3; }
          else { thr = 64 * (rep >> 2) + 16 * (rep & 3) + 4 * (rep & 3) + nq; val = nk; }
          sum += park[(val * 2 + h) * 256 + thr];
        }
      }
      row[e] = sum;
    }
  }
  __syncthreads();                         // park aliases the next slot's tiles
  FAB_STAMP(8, 3);
}

__device__ __forceinline__ void zero_row(const FusedAttnArgs& a, int slot) {
  const int n = (2 * a.ws[slot] - 1) * (2 * a.ws[slot] - 1) * 2;
  float* row = a.tpart[slot] + (size_t)blockIdx.x * n;
  for (int i = threadIdx.x; i < n; i += 256) row[i] = 0.f;
}

template <bool DROP>
__global__ __launch_bounds__(256, 2) void k_ln_qkv_window_attn_bwd(FusedAttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, nbx = gridDim.x >> 3;
  const int S = 1 << a.lgS;
  const int ni = xcd < a.B ? (a.B - xcd + 7) / 8 : 0;
  const int per = ni * S;                  // units per slot on this XCD
  unsigned done = 0u;                      // slots whose partial row this block wrote
  int lo3[3] = {0, 0, 0}, hi3[3] = {0, 0, 0};      // this block's unit range of each slot
  if (per > 0) {
    const int* nb = a.nblk[ni == (a.B + 7) / 8 ? 0 : 1];
    if (nb[0] > 0) {
      const int slot = j < nb[0] ? 0 : (j < nb[0] + nb[1] ? 1 : 2);
      const int jj = j - (slot == 0 ? 0 : (slot == 1 ? nb[0] : nb[0] + nb[1])), ns = nb[slot];
      lo3[slot] = (int)((long)per * jj / ns);
      hi3[slot] = (int)((long)per * (jj + 1) / ns);
    } else {
      const int cs[3] = {a.cost[0], a.cost[1], a.cost[2]};
      const long ctot = (long)per * (cs[0] + cs[1] + cs[2]);
      const int u0 = units_before(ctot * j / nbx, per, cs);
      const int u1 = j + 1 == nbx ? 3 * per : units_before(ctot * (j + 1) / nbx, per, cs);
      for (int slot = 0; slot < 3; ++slot) {
        lo3[slot] = u0 > slot * per ? u0 - slot * per : 0;
        hi3[slot] = (u1 < (slot + 1) * per ? u1 : (slot + 1) * per) - slot * per;
      }
    }
  }
  for (int slot = 0; slot < 3; ++slot) {
    if (lo3[slot] >= hi3[slot]) continue;  // block-uniform
    const int ws = a.ws[slot];
    if (ws == 8) run_units_bwd<8, DROP>(a, slot, xcd, lo3[slot], hi3[slot], smem);
    else if (ws == 4) run_units_bwd<4, DROP>(a, slot, xcd, lo3[slot], hi3[slot], smem);
    else run_units_bwd<2, DROP>(a, slot, xcd, lo3[slot], hi3[slot], smem);
    done |= 1u << slot;
  }
  for (int slot = 0; slot < 3; ++slot)
    if (!((done >> slot) & 1u)) zero_row(a, slot);
}


#ifdef FAB_DIAG
template <int WS>
__global__ __launch_bounds__(256, 2) void k_fab_diag(FusedAttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  run_units_bwd<WS, false>(a, 0, blockIdx.x & 7, blockIdx.x, blockIdx.x + 4, smem);
}
template __global__ void k_fab_diag<8>(FusedAttnArgs);
template __global__ void k_fab_diag<4>(FusedAttnArgs);
template __global__ void k_fab_diag<2>(FusedAttnArgs);
#endif

long bwd_grid(int B, int H, int W, int bpc) {
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
    n_cu = prop.multiProcessorCount > 8 ? prop.multiProcessorCount / 8 * 8 : 8;
  }
  long blocks = (long)bpc * n_cu;
  const long need = ((3 * (long)B * (H * W / 64) + 7) / 8) * 8;
  return blocks > need ? need : blocks;
}

int bwd_bpc() {
  static const int bpc = getenv("DPMN_FAB_BPC") ? atoi(getenv("DPMN_FAB_BPC")) : 2;
  return bpc;
}

}  // namespace

extern "C" {

int dpmn_ln_qkv_window_attn_bwd_part_rows(int B, int H, int W) {
  if (B <= 0 || H <= 0 || W <= 0) return 0;
  return (int)bwd_grid(B, H, W, bwd_bpc());
}

int dpmn_ln_qkv_window_attn_bwd_f32(const float* tq, const float* tkv, const float* lnq_w, const float* lnq_b, const float* lnkv_w,
                                    const float* lnkv_b, float eps, const float* wq, const float* bq, const float* wkv,
                                    const float* bkv, const float* const* bias_tables, const int* windows, const int* shifts,
                                    int n_groups, int heads_per_group, const float* dout, float* dq, float* dkv,
                                    float* const* dtable_parts, float p_drop, unsigned long long seed, void* workspace, int refold,
                                    int B, int H, int W, int C, dpmn_stream_t stream) {
  DPMN_REQUIRE(dout && dq && dkv && dtable_parts, "ln_qkv_window_attn_bwd: null pointer");
  DPMN_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "ln_qkv_window_attn_bwd: p_drop outside [0, 1)");
  FusedAttnArgs a{};
  static const int c8 = getenv("DPMN_FAB_COST8") ? atoi(getenv("DPMN_FAB_COST8")) : 330;
  static const int c4 = getenv("DPMN_FAB_COST4") ? atoi(getenv("DPMN_FAB_COST4")) : 190;
  const int cost_ws[3] = {c8, c4, c4};       // 368 / 200 / 200 MFMAs per wave and unit
  long blocks = 0;
  const int rc = fa_prepare(a, tq, tkv, lnq_w, lnq_b, lnkv_w, lnkv_b, eps, wq, bq, wkv, bkv, bias_tables, windows, shifts, n_groups,
                            heads_per_group, B, H, W, C, workspace, cost_ws, bwd_bpc(), &blocks);
  if (rc != DPMN_OK) return rc;
  DPMN_REQUIRE(blocks == bwd_grid(B, H, W, bwd_bpc()), "ln_qkv_window_attn_bwd: grid size differs from dpmn_ln_qkv_window_attn_bwd_part_rows");
  a.dout = dout; a.dq = dq; a.dkv = dkv; a.p_drop = p_drop; a.inv_keep = 1.0f / (1.0f - p_drop); a.seed = seed;
  for (int s = 0; s < 3; ++s) {
    DPMN_REQUIRE(dtable_parts[a.gid[s]], "ln_qkv_window_attn_bwd: null table-gradient partial buffer");
    a.tpart[s] = dtable_parts[a.gid[s]];
  }
  const size_t smem = (size_t)(FOLD_STRIDE + 4 * 64 * LDK + STAT + 64) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ln_qkv_window_attn_bwd<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ln_qkv_window_attn_bwd<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_set = true;
  }
  const double tokens = (double)B * H * W;
  double attn = 0.0;
  for (int g = 0; g < 3; ++g) attn += 14.0 * windows[g] * windows[g] * FD * 2 * tokens;   // 7 products of 2 N d FLOP per (token, head)
  hipStream_t st = as_stream(stream);
  if (refold) fa_fold(a, st);
  ProfScope prof(PT_ATTN_FUSED_BWD, st, 2.0 * tokens * FC * (3 * FC) + attn, 4.0 * (6.0 * tokens * FC + 3.0 * FC * FC));
  if (p_drop > 0.f) hipLaunchKernelGGL(k_ln_qkv_window_attn_bwd<true>, dim3((unsigned)blocks), dim3(256), smem, st, a);
  else hipLaunchKernelGGL(k_ln_qkv_window_attn_bwd<false>, dim3((unsigned)blocks), dim3(256), smem, st, a);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

#if FAB_TIMING
int dpmn_fab_timing_dump(unsigned long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_fab_t), sizeof(unsigned long long) * 512 * 9 * 8);
}
int dpmn_fab_timing_clear() {
  static unsigned long long z[512 * 9 * 8];
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_fab_t), z, sizeof(z));
}
#endif

}  // extern "C"
