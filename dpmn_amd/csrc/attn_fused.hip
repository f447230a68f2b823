// Fused LayerNorm + q/kv projection + multi-size window attention (SURVEY.md K3 + K4 + K5):
//   norm1_q / norm1_kv (pgrm.py:322-323)  ->  q = Linear(96,96), kv = Linear(96,192) (pgrm.py:188,194)
//   -> per channel group g (window 2 / 4 / 8): roll, window partition, 2 heads x 16, q*scale, QK^T + relative position
//      bias (+ shift mask), softmax, P.V, window-major write without un-roll (pgrm.py:197-266, quirk Q1).
// q and kv never exist in HBM: the unfused path wrote and re-read 3 x (B, L, 96) floats per block (56 MB at B = 48).
//
// Work unit = (group g, 64 consecutive window-major tokens of one image) = one 256-thread block; wave w owns the unit's
// token tile [16w, 16w+16).  Everything GEMM-shaped runs on v_mfma_f32_16x16x4_f32 and chains through the accumulator
// registers -- the MFMA D layout (lane (j = l&15, kq = l>>4) holds D[feature 4kq+r][token j]) is at once
//   * the B-operand layout of the next product's "token" side, and
//   * the A-operand layout of a [token][feature] matrix,
// so with the output-feature axis on A and the token axis on B:
//   1. X rows (gathered through the roll / window permutation of THIS group) go straight from global memory into the
//      B-operand registers: lane (j, kq) loads x[token j][16c + 4kq .. +3], c = 0..5 -- the reduction index is permuted
//      (k-step (c, s) <-> k = 16c + 4kq + s) identically on both operands.  LayerNorm = 24 in-lane values + two
//      xor-shuffles (16, 32); no LDS staging of activations at all.
//   2. projection: A = the group's 96 weight rows (q: 32, k: 32, v: 32) from an LDS copy shared by the 4 waves;
//      result lane (j, kq) holds q|k|v[token j][head h, d = 4kq+r].
//   3. S^T = K . Q^T: A = K[key][d], B = Q^T[d][query] -- both ARE the projection accumulators (own tile); the 8x8 group
//      needs the other waves' keys, exchanged through 9 KB of LDS.  4x4 windows = exactly one 16x16 tile per head,
//      2x2 windows = four windows on the tile's block diagonal (off-diagonal logits = -inf).
//   4. softmax over keys = in-lane values + xor-shuffles 16, 32; P stays in the accumulator registers.
//   5. O^T = V^T . P: B = P (registers), A = V^T read column-wise from the LDS copy of V (the only transpose).
// Algorithmic work per unit: projections 2*64*96*96 = 1.18 MFLOP + attention 4*N*16 FLOP per (token, head); bytes: the unit's
// 128 input rows (49 KB) + 8 KB of output -- the kernel is MFMA-bound (AI ~57 FLOP/B).
#include <cstdlib>
#include "attn_fused.h"

#ifndef FA_SKIP
#define FA_SKIP 0
#endif
#ifndef FA_SCHED
#define FA_SCHED 2      // timing ablations only (tools/variants): 1 no LayerNorm, 2 no projection MFMAs, 4 no attention, 8 no barrier, 16 no row loads
#endif

#ifndef FA_TIMING
#define FA_TIMING 0    // tools/fa_timeline.py: s_memtime stamps of the loop phases of wave 0 of the first 64 blocks
#endif
#if FA_TIMING
__device__ unsigned long long g_fa_t[512][9][8];
#define FA_STAMP(u, k) do { if (blockIdx.x < 512 && threadIdx.x == 0 && (u) < 9) g_fa_t[blockIdx.x][(u)][(k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define FA_STAMP(u, k) do {} while (0)
#endif

using namespace dpmn_fa;

namespace {


// Folded projection weights, once per call (they used to be rebuilt by every block at every slot change: 15 k cycles of a
// 120 k-cycle kernel).  LayerNorm's affine goes into the projection, y = W (gamma * xhat + beta) + b = (W diag gamma) xhat +
// (W beta + b), and the normalisation moves behind the MFMAs by linearity, W' xhat = rstd * (W' x - mean * rowsum(W')): the
// MFMAs eat RAW rows, the per-element LayerNorm arithmetic becomes a 2-fma fix-up of the accumulators.
// Output per group g: [96][LDW] W' (rows q 32 | k 32 | v 32) | b' [96] | rowsum(W') [96] | bias table * log2(e) [TBLPAD];
// the q rows of b' / rowsum carry head_dim ** -0.5 * log2(e).  grid (97, 3) x 64 threads: block (r, g) = row r, block 96 = table.
__global__ __launch_bounds__(64) void k_attn_fold(FusedAttnArgs a) {
  const int g = blockIdx.y, r = blockIdx.x, tid = threadIdx.x;
  float* dst = a.folded + (size_t)g * FOLD_STRIDE;
  if (r == FC) {
    int slot = 0;
    for (int s_ = 1; s_ < 3; ++s_) if (a.gid[s_] == g) slot = s_;
    const int n = (2 * a.ws[slot] - 1) * (2 * a.ws[slot] - 1) * 2;
    for (int i = tid; i < TBLPAD; i += 64) dst[FC * LDW + 2 * FC + i] = i < n ? a.table[slot][i] * LOG2E : 0.f;
    return;
  }
  __shared__ float part[24][2];
  const float* srcw = r < 32 ? a.wq + (size_t)(FCG * g + r) * FC
                             : (r < 64 ? a.wkv + (size_t)(FCG * g + r - 32) * FC : a.wkv + (size_t)(FC + FCG * g + r - 64) * FC);
  const float* gam = r < 32 ? a.lnq_w : a.lnkv_w;
  const float* bet = r < 32 ? a.lnq_b : a.lnkv_b;
  if (tid < 24) {
    const int c4 = 4 * tid;
    const f32x4 wv = *reinterpret_cast<const f32x4*>(srcw + c4);
    const f32x4 wg = wv * *reinterpret_cast<const f32x4*>(gam + c4), wb = wv * *reinterpret_cast<const f32x4*>(bet + c4);
    *reinterpret_cast<f32x4*>(dst + r * LDW + c4) = wg;
    part[tid][0] = (wg[0] + wg[1]) + (wg[2] + wg[3]);
    part[tid][1] = (wb[0] + wb[1]) + (wb[2] + wb[3]);
  } else if (tid == 24) {
    *reinterpret_cast<f32x4*>(dst + r * LDW + FC) = (f32x4){0.f, 0.f, 0.f, 0.f};      // row padding
  }
  __syncthreads();
  if (tid == 0) {
    float cw = 0.f, bb = r < 32 ? a.bq[FCG * g + r] : (r < 64 ? a.bkv[FCG * g + r - 32] : a.bkv[FC + FCG * g + r - 64]);
    for (int k = 0; k < 24; ++k) { cw += part[k][0]; bb += part[k][1]; }      // fixed order
    dst[FC * LDW + r] = r < 32 ? bb * QSCALE : bb;             // b' = b + W beta
    dst[FC * LDW + FC + r] = r < 32 ? cw * QSCALE : cw;        // rowsum(W diag gamma)
  }
}

// Units [first, last) of ONE slot (window size WS), decoded as unit i -> image xcd + 8 * (i / S), slab i % S.
// Software pipeline (vmcnt retires in order, so the ONLY global loads inside the loop are the row prefetches):
//   top: x = rows of unit i (arrived) -> LayerNorm -> projections (x dead) -> issue the loads of unit i+1 into the same
//   registers -> attention of unit i + stores.
template <int WS, bool TRAIN>
__device__ __forceinline__ void run_units(const FusedAttnArgs& a, int slot, int xcd, int first, int last, float* smem) {
  constexpr int N = WS * WS, TBL = (2 * WS - 1) * (2 * WS - 1);
  constexpr int KT = (WS == 8) ? 4 : 1;    // key tiles per query tile
  // LDS: [ Wsm [96][LDW] | pbias [2][96] | tbl [TBLPAD] ] = one contiguous copy of this group's block of the folded-weight
  // workspace (k_attn_fold below), then the K / V exchange buffers and the shift-mask regions
  float* Wsm = smem;                       // [96][LDW]: rows 0-31 W'q, 32-63 W'k, 64-95 W'v of this group (LayerNorm gamma folded in)
  float* pbias = Wsm + FC * LDW;           // [2][96]: folded biases b' of the q / k / v rows, then rowsum(W')
  float* tbl = pbias + 2 * FC;             // [TBL][2] relative position bias table, times log2(e)
  float* KVs = tbl + TBLPAD;               // 2 x { K [64][LDK], V [64][LDK] }: double-buffered across units (one barrier per unit)
  int* reg_all = reinterpret_cast<int*>(KVs + 4 * 64 * LDK);   // 2 x [64] shift-mask region of each token of the unit
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, kq = lane >> 4;
  const int H = a.H, W = a.W, L = H * W, S = L / 64;
  const int g = a.gid[slot], shift = a.shift[slot];

  // global loads in the order they are consumed (vmcnt retires in order): this group's folded weights, then the first unit's
  // rows, which fly while the weights go to LDS
  FA_STAMP(8, slot * 2);
  f32x4 xq[6], xkv[6];
  {
    const f32x4* src = reinterpret_cast<const f32x4*>(a.folded + (size_t)g * FOLD_STRIDE);
    f32x4* dst = reinterpret_cast<f32x4*>(smem);
    constexpr int NV4 = FOLD_STRIDE / 4, NIT = (NV4 + 255) / 256;
    f32x4 wv[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k) {        // the tail index is clamped, not predicated
      const int i = tid + 256 * k;
      wv[k] = src[i < NV4 ? i : NV4 - 1];
    }
    load_rows<WS>(a, xcd, first, shift, wave, lr, kq, xq, xkv);
    __syncthreads();                       // the previous slot's readers of the staged tables are done
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int i = tid + 256 * k;
      if (i < NV4) dst[i] = wv[k];
    }
  }
  __syncthreads();
  FA_STAMP(8, slot * 2 + 1);
  // relative position bias of (my query, my keys): the same in every window, hence in every unit (pgrm.py:234-238).
  // 4x4 / 2x2: 8 values, kept in registers.  8x8: 32 values -- their table index is linear in the key tile
  // (idx(kt, r) = idx(3, r) + 60 (3 - kt)), so four base pointers + immediate offsets replace them.
  constexpr int RBN = (WS == 8) ? 1 : 4;
  float rb[RBN][2];
  const float* tb_r[4];
  {
    const int n = (16 * wave + lr) % N, iq = n / WS, jq = n % WS;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int krow = (WS == 8 ? 48 : 16 * wave) + 4 * kq + r;
      const int nk = krow % N, ik = nk / WS, jk = nk % WS;
      const int idx = ((iq - ik + WS - 1) * (2 * WS - 1) + (jq - jk + WS - 1)) * 2;
      tb_r[r] = tbl + idx;
      if (WS != 8) {
        rb[r % RBN][0] = tbl[idx];
        rb[r % RBN][1] = tbl[idx + 1];
        if (WS == 2 && kq != (lr >> 2)) { rb[r % RBN][0] = -INFINITY; rb[r % RBN][1] = -INFINITY; }   // another 2x2 window of the tile
      }
    }
  }

  for (int i = first; i < last; ++i) {
    const int b = xcd + 8 * (i >> a.lgS), t = ((i & (S - 1)) << 6) + 16 * wave + lr;
    float* Ks = KVs + (i & 1) * 2 * 64 * LDK;
    float* Vs = Ks + 64 * LDK;
    int* reg_s = reg_all + (i & 1) * 64;
    if (shift > 0 && kq == 0) {
      int hr_, wc_;
      (void)source_row<WS>(t, H, W, a.lgW, shift, hr_, wc_);
      const int rh = hr_ < H - WS ? 0 : (hr_ < H - shift ? 1 : 2), rw = wc_ < W - WS ? 0 : (wc_ < W - shift ? 1 : 2);
      reg_s[16 * wave + lr] = 3 * rh + rw;
    }
    FA_STAMP(i - first, 0);
    float mq = 0.f, rq = 1.f, mk = 0.f, rk = 1.f;
    if (!(FA_SKIP & 1)) {
      row_stats(xq, a.eps, mq, rq);
      row_stats(xkv, a.eps, mk, rk);
    }
    const float rqs = rq * QSCALE, nmq = -mq * rq, nmk = -mk * rk;     // (b' and rowsum(W') of the q rows are pre-scaled)

    FA_STAMP(i - first, 1);
    // ---- projections: six independent accumulator tiles (q, k, v x 2 heads) interleaved over the 24 k-steps
    f32x4 qa[2], ka[2], va[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) { qa[h] = (f32x4){0.f, 0.f, 0.f, 0.f}; ka[h] = qa[h]; va[h] = qa[h]; }
    if (FA_SKIP & 2) { qa[0] = xq[0] + xq[2]; qa[1] = xq[1] + xq[3]; ka[0] = xkv[0] + xq[4]; ka[1] = xkv[1] + xq[5]; va[0] = xkv[2] + xkv[4]; va[1] = xkv[3] + xkv[5]; }
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
    FA_STAMP(i - first, 2);
#if FA_SCHED
    // issue order of the block above: the row statistics (vector ALU, independent of the MFMAs: they eat RAW rows) are
    // dealt into the shadows of the 144 projection MFMAs, two vector instructions behind each
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
      for (int m = 0; m < 24; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, FA_SCHED, 0);
      }
    }
#endif
    // y = rstd * acc - (rstd * mean) * rowsum(W') + b'   (two fma per value; the q rows of b' / rowsum(W') / rstd carry
    // head_dim ** -0.5 (pgrm.py:230-231) times log2(e): the softmax below is exp2(s - max)).  v, k, q in turn: short live ranges.
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
      if (WS == 8) *reinterpret_cast<f32x4*>(Ks + (16 * wave + lr) * LDK + f) = ka[h];
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int f = 16 * h + 4 * kq;
      const f32x4 cq = *reinterpret_cast<const f32x4*>(pbias + FC + f), bq4 = *reinterpret_cast<const f32x4*>(pbias + f);
#pragma unroll
      for (int e = 0; e < 4; ++e) qa[h][e] = fmaf(qa[h][e], rqs, fmaf(nmq, cq[e], bq4[e]));
    }
    if (TRAIN && a.q_out) {
      // training forward: q / k / v of this (token, group) go to HBM in raster token order, exactly the tensors the unfused
      // q / kv Linear layers produce (pgrm.py:188,194) -- the backward kernels read them; q without the folded softmax scale
      int hr_, wc_;
      const size_t src = (size_t)b * L + source_row<WS>(t, H, W, a.lgW, shift, hr_, wc_);
      constexpr float UNSCALE = 1.0f / QSCALE;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int f = FCG * g + 16 * h + 4 * kq;
        *reinterpret_cast<f32x4*>(a.q_out + src * FC + f) = qa[h] * UNSCALE;
        *reinterpret_cast<f32x4*>(a.kv_out + src * (2 * FC) + f) = ka[h];
        *reinterpret_cast<f32x4*>(a.kv_out + src * (2 * FC) + FC + f) = va[h];
      }
    }
    // ---- the row registers are dead: send for unit i+1 now, the loads fly during this unit's attention (and the partner
    // wave's projection).  The index is clamped, not predicated: a load inside a branch makes hipcc drain vmcnt(0) at the join.
    if (!(FA_SKIP & 16)) load_rows<WS>(a, xcd, i + 1 < last ? i + 1 : i, shift, wave, lr, kq, xq, xkv);
    FA_STAMP(i - first, 3);
    if (!(FA_SKIP & 8)) __syncthreads();
    FA_STAMP(i - first, 4);
    if (FA_SKIP & 4) {
      float* dst = a.out + ((size_t)b * L + t) * FC + FCG * g + 4 * kq;
      *reinterpret_cast<f32x4*>(dst) = qa[0] + ka[0] + va[0];
      *reinterpret_cast<f32x4*>(dst + 16) = qa[1] + ka[1] + va[1];
      continue;
    }

    unsigned masked = 0u;                  // bit (4 kt + r): key in another shift-mask region than my query (pgrm.py:240-243)
    if (shift > 0) {
      const int my_reg = reg_s[16 * wave + lr];
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          masked |= (reg_s[(WS == 8 ? 16 * kt : 16 * wave) + 4 * kq + r] != my_reg ? 1u : 0u) << (4 * kt + r);
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      f32x4 sacc[KT];
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        f32x4 kf = ka[h];
        if (WS == 8) kf = *reinterpret_cast<const f32x4*>(Ks + (16 * kt + lr) * LDK + 16 * h + 4 * kq);
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = mfma16(kf[s], qa[h][s], acc);
        sacc[kt] = acc;
      }
      // + relative position bias, shift mask; softmax over the keys of query column lr
      float mx = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = sacc[kt][r] + (WS == 8 ? tb_r[r][60 * (3 - kt) + h] : rb[r % RBN][h]);
          if ((masked >> (4 * kt + r)) & 1u) v += -100.0f * LOG2E;
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
      if (TRAIN) {
        // attn_drop on the probabilities (the denominator is over the undropped row); element index of (b, g, h, query t, key m)
        // as documented for dpmn_window_attn_f32: ((((b G + g) heads + h) L + t) N + m)
        if (a.p_drop > 0.f) {
          const unsigned long long e0 = ((((unsigned long long)b * 3 + g) * 2 + h) * L + t) * N;
          const unsigned long long z0 = drop_z0(a.seed, e0 + (WS == 2 ? 0 : 4 * kq));      // + (16 kt + r) * PHI: constant adds
#pragma unroll
          for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              sacc[kt][r] *= drop_scale_z(z0 + (unsigned long long)((WS == 8 ? 16 * kt : 0) + r) * DROP_PHI, a.p_drop, a.inv_keep);
        }
      }
      // O^T = V^T . P: two accumulator chains
      f32x4 o0 = (f32x4){0.f, 0.f, 0.f, 0.f}, o1 = o0;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int s = 0; s < 4; s += 2) {
          const int krow = (WS == 8 ? 16 * kt : 16 * wave) + 4 * kq + s;
          o0 = mfma16(Vs[krow * LDK + 16 * h + lr], sacc[kt][s], o0);
          o1 = mfma16(Vs[(krow + 1) * LDK + 16 * h + lr], sacc[kt][s + 1], o1);
        }
      o0 += o1;
      float* dst = a.out + ((size_t)b * L + t) * FC + FCG * g + 16 * h + 4 * kq;
      *reinterpret_cast<float4*>(dst) = make_float4(o0[0] * inv, o0[1] * inv, o0[2] * inv, o0[3] * inv);
      FA_STAMP(i - first, 5 + h);
    }
  }
  FA_STAMP(8, 6 + (slot == 2 ? 1 : 0));
}


// Persistent: 2 blocks per CU, each walks a contiguous, cost-balanced range of the unit list of "its" XCD (blocks are
// dealt to XCDs round-robin -- observed placement, used for speed only): images b with b % 8 == xcd, ordered slot-major, so
//   * the three groups' gathers of an image's rows (each row is needed once per group) meet in one XCD's L2,
//   * the group's weight slice is staged once per block and slot (<= 3 times), not once per unit,
//   * the next unit's rows are loaded while the current unit computes, and workgroup dispatch cost is paid 512 times, not 2304.
template <bool TRAIN>
__global__ __launch_bounds__(256, 2) void k_ln_qkv_window_attn(FusedAttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, nbx = gridDim.x >> 3;
  const int S = 1 << a.lgS;
  const int ni = xcd < a.B ? (a.B - xcd + 7) / 8 : 0;
  const int per = ni * S;                  // units per slot on this XCD
  if (per == 0) return;
  if (FA_SKIP & 64) return;
  const int* nb = a.nblk[ni == (a.B + 7) / 8 ? 0 : 1];
  if (nb[0] > 0) {
    // one slot per block: a block never changes the staged weights, and the units of a slot are dealt evenly (+-1) to its
    // blocks; the block counts per slot come from the host (minimum of the predicted finish time of the slowest block)
    const int slot = j < nb[0] ? 0 : (j < nb[0] + nb[1] ? 1 : 2);
    const int jj = j - (slot == 0 ? 0 : (slot == 1 ? nb[0] : nb[0] + nb[1])), ns = nb[slot];
    const int lo = (int)((long)per * jj / ns), hi = (int)((long)per * (jj + 1) / ns);
    if (lo >= hi) return;
    const int ws = a.ws[slot];
    if (ws == 8) run_units<8, TRAIN>(a, slot, xcd, lo, hi, smem);
    else if (ws == 4) run_units<4, TRAIN>(a, slot, xcd, lo, hi, smem);
    else run_units<2, TRAIN>(a, slot, xcd, lo, hi, smem);
    return;
  }
  const int cs[3] = {a.cost[0], a.cost[1], a.cost[2]};
  const long ctot = (long)per * (cs[0] + cs[1] + cs[2]);
  const int u0 = units_before(ctot * j / nbx, per, cs);
  const int u1 = j + 1 == nbx ? 3 * per : units_before(ctot * (j + 1) / nbx, per, cs);
  for (int slot = 0; slot < 3; ++slot) {
    const int lo = u0 > slot * per ? u0 - slot * per : 0;
    const int hi = (u1 < (slot + 1) * per ? u1 : (slot + 1) * per) - slot * per;
    if (lo >= hi) continue;                // block-uniform
    const int ws = a.ws[slot];
    if (ws == 8) run_units<8, TRAIN>(a, slot, xcd, lo, hi, smem);
    else if (ws == 4) run_units<4, TRAIN>(a, slot, xcd, lo, hi, smem);
    else run_units<2, TRAIN>(a, slot, xcd, lo, hi, smem);
  }
}

}  // namespace

extern "C" {

int dpmn_ln_qkv_window_attn_supported(int C, int n_groups, int heads_per_group, const int* windows, int H, int W) {
  if (C != FC || n_groups != 3 || heads_per_group != 2 || !windows || (H * W) % 64 != 0) return 0;
  if ((H & (H - 1)) || (W & (W - 1)) || H < 8 || W < 8) return 0;      // index math uses shifts / masks (16x64 and 32x128 token grids)
  static const int off = getenv("DPMN_ATTN_FUSED") && atoi(getenv("DPMN_ATTN_FUSED")) == 0;
  if (off) return 0;
  for (int g = 0; g < 3; ++g) {
    const int ws = windows[g];
    if (!(ws == 2 || ws == 4 || ws == 8) || H % ws || W % ws) return 0;
  }
  return 1;
}

}  // extern "C" (reopened below)

namespace dpmn_fa {

void fa_fold(const FusedAttnArgs& a, hipStream_t st) { hipLaunchKernelGGL(k_attn_fold, dim3(FC + 1, 3), dim3(64), 0, st, a); }

int fa_prepare(FusedAttnArgs& a, const float* tq, const float* tkv, const float* lnq_w, const float* lnq_b, const float* lnkv_w,
               const float* lnkv_b, float eps, const float* wq, const float* bq, const float* wkv, const float* bkv,
               const float* const* bias_tables, const int* windows, const int* shifts, int n_groups, int heads_per_group, int B, int H,
               int W, int C, void* workspace, const int* cost_ws, int blocks_per_cu, long* blocks_out) {
  DPMN_REQUIRE(tq && tkv && lnq_w && lnq_b && lnkv_w && lnkv_b && wq && bq && wkv && bkv && bias_tables && windows && shifts,
               "ln_qkv_window_attn: null pointer");
  DPMN_REQUIRE(B > 0, "ln_qkv_window_attn: empty batch");
  DPMN_REQUIRE(!(H & (H - 1)) && !(W & (W - 1)) && H >= 8 && W >= 8, "ln_qkv_window_attn: token grid sides must be powers of two >= 8");
  DPMN_REQUIRE(C == FC && n_groups == 3 && heads_per_group == 2 && (H * W) % 64 == 0,
               "ln_qkv_window_attn: built for dim 96 = 3 groups x 2 heads x 16 (config 1/2/3); other shapes use the unfused kernels");
  a.tq = tq; a.tkv = tkv; a.lnq_w = lnq_w; a.lnq_b = lnq_b; a.lnkv_w = lnkv_w; a.lnkv_b = lnkv_b;
  a.wq = wq; a.bq = bq; a.wkv = wkv; a.bkv = bkv; a.B = B; a.H = H; a.W = W; a.eps = eps;
  for (a.lgW = 0; (1 << a.lgW) < W; ++a.lgW) {}
  for (a.lgS = 0; (64 << a.lgS) < H * W; ++a.lgS) {}
  int order[3] = {0, 1, 2};      // largest windows first: their units carry the most MFMA work, the short ones fill the tail
  for (int i = 0; i < 3; ++i)
    for (int j = i + 1; j < 3; ++j)
      if (windows[order[j]] > windows[order[i]]) { const int t_ = order[i]; order[i] = order[j]; order[j] = t_; }
  for (int s = 0; s < 3; ++s) {
    const int g = order[s], ws = windows[g];
    DPMN_REQUIRE((ws == 2 || ws == 4 || ws == 8) && H % ws == 0 && W % ws == 0 && shifts[g] >= 0 && shifts[g] < ws,
                 "ln_qkv_window_attn: windows must be 2, 4 or 8 and divide the token grid (padding path of pgrm.py:200-207 not built)");
    DPMN_REQUIRE(bias_tables[g], "ln_qkv_window_attn: null bias table");
    a.gid[s] = g; a.ws[s] = ws; a.shift[s] = shifts[g]; a.table[s] = bias_tables[g];
    a.cost[s] = ws == 8 ? cost_ws[0] : (ws == 4 ? cost_ws[1] : cost_ws[2]);
  }
  DPMN_REQUIRE(workspace && ((uintptr_t)workspace & 15) == 0, "ln_qkv_window_attn: workspace (dpmn_ln_qkv_window_attn_workspace_bytes, 16-byte aligned) missing");
  a.folded = static_cast<float*>(workspace);
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return dpmn_set_error(DPMN_ERR_LAUNCH, "ln_qkv_window_attn: device query failed");
    n_cu = prop.multiProcessorCount > 8 ? prop.multiProcessorCount / 8 * 8 : 8;
  }
  const long slabs = (long)B * (H * W / 64);
  long blocks = (long)blocks_per_cu * n_cu;          // resident blocks per CU x CUs; a multiple of 8
  const long need = ((3 * slabs + 7) / 8) * 8;
  if (blocks > need) blocks = need;
  {
    const int nbx = (int)(blocks / 8), S = H * W / 64, PRE = 50;
    for (int w = 0; w < 2; ++w) {
      const int ni = w == 0 ? (B + 7) / 8 : B / 8, per = ni * S;
      a.nblk[w][0] = 0;
      if (per == 0 || nbx < 3) continue;
      long best = -1;
      for (int n0 = 1; n0 <= nbx - 2; ++n0)
        for (int n1 = 1; n0 + n1 <= nbx - 1; ++n1) {
          const int n[3] = {n0, n1, nbx - n0 - n1};
          long worst = 0, sum = 0;
          for (int s_ = 0; s_ < 3; ++s_) {
            const long t_ = PRE + (long)((per + n[s_] - 1) / n[s_]) * a.cost[s_];
            worst = t_ > worst ? t_ : worst;
            sum += t_ * n[s_];
          }
          const long key = worst * 1000000 + sum / nbx;
          if (best < 0 || key < best) { best = key; a.nblk[w][0] = n0; a.nblk[w][1] = n1; a.nblk[w][2] = n[2]; }
        }
    }
    static const int contiguous = getenv("DPMN_FA_CONTIG") ? atoi(getenv("DPMN_FA_CONTIG")) : 0;
    if (contiguous) a.nblk[0][0] = a.nblk[1][0] = 0;
  }
  *blocks_out = blocks;
  return DPMN_OK;
}

}  // namespace dpmn_fa

static int fused_attn_launch(const float* tq, const float* tkv, const float* lnq_w, const float* lnq_b, const float* lnkv_w,
                             const float* lnkv_b, float eps, const float* wq, const float* bq, const float* wkv,
                             const float* bkv, const float* const* bias_tables, const int* windows, const int* shifts,
                             int n_groups, int heads_per_group, float* out, int B, int H, int W, int C, dpmn_stream_t stream,
                             void* workspace, int refold, bool train, float* q_out, float* kv_out, float p_drop, unsigned long long seed) {
  DPMN_REQUIRE(out, "ln_qkv_window_attn: null pointer");
  DPMN_REQUIRE(!train || (((q_out != nullptr) == (kv_out != nullptr)) && p_drop >= 0.f && p_drop < 1.f),
               "ln_qkv_window_attn_train: q_out / kv_out must both be given or both be null (the recomputing backward needs neither), p_drop in [0, 1)");
  FusedAttnArgs a{};
  const int cost_ws[3] = {183, 151, 146};      // cycles / 100 per unit at B = 48 (tools/fa_timeline.py, round 3)
  static const int bpc = getenv("DPMN_FA_BPC") ? atoi(getenv("DPMN_FA_BPC")) : 2;      // 2 resident blocks per CU (78 KB of LDS each)
  long blocks = 0;
  const int rc = fa_prepare(a, tq, tkv, lnq_w, lnq_b, lnkv_w, lnkv_b, eps, wq, bq, wkv, bkv, bias_tables, windows, shifts, n_groups,
                            heads_per_group, B, H, W, C, workspace, cost_ws, bpc, &blocks);
  if (rc != DPMN_OK) return rc;
  a.q_out = q_out; a.kv_out = kv_out; a.p_drop = p_drop; a.inv_keep = train ? 1.0f / (1.0f - p_drop) : 1.0f; a.seed = seed;
  a.out = out;
  const size_t smem = (size_t)(FOLD_STRIDE + 4 * 64 * LDK + 2 * 64) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ln_qkv_window_attn<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ln_qkv_window_attn<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_set = true;
  }
  const double tokens = (double)B * H * W;
  double attn = 0.0;
  for (int g = 0; g < 3; ++g) attn += 4.0 * windows[g] * windows[g] * FD * 2 * tokens;
  hipStream_t st = as_stream(stream);
  if (refold) fa_fold(a, st);
  ProfScope prof(PT_ATTN_FUSED, st, 2.0 * tokens * FC * (3 * FC) + attn, 4.0 * (3.0 * tokens * FC + 3.0 * FC * FC));
  if (train) hipLaunchKernelGGL(k_ln_qkv_window_attn<true>, dim3((unsigned)blocks), dim3(256), smem, st, a);
  else hipLaunchKernelGGL(k_ln_qkv_window_attn<false>, dim3((unsigned)blocks), dim3(256), smem, st, a);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

extern "C" {

int dpmn_ln_qkv_window_attn_f32(const float* tq, const float* tkv, const float* lnq_w, const float* lnq_b, const float* lnkv_w,
                                const float* lnkv_b, float eps, const float* wq, const float* bq, const float* wkv,
                                const float* bkv, const float* const* bias_tables, const int* windows, const int* shifts,
                                int n_groups, int heads_per_group, float* out, void* workspace, int refold, int B, int H, int W,
                                int C, dpmn_stream_t stream) {
  return fused_attn_launch(tq, tkv, lnq_w, lnq_b, lnkv_w, lnkv_b, eps, wq, bq, wkv, bkv, bias_tables, windows, shifts, n_groups,
                           heads_per_group, out, B, H, W, C, stream, workspace, refold, false, nullptr, nullptr, 0.f, 0ull);
}

size_t dpmn_ln_qkv_window_attn_workspace_bytes(void) { return sizeof(float) * 3 * FOLD_STRIDE; }

int dpmn_ln_qkv_window_attn_train_f32(const float* tq, const float* tkv, const float* lnq_w, const float* lnq_b, const float* lnkv_w,
                                      const float* lnkv_b, float eps, const float* wq, const float* bq, const float* wkv,
                                      const float* bkv, const float* const* bias_tables, const int* windows, const int* shifts,
                                      int n_groups, int heads_per_group, float* out, float* q_out, float* kv_out, float p_drop,
                                      unsigned long long seed, void* workspace, int B, int H, int W, int C, dpmn_stream_t stream) {
  return fused_attn_launch(tq, tkv, lnq_w, lnq_b, lnkv_w, lnkv_b, eps, wq, bq, wkv, bkv, bias_tables, windows, shifts, n_groups,
                           heads_per_group, out, B, H, W, C, stream, workspace, 1, true, q_out, kv_out, p_drop, seed);
}

#if FA_TIMING
int dpmn_fa_timing_dump(unsigned long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_fa_t), sizeof(unsigned long long) * 512 * 9 * 8);
}
int dpmn_fa_timing_clear() {
  static unsigned long long z[512 * 9 * 8];
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_fa_t), z, sizeof(z));
}
#endif

}  // extern "C"
