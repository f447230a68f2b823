// fp32 MFMA GEMM family for the DPMN token path (PGRM linears, TATT linears).
//
//   y[m][n] = epi( pro(x)[m][:] . w[n][:] + bias[n] )          x: (M,K) row-major, w: (N,K) row-major
//
// Layout / mapping (gfx950, wave64, v_mfma_f32_16x16x4_f32):
//   * the MFMA "A" operand carries W rows (output features) and the "B" operand carries X rows
//     (tokens), so a lane ends up with 4 CONSECUTIVE output features of one token -> one 16-byte
//     store per 16x16 tile and float4 bias / residual loads in the epilogue;
//   * operands come from LDS tiles with K contiguous and a +4 float row pad; one ds_read_b128
//     feeds 4 MFMA k-steps (lane-quad kq at step s consumes k = 4*kq + s of a 16-deep chunk --
//     a k permutation applied identically to both operands);
//   * whole-K kernels (K <= 192) keep the full X tile resident so LayerNorm (two-pass, like
//     at::layer_norm) or the SK select runs as a prologue on the tile already in LDS;
//   * the k-loop kernel double-buffers 32-deep chunks through registers.
// Reference call sites replaced: pgrm.py:188,194 (q/kv Linear after norm1_q/norm1_kv 322-323),
// pgrm.py:82 (SKConv.proj), 92-95 (select + proj_head + residual), 30-31 (fc1+GELU after norm2 330),
// 39 (fc2), tatt.py:209 / transformer_v2.py linears.
#include "gemm_body.h"

namespace {

// ---------------------------------------------------------------------------------- whole-K, W-stationary
// Persistent blocks: the (BN=96 x K) weight tile is loaded into LDS ONCE per block; the block then walks over
// BM=32-token tiles.  Tile i+1 is fetched into registers while tile i runs on the MFMA pipe; the prologue
// transform (LayerNorm two-pass in registers across the 8 threads of a row / SK select / add) is applied on
// the way from registers to the LDS tile.  Two barriers per tile, 3 blocks per CU (51 KB LDS at K=96)
// so one block's VALU prologue/epilogue overlaps the other's MFMAs.
// Block: 256 threads; waves 2(m: 16 tokens) x 2(n: 48 features): acc[3 n-tiles][1 m-tile].
constexpr int WS_BM = 32, WS_BN = 96;
#ifndef WSTAT_NBUF
#define WSTAT_NBUF 1   // single X buffer + second barrier: 51 KB LDS at K = 96 -> 3 blocks per CU (measured 5-8 % faster than 2 x 64 KB)
#endif

// EPI (with FULL): 0 = the generic epilogue above (every option a run-time branch); 1 = (bias), 2 = (bias) + GELU, 3 / 5 = (bias) + two / one residuals, 4 = (bias) + GELU column sums as
// straight-line code: three unconditional float4 stores per lane.  With no branch between a tile's loads, its stores and
// the next tile's loads, hipcc counts its vmcnt waits instead of draining to 0 at the top of every tile -- on gfx9 stores
// count on vmcnt too, so the drain also waited for the previous tile's stores to reach memory.
template <int EPI>
__device__ __forceinline__ void epilogue_fast(f32x4 (&acc)[3][1], int m0, int n0, int ldy, float* y, const float4 (&bias4)[3],
                                              const float* res1, const float* res2, float* red, int bn_cols, int n_block0) {
  const int lane = threadIdx.x & 63;
  const int lm = lane & 15, lq = lane >> 4;
  const size_t off = (size_t)(m0 + lm) * ldy + n0 + lq * 4;
  float4 r1[3], r2[3];
  if (EPI == 3 || EPI == 5) {         // 3: both residuals (SwinTransformerBlock shortcut + SKConv feats, pgrm.py:96,329); 5: one
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) {
      r1[nt] = *reinterpret_cast<const float4*>(res1 + off + nt * 16);
      if (EPI == 3) r2[nt] = *reinterpret_cast<const float4*>(res2 + off + nt * 16);
    }
  }
#pragma unroll
  for (int nt = 0; nt < 3; ++nt) {
    const float4 b4 = bias4[nt];      // the lane's 12 bias values, loaded once per block
    float v[4] = {acc[nt][0][0] + b4.x, acc[nt][0][1] + b4.y, acc[nt][0][2] + b4.z, acc[nt][0][3] + b4.w};
    if (EPI == 2) {
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
    }
    if (EPI == 3 || EPI == 5) { v[0] += r1[nt].x; v[1] += r1[nt].y; v[2] += r1[nt].z; v[3] += r1[nt].w; }
    if (EPI == 3) { v[0] += r2[nt].x; v[1] += r2[nt].y; v[2] += r2[nt].z; v[3] += r2[nt].w; }
    *reinterpret_cast<float4*>(y + off + nt * 16) = make_float4(v[0], v[1], v[2], v[3]);
    if (EPI == 4) {                   // SKConv global-average-pool partials: column sums of GELU(y) over the tile's rows
      const int wave = threadIdx.x >> 6;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float c = gelu_erf(v[r]);
        c += xshfl<1>(c); c += xshfl<2>(c); c += xshfl<4>(c); c += xshfl<8>(c);
        if (lm == 0) red[wave * bn_cols + (n0 - n_block0) + nt * 16 + lq * 4 + r] = c;
      }
    }
  }
}

template <int K, int PRO, int TH, bool FULL = false, int EPI = 0>   // TH threads: BM = TH/8 token rows per tile (32 or 64), waves (BM/16)(m) x 2(n)
__global__ __launch_bounds__(TH) void k_gemm_wstat(const float* __restrict__ x, int ldx, const float* __restrict__ w,
                                                    float* __restrict__ y, int ldy, int M, int N, ProArgs p, EpiArgs e) {
  constexpr int BM = TH / 8, BN = WS_BN, LDK = K + PAD;
  constexpr int WMN = BM / 16;
  constexpr int VPT = K / 32;                       // float4 per thread per tile (8 threads per row)
  constexpr int NRAW = (PRO == PRO_SKSEL) ? 4 : (PRO == PRO_ADD ? 2 : 1);
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ws = smem;                       // [BN][LDK]
  float* Xs = Ws + BN * LDK;              // [2][BM][LDK]
  float* red = Xs + WSTAT_NBUF * BM * LDK;   // [4][BN] colsum scratch
  float* lng = red + (TH / 64) * BN;      // [K] LayerNorm gamma, [K] beta   (red: one row of BN column sums per wave)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n_blk = blockIdx.y * BN;
  const int tiles = (M + BM - 1) / BM;

  constexpr int KV = K / 4;
  if constexpr (FULL) {
    // the W tile's float4s of a thread are loaded back to back (unconditional: every row is inside N) and stored afterwards,
    // instead of one load -> wait -> LDS store round trip per element
    constexpr int WL = (BN * KV + TH - 1) / TH;
    float4 wv[WL];
#pragma unroll
    for (int u = 0; u < WL; ++u) {
      const int i = min(tid + u * TH, BN * KV - 1);
      wv[u] = *reinterpret_cast<const float4*>(w + (size_t)(n_blk + i / KV) * K + (i % KV) * 4);
    }
#pragma unroll
    for (int u = 0; u < WL; ++u) {
      const int i = tid + u * TH;
      if (i < BN * KV) *reinterpret_cast<float4*>(Ws + (i / KV) * LDK + (i % KV) * 4) = wv[u];
    }
  } else {
    for (int i = tid; i < BN * KV; i += TH) {
      const int r = i / KV, c = (i % KV) * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (n_blk + r < N) v = *reinterpret_cast<const float4*>(w + (size_t)(n_blk + r) * K + c);
      *reinterpret_cast<float4*>(Ws + r * LDK + c) = v;
    }
  }
  if (PRO == PRO_LN)
    for (int i = tid; i < K; i += TH) { lng[i] = p.ln_w[i]; lng[K + i] = p.ln_b[i]; }

  const int srow = tid >> 3, spart = tid & 7;       // staging: row in tile, eighth of the row
  const int scol = spart * (K / 8);
  float4 rawA[NRAW][VPT], rawB[NRAW][VPT];   // two tiles in flight: HBM latency under load exceeds one tile of MFMA work

  auto issue = [&](float4 (&raw)[NRAW][VPT], int tile) {
    const int m = tile * BM + srow;
    const bool ok = FULL || m < M;
#pragma unroll
    for (int v = 0; v < VPT; ++v) {
      if (PRO == PRO_SKSEL) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
          raw[g][v] = (ok && g < p.groups) ? *reinterpret_cast<const float4*>(x + (size_t)m * ldx + g * K + scol + v * 4)
                                           : make_float4(0.f, 0.f, 0.f, 0.f);
      } else if (PRO == PRO_CAT2) {
        const int col = scol + v * 4;
        const float* src = col < p.k1 ? x + (size_t)m * p.k1 + col : p.x2 + (size_t)m * (K - p.k1) + (col - p.k1);
        raw[0][v] = ok ? *reinterpret_cast<const float4*>(src) : make_float4(0.f, 0.f, 0.f, 0.f);
      } else {
        raw[0][v] = ok ? *reinterpret_cast<const float4*>(x + (size_t)m * ldx + scol + v * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (PRO == PRO_ADD)
          raw[NRAW - 1][v] = ok ? *reinterpret_cast<const float4*>(p.addv + (size_t)m * ldx + scol + v * 4)
                                : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  };
  auto commit = [&](float4 (&raw)[NRAW][VPT], int tile, int buf) {
    float* dst = Xs + (size_t)(WSTAT_NBUF == 1 ? 0 : buf) * BM * LDK + srow * LDK + scol;
    float vals[VPT * 4];
    if (PRO == PRO_SKSEL) {
      const int m = tile * BM + srow;
      const int b = (m < M ? m : M - 1) / p.rows_per_image;
#pragma unroll
      for (int v = 0; v < VPT; ++v) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int g = 0; g < 4; ++g)
          if (g < p.groups) {
            const float4 a = *reinterpret_cast<const float4*>(p.sel + ((size_t)b * p.groups + g) * K + scol + v * 4);
            acc.x += a.x * raw[g][v].x; acc.y += a.y * raw[g][v].y; acc.z += a.z * raw[g][v].z; acc.w += a.w * raw[g][v].w;
          }
        vals[v * 4] = acc.x; vals[v * 4 + 1] = acc.y; vals[v * 4 + 2] = acc.z; vals[v * 4 + 3] = acc.w;
      }
    } else {
#pragma unroll
      for (int v = 0; v < VPT; ++v) {
        float4 t = raw[0][v];
        if (PRO == PRO_ADD) { const float4 a = raw[NRAW - 1][v]; t.x += a.x; t.y += a.y; t.z += a.z; t.w += a.w; }
        vals[v * 4] = t.x; vals[v * 4 + 1] = t.y; vals[v * 4 + 2] = t.z; vals[v * 4 + 3] = t.w;
      }
    }
    if (PRO == PRO_LN) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < VPT * 4; ++i) s += vals[i];
      s += xshfl<1>(s); s += xshfl<2>(s); s += xshfl<4>(s);
      const float mean = s * (1.0f / K);
      float q = 0.f;
#pragma unroll
      for (int i = 0; i < VPT * 4; ++i) { const float d = vals[i] - mean; q += d * d; }
      q += xshfl<1>(q); q += xshfl<2>(q); q += xshfl<4>(q);
      const float rstd = 1.0f / sqrtf(q * (1.0f / K) + p.eps);
#pragma unroll
      for (int i = 0; i < VPT * 4; ++i) vals[i] = (vals[i] - mean) * rstd * lng[scol + i] + lng[K + scol + i];
    }
#pragma unroll
    for (int v = 0; v < VPT; ++v)
      *reinterpret_cast<float4*>(dst + v * 4) = make_float4(vals[v * 4], vals[v * 4 + 1], vals[v * 4 + 2], vals[v * 4 + 3]);
  };

  const int wm = wave % WMN, wn = wave / WMN;
  const int lr = lane & 15, kq = lane >> 4;
  float4 bias4[3] = {};
  if constexpr (EPI != 0) {
#pragma unroll
    for (int nt = 0; nt < 3; ++nt)
      if (e.bias) bias4[nt] = *reinterpret_cast<const float4*>(e.bias + n_blk + wn * 48 + nt * 16 + kq * 4);
  }
  const int stride = gridDim.x;
  int tile = blockIdx.x;
  if (FULL) {                            // (grid.x <= tiles, so `tile` itself is valid)
    issue(rawA, tile);
    issue(rawB, min(tile + stride, tiles - 1));
  } else {
    if (tile < tiles) issue(rawA, tile);
    if (tile + stride < tiles) issue(rawB, tile + stride);
  }
  __syncthreads();                       // Ws / lng visible
  if (FULL) {
    commit(rawA, tile, 0);
    issue(rawA, min(tile + 2 * stride, tiles - 1));
  } else {
    if (tile < tiles) commit(rawA, tile, 0);
    if (tile + 2 * stride < tiles) issue(rawA, tile + 2 * stride);
  }
  int buf = 0;
  // one pipeline step: MFMA on Xs[buf] (tile), commit tile+stride from RAWN into Xs[buf^1], refill RAWN with tile+3*stride
// experiment hooks (tools/variants): -DWSTAT_NOMFMA runs one k-chunk only, -DWSTAT_NOEPI skips the epilogue
#ifdef WSTAT_NOMFMA
#define WSTAT_KLIM 16
#else
#define WSTAT_KLIM K
#endif
#ifdef WSTAT_NOEPI
#define WSTAT_EPI_GUARD if (acc[0][0][0] == 12345.678f)
#else
#define WSTAT_EPI_GUARD
#endif
#define WSTAT_STEP(RAWN)                                                                                     \
  {                                                                                                          \
    __syncthreads(); /* Xs[buf] committed by everyone; previous MFMA reads of Xs[buf^1] done */               \
    f32x4 acc[3][1];                                                                                         \
    _Pragma("unroll") for (int i = 0; i < 3; ++i) acc[i][0] = (f32x4){0.f, 0.f, 0.f, 0.f};                    \
    const float* xa = Xs + (size_t)(WSTAT_NBUF == 1 ? 0 : buf) * BM * LDK + (wm * 16 + lr) * LDK + kq * 4;     \
    const float* wa = Ws + (wn * 48 + lr) * LDK + kq * 4;                                                    \
    _Pragma("unroll") for (int kc = 0; kc < WSTAT_KLIM; kc += 16) {                                          \
      const f32x4 xf = *reinterpret_cast<const f32x4*>(xa + kc);                                             \
      f32x4 wf[3];                                                                                           \
      _Pragma("unroll") for (int i = 0; i < 3; ++i) wf[i] = *reinterpret_cast<const f32x4*>(wa + i * 16 * LDK + kc); \
      _Pragma("unroll") for (int s4 = 0; s4 < 4; ++s4)                                                       \
        _Pragma("unroll") for (int i = 0; i < 3; ++i) acc[i][0] = mfma16(wf[i][s4], xf[s4], acc[i][0]);       \
    }                                                                                                        \
    /* commit BEFORE the epilogue's stores: vmcnt retires in order, waiting for loads issued after stores   \
       would also wait for those stores */                                                                   \
    if (WSTAT_NBUF == 1) __syncthreads(); /* single X buffer: everyone is done reading it */                 \
    if constexpr (FULL) { /* unconditional (tile index clamped): a branch here costs the counted vmcnt waits */ \
      commit(RAWN, min(tile + stride, tiles - 1), buf ^ 1);                                                  \
      issue(RAWN, min(tile + 3 * stride, tiles - 1));                                                        \
    } else {                                                                                                 \
      if (tile + stride < tiles) commit(RAWN, tile + stride, buf ^ 1);                                       \
      if (tile + 3 * stride < tiles) issue(RAWN, tile + 3 * stride);                                         \
    }                                                                                                        \
    if constexpr (EPI != 0) {                                                                                \
      WSTAT_EPI_GUARD epilogue_fast<EPI>(acc, tile * BM + wm * 16, n_blk + wn * 48, ldy, y, bias4, e.res1, e.res2, red, BN, n_blk); \
    } else {                                                                                                 \
      WSTAT_EPI_GUARD epilogue<3, 1, FULL>(acc, tile * BM + wm * 16, n_blk + wn * 48, M, N, ldy, y, e, red, BN, n_blk); \
    }                                                                                                        \
    if ((EPI == 0 && e.colsum) || EPI == 4) {                                                                \
      __syncthreads();                                                                                       \
      /* one partial per 32 rows (the consumer's contract): BM/32 per tile, each summed over the 2 waves of its rows */ \
      for (int c = tid; c < BN * (BM / 32); c += TH) {                                                       \
        const int part = c / BN, col = c - part * BN, wn_c = col / 48;                                       \
        const float s_ = red[(wn_c * WMN + 2 * part) * BN + col] + red[(wn_c * WMN + 2 * part + 1) * BN + col]; \
        if (n_blk + col < N) e.colsum[((size_t)tile * (BM / 32) + part) * N + n_blk + col] = s_;             \
      }                                                                                                      \
    }                                                                                                        \
    buf ^= 1;                                                                                                \
    tile += stride;                                                                                          \
  }
  while (tile < tiles) {
    WSTAT_STEP(rawB)
    if (tile >= tiles) break;
    WSTAT_STEP(rawA)
  }
#undef WSTAT_STEP
}

// ---------------------------------------------------------------------------------- k-loop
// Block 256 threads; tile BM=64 x BN=96, BK=32, register-prefetch double buffering.
// Batched-K mode (kb_len > 0): the reduction axis is split into segments of kb_len that live in different images,
// element (row, k) sits at base + (k / kb_len) * batch_stride + row * kb_len + k % kb_len -- the pointwise-conv
// weight gradient dWp = sum_b dz_b . g_b^T over the raw (B, Ch, L) views (pgrm.py:37).
__global__ __launch_bounds__(256) void k_gemm_kloop(const float* __restrict__ x, int ldx, const float* __restrict__ w, int ldw,
                                                     float* __restrict__ y, int ldy, int M, int N, int K, EpiArgs e, int kb_len,
                                                     long x_bstride, long w_bstride) {
  constexpr int BM = 64, BN = 96, BK = 32, LDK = BK + PAD;
  __shared__ __attribute__((aligned(16))) float Xs[2][BM * LDK];
  __shared__ __attribute__((aligned(16))) float Ws[2][BN * LDK];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // Split-K launches (the pointwise-conv weight gradient: 6 x 4 tiles x 32 k splits): workgroups are dealt to the 8 XCDs round-robin
  // by linear id, so the 24 tiles of one k split -- which read the same 2 x 2.4 MB operand slices -- landed on 8 different L2s and
  // every slice crossed the fabric 4 / 6 times (750 MB per launch = the whole 185 us at HBM rate).  XCD c takes the splits
  // z = c, c + 8, ... and runs their tiles back to back.
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (gridDim.z > 1 && (gridDim.z & 7) == 0) {
    const int tiles = gridDim.x * gridDim.y;
    const int lid = bx + gridDim.x * (by + gridDim.y * bz);
    const int c = lid & 7, j = lid >> 3;
    const int zq = j / tiles, t = j - zq * tiles;
    bz = c + 8 * zq; by = t / (int)gridDim.x; bx = t - by * (int)gridDim.x;
  }
  const int m_blk = bx * BM, n_blk = by * BN;
  y += (size_t)bz * e.zstride;
  // loader mapping: 8 float4 per row
  const int lrow = tid >> 3, lcol = (tid & 7) * 4;   // rows 0..31 (+32 per pass)
  // Two named register sets (a, b), same scheme as k_gemm_pw: the loads of k-step kt+2 are issued at the end of step kt and
  // stored to LDS at the end of step kt+1, so they are in flight for a whole step of MFMA work and the waits are partial
  // vmcnt.  Every load is unconditional: rows past M / N are clamped to the last row (their accumulator rows are never
  // stored) and refills past the last k-step re-read it -- a predicated load is a branch, and hipcc drains vmcnt(0) at joins.
  float4 ax0, ax1, aw0, aw1, aw2, bx0, bx1, bw0, bw1, bw2;
  const int xm0 = min(m_blk + lrow, M - 1), xm1 = min(m_blk + lrow + 32, M - 1);
  const int wn0 = min(n_blk + lrow, N - 1), wn1 = min(n_blk + lrow + 32, N - 1), wn2 = min(n_blk + lrow + 64, N - 1);
#define KL_GLOAD(P, k0_)                                                                               \
  do {                                                                                                 \
    int kk = (k0_);                                                                                    \
    size_t xoff = 0, woff = 0;                                                                         \
    if (kb_len > 0) { const int kb = kk / kb_len; kk -= kb * kb_len; xoff = (size_t)kb * x_bstride; woff = (size_t)kb * w_bstride; } \
    P##x0 = *reinterpret_cast<const float4*>(x + xoff + (size_t)xm0 * ldx + kk + lcol);                \
    P##x1 = *reinterpret_cast<const float4*>(x + xoff + (size_t)xm1 * ldx + kk + lcol);                \
    P##w0 = *reinterpret_cast<const float4*>(w + woff + (size_t)wn0 * ldw + kk + lcol);                \
    P##w1 = *reinterpret_cast<const float4*>(w + woff + (size_t)wn1 * ldw + kk + lcol);                \
    P##w2 = *reinterpret_cast<const float4*>(w + woff + (size_t)wn2 * ldw + kk + lcol);                \
  } while (0)
#define KL_SSTORE(P, buf)                                                                              \
  do {                                                                                                 \
    *reinterpret_cast<float4*>(&Xs[buf][lrow * LDK + lcol]) = P##x0;                                   \
    *reinterpret_cast<float4*>(&Xs[buf][(lrow + 32) * LDK + lcol]) = P##x1;                            \
    *reinterpret_cast<float4*>(&Ws[buf][lrow * LDK + lcol]) = P##w0;                                   \
    *reinterpret_cast<float4*>(&Ws[buf][(lrow + 32) * LDK + lcol]) = P##w1;                            \
    *reinterpret_cast<float4*>(&Ws[buf][(lrow + 64) * LDK + lcol]) = P##w2;                            \
  } while (0)

  const int wm = wave & 1, wn = wave >> 1;
  const int lr = lane & 15, kq = lane >> 4;
  f32x4 acc[3][2];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#define KL_MMA(buf)                                                                                    \
  do {                                                                                                 \
    const float* xa = &Xs[buf][(wm * 32 + lr) * LDK + kq * 4];                                         \
    const float* wa = &Ws[buf][(wn * 48 + lr) * LDK + kq * 4];                                         \
    _Pragma("unroll") for (int kc = 0; kc < BK; kc += 16) {                                            \
      float4 xf[2], wf[3];                                                                             \
      _Pragma("unroll") for (int j = 0; j < 2; ++j) xf[j] = *reinterpret_cast<const float4*>(xa + j * 16 * LDK + kc); \
      _Pragma("unroll") for (int i = 0; i < 3; ++i) wf[i] = *reinterpret_cast<const float4*>(wa + i * 16 * LDK + kc); \
      _Pragma("unroll") for (int i = 0; i < 3; ++i)                                                    \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                \
          acc[i][j] = mfma16(wf[i].x, xf[j].x, acc[i][j]);                                             \
          acc[i][j] = mfma16(wf[i].y, xf[j].y, acc[i][j]);                                             \
          acc[i][j] = mfma16(wf[i].z, xf[j].z, acc[i][j]);                                             \
          acc[i][j] = mfma16(wf[i].w, xf[j].w, acc[i][j]);                                             \
        }                                                                                              \
    }                                                                                                  \
  } while (0)

  const int nk_all = K / BK;
  const int cps = (nk_all + gridDim.z - 1) / gridDim.z;      // K chunks per split (grid.z > 1 only with e.atomic)
  const int kt0 = bz * cps;
  const int nk = min(nk_all, kt0 + cps);
  if (kt0 >= nk) return;
  KL_GLOAD(a, kt0 * BK);
  KL_GLOAD(b, min(kt0 + 1, nk - 1) * BK);
  KL_SSTORE(a, 0);
  __syncthreads();
  for (int kt = kt0; kt < nk; kt += 2) {
    // even step: multiply k-step kt (buffer 0); set b holds kt+1; set a is refilled with kt+2
    KL_GLOAD(a, min(kt + 2, nk - 1) * BK);
    KL_MMA(0);
    KL_SSTORE(b, 1);
    __syncthreads();
    if (kt + 1 >= nk) break;          // odd number of k-steps (uniform)
    // odd step: multiply kt+1 (buffer 1); set a holds kt+2; set b is refilled with kt+3
    KL_GLOAD(b, min(kt + 3, nk - 1) * BK);
    KL_MMA(1);
    KL_SSTORE(a, 0);
    __syncthreads();
  }
#undef KL_GLOAD
#undef KL_SSTORE
#undef KL_MMA
  // interior tile, bias + one residual, no activation (Mlp.fc2 + shortcut, pgrm.py:39,330): the 6 residual loads are issued
  // together and the stores follow -- the generic epilogue below does load -> wait -> store per 16 x 16 tile
  if (e.bias && e.res1 && !e.res2 && !e.colsum && !e.atomic && e.act == ACT_NONE && m_blk + BM <= M && n_blk + BN <= N && (ldy & 3) == 0) {
    const int lm = lane & 15, lq = lane >> 4;
    float4 rr[3][2];
#pragma unroll
    for (int nt = 0; nt < 3; ++nt)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
        rr[nt][mt] = *reinterpret_cast<const float4*>(e.res1 + (size_t)(m_blk + wm * 32 + mt * 16 + lm) * ldy + n_blk + wn * 48 + nt * 16 + lq * 4);
    if (e.p_elem > 0.f || e.p_row > 0.f) {
      // the masks dpmn_dropout_f32 would apply to the stored Linear output (same element indices, same order of the two factors)
      const float ike = 1.0f / (1.0f - e.p_elem), ikr = 1.0f / (1.0f - e.p_row);
#pragma unroll
      for (int nt = 0; nt < 3; ++nt) {
        const float4 b4 = *reinterpret_cast<const float4*>(e.bias + n_blk + wn * 48 + nt * 16 + lq * 4);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          const size_t off = (size_t)(m_blk + wm * 32 + mt * 16 + lm) * ldy + n_blk + wn * 48 + nt * 16 + lq * 4;
          float o[4] = {acc[nt][mt][0] + b4.x, acc[nt][mt][1] + b4.y, acc[nt][mt][2] + b4.z, acc[nt][mt][3] + b4.w};
          if (e.p_elem > 0.f) {
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] *= drop_scale_z(drop_z0(e.seed_elem, (unsigned long long)off) + (unsigned long long)r * DROP_PHI, e.p_elem, ike);
          }
          if (e.p_row > 0.f) {
            const float mr = drop_scale(e.seed_row, (unsigned long long)(off / (size_t)e.row_len), e.p_row, ikr);
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] *= mr;
          }
          *reinterpret_cast<float4*>(y + off) = make_float4(o[0] + rr[nt][mt].x, o[1] + rr[nt][mt].y, o[2] + rr[nt][mt].z, o[3] + rr[nt][mt].w);
        }
      }
      return;
    }
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) {
      const float4 b4 = *reinterpret_cast<const float4*>(e.bias + n_blk + wn * 48 + nt * 16 + lq * 4);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
        *reinterpret_cast<float4*>(y + (size_t)(m_blk + wm * 32 + mt * 16 + lm) * ldy + n_blk + wn * 48 + nt * 16 + lq * 4) =
            make_float4(acc[nt][mt][0] + b4.x + rr[nt][mt].x, acc[nt][mt][1] + b4.y + rr[nt][mt].y,
                        acc[nt][mt][2] + b4.z + rr[nt][mt].z, acc[nt][mt][3] + b4.w + rr[nt][mt].w);
    }
    return;
  }
  epilogue<3, 2>(acc, m_blk + wm * 32, n_blk + wn * 48, M, N, ldy, y, e, nullptr, BN, n_blk);
}


// ---------------------------------------------------------------------------------- k-loop, 128 x 128 tiles (pointwise-conv weight gradient)
// dW (M, N) = sum_b X_b (M, L) . Y_b (N, L)^T over the raw (B, Ch, L) views (pgrm.py:37: X = dz, Y = g, M = N = Ch = 384, L = 1024,
// B = 48: 14.5 GFLOP, the largest single GEMM of the backward).  The 64 x 96 k-loop above ran it at 78 TFLOP/s; this is the same
// pipeline (two named register sets, loads of step kt + 2 in flight during step kt) on the implicit-GEMM conv's 128 x 128 tile
// (waves 2 x 2, 64 x 64 each: half the LDS operand reads per MFMA).  The reduction (b, s) is cut into `splits` contiguous ranges
// of 32-wide chunks (a range may cross image boundaries: the chunk -> (image, offset) decode is per chunk); split z STORES its
// tile at y + z * zstride, the caller adds the splits in order.  grid = (M / 128 * N / 128 * splits) workgroups, tile fastest within
// an XCD's share as in k_gemm_kloop (XCD c takes the splits c, c + 8, ...: the 9 tiles of a split read the same operand slices).
__global__ __launch_bounds__(256, 2) void k_gemm_kloop128(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y,
                                                           int M, int N, int L, int nchunks, int splits, long bstride, long zstride) {
  constexpr int BM = 128, BN = 128, BK = 32, LDK = BK + PAD;
  __shared__ __attribute__((aligned(16))) float Xs[2][BM * LDK];
  __shared__ __attribute__((aligned(16))) float Ws[2][BN * LDK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_m = M / BM, tiles = tiles_m * (N / BN);
  int lid = blockIdx.x, bz, t;
  if ((splits & 7) == 0) { const int c = lid & 7, j = lid >> 3; const int zq = j / tiles; t = j - zq * tiles; bz = c + 8 * zq; }
  else { bz = lid / tiles; t = lid - bz * tiles; }
  const int m_blk = (t % tiles_m) * BM, n_blk = (t / tiles_m) * BN;
  y += (size_t)bz * zstride;
  const int lrow = tid >> 3, lcol = (tid & 7) * 4;
  float4 ax0, ax1, ax2, ax3, aw0, aw1, aw2, aw3, bx0, bx1, bx2, bx3, bw0, bw1, bw2, bw3;
  const int cpi = L / BK;                                    // chunks per image
#define K8_GLOAD(P, kt_)                                                                                \
  do {                                                                                                  \
    const int kb = (kt_) / cpi, kk = ((kt_) - kb * cpi) * BK + lcol;                                     \
    const float* xb = x + (size_t)kb * bstride + (size_t)(m_blk + lrow) * L + kk;                        \
    const float* wb = w + (size_t)kb * bstride + (size_t)(n_blk + lrow) * L + kk;                        \
    P##x0 = *reinterpret_cast<const float4*>(xb);                                                        \
    P##x1 = *reinterpret_cast<const float4*>(xb + (size_t)32 * L);                                       \
    P##x2 = *reinterpret_cast<const float4*>(xb + (size_t)64 * L);                                       \
    P##x3 = *reinterpret_cast<const float4*>(xb + (size_t)96 * L);                                       \
    P##w0 = *reinterpret_cast<const float4*>(wb);                                                        \
    P##w1 = *reinterpret_cast<const float4*>(wb + (size_t)32 * L);                                       \
    P##w2 = *reinterpret_cast<const float4*>(wb + (size_t)64 * L);                                       \
    P##w3 = *reinterpret_cast<const float4*>(wb + (size_t)96 * L);                                       \
  } while (0)
#define K8_SSTORE(P, buf)                                                                               \
  do {                                                                                                  \
    *reinterpret_cast<float4*>(&Xs[buf][lrow * LDK + lcol]) = P##x0;                                    \
    *reinterpret_cast<float4*>(&Xs[buf][(lrow + 32) * LDK + lcol]) = P##x1;                             \
    *reinterpret_cast<float4*>(&Xs[buf][(lrow + 64) * LDK + lcol]) = P##x2;                             \
    *reinterpret_cast<float4*>(&Xs[buf][(lrow + 96) * LDK + lcol]) = P##x3;                             \
    *reinterpret_cast<float4*>(&Ws[buf][lrow * LDK + lcol]) = P##w0;                                    \
    *reinterpret_cast<float4*>(&Ws[buf][(lrow + 32) * LDK + lcol]) = P##w1;                             \
    *reinterpret_cast<float4*>(&Ws[buf][(lrow + 64) * LDK + lcol]) = P##w2;                             \
    *reinterpret_cast<float4*>(&Ws[buf][(lrow + 96) * LDK + lcol]) = P##w3;                             \
  } while (0)
  const int wm = wave & 1, wn = wave >> 1;
  const int lr = lane & 15, kq = lane >> 4;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#define K8_MMA(buf)                                                                                     \
  do {                                                                                                  \
    const float* xa = &Xs[buf][(wm * 64 + lr) * LDK + kq * 4];                                          \
    const float* wa = &Ws[buf][(wn * 64 + lr) * LDK + kq * 4];                                          \
    _Pragma("unroll") for (int kc = 0; kc < BK; kc += 16) {                                             \
      f32x4 xf[4], wf[4];                                                                               \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) xf[j] = *reinterpret_cast<const f32x4*>(xa + j * 16 * LDK + kc); \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) wf[i] = *reinterpret_cast<const f32x4*>(wa + i * 16 * LDK + kc); \
      _Pragma("unroll") for (int s4 = 0; s4 < 4; ++s4)                                                  \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                   \
          _Pragma("unroll") for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(wf[i][s4], xf[j][s4], acc[i][j]); \
    }                                                                                                   \
  } while (0)
  // split bz takes the chunks [kt0, nk): equal shares up to one chunk
  const int kt0 = (int)((long)nchunks * bz / splits), nk = (int)((long)nchunks * (bz + 1) / splits);
  if (kt0 < nk) {
    K8_GLOAD(a, kt0);
    K8_GLOAD(b, min(kt0 + 1, nk - 1));
    K8_SSTORE(a, 0);
    __syncthreads();
    for (int kt = kt0; kt < nk; kt += 2) {
      K8_GLOAD(a, min(kt + 2, nk - 1));
      K8_MMA(0);
      K8_SSTORE(b, 1);
      __syncthreads();
      if (kt + 1 >= nk) break;
      K8_GLOAD(b, min(kt + 3, nk - 1));
      K8_MMA(1);
      K8_SSTORE(a, 0);
      __syncthreads();
    }
  }
#undef K8_GLOAD
#undef K8_SSTORE
#undef K8_MMA
  // lane holds y[m = .. + lr][n = .. + 4 kq + r]
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i)
      *reinterpret_cast<f32x4*>(y + (size_t)(m_blk + wm * 64 + j * 16 + lr) * N + n_blk + wn * 64 + i * 16 + kq * 4) = acc[i][j];
}

// ---------------------------------------------------------------------------------- batched NN
// z[b][co][s] = sum_c w[co][c] * g[b][c][s] + bias[co]      (pointwise 1x1 conv, pgrm.py:37)
// g, z are the raw (B, Ch, L) views of token buffers (quirk Q2).  Output is s-contiguous, so the
// MFMA "A" operand carries s (from the [k][s] tile via ds_read_b32) and "B" carries co.
// Block 256 threads, tile 128 (s) x BC (co), BK = 16; waves 2(s) x 2(co): 64 x BC/2 each.  BC = 192 when Ch is a multiple
// of 192 (Ch = 384: 8 x 2 x 48 = 768 tiles = exactly 3 per CU at 3 resident blocks, and 96 MFMAs per barrier), else 128.
template <int BC>
__global__ __launch_bounds__(256, 2) void k_gemm_pw(const float* __restrict__ g, const float* __restrict__ w,
                                                     const float* __restrict__ bias, float* __restrict__ z, int Ch, int L) {
  constexpr int BS = 128, BK = 16, LDS_G = BS + 4, LDS_W = BK + PAD, NJ = BC / 32, WP = BC / 64;
  __shared__ __attribute__((aligned(16))) float Gs[2][BK * LDS_G];
  __shared__ __attribute__((aligned(16))) float Wsm[2][BC * LDS_W];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int s_blk = blockIdx.x * BS, c_blk = blockIdx.y * BC, b = blockIdx.z;
  const float* gb = g + (size_t)b * Ch * L;
  float* zb = z + (size_t)b * Ch * L;

  // loaders: G chunk = 16 rows(k) x 128 s = 512 float4 -> 2 per thread ; W chunk = BC rows x 16 k = 4*BC float4 -> WP per thread.
  // Two named register sets (a, b) hold the k-steps kt+1 and kt+2: a step's loads are issued two steps before they are
  // stored to LDS, so their latency hides behind two steps of MFMA work (tools/ubench/pw_steps.hip: 116.9 -> 121.5 TFLOP/s;
  // __launch_bounds__(256, 2) keeps the wave at 2 per SIMD with the accumulators in VGPRs).  Indexed arrays here end up in
  // scratch, hence the token-pasted names; w2 only exists for BC = 192.
  const int grow = tid >> 5, gcol = (tid & 31) * 4;   // rows 0..7 (+8)
  const int wrow = tid >> 2, wcol = (tid & 3) * 4;    // rows 0..63 (+64)
  float4 ag0, ag1, aw0, aw1, aw2, bg0, bg1, bw0, bw1, bw2;
#define PW_GLOAD(P, k0)                                                                                \
  do {                                                                                                 \
    P##g0 = *reinterpret_cast<const float4*>(gb + (size_t)((k0) + grow) * L + s_blk + gcol);           \
    P##g1 = *reinterpret_cast<const float4*>(gb + (size_t)((k0) + grow + 8) * L + s_blk + gcol);       \
    P##w0 = *reinterpret_cast<const float4*>(w + (size_t)(c_blk + wrow) * Ch + (k0) + wcol);           \
    P##w1 = *reinterpret_cast<const float4*>(w + (size_t)(c_blk + wrow + 64) * Ch + (k0) + wcol);      \
    if (WP == 3) P##w2 = *reinterpret_cast<const float4*>(w + (size_t)(c_blk + wrow + 128) * Ch + (k0) + wcol); \
  } while (0)
#define PW_SSTORE(P, buf)                                                                              \
  do {                                                                                                 \
    *reinterpret_cast<float4*>(&Gs[buf][grow * LDS_G + gcol]) = P##g0;                                 \
    *reinterpret_cast<float4*>(&Gs[buf][(grow + 8) * LDS_G + gcol]) = P##g1;                           \
    *reinterpret_cast<float4*>(&Wsm[buf][wrow * LDS_W + wcol]) = P##w0;                                \
    *reinterpret_cast<float4*>(&Wsm[buf][(wrow + 64) * LDS_W + wcol]) = P##w1;                         \
    if (WP == 3) *reinterpret_cast<float4*>(&Wsm[buf][(wrow + 128) * LDS_W + wcol]) = P##w2;           \
  } while (0)

  const int ws_ = wave & 1, wc_ = wave >> 1;
  const int lr = lane & 15, kq = lane >> 4;
  f32x4 acc[4][NJ];   // [s tile][co tile]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

#define PW_MMA(buf)                                                                                    \
  do {                                                                                                 \
    f32x4 wf[NJ];                                                                                      \
    _Pragma("unroll") for (int j = 0; j < NJ; ++j)                                                     \
      wf[j] = *reinterpret_cast<const f32x4*>(&Wsm[buf][(wc_ * (BC / 2) + j * 16 + lr) * LDS_W + kq * 4]); \
    const float* gp = &Gs[buf][(kq * 4) * LDS_G + ws_ * 64 + lr];                                      \
    _Pragma("unroll") for (int st = 0; st < 4; ++st) {                                                 \
      const float a0 = gp[st * LDS_G], a1 = gp[st * LDS_G + 16], a2 = gp[st * LDS_G + 32], a3 = gp[st * LDS_G + 48]; \
      _Pragma("unroll") for (int j = 0; j < NJ; ++j) {                                                 \
        const float bv = wf[j][st];                                                                    \
        acc[0][j] = mfma16(a0, bv, acc[0][j]);                                                         \
        acc[1][j] = mfma16(a1, bv, acc[1][j]);                                                         \
        acc[2][j] = mfma16(a2, bv, acc[2][j]);                                                         \
        acc[3][j] = mfma16(a3, bv, acc[3][j]);                                                         \
      }                                                                                                \
    }                                                                                                  \
  } while (0)

  const int nk = Ch / BK;      // even: Ch is a multiple of 128
  PW_GLOAD(a, 0);
  PW_GLOAD(b, BK);
  PW_SSTORE(a, 0);
  __syncthreads();
  for (int kt = 0; kt < nk; kt += 2) {
    // even step: multiply k-step kt (buffer 0); set b holds kt+1; set a is refilled with kt+2
    // (the refills past the end re-read the last k-step instead of being skipped: a conditional load makes hipcc drain
    // vmcnt(0) at the join, which would also wait for the loads issued a moment ago)
    PW_GLOAD(a, min(kt + 2, nk - 1) * BK);
    PW_MMA(0);
    PW_SSTORE(b, 1);
    __syncthreads();
    // odd step: multiply kt+1 (buffer 1); set a holds kt+2; set b is refilled with kt+3
    PW_GLOAD(b, min(kt + 3, nk - 1) * BK);
    PW_MMA(1);
    PW_SSTORE(a, 0);
    __syncthreads();
  }
#undef PW_GLOAD
#undef PW_SSTORE
#undef PW_MMA
  // lane holds z[co = c0 + j*16 + (l&15)][s = s0 + i*16 + (l>>4)*4 + r]
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int co = c_blk + wc_ * (BC / 2) + j * 16 + lr;
    const float bv = bias[co];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int s = s_blk + ws_ * 64 + i * 16 + kq * 4;
      *reinterpret_cast<float4*>(zb + (size_t)co * L + s) =
          make_float4(acc[i][j][0] + bv, acc[i][j][1] + bv, acc[i][j][2] + bv, acc[i][j][3] + bv);
    }
  }
}


// ---------------------------------------------------------------------------------- pointwise GEMM, bf16 operands
// dpmn_set_compute_dtype(1): the same 128 x BC tile on v_mfma_f32_16x16x32_bf16 (fp32 accumulation, fp32 tensors in HBM).  G is
// k-major in memory (rows = channels, s contiguous) and the MFMA wants 8 consecutive k per lane, so the chunk is staged as
// k-PAIRS: dword (p, s) = (bf16 G[2p][s], bf16 G[2p+1][s]) -- written with one ds_write_b128 per thread and pair (a thread owns 4
// consecutive s of two adjacent channels), read with four conflict-free ds_read_b32 per operand tile (pairs 4 kq .. 4 kq + 3 of
// column s).  W rows are k-contiguous: [BC][32 + 8] bf16, one ds_read_b128 per tile.  32-deep chunks, the two-set register
// prefetch of the fp32 kernel.
template <int BC>
__global__ __launch_bounds__(256, 2) void k_gemm_pw_bf16(const float* __restrict__ g, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ z, int Ch, int L) {
  constexpr int BS = 128, BK = 32, LDP = BS + 4, LDWB = BK + 8, NJ = BC / 32, WP = BC / 32;
  __shared__ __attribute__((aligned(16))) unsigned Gp[2][(BK / 2) * LDP];        // [16 pairs][128 s + pad] dwords
  __shared__ __attribute__((aligned(16))) unsigned short Wb[2][BC * LDWB];       // [BC][40] bf16
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int s_blk = blockIdx.x * BS, c_blk = blockIdx.y * BC, b = blockIdx.z;
  const float* gb = g + (size_t)b * Ch * L;
  float* zb = z + (size_t)b * Ch * L;
  const int gp_ = tid >> 5, gcol = (tid & 31) * 4;      // pair rows (2 gp_, 2 gp_ + 1) and (+16, +17)
  const int wrow = tid >> 3, wcol = (tid & 7) * 4;      // rows 0..31 (+32 per pass)
  float4 ag0, ag1, ag2, ag3, aw0, aw1, aw2, aw3, aw4, aw5, bg0, bg1, bg2, bg3, bw0, bw1, bw2, bw3, bw4, bw5;
#define PB_GLOAD(P, k0)                                                                                \
  do {                                                                                                 \
    P##g0 = *reinterpret_cast<const float4*>(gb + (size_t)((k0) + 2 * gp_) * L + s_blk + gcol);        \
    P##g1 = *reinterpret_cast<const float4*>(gb + (size_t)((k0) + 2 * gp_ + 1) * L + s_blk + gcol);    \
    P##g2 = *reinterpret_cast<const float4*>(gb + (size_t)((k0) + 2 * gp_ + 16) * L + s_blk + gcol);   \
    P##g3 = *reinterpret_cast<const float4*>(gb + (size_t)((k0) + 2 * gp_ + 17) * L + s_blk + gcol);   \
    P##w0 = *reinterpret_cast<const float4*>(w + (size_t)(c_blk + wrow) * Ch + (k0) + wcol);           \
    P##w1 = *reinterpret_cast<const float4*>(w + (size_t)(c_blk + wrow + 32) * Ch + (k0) + wcol);      \
    P##w2 = *reinterpret_cast<const float4*>(w + (size_t)(c_blk + wrow + 64) * Ch + (k0) + wcol);      \
    P##w3 = *reinterpret_cast<const float4*>(w + (size_t)(c_blk + wrow + 96) * Ch + (k0) + wcol);      \
    if (WP == 6) {                                                                                     \
      P##w4 = *reinterpret_cast<const float4*>(w + (size_t)(c_blk + wrow + 128) * Ch + (k0) + wcol);   \
      P##w5 = *reinterpret_cast<const float4*>(w + (size_t)(c_blk + wrow + 160) * Ch + (k0) + wcol);   \
    }                                                                                                  \
  } while (0)
#define PB_PAIR(lo, hi) __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2__){lo, hi}, bf16x2))
#define PB_SSTORE(P, buf)                                                                              \
  do {                                                                                                 \
    *reinterpret_cast<uint4*>(&Gp[buf][gp_ * LDP + gcol]) =                                            \
        make_uint4(PB_PAIR(P##g0.x, P##g1.x), PB_PAIR(P##g0.y, P##g1.y), PB_PAIR(P##g0.z, P##g1.z), PB_PAIR(P##g0.w, P##g1.w)); \
    *reinterpret_cast<uint4*>(&Gp[buf][(gp_ + 8) * LDP + gcol]) =                                      \
        make_uint4(PB_PAIR(P##g2.x, P##g3.x), PB_PAIR(P##g2.y, P##g3.y), PB_PAIR(P##g2.z, P##g3.z), PB_PAIR(P##g2.w, P##g3.w)); \
    *reinterpret_cast<uint2*>(&Wb[buf][wrow * LDWB + wcol]) = pack_bf16x4(P##w0.x, P##w0.y, P##w0.z, P##w0.w);          \
    *reinterpret_cast<uint2*>(&Wb[buf][(wrow + 32) * LDWB + wcol]) = pack_bf16x4(P##w1.x, P##w1.y, P##w1.z, P##w1.w);   \
    *reinterpret_cast<uint2*>(&Wb[buf][(wrow + 64) * LDWB + wcol]) = pack_bf16x4(P##w2.x, P##w2.y, P##w2.z, P##w2.w);   \
    *reinterpret_cast<uint2*>(&Wb[buf][(wrow + 96) * LDWB + wcol]) = pack_bf16x4(P##w3.x, P##w3.y, P##w3.z, P##w3.w);   \
    if (WP == 6) {                                                                                     \
      *reinterpret_cast<uint2*>(&Wb[buf][(wrow + 128) * LDWB + wcol]) = pack_bf16x4(P##w4.x, P##w4.y, P##w4.z, P##w4.w); \
      *reinterpret_cast<uint2*>(&Wb[buf][(wrow + 160) * LDWB + wcol]) = pack_bf16x4(P##w5.x, P##w5.y, P##w5.z, P##w5.w); \
    }                                                                                                  \
  } while (0)
  typedef float f32x2__ __attribute__((ext_vector_type(2)));
  typedef unsigned u32x4__ __attribute__((ext_vector_type(4)));
  const int ws_ = wave & 1, wc_ = wave >> 1;
  const int lr = lane & 15, kq = lane >> 4;
  f32x4 acc[4][NJ];   // [s tile][co tile]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#define PB_MMA(buf)                                                                                    \
  do {                                                                                                 \
    bf16x8 wf[NJ];                                                                                     \
    _Pragma("unroll") for (int j = 0; j < NJ; ++j)                                                     \
      wf[j] = *reinterpret_cast<const bf16x8*>(&Wb[buf][(wc_ * (BC / 2) + j * 16 + lr) * LDWB + kq * 8]); \
    const unsigned* gp = &Gp[buf][(kq * 4) * LDP + ws_ * 64 + lr];                                     \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                    \
      const u32x4__ av = {gp[i * 16], gp[LDP + i * 16], gp[2 * LDP + i * 16], gp[3 * LDP + i * 16]};   \
      const bf16x8 a8 = __builtin_bit_cast(bf16x8, av);                                                \
      _Pragma("unroll") for (int j = 0; j < NJ; ++j) acc[i][j] = mfma16_bf16(a8, wf[j], acc[i][j]);    \
    }                                                                                                  \
  } while (0)
  const int nk = Ch / BK;      // >= 2 (Ch is a multiple of 128)
  PB_GLOAD(a, 0);
  PB_GLOAD(b, BK);
  PB_SSTORE(a, 0);
  __syncthreads();
  for (int kt = 0; kt < nk; kt += 2) {
    PB_GLOAD(a, min(kt + 2, nk - 1) * BK);
    PB_MMA(0);
    PB_SSTORE(b, 1);
    __syncthreads();
    if (kt + 1 < nk) {
      PB_GLOAD(b, min(kt + 3, nk - 1) * BK);
      PB_MMA(1);
      PB_SSTORE(a, 0);
      __syncthreads();
    }
  }
#undef PB_GLOAD
#undef PB_SSTORE
#undef PB_PAIR
#undef PB_MMA
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int co = c_blk + wc_ * (BC / 2) + j * 16 + lr;
    const float bv = bias[co];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int s_ = s_blk + ws_ * 64 + i * 16 + kq * 4;
      *reinterpret_cast<float4*>(zb + (size_t)co * L + s_) =
          make_float4(acc[i][j][0] + bv, acc[i][j][1] + bv, acc[i][j][2] + bv, acc[i][j][3] + bv);
    }
  }
}

// ---------------------------------------------------------------------------------- whole-K, rows straight into the B operand
// The scheme of the fused attention kernel's projection (attn_fused.hip) for the K <= 192 token GEMMs: a WAVE owns a 16-token
// tile; lane (j = l & 15, kq = l >> 4) loads x[token j][16 c + 4 kq .. + 3] straight from global memory into the MFMA B-operand
// registers (the k index is permuted identically on both operands), A = the block's 96 weight rows from an LDS copy, and the
// accumulator layout (lane (j, kq) holds y[token j][16 nt + 4 kq + r]) is stored with one float4 per n tile.  No X tile in LDS,
// no barrier after the weight staging: the four waves of a block walk their tiles independently, 3-4 blocks per CU.
// (k_gemm_wstat stages X through LDS with two barriers per 32/64-row tile: 21 us for M = 49152, N = K = 96 -- 0.25 of either
// roof -- against ~6 us of MFMA time and ~6 us of HBM time.)
// PRO_LN: LayerNorm folded as in the attention kernel -- gamma into the weights while they are staged, the normalisation
// behind the MFMAs: y = rstd * (W' x - mean * rowsum(W')) + (b + W beta).
// RR_EPI: 1 bias, 2 bias + GELU, 3 bias + two residuals, 5 bias + one residual.
template <int K, int PRO, int EPI>
__global__ __launch_bounds__(256, ((K <= 96 && !(PRO == PRO_SKSEL && K > 32)) ? 3 : 2)) void k_gemm_rowreg(const float* __restrict__ x, int ldx, const float* __restrict__ w,
                                                      float* __restrict__ y, int ldy, int M, int N, ProArgs p, EpiArgs e) {
  constexpr int KC = K / 16, LDW = K + PAD, BN = 96, NT = 6;
  constexpr int G = PRO == PRO_SKSEL ? 3 : 1;              // SKConv select: x = sum_g A[b][g] * cat[:, g K : (g + 1) K]
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ws = smem;                  // [96][LDW]
  float* pb = Ws + BN * LDW;         // [96] bias (b' under PRO_LN), [96] rowsum(W') (PRO_LN)
  float* scr = pb + 2 * BN;          // PRO_LN: [96][K / 4][2] partial sums of the fold
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, kq = lane >> 4;
  const int n_blk = blockIdx.y * BN;
  const int tiles = M / 16;
  const int stride = gridDim.x * 4;                        // in units of SUB consecutive 16-row tiles (SUB = 2 under RR_EPI 4)
  int tile = (blockIdx.x * 4 + wave) * (EPI == 4 ? 2 : 1);

  // global loads in the order they are consumed: weights, then the first tile's rows
  constexpr int KV = K / 4, WL = (BN * KV + 255) / 256;
  float4 wv[WL];
#pragma unroll
  for (int u = 0; u < WL; ++u) {
    const int i = min(tid + u * 256, BN * KV - 1);
    wv[u] = *reinterpret_cast<const float4*>(w + (size_t)(n_blk + i / KV) * K + (i % KV) * 4);
  }
  f32x4 xr[G][KC];
  auto load_rows = [&](int t_) {
    const size_t m = (size_t)(t_ < tiles ? t_ : tiles - 1) * 16 + lr;      // clamped, never predicated
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int c = 0; c < KC; ++c) xr[g][c] = *reinterpret_cast<const f32x4*>(x + m * ldx + g * K + 16 * c + 4 * kq);
  };
  load_rows(tile);
#pragma unroll
  for (int u = 0; u < WL; ++u) {
    const int i = tid + u * 256;
    if (i < BN * KV) {
      const int r = i / KV, c4 = (i % KV) * 4;
      float4 v = wv[u];
      if (PRO == PRO_LN) {
        const float4 gm = *reinterpret_cast<const float4*>(p.ln_w + c4), bt = *reinterpret_cast<const float4*>(p.ln_b + c4);
        const float4 wb = make_float4(v.x * bt.x, v.y * bt.y, v.z * bt.z, v.w * bt.w);
        v = make_float4(v.x * gm.x, v.y * gm.y, v.z * gm.z, v.w * gm.w);
        scr[(r * KV + c4 / 4) * 2] = (v.x + v.y) + (v.z + v.w);
        scr[(r * KV + c4 / 4) * 2 + 1] = (wb.x + wb.y) + (wb.z + wb.w);
      }
      *reinterpret_cast<float4*>(Ws + r * LDW + c4) = v;
    }
  }
  if (PRO == PRO_LN) __syncthreads();
  if (tid < BN) {
    float bb = e.bias ? e.bias[n_blk + tid] : 0.f, cw = 0.f;
    if (PRO == PRO_LN)
      for (int k = 0; k < KV; ++k) { cw += scr[(tid * KV + k) * 2]; bb += scr[(tid * KV + k) * 2 + 1]; }      // fixed order
    pb[tid] = bb;
    pb[BN + tid] = cw;
  }
  __syncthreads();

  // RR_EPI 4 (SKConv projection + global-average-pool partials, pgrm.py:84-86): a wave takes PAIRS of 16-row tiles and writes one
  // partial (column sums of GELU(y) over the pair's 32 rows) per pair -- the consumer's one-partial-per-32-rows contract
  constexpr int SUB = EPI == 4 ? 2 : 1;
  float cs[NT][4];
  for (; tile < tiles; tile = ((tile % SUB) + 1 < SUB) ? tile + 1 : (tile / SUB + stride) * SUB) {
    const size_t m = (size_t)tile * 16 + lr;
    const int next_tile = ((tile % SUB) + 1 < SUB) ? tile + 1 : (tile / SUB + stride) * SUB;
    const size_t yoff = m * ldy + n_blk + 4 * kq;
    f32x4 r1[NT], r2[NT];
    if (EPI == 3 || EPI == 5) {            // residual rows: in flight during the MFMAs
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        r1[nt] = *reinterpret_cast<const f32x4*>(e.res1 + yoff + 16 * nt);
        if (EPI == 3) r2[nt] = *reinterpret_cast<const f32x4*>(e.res2 + yoff + 16 * nt);
      }
    }
    f32x4 xb[KC];
    float mean = 0.f, rstd = 1.f;
    if (PRO == PRO_SKSEL) {
      const size_t b = m / p.rows_per_image;
#pragma unroll
      for (int c = 0; c < KC; ++c) {
        xb[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const f32x4 a = *reinterpret_cast<const f32x4*>(p.sel + (b * G + g) * K + 16 * c + 4 * kq);
          xb[c][0] += a[0] * xr[g][c][0]; xb[c][1] += a[1] * xr[g][c][1]; xb[c][2] += a[2] * xr[g][c][2]; xb[c][3] += a[3] * xr[g][c][3];
        }
      }
    } else {
#pragma unroll
      for (int c = 0; c < KC; ++c) xb[c] = xr[0][c];
    }
    // the next tile's rows are requested NOW (second register set): they fly during this tile's MFMAs and epilogue.  With one
    // tile per wave every wave of the chip loads, multiplies and stores in lockstep -- load / MFMA / store phases in sequence,
    // 18 us for 6 us of MFMA and 6 us of HBM time -- so the launcher gives a wave several tiles and this pipeline overlaps them
    __builtin_amdgcn_sched_barrier(0);
    load_rows(next_tile);
    __builtin_amdgcn_sched_barrier(0);
    if (PRO != PRO_SKSEL) {
      if (PRO == PRO_LN) {                 // two-pass row statistics over the 4 kq partners of the row (like nn.LayerNorm)
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int c = 0; c < KC; ++c) { s0 += xb[c][0] + xb[c][1]; s1 += xb[c][2] + xb[c][3]; }
        float s_ = s0 + s1;
        s_ += xshfl<16>(s_); s_ += xshfl<32>(s_);
        mean = s_ * (1.0f / K);
        float q0 = 0.f, q1 = 0.f, q2 = 0.f, q3 = 0.f;
#pragma unroll
        for (int c = 0; c < KC; ++c) {
          const float d0 = xb[c][0] - mean, d1 = xb[c][1] - mean, d2 = xb[c][2] - mean, d3 = xb[c][3] - mean;
          q0 = fmaf(d0, d0, q0); q1 = fmaf(d1, d1, q1); q2 = fmaf(d2, d2, q2); q3 = fmaf(d3, d3, q3);
        }
        float q = (q0 + q1) + (q2 + q3);
        q += xshfl<16>(q); q += xshfl<32>(q);
        rstd = 1.0f / sqrtf(q * (1.0f / K) + p.eps);
      }
    }
    f32x4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // weight operands double-buffered one k-chunk ahead; the scheduling fences keep hipcc from hoisting ALL the chunks' LDS reads
    // to the top of the tile (6 x 24 registers: it then spills the row and residual registers to scratch)
    f32x4 wf[2][NT];
    const float* wa = Ws + lr * LDW + 4 * kq;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) wf[0][nt] = *reinterpret_cast<const f32x4*>(wa + 16 * nt * LDW);
#pragma unroll
    for (int c = 0; c < KC; ++c) {
      if (c + 1 < KC) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) wf[(c + 1) & 1][nt] = *reinterpret_cast<const f32x4*>(wa + 16 * nt * LDW + 16 * (c + 1));
      }
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma16(wf[c & 1][nt][s4], xb[c][s4], acc[nt]);
      __builtin_amdgcn_sched_barrier(0);
    }
    const float nm = -mean * rstd;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(pb + 16 * nt + 4 * kq);
      f32x4 v;
      if (PRO == PRO_LN) {
        const f32x4 cw = *reinterpret_cast<const f32x4*>(pb + BN + 16 * nt + 4 * kq);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaf(acc[nt][r], rstd, fmaf(nm, cw[r], b4[r]));
      } else {
        v = acc[nt] + b4;
      }
      if (EPI == 2) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
      }
      if (EPI == 3 || EPI == 5) v += r1[nt];
      if (EPI == 3) v += r2[nt];
      *reinterpret_cast<f32x4*>(y + yoff + 16 * nt) = v;
      if (EPI == 4) {
#pragma unroll
        for (int r = 0; r < 4; ++r) cs[nt][r] = (tile % SUB == 0 ? 0.f : cs[nt][r]) + gelu_erf(v[r]);
      }
    }
    if (EPI == 4 && tile % SUB == SUB - 1) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        f32x4 c4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float c = cs[nt][r];
          c += xshfl<1>(c); c += xshfl<2>(c); c += xshfl<4>(c); c += xshfl<8>(c);
          c4[r] = c;
        }
        if (lr == 0) *reinterpret_cast<f32x4*>(e.colsum + (size_t)(tile / SUB) * N + n_blk + 16 * nt + 4 * kq) = c4;
      }
    }
  }
}


// ---------------------------------------------------------------------------------- SKConv select -> x1 -> LayerNorm2 -> fc1
// The second half of a Swin block up to the Mlp's first Linear (pgrm.py:91-96, 327-331, 31) in ONE launch: per 16-token tile of
// a wave
//   sel  = sum_g A[b][g] * cat[:, g CG : (g + 1) CG]                  (the SKConv's softmax-weighted group sum)
//   x1   = proj_head(sel) + b_head + feats + shortcut                   (both residuals, = k_gemm_rowreg<CG, PRO_SKSEL, 3>: same
//                                                                        instructions in the same order, bitwise-equal x1)
//   y    = fc1(LayerNorm2(x1))                                          (= k_gemm_rowreg<C, PRO_LN, 1> on the x1 REGISTERS: the MFMA D
//                                                                        layout of x1 is the B-operand layout of the next product)
// x1 is written once (fc2's residual) by the blocks of the first column group and never read back; the other column groups of the
// N = 4 C outputs recompute the cheap first product (K = CG against K = C) instead: one launch and one 4 M C read less per
// block than sk_select + ln_linear.  Measured (B = 48, tools/prof_skmlp.py): 62 us against 60-62 us for the two launches it
// replaces (22 + 44 in the pipeline), 70 us with the two training outputs against four launches (select 9 + proj_head 22 +
// LayerNorm 10 + fc1 45): the time of these K <= 96 products is the MFMA time PLUS the vector / LDS issue time of the waves of a
// SIMD (DESIGN.md "What bounds these fp32 kernels"), ~2x the MFMA floor in both forms -- fusing removes launches, not that.
// SAVE (training forward): the first column group also writes sel (M, CG) and n2 = LayerNorm2(x1) (M, C) for the backward.
template <int C, int CG, bool SAVE, int OCC>      // OCC: resident blocks per CU (3: <= 168 registers, weight fragments single-buffered)
__global__ __launch_bounds__(256, OCC) void k_sk_mlp_in(const float* __restrict__ cat, const float* __restrict__ sel, int rows_per_image,
                                                      const float* __restrict__ w_head, const float* __restrict__ b_head,
                                                      const float* __restrict__ feats, const float* __restrict__ shortcut, float* __restrict__ x1,
                                                      const float* __restrict__ ln_w, const float* __restrict__ ln_b, float eps,
                                                      const float* __restrict__ w_fc1, const float* __restrict__ b_fc1, float* __restrict__ y,
                                                      int M, int N, float* __restrict__ v_out, float* __restrict__ n2_out, float p_row,
                                                      unsigned long long seed_row) {
  // p_row > 0 (training): timm DropPath on the attention branch (pgrm.py:329) -- x1 = shortcut + m_b (proj_head(sel) + b_head + feats),
  // one counter-based draw m_b in {0, 1 / (1 - p)} per batch sample, the mask the unfused dpmn_dropout_f32 applies
  constexpr int KC = C / 16, KV = C / 4, LDW = C + PAD, BN = 96, NT = 6, G = C / CG, HC = CG / 16, LDH = CG + PAD;
  static_assert(C == 96 && G == 3, "built for dim 96, three window groups");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ws = smem;                  // [96][LDW]  fc1 rows of this column group, gamma folded in
  float* pb = Ws + BN * LDW;         // [96] b' = b + W beta, [96] rowsum(W'), [96] b_head
  float* Wh = pb + 3 * BN;           // [96][LDH]  proj_head (staged behind the fold's scratch, which lives here first)
  float* scr = Wh;                   // [96][KV] partial sums of W beta (dead before Wh is written); 53.4 KB in all: three blocks per CU
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, kq = lane >> 4;
  const int n_blk = blockIdx.y * BN;
  const int tiles = M / 16;
  const int stride = gridDim.x * 4;
  int tile = blockIdx.x * 4 + wave;
  const bool write_x1 = blockIdx.y == 0;

  constexpr int WL = (BN * KV + 255) / 256;
  float4 wv[WL];
#pragma unroll
  for (int u = 0; u < WL; ++u) {
    const int i = min(tid + u * 256, BN * KV - 1);
    wv[u] = *reinterpret_cast<const float4*>(w_fc1 + (size_t)(n_blk + i / KV) * C + (i % KV) * 4);
  }
  f32x4 xr[G][HC];
  auto load_rows = [&](int t_) {
    const size_t m = (size_t)(t_ < tiles ? t_ : tiles - 1) * 16 + lr;      // clamped, never predicated
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int c = 0; c < HC; ++c) xr[g][c] = *reinterpret_cast<const f32x4*>(cat + m * C + g * CG + 16 * c + 4 * kq);
  };
  // the residual rows of a tile are requested as soon as the previous tile has consumed its own (same registers), BEFORE that
  // tile's stores: vmcnt retires loads and stores in order, so a load issued behind the y stores of the previous tile would make
  // its consumer wait for those stores to reach memory as well
  f32x4 r1[NT], r2[NT];
  auto load_res = [&](int t_) {
    const size_t ro = ((size_t)(t_ < tiles ? t_ : tiles - 1) * 16 + lr) * C + 4 * kq;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      r1[nt] = *reinterpret_cast<const f32x4*>(feats + ro + 16 * nt);
      r2[nt] = *reinterpret_cast<const f32x4*>(shortcut + ro + 16 * nt);
    }
  };
  load_rows(tile);
  load_res(tile);
#pragma unroll
  for (int u = 0; u < WL; ++u) {
    const int i = tid + u * 256;
    if (i < BN * KV) {
      const int r = i / KV, c4 = (i % KV) * 4;
      float4 v = wv[u];
      if (!SAVE) {        // eval: LayerNorm folded into the weights (gamma) and the bias (W beta), as k_gemm_rowreg<C, PRO_LN>
        const float4 gm = *reinterpret_cast<const float4*>(ln_w + c4), bt = *reinterpret_cast<const float4*>(ln_b + c4);
        const float4 wb = make_float4(v.x * bt.x, v.y * bt.y, v.z * bt.z, v.w * bt.w);
        v = make_float4(v.x * gm.x, v.y * gm.y, v.z * gm.z, v.w * gm.w);
        scr[r * KV + c4 / 4] = (wb.x + wb.y) + (wb.z + wb.w);
      }
      *reinterpret_cast<float4*>(Ws + r * LDW + c4) = v;
    }
  }
  __syncthreads();
  if (tid < BN) {
    float bb = b_fc1 ? b_fc1[n_blk + tid] : 0.f, cw = 0.f;
    if (!SAVE)
    for (int k = 0; k < KV; ++k) {             // fixed order, the same sums as k_gemm_rowreg's fold
      const float4 v = *reinterpret_cast<const float4*>(Ws + tid * LDW + 4 * k);
      cw += (v.x + v.y) + (v.z + v.w);
      bb += scr[tid * KV + k];
    }
    pb[tid] = bb;
    pb[BN + tid] = cw;
    pb[2 * BN + tid] = b_head ? b_head[tid] : 0.f;
  }
  __syncthreads();
  for (int i = tid; i < C * (CG / 4); i += 256) {      // proj_head: (C, CG) row-major
    const int r = i / (CG / 4), c4 = (i % (CG / 4)) * 4;
    *reinterpret_cast<float4*>(Wh + r * LDH + c4) = *reinterpret_cast<const float4*>(w_head + (size_t)r * CG + c4);
  }
  __syncthreads();

  for (; tile < tiles; tile += stride) {
    const size_t m = (size_t)tile * 16 + lr;
    const size_t roff = m * C + 4 * kq;
    f32x4 xb[HC];
    {
      const size_t b = m / rows_per_image;
#pragma unroll
      for (int c = 0; c < HC; ++c) {
        xb[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const f32x4 a = *reinterpret_cast<const f32x4*>(sel + (b * G + g) * CG + 16 * c + 4 * kq);
          xb[c][0] += a[0] * xr[g][c][0]; xb[c][1] += a[1] * xr[g][c][1]; xb[c][2] += a[2] * xr[g][c][2]; xb[c][3] += a[3] * xr[g][c][3];
        }
      }
    }
    if (SAVE && write_x1) {
#pragma unroll
      for (int c = 0; c < HC; ++c) *reinterpret_cast<f32x4*>(v_out + m * CG + 16 * c + 4 * kq) = xb[c];
    }
    __builtin_amdgcn_sched_barrier(0);
    load_rows(tile + stride);                  // the next tile's rows fly during this tile's MFMAs
    __builtin_amdgcn_sched_barrier(0);
    // ---- x1 = proj_head(sel) + b_head + feats + shortcut
    f32x4 x1r[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) x1r[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    {
      const float* ha = Wh + lr * LDH + 4 * kq;
#pragma unroll
      for (int c = 0; c < HC; ++c) {
        f32x4 hf[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) hf[nt] = *reinterpret_cast<const f32x4*>(ha + 16 * nt * LDH + 16 * c);
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) x1r[nt] = mfma16(hf[nt][s4], xb[c][s4], x1r[nt]);
      }
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(pb + 2 * BN + 16 * nt + 4 * kq);
      f32x4 v = x1r[nt] + b4;
      v += r1[nt];
      if (p_row > 0.f) v *= drop_scale(seed_row, (unsigned long long)(m / rows_per_image), p_row, 1.0f / (1.0f - p_row));
      v += r2[nt];
      x1r[nt] = v;
    }
    __builtin_amdgcn_sched_barrier(0);
    load_res(tile + stride);                   // the next tile's residual rows, ahead of this tile's stores
    __builtin_amdgcn_sched_barrier(0);
    if (write_x1) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) *reinterpret_cast<f32x4*>(x1 + roff + 16 * nt) = x1r[nt];
    }
    // ---- LayerNorm2 statistics of the row (two passes over the 4 kq partners, as nn.LayerNorm / k_gemm_rowreg)
    float mean, rstd;
    {
      float s0 = 0.f, s1 = 0.f;
#pragma unroll
      for (int c = 0; c < KC; ++c) { s0 += x1r[c][0] + x1r[c][1]; s1 += x1r[c][2] + x1r[c][3]; }
      float s_ = s0 + s1;
      s_ += xshfl<16>(s_); s_ += xshfl<32>(s_);
      mean = s_ * (1.0f / C);
      float q0 = 0.f, q1 = 0.f, q2 = 0.f, q3 = 0.f;
#pragma unroll
      for (int c = 0; c < KC; ++c) {
        const float d0 = x1r[c][0] - mean, d1 = x1r[c][1] - mean, d2 = x1r[c][2] - mean, d3 = x1r[c][3] - mean;
        q0 = fmaf(d0, d0, q0); q1 = fmaf(d1, d1, q1); q2 = fmaf(d2, d2, q2); q3 = fmaf(d3, d3, q3);
      }
      float q = (q0 + q1) + (q2 + q3);
      q += xshfl<16>(q); q += xshfl<32>(q);
      rstd = 1.0f / sqrtf(q * (1.0f / C) + eps);
    }
    if (SAVE) {
      // training forward: the NORMALISED row n2 = (x1 - mean) * rstd * gamma + beta (the expression of dpmn_layernorm_f32) feeds the
      // MFMAs with the unfolded weights -- the same arithmetic as LayerNorm + Linear launches.  (The folded form of the eval path
      // subtracts mean * rowsum(W') from W' x behind the MFMAs: exact enough for the forward, but on the text-prior branch, whose
      // tokens carry a large common offset, that cancellation raised the gradient error vs the oracle from 4e-5 to 2e-4.)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const f32x4 gm = *reinterpret_cast<const f32x4*>(ln_w + 16 * nt + 4 * kq), bt = *reinterpret_cast<const f32x4*>(ln_b + 16 * nt + 4 * kq);
#pragma unroll
        for (int r = 0; r < 4; ++r) x1r[nt][r] = (x1r[nt][r] - mean) * rstd * gm[r] + bt[r];
        if (write_x1) *reinterpret_cast<f32x4*>(n2_out + roff + 16 * nt) = x1r[nt];
      }
    }
    // ---- y = rstd * (W' x1 - mean * rowsum(W')) + b'
    f32x4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* wa = Ws + lr * LDW + 4 * kq;
    if constexpr (OCC >= 3) {
#pragma unroll
      for (int c = 0; c < KC; ++c) {
        f32x4 wf1[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) wf1[nt] = *reinterpret_cast<const f32x4*>(wa + 16 * nt * LDW + 16 * c);
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma16(wf1[nt][s4], x1r[c][s4], acc[nt]);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
    f32x4 wf[2][NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) wf[0][nt] = *reinterpret_cast<const f32x4*>(wa + 16 * nt * LDW);
#pragma unroll
    for (int c = 0; c < KC; ++c) {
      if (c + 1 < KC) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) wf[(c + 1) & 1][nt] = *reinterpret_cast<const f32x4*>(wa + 16 * nt * LDW + 16 * (c + 1));
      }
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma16(wf[c & 1][nt][s4], x1r[c][s4], acc[nt]);
      __builtin_amdgcn_sched_barrier(0);
    }
    }
    const float nm = -mean * rstd;
    const size_t yoff = m * N + n_blk + 4 * kq;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(pb + 16 * nt + 4 * kq);
      const f32x4 cw = *reinterpret_cast<const f32x4*>(pb + BN + 16 * nt + 4 * kq);
      f32x4 v;
      if (SAVE) v = acc[nt] + b4;
      else {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaf(acc[nt][r], rstd, fmaf(nm, cw[r], b4[r]));
      }
      *reinterpret_cast<f32x4*>(y + yoff + 16 * nt) = v;
    }
  }
}

template <int K, int PRO, int EPI>
int launch_rowreg(const float* x, int ldx, const float* w, float* y, int ldy, int M, int N, const ProArgs& p, const EpiArgs& e,
                  hipStream_t st) {
  const size_t smem = (size_t)(96 * (K + PAD) + 2 * 96 + (PRO == PRO_LN ? 96 * (K / 4) * 2 : 0)) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_rowreg<K, PRO, EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_set = true;
  }
  const int tiles = M / (EPI == 4 ? 32 : 16), ny = N / 96;      // work items of a wave (pairs of 16-row tiles under RR_EPI 4)
  // resident waves: LDS-limited blocks per CU x 4; every wave gets the same number of tiles when that divides evenly
  static const int rr_blocks = getenv("DPMN_RR_BLOCKS") ? atoi(getenv("DPMN_RR_BLOCKS")) : 0;
  const bool x3 = (PRO == PRO_NONE || PRO == PRO_LN) && (K == 96 || K == 192) && x3_on(128);
  // (mode 2: the weight planes are 1.5x the fp32 copy -- two blocks per CU at K = 96, one at K = 192 -- and a tile's MFMAs take 0.4x the
  //  time, so two tiles per wave are enough to overlap: 96 -> 384 at M = 49152 30.3 vs 33.2 us)
  const int per_cu = x3 ? (K <= 96 ? 2 : 1) : (smem <= 40 * 1024 ? 3 : (smem <= 80 * 1024 ? 2 : 1));
  int gx = (rr_blocks > 0 ? rr_blocks : 256 * per_cu) / ny;
  if (gx < 1) gx = 1;
  // at least 3 tiles per wave, so that the load / MFMA / store pipeline of a wave has something to overlap
  static const int rr_tpw_env = getenv("DPMN_RR_TPW") ? atoi(getenv("DPMN_RR_TPW")) : 0;
  const int rr_tpw = rr_tpw_env > 0 ? rr_tpw_env : (x3 ? 2 : 3);
  while (gx > 256 / ny && gx > 1 && (long)gx * 4 * rr_tpw > tiles) gx -= 256 / ny > 0 ? 256 / ny : 1;
  if (gx * 4 > tiles) gx = cdiv(tiles, 4);
  ProfScope prof(PRO == PRO_LN ? PT_GEMM_WSTAT_LN : PT_GEMM_WSTAT, st, 2.0 * M * (double)N * K,
                 4.0 * ((double)M * K * (PRO == PRO_SKSEL ? 3 : 1) + (double)M * N * (1 + (EPI == 3 ? 2 : (EPI == 5 ? 1 : 0))) + (double)N * K));
  if constexpr ((PRO == PRO_NONE || PRO == PRO_LN) && (K == 96 || K == 192)) {
    if (x3) {                  // mode 2: weights split once per block into bf16 planes, rows split in registers (gemm_rowreg_x3.hip)
      (void)dpmn_gemm::x3_launch_rowreg(K, PRO, EPI, x, ldx, w, y, ldy, M, N, p, e, gx, st);
      DPMN_CHECK_LAUNCH();
      return DPMN_OK;
    }
  }
  hipLaunchKernelGGL((k_gemm_rowreg<K, PRO, EPI>), dim3(gx, ny), dim3(256), smem, st, x, ldx, w, y, ldy, M, N, p, e);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

// rows-in-registers path for the shapes it is built for; returns -1 when the call does not qualify
template <int K, int PRO>
int try_rowreg(const float* x, int ldx, const float* w, float* y, int ldy, int M, int N, const ProArgs& p, const EpiArgs& e,
               hipStream_t st) {
  static const int on = getenv("DPMN_ROWREG") ? atoi(getenv("DPMN_ROWREG")) : 1;
  if (!on || M % 16 || N % 96 || e.atomic || ldy % 4 || ldx % 4 || M < 1024) return -1;
  if constexpr (PRO == PRO_SKSEL && (K == 32 || K == 64)) {
    if (p.groups != 3 || p.rows_per_image % 16 || !(e.res1 && e.res2) || e.act != ACT_NONE || e.colsum) return -1;
    return launch_rowreg<K, PRO_SKSEL, 3>(x, ldx, w, y, ldy, M, N, p, e, st);
  } else if constexpr ((PRO == PRO_NONE || PRO == PRO_LN) && (K == 96 || K == 192)) {
    if (e.colsum) {
      if constexpr (PRO == PRO_NONE) {
        if (M % 32 == 0 && !e.res1 && !e.res2 && e.act == ACT_NONE) return launch_rowreg<K, PRO_NONE, 4>(x, ldx, w, y, ldy, M, N, p, e, st);
      }
      return -1;
    }
    if (e.act == ACT_GELU && !e.res1 && !e.res2) return launch_rowreg<K, PRO, 2>(x, ldx, w, y, ldy, M, N, p, e, st);
    if (e.act != ACT_NONE) return -1;
    if (e.res1 && e.res2) return launch_rowreg<K, PRO, 3>(x, ldx, w, y, ldy, M, N, p, e, st);
    if (e.res1) return launch_rowreg<K, PRO, 5>(x, ldx, w, y, ldy, M, N, p, e, st);
    if (e.res2) return -1;
    return launch_rowreg<K, PRO, 1>(x, ldx, w, y, ldy, M, N, p, e, st);
  }
  return -1;
}

template <int K, int PRO, int TH, bool FULL = false, int EPI = 0>
int launch_wholeK_th(const float* x, int ldx, const float* w, float* y, int ldy, int M, int N, const ProArgs& p,
                     const EpiArgs& e, hipStream_t st, int target_blocks) {
  constexpr int BM = TH / 8;
  const size_t smem = (size_t)((WS_BN + WSTAT_NBUF * BM) * (K + PAD) + (TH / 64) * WS_BN + 2 * K) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_wstat<K, PRO, TH, FULL, EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_set = true;
  }
  const int tiles = cdiv(M, BM), ny = cdiv(N, WS_BN);
  static const int force_blocks = getenv("DPMN_WSTAT_BLOCKS") ? atoi(getenv("DPMN_WSTAT_BLOCKS")) : 0;      // experiment knob
  if (force_blocks > 0) target_blocks = force_blocks;
  int gx = target_blocks / ny;
  if (gx < 1) gx = 1;
  if (gx > tiles) gx = tiles;
  dim3 grid(gx, ny);
  ProfScope prof(PRO == PRO_LN ? PT_GEMM_WSTAT_LN : PT_GEMM_WSTAT, st, 2.0 * M * (double)N * K, 4.0 * ((double)M * K + (double)M * N + (double)N * K));
  hipLaunchKernelGGL((k_gemm_wstat<K, PRO, TH, FULL, EPI>), grid, dim3(TH), smem, st, x, ldx, w, y, ldy, M, N, p, e);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

template <int K, int PRO>
int launch_wholeK(const float* x, int ldx, const float* w, float* y, int ldy, int M, int N, const ProArgs& p,
                  const EpiArgs& e, hipStream_t st) {
  // 512-thread blocks (64-token tiles, 8 waves sharing one W tile): 64 KB LDS at K = 96 -> 2 blocks = 16 waves per CU;
  // 256-thread blocks (32-token tiles): 51 KB -> 3 blocks = 12 waves per CU, and the only variant with the per-32-row
  // column-sum epilogue (SKConv GAP partials)
  static const int big = getenv("DPMN_WSTAT_TH") ? atoi(getenv("DPMN_WSTAT_TH")) : 512;
  {
    const int rc = try_rowreg<K, PRO>(x, ldx, w, y, ldy, M, N, p, e, st);
    if (rc != -1) return rc;
  }
  if constexpr (K <= 128) {
    if (big == 512 && !e.colsum && M >= 4096) {
      if (M % 64 == 0 && N % WS_BN == 0 && !e.atomic) {    // every tile interior: the predicate-free instantiations
        const bool plain = !e.res1 && !e.res2 && ldy % 4 == 0;      // bias optional (data-gradient GEMMs have none)
        if constexpr (K == 96 && (PRO == PRO_NONE || PRO == PRO_LN)) {
          // one column block and a 64-row tile count that does not divide over 512 resident blocks (M = 49152: 768 tiles =
          // 1.5 per block): 32-row tiles over 768 blocks (3 per CU) are balanced, 2 tiles each
          const int tiles64 = M / 64;
          if (N == WS_BN && plain && e.act == ACT_NONE && tiles64 % 512 != 0 && tiles64 < 2048 && (M / 32) % 768 == 0)
            return launch_wholeK_th<K, PRO, 256, true, 1>(x, ldx, w, y, ldy, M, N, p, e, st, 768);
        }
        if (plain && e.act == ACT_NONE) return launch_wholeK_th<K, PRO, 512, true, 1>(x, ldx, w, y, ldy, M, N, p, e, st, 512);
        if (plain && e.act == ACT_GELU) return launch_wholeK_th<K, PRO, 512, true, 2>(x, ldx, w, y, ldy, M, N, p, e, st, 512);
        if constexpr (PRO == PRO_SKSEL || PRO == PRO_NONE) {
          if (e.res1 && e.res2 && e.act == ACT_NONE && ldy % 4 == 0)
            return launch_wholeK_th<K, PRO, 512, true, 3>(x, ldx, w, y, ldy, M, N, p, e, st, 512);
        }
        if constexpr (PRO == PRO_NONE) {
          if (e.res1 && !e.res2 && e.act == ACT_NONE && ldy % 4 == 0)
            return launch_wholeK_th<K, PRO, 512, true, 5>(x, ldx, w, y, ldy, M, N, p, e, st, 512);
        }
        return launch_wholeK_th<K, PRO, 512, true>(x, ldx, w, y, ldy, M, N, p, e, st, 512);
      }
      return launch_wholeK_th<K, PRO, 512>(x, ldx, w, y, ldy, M, N, p, e, st, 512);
    }
  }
  if constexpr (K == 96 && PRO == PRO_LN) {        // experiment: 32-row tiles with the straight-line epilogues (DPMN_WSTAT_TH=256)
    if (big == 256 && !e.colsum && M % 32 == 0 && N % WS_BN == 0 && !e.atomic && !e.res1 && !e.res2 && ldy % 4 == 0) {
      if (e.act == ACT_NONE) return launch_wholeK_th<K, PRO, 256, true, 1>(x, ldx, w, y, ldy, M, N, p, e, st, 768);
      if (e.act == ACT_GELU) return launch_wholeK_th<K, PRO, 256, true, 2>(x, ldx, w, y, ldy, M, N, p, e, st, 768);
    }
  }
  if constexpr (PRO == PRO_NONE && K == 96) {      // SKConv projection + GAP partials, straight-line (pgrm.py:84-86)
    if (e.colsum && M % 32 == 0 && N % WS_BN == 0 && !e.atomic && !e.res1 && !e.res2 && e.act == ACT_NONE && ldy % 4 == 0) {
      static const int cs512 = getenv("DPMN_COLSUM_TH") ? atoi(getenv("DPMN_COLSUM_TH")) : 512;
      if (cs512 == 512 && M % 64 == 0 && M >= 4096)     // 64-row tiles, two 32-row partials each: half the barriers
        return launch_wholeK_th<K, PRO, 512, true, 4>(x, ldx, w, y, ldy, M, N, p, e, st, 512);
      return launch_wholeK_th<K, PRO, 256, true, 4>(x, ldx, w, y, ldy, M, N, p, e, st, WSTAT_NBUF == 1 ? 768 : 512);
    }
  }
  return launch_wholeK_th<K, PRO, 256>(x, ldx, w, y, ldy, M, N, p, e, st, WSTAT_NBUF == 1 ? 768 : 512);
}

template <int PRO>
int dispatch_wholeK(int K, const float* x, int ldx, const float* w, float* y, int ldy, int M, int N, const ProArgs& p,
                    const EpiArgs& e, hipStream_t st) {
  switch (K) {
    case 32: return launch_wholeK<32, PRO>(x, ldx, w, y, ldy, M, N, p, e, st);
    case 64: return launch_wholeK<64, PRO>(x, ldx, w, y, ldy, M, N, p, e, st);
    case 96: return launch_wholeK<96, PRO>(x, ldx, w, y, ldy, M, N, p, e, st);
    case 128: return launch_wholeK<128, PRO>(x, ldx, w, y, ldy, M, N, p, e, st);
    case 192: return launch_wholeK<192, PRO>(x, ldx, w, y, ldy, M, N, p, e, st);
    default: return dpmn_set_error(DPMN_ERR_ARG, "whole-K GEMM supports K in {32,64,96,128,192}");
  }
}

}  // namespace

// ================================================================================== C ABI
extern "C" {

int dpmn_linear_f32(const float* x, const float* w, const float* bias, const float* res1, const float* res2, float* y,
                    int M, int N, int K, int act, float slope, dpmn_stream_t stream) {
  DPMN_REQUIRE(x && w && y && M > 0 && N > 0 && K > 0, "linear: null pointer or empty shape");
  DPMN_REQUIRE(N % 4 == 0, "linear: N must be a multiple of 4");
  EpiArgs e{bias, res1, res2, nullptr, act, slope};
  ProArgs p{};
  if (K == 32 || K == 64 || K == 96 || K == 128 || K == 192)
    return dispatch_wholeK<PRO_NONE>(K, x, K, w, y, N, M, N, p, e, as_stream(stream));
  DPMN_REQUIRE(K % 32 == 0, "linear: K must be a multiple of 32");
  dim3 grid(cdiv(M, 64), cdiv(N, 96));
  ProfScope prof(PT_GEMM_KLOOP, as_stream(stream), 2.0 * M * (double)N * K, 4.0 * ((double)M * K + (double)M * N * (res1 ? 2 : 1) + (double)N * K));
  if (x3_on(8)) (void)dpmn_gemm::x3_launch_kloop(x, K, w, K, y, N, M, N, K, e, 0, 0L, 0L, grid, as_stream(stream));
  else hipLaunchKernelGGL(k_gemm_kloop, grid, dim3(256), 0, as_stream(stream), x, K, w, K, y, N, M, N, K, e, 0, 0L, 0L);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

// y = res + Dropout(x w^T + bias) with the masks of dpmn_dropout_f32 (element dropout p_elem and / or per-sample DropPath p_row, row_len =
// elements per sample): Mlp.fc2 -> Mlp.drop -> DropPath -> + shortcut (pgrm.py:39-40, 330) in the GEMM's epilogue.  Whole 64 x 96
// tiles only (the k-loop kernel's interior epilogue); returns DPMN_ERR_ARG otherwise (the caller composes Linear + dropout).
int dpmn_linear_drop_f32(const float* x, const float* w, const float* bias, const float* res, float* y, int M, int N, int K, float p_elem,
                         unsigned long long seed_elem, float p_row, unsigned long long seed_row, long row_len, dpmn_stream_t stream) {
  DPMN_REQUIRE(x && w && bias && res && y && M > 0 && M % 64 == 0 && N % 96 == 0 && K % 32 == 0 && K > 192,
               "linear_drop: whole 64 x 96 tiles, K a multiple of 32 above 192 (the k-loop GEMM)");
  DPMN_REQUIRE(p_elem >= 0.f && p_elem < 1.f && p_row >= 0.f && p_row < 1.f && (p_row == 0.f || row_len > 0), "linear_drop: bad rates");
  EpiArgs e{bias, res, nullptr, nullptr, ACT_NONE, 0.f};
  e.p_elem = p_elem; e.p_row = p_row; e.seed_elem = seed_elem; e.seed_row = seed_row; e.row_len = row_len;
  dim3 grid(cdiv(M, 64), cdiv(N, 96));
  ProfScope prof(PT_GEMM_KLOOP, as_stream(stream), 2.0 * M * (double)N * K, 4.0 * ((double)M * K + (double)M * N * 2 + (double)N * K));
  if (x3_on(8)) (void)dpmn_gemm::x3_launch_kloop(x, K, w, K, y, N, M, N, K, e, 0, 0L, 0L, grid, as_stream(stream));
  else hipLaunchKernelGGL(k_gemm_kloop, grid, dim3(256), 0, as_stream(stream), x, K, w, K, y, N, M, N, K, e, 0, 0L, 0L);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

int dpmn_add_linear_f32(const float* x, const float* addv, const float* w, const float* bias, float* y, int M, int N,
                        int K, int act, dpmn_stream_t stream) {
  DPMN_REQUIRE(x && addv && w && y && M > 0 && N % 4 == 0, "add_linear: bad arguments");
  EpiArgs e{bias, nullptr, nullptr, nullptr, act, 0.f};
  ProArgs p{};
  p.addv = addv;
  return dispatch_wholeK<PRO_ADD>(K, x, K, w, y, N, M, N, p, e, as_stream(stream));
}

int dpmn_cat2_linear_f32(const float* x1, int k1, const float* x2, int k2, const float* w, const float* bias, float* y,
                         int M, int N, int act, dpmn_stream_t stream) {
  DPMN_REQUIRE(x1 && x2 && w && y && M > 0 && N % 4 == 0 && k1 % 4 == 0 && k2 % 4 == 0, "cat2_linear: bad arguments");
  EpiArgs e{bias, nullptr, nullptr, nullptr, act, 0.f};
  ProArgs p{};
  p.x2 = x2; p.k1 = k1;
  return dispatch_wholeK<PRO_CAT2>(k1 + k2, x1, k1 + k2, w, y, N, M, N, p, e, as_stream(stream));
}

int dpmn_ln_linear_f32(const float* x, const float* ln_w, const float* ln_b, float eps, const float* w,
                       const float* bias, float* y, int M, int N, int K, int act, dpmn_stream_t stream) {
  DPMN_REQUIRE(x && ln_w && ln_b && w && y && M > 0 && N % 4 == 0, "ln_linear: bad arguments");
  EpiArgs e{bias, nullptr, nullptr, nullptr, act, 0.f};
  ProArgs p{};
  p.ln_w = ln_w; p.ln_b = ln_b; p.eps = eps;
  return dispatch_wholeK<PRO_LN>(K, x, K, w, y, N, M, N, p, e, as_stream(stream));
}

int dpmn_sk_proj_f32(const float* cat, const float* w, const float* bias, float* feats, float* colsum_partials, int M,
                     int C, dpmn_stream_t stream) {
  DPMN_REQUIRE(cat && w && feats && colsum_partials && C % 4 == 0, "sk_proj: bad arguments");
  EpiArgs e{bias, nullptr, nullptr, colsum_partials, ACT_NONE, 0.f};
  ProArgs p{};
  return dispatch_wholeK<PRO_NONE>(C, cat, C, w, feats, C, M, C, p, e, as_stream(stream));
}

int dpmn_sk_select_f32(const float* cat, const float* attn_vec, const float* w_head, const float* b_head,
                       const float* feats, const float* shortcut, float* out, int M, int rows_per_image, int C,
                       int groups, dpmn_stream_t stream) {
  DPMN_REQUIRE(cat && attn_vec && w_head && feats && shortcut && out && C % groups == 0, "sk_select: bad arguments");
  EpiArgs e{b_head, feats, shortcut, nullptr, ACT_NONE, 0.f};
  ProArgs p{};
  p.sel = attn_vec; p.rows_per_image = rows_per_image; p.groups = groups;
  return dispatch_wholeK<PRO_SKSEL>(C / groups, cat, C, w_head, out, C, M, C, p, e, as_stream(stream));
}

int dpmn_sk_mlp_in_f32(const float* cat, const float* attn_vec, const float* w_head, const float* b_head, const float* feats,
                       const float* shortcut, float* x1, const float* ln_w, const float* ln_b, float eps, const float* w_fc1,
                       const float* b_fc1, float* y, float* v_out, float* n2_out, int M, int rows_per_image, int C, int groups, int N,
                       dpmn_stream_t stream) {
  return dpmn_sk_mlp_in_drop_f32(cat, attn_vec, w_head, b_head, feats, shortcut, x1, ln_w, ln_b, eps, w_fc1, b_fc1, y, v_out, n2_out, M,
                                 rows_per_image, C, groups, N, 0.f, 0ull, stream);
}

int dpmn_sk_mlp_in_drop_f32(const float* cat, const float* attn_vec, const float* w_head, const float* b_head, const float* feats,
                            const float* shortcut, float* x1, const float* ln_w, const float* ln_b, float eps, const float* w_fc1,
                            const float* b_fc1, float* y, float* v_out, float* n2_out, int M, int rows_per_image, int C, int groups, int N,
                            float p_row, unsigned long long seed_row, dpmn_stream_t stream) {
  DPMN_REQUIRE(p_row >= 0.f && p_row < 1.f, "sk_mlp_in: DropPath rate in [0, 1)");
  DPMN_REQUIRE(cat && attn_vec && w_head && feats && shortcut && x1 && ln_w && ln_b && w_fc1 && y, "sk_mlp_in: null pointer");
  DPMN_REQUIRE((v_out == nullptr) == (n2_out == nullptr), "sk_mlp_in: the two training outputs go together");
  DPMN_REQUIRE(C == 96 && groups == 3 && N % 96 == 0 && M % 16 == 0 && rows_per_image % 16 == 0 && M >= 1024,
               "sk_mlp_in: built for dim 96, three window groups, whole 16-token tiles (dpmn_sk_mlp_in_supported)");
  constexpr int Cc = 96, CG = 32;
  const size_t smem = (size_t)(96 * (Cc + PAD) + 3 * 96 + 96 * (CG + PAD)) * sizeof(float);      // 53376 B: proj_head is the larger tenant of its region
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sk_mlp_in<Cc, CG, false, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sk_mlp_in<Cc, CG, true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sk_mlp_in<Cc, CG, false, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sk_mlp_in<Cc, CG, true, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_set = true;
  }
  static const int occ = getenv("DPMN_SKMLP_OCC") ? atoi(getenv("DPMN_SKMLP_OCC")) : 2;
  const int tiles = M / 16, ny = N / 96;
  static const int blocks = getenv("DPMN_SKMLP_BLOCKS") ? atoi(getenv("DPMN_SKMLP_BLOCKS")) : 256 * (occ >= 3 ? 3 : 2);
  int gx = blocks / ny;
  if (gx < 1) gx = 1;
  if (gx * 4 > tiles) gx = cdiv(tiles, 4);
  hipStream_t st = as_stream(stream);
  // algorithmic work of the two products; compulsory bytes: cat, feats, shortcut in, x1 and y out, both weight matrices
  ProfScope prof(PT_GEMM_WSTAT_LN, st, 2.0 * M * ((double)N * Cc + (double)Cc * CG),
                 4.0 * ((double)M * Cc * 4 + (double)M * N + (double)N * Cc + (double)Cc * CG));
#define SKMLP_LAUNCH(SAVE_, OCC_) hipLaunchKernelGGL((k_sk_mlp_in<Cc, CG, SAVE_, OCC_>), dim3(gx, ny), dim3(256), smem, st, cat, attn_vec, rows_per_image, \
                                                    w_head, b_head, feats, shortcut, x1, ln_w, ln_b, eps, w_fc1, b_fc1, y, M, N, v_out, n2_out, p_row, seed_row)
  if (x3_on(128))
    (void)dpmn_gemm::x3_launch_sk_mlp_in(cat, attn_vec, rows_per_image, w_head, b_head, feats, shortcut, x1, ln_w, ln_b, eps, w_fc1, b_fc1, y, M, N,
                                         v_out, n2_out, p_row, seed_row, gx, st);
  else if (v_out) { if (occ >= 3) SKMLP_LAUNCH(true, 3); else SKMLP_LAUNCH(true, 2); }
  else { if (occ >= 3) SKMLP_LAUNCH(false, 3); else SKMLP_LAUNCH(false, 2); }
#undef SKMLP_LAUNCH
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

int dpmn_sk_mlp_in_supported(int M, int rows_per_image, int C, int groups, int N) {
  return C == 96 && groups == 3 && N % 96 == 0 && M % 16 == 0 && rows_per_image % 16 == 0 && M >= 1024 && !g_dpmn_bf16;
}

int dpmn_pointwise_wgrad_f32(const float* dz, const float* g, float* dw, int B, int Ch, int L, dpmn_stream_t stream) {
  // dw[co][c] += sum_{b,s} dz[b][co][s] * g[b][c][s]  -- x = dz, w = g, reduction over (b, s)
  DPMN_REQUIRE(dz && g && dw && L % 32 == 0 && Ch % 4 == 0, "pointwise_wgrad: bad arguments");
  EpiArgs e{nullptr, nullptr, nullptr, nullptr, ACT_NONE, 0.f, 1};   // split over (b, s), atomic accumulation
  dim3 grid(cdiv(Ch, 64), cdiv(Ch, 96), 32);
  if (x3_on(8)) (void)dpmn_gemm::x3_launch_kloop(dz, L, g, L, dw, Ch, Ch, Ch, B * L, e, L,
                     (long)Ch * L, (long)Ch * L, grid, as_stream(stream));
  else hipLaunchKernelGGL(k_gemm_kloop, grid, dim3(256), 0, as_stream(stream), dz, L, g, L, dw, Ch, Ch, Ch, B * L, e, L,
                     (long)Ch * L, (long)Ch * L);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

// the same without atomics: the k splits store their (Ch, Ch) partial results in ws, added in split order (dpmn_rows_reduce_f32).
// Ch a multiple of 128, L of 32: 128 x 128 tiles with ~504 workgroups (k_gemm_kloop128: 56 splits for Ch = 384); otherwise the
// 64 x 96 k-loop with 32 splits.  Workspace: dpmn_pointwise_wgrad_det_bytes(Ch, L).
static int pw_wgrad_splits(int Ch, int L) {
  static const int nt_on = getenv("DPMN_PW_WGRAD_128") ? atoi(getenv("DPMN_PW_WGRAD_128")) : 1;
  if (!nt_on || Ch % 128 != 0 || L % 32 != 0) return 0;
  const int tiles = (Ch / 128) * (Ch / 128);
  int s = 504 / tiles;                 // just under the 512 resident workgroups (2 per CU)
  s &= ~7;                             // a multiple of 8: the XCD-local order
  return s >= 8 ? s : 0;
}

size_t dpmn_pointwise_wgrad_det_bytes(int Ch, int L) {
  const int s = pw_wgrad_splits(Ch, L);
  return (size_t)(s ? s : 32) * Ch * Ch * sizeof(float);
}

int dpmn_pointwise_wgrad_det_f32(const float* dz, const float* g, float* dw, int B, int Ch, int L, float* ws, size_t ws_bytes,
                                 dpmn_stream_t stream) {
  DPMN_REQUIRE(dz && g && dw && ws && L % 32 == 0 && Ch % 4 == 0, "pointwise_wgrad_det: bad arguments");
  const int s128 = pw_wgrad_splits(Ch, L);
  const int S = s128 ? s128 : 32;
  if ((size_t)S * Ch * Ch * sizeof(float) > ws_bytes) return dpmn_set_error(DPMN_ERR_WORKSPACE, "pointwise_wgrad_det: workspace too small");
  if (s128) {
    const int nchunks = B * (L / 32);
    ProfScope prof(PT_GEMM_KLOOP, as_stream(stream), 2.0 * Ch * (double)Ch * B * L, 4.0 * (2.0 * B * Ch * (double)L + (double)Ch * Ch));
    if (x3_on(32)) (void)dpmn_gemm::x3_launch_kloop128(dz, g, ws, Ch, Ch, L, nchunks, S, (long)Ch * L, (long)Ch * Ch, as_stream(stream));
    else hipLaunchKernelGGL(k_gemm_kloop128, dim3((Ch / 128) * (Ch / 128) * S), dim3(256), 0, as_stream(stream), dz, g, ws, Ch, Ch, L, nchunks, S,
                       (long)Ch * L, (long)Ch * Ch);
    DPMN_CHECK_LAUNCH();
    return dpmn_rows_reduce_f32(ws, dw, nullptr, Ch * Ch, 0, S, stream);
  }
  EpiArgs e{nullptr, nullptr, nullptr, nullptr, ACT_NONE, 0.f, 0, (long)Ch * Ch};
  dim3 grid(cdiv(Ch, 64), cdiv(Ch, 96), S);
  if (x3_on(8)) (void)dpmn_gemm::x3_launch_kloop(dz, L, g, L, ws, Ch, Ch, Ch, B * L, e, L,
                     (long)Ch * L, (long)Ch * L, grid, as_stream(stream));
  else hipLaunchKernelGGL(k_gemm_kloop, grid, dim3(256), 0, as_stream(stream), dz, L, g, L, ws, Ch, Ch, Ch, B * L, e, L,
                     (long)Ch * L, (long)Ch * L);
  DPMN_CHECK_LAUNCH();
  return dpmn_rows_reduce_f32(ws, dw, nullptr, Ch * Ch, 0, S, stream);
}

int dpmn_pointwise_f32(const float* g, const float* w, const float* bias, float* z, int B, int Ch, int L,
                       dpmn_stream_t stream) {
  DPMN_REQUIRE(g && w && bias && z && Ch % 128 == 0 && L % 128 == 0, "pointwise: Ch and L must be multiples of 128");
  static const int pw_bc = getenv("DPMN_PW_BC") ? atoi(getenv("DPMN_PW_BC")) : 192;
  ProfScope prof(PT_GEMM_PW, as_stream(stream), 2.0 * Ch * Ch * (double)L * B, 4.0 * (2.0 * B * Ch * (double)L + (double)Ch * Ch + Ch));
  if (x3_on(4)) {
    // fp32 product through six bf16 MFMAs of a three-term operand split (dpmn_set_compute_dtype(2), gemm_x3.hip)
    if (dpmn_gemm::x3_launch_pw(g, w, bias, z, B, Ch, L, as_stream(stream)) != 0) return dpmn_set_error(DPMN_ERR_LAUNCH, "pointwise: bf16x3 launch failed");
  } else if (g_dpmn_bf16 && Ch % 192 == 0)
    hipLaunchKernelGGL((k_gemm_pw_bf16<192>), dim3(L / 128, Ch / 192, B), dim3(256), 0, as_stream(stream), g, w, bias, z, Ch, L);
  else if (g_dpmn_bf16)
    hipLaunchKernelGGL((k_gemm_pw_bf16<128>), dim3(L / 128, Ch / 128, B), dim3(256), 0, as_stream(stream), g, w, bias, z, Ch, L);
  else if (Ch % 192 == 0 && pw_bc == 192)
    hipLaunchKernelGGL((k_gemm_pw<192>), dim3(L / 128, Ch / 192, B), dim3(256), 0, as_stream(stream), g, w, bias, z, Ch, L);
  else
    hipLaunchKernelGGL((k_gemm_pw<128>), dim3(L / 128, Ch / 128, B), dim3(256), 0, as_stream(stream), g, w, bias, z, Ch, L);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

}  // extern "C"
