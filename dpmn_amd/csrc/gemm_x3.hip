// "f32 via bf16x3" instantiations of the token GEMMs (dpmn_set_compute_dtype(2); the split is common.h x3_split2t): fp32 tensors in
// HBM, both operands split exactly into three bf16 planes on the way into LDS, six v_mfma_f32_16x16x32_bf16 per product (417 TFLOP/s
// of fp32-equivalent work at the bf16 peak against the 157 TFLOP/s of v_mfma_f32_16x16x4_f32), fp32 accumulation and epilogues.
// Kernels: the pointwise conv of the Mlp (pgrm.py:37), the k-loop GEMM (Mlp.fc2 pgrm.py:39, data gradients of fc1 / the pointwise conv)
// and the 128 x 128 k-loop of the pointwise conv's weight gradient.  gemm.hip routes a launch here when the mode is set.
#include "gemm_body.h"

namespace {

// ---------------------------------------------------------------------------------- pointwise GEMM
// z[b][co][s] = sum_c w[co][c] * g[b][c][s] + bias[co] on the raw (B, Ch, L) views (quirk Q2).  128 (s) x BC (co) tiles, 32-deep chunks.
// G is k-major in memory (rows = channels, s contiguous) and the MFMA wants 8 consecutive k per lane, so a plane of the chunk is staged
// as k-PAIRS: dword (p, s) = (bf16 G[2p][s], bf16 G[2p+1][s]) -- one ds_write_b128 per thread and pair, four conflict-free ds_read_b32
// per operand tile; W rows are k-contiguous: [BC][32 + 8] bf16 per plane, one ds_read_b128 per tile.
// Persistent: one 512-thread block per CU walks its tiles (768 tiles of 128 x 192 at B = 48, Ch = 384 = exactly 3 per CU; a
// 2-blocks-per-CU launch of the same tiles needs two rounds, the second half empty).  Both LDS buffers fit (2 x 71 KB): chunk
// k + 1 is split and stored while the other waves still multiply chunk k -- one barrier per chunk; its rows are loaded BEFORE the
// MFMA block of chunk k (scheduling barriers keep hipcc from sinking the loads to their use).  Wave (ws_, wc_) = 64 (s) x 48 (co)
// of the tile.  Tile order: the co blocks of one (image, s tile) run side by side on one XCD (they share the G rows).
template <int BC>
__global__ __launch_bounds__(512, 1) void k_gemm_pw_bf16x3(const float* g, const float* w, const float* __restrict__ bias, float* z, int Ch,
                                                            int L, int B) {
  constexpr int BS = 128, BK = 32, LDP = BS + 4, LDWB = BK + 16, NJ = BC / 64, TH = 512;      // W rows 96 bytes: conflict-free ds_read_b128 (conv_body.h)
  constexpr int GPL = (BK / 2) * LDP, WPL = BC * LDWB;            // one plane of G (dwords) / of W (bf16)
  constexpr int BUF = 3 * GPL + 3 * WPL / 2;                      // dwords per buffer
  constexpr int WQ = BC * BK / 4 / TH;                            // float4 of the W chunk per thread (3 at BC = 192, 2 at 128)
  static_assert(BC * BK / 4 % TH == 0 && WQ <= 3, "W chunk: whole float4 per thread");
  extern __shared__ __attribute__((aligned(16))) unsigned x3smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ns = L / BS, nco = Ch / BC;
  const int tiles = ns * nco * B;
  const int gp_ = tid >> 5, gcol = (tid & 31) * 4;      // pair row gp_ (channels 2 gp_, 2 gp_ + 1), 4 consecutive s
  const int wrow = tid >> 3, wcol = (tid & 7) * 4;      // W rows wrow (+64 per pass), 4 consecutive k
  const int ws_ = wave & 1, wc_ = wave >> 1;            // 2 (s) x 4 (co) waves
  const int lr = lane & 15, kq = lane >> 4;
  typedef unsigned u32x4__ __attribute__((ext_vector_type(4)));
  const int nk = Ch / BK;
  const int go = gp_ * LDP + gcol;                       // LDS offsets of this thread's stores
  const int wo = wrow * LDWB + wcol;
  const bool xcd_order = tiles % 8 == 0 && gridDim.x % 8 == 0;
  int round = 0;
  for (int t = blockIdx.x; t < tiles; t += gridDim.x, ++round) {
    // XCD x takes a contiguous eighth of the (image, s tile, co block) list
    const int lt = xcd_order ? (int)(blockIdx.x & 7) * (tiles / 8) + round * (int)(gridDim.x / 8) + (int)(blockIdx.x >> 3) : t;
    const int cb = lt % nco, sb = (lt / nco) % ns, b = lt / (nco * ns);
    const int s_blk = sb * BS, c_blk = cb * BC;
    const float* gsrc = g + (size_t)b * Ch * L + (size_t)(2 * gp_) * L + s_blk + gcol;      // + k0 * L
    const float* wsrc = w + (size_t)(c_blk + wrow) * Ch + wcol;                              // + k0 (+ 64 q rows)
    float* zb = z + (size_t)b * Ch * L;
    // two named register sets (a, b) hold the chunks kt + 1 and kt + 2: a set is loaded right after the split + store that frees it,
    // a whole barrier + MFMA block before it is consumed, so the split / LDS stores of chunk kt + 1 can be dealt into the MFMA block
    // of chunk kt (no scheduling barrier between them) instead of running, exposed, between the last MFMA and the barrier
    float4 ag0, ag1, aw0, aw1, aw2, bg0, bg1, bw0, bw1, bw2;
#define X3_GLOAD(P, k0)                                                                           \
    do {                                                                                          \
      P##g0 = *reinterpret_cast<const float4*>(gsrc + (size_t)(k0) * L);                          \
      P##g1 = *reinterpret_cast<const float4*>(gsrc + (size_t)((k0) + 1) * L);                    \
      P##w0 = *reinterpret_cast<const float4*>(wsrc + (k0));                                      \
      P##w1 = *reinterpret_cast<const float4*>(wsrc + (size_t)64 * Ch + (k0));                    \
      if (WQ > 2) P##w2 = *reinterpret_cast<const float4*>(wsrc + (size_t)128 * Ch + (k0));       \
    } while (0)
#define X3_WST(buf, q, V)                                                                         \
    do {                                                                                          \
      uint2 h2, m2, l2;                                                                           \
      x3_split2t(V.x, V.y, h2.x, m2.x, l2.x);                                                      \
      x3_split2t(V.z, V.w, h2.y, m2.y, l2.y);                                                      \
      unsigned short* d_ = reinterpret_cast<unsigned short*>((buf) + 3 * GPL) + wo + (q) * 64 * LDWB; \
      *reinterpret_cast<uint2*>(d_) = h2;                                                         \
      *reinterpret_cast<uint2*>(d_ + WPL) = m2;                                                   \
      *reinterpret_cast<uint2*>(d_ + 2 * WPL) = l2;                                               \
    } while (0)
#define X3_SSTORE(P, buf)                                                                         \
    do {                                                                                          \
      uint4 h4, m4, l4;                                                                           \
      x3_split2t(P##g0.x, P##g1.x, h4.x, m4.x, l4.x);                                              \
      x3_split2t(P##g0.y, P##g1.y, h4.y, m4.y, l4.y);                                              \
      x3_split2t(P##g0.z, P##g1.z, h4.z, m4.z, l4.z);                                              \
      x3_split2t(P##g0.w, P##g1.w, h4.w, m4.w, l4.w);                                              \
      *reinterpret_cast<uint4*>((buf) + go) = h4;                                                 \
      *reinterpret_cast<uint4*>((buf) + GPL + go) = m4;                                           \
      *reinterpret_cast<uint4*>((buf) + 2 * GPL + go) = l4;                                       \
      X3_WST(buf, 0, P##w0); X3_WST(buf, 1, P##w1);                                               \
      if (WQ > 2) X3_WST(buf, 2, P##w2);                                                          \
    } while (0)
#define X3_TERM(PA, PW)                                                               \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                 \
          _Pragma("unroll") for (int j = 0; j < NJ; ++j) acc[i][j] = mfma16_bf16(a3[i][PA], wf[PW][j], acc[i][j]);
#define X3_MMA(cur)                                                                               \
    do {                                                                                          \
      const unsigned short* Wb = reinterpret_cast<const unsigned short*>((cur) + 3 * GPL);        \
      bf16x8 wf[3][NJ];                                                                           \
      _Pragma("unroll") for (int pl = 0; pl < 3; ++pl)                                            \
        _Pragma("unroll") for (int j = 0; j < NJ; ++j)                                            \
          wf[pl][j] = *reinterpret_cast<const bf16x8*>(Wb + pl * WPL + (wc_ * (BC / 4) + j * 16 + lr) * LDWB + kq * 8); \
      const unsigned* gp = (cur) + (kq * 4) * LDP + ws_ * 64 + lr;                                \
      bf16x8 a3[4][3];                                                                            \
      _Pragma("unroll") for (int i = 0; i < 4; ++i)                                               \
        _Pragma("unroll") for (int pl = 0; pl < 3; ++pl) {                                        \
          const unsigned* q = gp + pl * GPL + i * 16;                                             \
          const u32x4__ av = {q[0], q[LDP], q[2 * LDP], q[3 * LDP]};                              \
          a3[i][pl] = __builtin_bit_cast(bf16x8, av);                                             \
        }                                                                                         \
      /* six terms, smallest first; consecutive MFMAs go to different accumulators */             \
      X3_TERM(2, 0) X3_TERM(1, 1) X3_TERM(0, 2) X3_TERM(1, 0) X3_TERM(0, 1) X3_TERM(0, 0)         \
    } while (0)
    f32x4 acc[4][NJ];   // [s tile][co tile]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    unsigned* buf0 = x3smem;
    unsigned* buf1 = x3smem + BUF;
    __syncthreads();                       // the previous tile's last chunk is consumed
    X3_GLOAD(a, 0);
    X3_SSTORE(a, buf0);
    X3_GLOAD(a, min(1, nk - 1) * BK);      // (past the end: clamped re-reads / a spare store, never a conditional load)
    X3_GLOAD(b, min(2, nk - 1) * BK);
    __syncthreads();
    for (int kt = 0; kt < nk; kt += 2) {
      X3_MMA(buf0);
      X3_SSTORE(a, buf1);                  // chunk kt + 1; buf1's readers passed the previous barrier
      X3_GLOAD(a, min(kt + 3, nk - 1) * BK);
      __syncthreads();
      if (kt + 1 >= nk) break;
      X3_MMA(buf1);
      X3_SSTORE(b, buf0);                  // chunk kt + 2
      X3_GLOAD(b, min(kt + 4, nk - 1) * BK);
      __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int co = c_blk + wc_ * (BC / 4) + j * 16 + lr;
      const float bv = bias[co];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int s_ = s_blk + ws_ * 64 + i * 16 + kq * 4;
        *reinterpret_cast<float4*>(zb + (size_t)co * L + s_) =
            make_float4(acc[i][j][0] + bv, acc[i][j][1] + bv, acc[i][j][2] + bv, acc[i][j][3] + bv);
      }
    }
  }
#undef X3_GLOAD
#undef X3_WST
#undef X3_SSTORE
#undef X3_TERM
#undef X3_MMA
}


// ---------------------------------------------------------------------------------- k-loop, 64 x 96 tiles
// k_gemm_kloop's contract (gemm.hip): y = epi(x w^T), x (M, K) / w (N, K) row-major with k contiguous, batched-K mode for reductions
// that run over several images, split launches that store at y + z * zstride.  One LDS buffer of three planes per operand
// (160 rows x 96 bytes x 3 = 46 KB: three blocks per CU), the next chunk's rows in registers during the MFMA block.
__global__ __launch_bounds__(256, 3) void k_gemm_kloop_x3(const float* __restrict__ x, int ldx, const float* __restrict__ w, int ldw,
                                                        float* __restrict__ y, int ldy, int M, int N, int K, EpiArgs e, int kb_len,
                                                        long x_bstride, long w_bstride) {
  constexpr int BM = 64, BN = 96, BK = 32, LDKB = BK + 16, XPL = BM * LDKB, WPL = BN * LDKB;      // 96-byte rows: conflict-free ds_read_b128 (conv_body.h)
  __shared__ __attribute__((aligned(16))) unsigned short Xs[3 * XPL];
  __shared__ __attribute__((aligned(16))) unsigned short Ws[3 * WPL];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (gridDim.z > 1 && (gridDim.z & 7) == 0) {       // XCD c takes the splits c, c + 8, ... (as k_gemm_kloop)
    const int tiles = gridDim.x * gridDim.y;
    const int lid = bx + gridDim.x * (by + gridDim.y * bz);
    const int c = lid & 7, j = lid >> 3;
    const int zq = j / tiles, t = j - zq * tiles;
    bz = c + 8 * zq; by = t / (int)gridDim.x; bx = t - by * (int)gridDim.x;
  }
  const int m_blk = bx * BM, n_blk = by * BN;
  y += (size_t)bz * e.zstride;
  const int lrow = tid >> 3, lcol = (tid & 7) * 4;
  const int xm0 = min(m_blk + lrow, M - 1), xm1 = min(m_blk + lrow + 32, M - 1);
  const int wn0 = min(n_blk + lrow, N - 1), wn1 = min(n_blk + lrow + 32, N - 1), wn2 = min(n_blk + lrow + 64, N - 1);
  float4 x0, x1, w0, w1, w2;
  auto gload = [&](int k0_) {
    int kk = k0_;
    size_t xoff = 0, woff = 0;
    if (kb_len > 0) { const int kb = kk / kb_len; kk -= kb * kb_len; xoff = (size_t)kb * x_bstride; woff = (size_t)kb * w_bstride; }
    x0 = *reinterpret_cast<const float4*>(x + xoff + (size_t)xm0 * ldx + kk + lcol);
    x1 = *reinterpret_cast<const float4*>(x + xoff + (size_t)xm1 * ldx + kk + lcol);
    w0 = *reinterpret_cast<const float4*>(w + woff + (size_t)wn0 * ldw + kk + lcol);
    w1 = *reinterpret_cast<const float4*>(w + woff + (size_t)wn1 * ldw + kk + lcol);
    w2 = *reinterpret_cast<const float4*>(w + woff + (size_t)wn2 * ldw + kk + lcol);
  };
  auto put = [&](unsigned short* base, int plane, int row, const float4& v) {
    uint2 h, m, l;
    x3_split4t(v, h, m, l);
    unsigned short* d_ = base + row * LDKB + lcol;
    *reinterpret_cast<uint2*>(d_) = h;
    *reinterpret_cast<uint2*>(d_ + plane) = m;
    *reinterpret_cast<uint2*>(d_ + 2 * plane) = l;
  };
  auto sstore = [&]() {
    put(Xs, XPL, lrow, x0); put(Xs, XPL, lrow + 32, x1);
    put(Ws, WPL, lrow, w0); put(Ws, WPL, lrow + 32, w1); put(Ws, WPL, lrow + 64, w2);
  };
  const int wm = wave & 1, wn = wave >> 1;
  const int lr = lane & 15, kq = lane >> 4;
  f32x4 acc[3][2];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int nk_all = K / BK;
  const int cps = (nk_all + gridDim.z - 1) / gridDim.z;
  const int kt0 = bz * cps;
  const int nk = min(nk_all, kt0 + cps);
  if (kt0 >= nk) return;
  gload(kt0 * BK);
  sstore();
  __syncthreads();
  const unsigned short* xa = Xs + (wm * 32 + lr) * LDKB + kq * 8;
  const unsigned short* wa = Ws + (wn * 48 + lr) * LDKB + kq * 8;
  for (int kt = kt0; kt < nk; ++kt) {
    gload(min(kt + 1, nk - 1) * BK);                 // unconditional: the refill past the end re-reads the last chunk
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 xf[3][2], wf[3][3];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
      for (int j = 0; j < 2; ++j) xf[pl][j] = *reinterpret_cast<const bf16x8*>(xa + pl * XPL + j * 16 * LDKB);
#pragma unroll
      for (int i = 0; i < 3; ++i) wf[pl][i] = *reinterpret_cast<const bf16x8*>(wa + pl * WPL + i * 16 * LDKB);
    }
#define KL_X3_TERM(PX, PW)                                                   \
    _Pragma("unroll") for (int i = 0; i < 3; ++i)                            \
      _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[i][j] = mfma16_bf16(wf[PW][i], xf[PX][j], acc[i][j]);
    KL_X3_TERM(2, 0) KL_X3_TERM(1, 1) KL_X3_TERM(0, 2) KL_X3_TERM(1, 0) KL_X3_TERM(0, 1) KL_X3_TERM(0, 0)
#undef KL_X3_TERM
    __syncthreads();                                 // every wave has read chunk kt
    sstore();
    __syncthreads();
  }
  // Mlp.fc2 + shortcut (+ Dropout / DropPath): the interior-tile epilogue of k_gemm_kloop, same element order and mask indices
  if (e.bias && e.res1 && !e.res2 && !e.colsum && !e.atomic && e.act == ACT_NONE && m_blk + BM <= M && n_blk + BN <= N && (ldy & 3) == 0) {
    const int lm = lane & 15, lq = lane >> 4;
    const float ike = 1.0f / (1.0f - e.p_elem), ikr = 1.0f / (1.0f - e.p_row);
    float4 rr[3][2];
#pragma unroll
    for (int nt = 0; nt < 3; ++nt)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
        rr[nt][mt] = *reinterpret_cast<const float4*>(e.res1 + (size_t)(m_blk + wm * 32 + mt * 16 + lm) * ldy + n_blk + wn * 48 + nt * 16 + lq * 4);
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
  epilogue<3, 2>(acc, m_blk + wm * 32, n_blk + wn * 48, M, N, ldy, y, e, nullptr, BN, n_blk);
}

// ---------------------------------------------------------------------------------- k-loop, 128 x 128 tiles (pointwise-conv weight gradient)
// k_gemm_kloop128's contract: dW (M, N) = sum_b X_b (M, L) . Y_b (N, L)^T over the raw (B, Ch, L) views, the reduction (b, s) cut into
// `splits` contiguous ranges of 32-wide chunks, split z stores its tile at y + z * zstride.  One LDS buffer (74 KB), two blocks per CU.
__global__ __launch_bounds__(256, 2) void k_gemm_kloop128_x3(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y,
                                                              int M, int N, int L, int nchunks, int splits, long bstride, long zstride) {
  constexpr int BM = 128, BN = 128, BK = 32, LDKB = BK + 16, PL = BM * LDKB;
  __shared__ __attribute__((aligned(16))) unsigned short Xs[3 * PL];
  __shared__ __attribute__((aligned(16))) unsigned short Ws[3 * PL];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_m = M / BM, tiles = tiles_m * (N / BN);
  int lid = blockIdx.x, bz, t;
  if ((splits & 7) == 0) { const int c = lid & 7, j = lid >> 3; const int zq = j / tiles; t = j - zq * tiles; bz = c + 8 * zq; }
  else { bz = lid / tiles; t = lid - bz * tiles; }
  const int m_blk = (t % tiles_m) * BM, n_blk = (t / tiles_m) * BN;
  y += (size_t)bz * zstride;
  const int lrow = tid >> 3, lcol = (tid & 7) * 4;
  const int cpi = L / BK;                                    // chunks per image
  float4 xr[4], wr[4];
  auto gload = [&](int kt_) {
    const int kb = kt_ / cpi, kk = (kt_ - kb * cpi) * BK + lcol;
    const float* xb = x + (size_t)kb * bstride + (size_t)(m_blk + lrow) * L + kk;
    const float* wb = w + (size_t)kb * bstride + (size_t)(n_blk + lrow) * L + kk;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      xr[p] = *reinterpret_cast<const float4*>(xb + (size_t)(32 * p) * L);
      wr[p] = *reinterpret_cast<const float4*>(wb + (size_t)(32 * p) * L);
    }
  };
  auto put = [&](unsigned short* base, int row, const float4& v) {
    uint2 h, m, l;
    x3_split4t(v, h, m, l);
    unsigned short* d_ = base + row * LDKB + lcol;
    *reinterpret_cast<uint2*>(d_) = h;
    *reinterpret_cast<uint2*>(d_ + PL) = m;
    *reinterpret_cast<uint2*>(d_ + 2 * PL) = l;
  };
  auto sstore = [&]() {
#pragma unroll
    for (int p = 0; p < 4; ++p) { put(Xs, lrow + 32 * p, xr[p]); put(Ws, lrow + 32 * p, wr[p]); }
  };
  const int wm = wave & 1, wn = wave >> 1;
  const int lr = lane & 15, kq = lane >> 4;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // split bz takes the chunks [kt0, nk): equal shares up to one chunk
  const int kt0 = (int)((long)nchunks * bz / splits), nk = (int)((long)nchunks * (bz + 1) / splits);
  if (kt0 < nk) {
    gload(kt0);
    sstore();
    __syncthreads();
    const unsigned short* xa = Xs + (wm * 64 + lr) * LDKB + kq * 8;
    const unsigned short* wa = Ws + (wn * 64 + lr) * LDKB + kq * 8;
    for (int kt = kt0; kt < nk; ++kt) {
      gload(min(kt + 1, nk - 1));
      __builtin_amdgcn_sched_barrier(0);
      bf16x8 w0[4], w1[4], w2[4], xp[4], xq[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        w0[i] = *reinterpret_cast<const bf16x8*>(wa + i * 16 * LDKB);
        w1[i] = *reinterpret_cast<const bf16x8*>(wa + PL + i * 16 * LDKB);
        w2[i] = *reinterpret_cast<const bf16x8*>(wa + 2 * PL + i * 16 * LDKB);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) xp[j] = *reinterpret_cast<const bf16x8*>(xa + j * 16 * LDKB);
#pragma unroll
      for (int j = 0; j < 4; ++j) xq[j] = *reinterpret_cast<const bf16x8*>(xa + PL + j * 16 * LDKB);
#define K8_X3_TERM(XF, WF)                                                   \
      _Pragma("unroll") for (int i = 0; i < 4; ++i)                          \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) acc[i][j] = mfma16_bf16(WF[i], XF[j], acc[i][j]);
      K8_X3_TERM(xp, w2) K8_X3_TERM(xp, w1) K8_X3_TERM(xp, w0)
#pragma unroll
      for (int j = 0; j < 4; ++j) xp[j] = *reinterpret_cast<const bf16x8*>(xa + 2 * PL + j * 16 * LDKB);
      K8_X3_TERM(xq, w1) K8_X3_TERM(xq, w0)
      K8_X3_TERM(xp, w0)
#undef K8_X3_TERM
      __syncthreads();
      sstore();
      __syncthreads();
    }
  }
  // lane holds y[m = .. + lr][n = .. + 4 kq + r]  (A operand = W rows, B operand = X rows, as k_gemm_kloop128)
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i)
      *reinterpret_cast<f32x4*>(y + (size_t)(m_blk + wm * 64 + j * 16 + lr) * N + n_blk + wn * 64 + i * 16 + kq * 4) = acc[i][j];
}

int cu_count() {
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    hipDeviceProp_t prop;
    n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
  }
  return n_cu;
}

}  // namespace

namespace dpmn_gemm {
int x3_launch_kloop(const float* x, int ldx, const float* w, int ldw, float* y, int ldy, int M, int N, int K, const EpiArgs& e, int kb_len,
                    long x_bstride, long w_bstride, dim3 grid, hipStream_t st) {
  hipLaunchKernelGGL(k_gemm_kloop_x3, grid, dim3(256), 0, st, x, ldx, w, ldw, y, ldy, M, N, K, e, kb_len, x_bstride, w_bstride);
  return 0;
}
int x3_launch_kloop128(const float* x, const float* w, float* y, int M, int N, int L, int nchunks, int splits, long bstride, long zstride,
                       hipStream_t st) {
  hipLaunchKernelGGL(k_gemm_kloop128_x3, dim3((M / 128) * (N / 128) * splits), dim3(256), 0, st, x, w, y, M, N, L, nchunks, splits, bstride, zstride);
  return 0;
}
int x3_launch_pw(const float* g, const float* w, const float* bias, float* z, int B, int Ch, int L, hipStream_t st) {
  constexpr int LDP_ = 132, LDWB_ = 48;
  const int bc = Ch % 192 == 0 ? 192 : 128;
  const size_t smem = (size_t)2 * 3 * (16 * LDP_ * 4 + bc * LDWB_ * 2);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_pw_bf16x3<192>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 3 * (16 * LDP_ * 4 + 192 * LDWB_ * 2));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_pw_bf16x3<128>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 3 * (16 * LDP_ * 4 + 128 * LDWB_ * 2));
    attr_set = true;
  }
  const long tiles = (long)(L / 128) * (Ch / bc) * B;
  const int n_cu = cu_count();
  const unsigned grid = (unsigned)(tiles < n_cu ? tiles : n_cu);
  if (bc == 192) hipLaunchKernelGGL((k_gemm_pw_bf16x3<192>), dim3(grid), dim3(512), smem, st, g, w, bias, z, Ch, L, B);
  else hipLaunchKernelGGL((k_gemm_pw_bf16x3<128>), dim3(grid), dim3(512), smem, st, g, w, bias, z, Ch, L, B);
  return 0;
}
}  // namespace dpmn_gemm
