// "f32 via bf16x3" forms of the whole-K token GEMMs whose rows go straight from global memory into the MFMA B operand (gemm.hip
// k_gemm_rowreg, k_sk_mlp_in; dpmn_set_compute_dtype(2)): y = x W^T for K = 96 / 192 with the LayerNorm prologue and the bias / GELU /
// residual / column-sum epilogues of the fp32 kernels.
//
// The WEIGHTS are split once per block while they are staged (three bf16 planes in LDS; gamma of a folded LayerNorm goes in before the
// split), the token rows once per tile in registers -- per 16-token tile and 96 outputs 12 K / 32 splits per lane against 108 bf16
// MFMAs at K = 96 (fp32 kernel: 144 MFMAs of twice the latency).  k index of the bf16 MFMA: lane (lr, kq) supplies, for the 32-chunk C2,
// the channels {32 C2 + 4 kq + r} and {32 C2 + 16 + 4 kq + r}, r = 0 .. 3 -- exactly the two float4 the fp32 kernel's lane loads for its
// chunks 2 C2 and 2 C2 + 1 (and, in k_sk_mlp_in, the accumulator registers of output tiles 2 C2, 2 C2 + 1 of the first product) -- so
// the global loads are unchanged and the weight planes are stored PERMUTED: channel 16 h + 4 q + r of a chunk sits at 8 q + 4 h + r, a
// lane's eight values are 16 contiguous bytes.  Plane rows are K + 8 halves (an odd multiple of 16 bytes): ds_read_b128 conflict-free.
#include <cstdlib>
#include "gemm_body.h"

namespace {
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// the lane's eight channels of chunk C2 (two float4) -> the three plane registers of the B operand
__device__ __forceinline__ void split_rows8(const f32x4& lo, const f32x4& hi, bf16x8& h, bf16x8& m, bf16x8& l) {
  u32x4 H, M_, L;
  unsigned a, b, c;
  x3_split2t(lo[0], lo[1], a, b, c); H[0] = a; M_[0] = b; L[0] = c;
  x3_split2t(lo[2], lo[3], a, b, c); H[1] = a; M_[1] = b; L[1] = c;
  x3_split2t(hi[0], hi[1], a, b, c); H[2] = a; M_[2] = b; L[2] = c;
  x3_split2t(hi[2], hi[3], a, b, c); H[3] = a; M_[3] = b; L[3] = c;
  h = __builtin_bit_cast(bf16x8, H); m = __builtin_bit_cast(bf16x8, M_); l = __builtin_bit_cast(bf16x8, L);
}

// four consecutive channels c4 .. c4 + 3 of weight row r -> their permuted place in the three planes (8 bytes each)
template <int K>
__device__ __forceinline__ void stage_w4(unsigned short* Wb, int r, int c4, const float4& v) {
  constexpr int LDB = K + 8, PL = 96 * LDB;
  uint2 h, m, l;
  x3_split4t(v, h, m, l);
  const int pos = (c4 & ~31) + 8 * ((c4 >> 2) & 3) + 4 * ((c4 >> 4) & 1);
  unsigned short* d = Wb + r * LDB + pos;
  *reinterpret_cast<uint2*>(d) = h;
  *reinterpret_cast<uint2*>(d + PL) = m;
  *reinterpret_cast<uint2*>(d + 2 * PL) = l;
}

// acc[nt] += W[16 nt .. + 15][chunk C2] . rows, six bf16 MFMAs per output tile; NT output tiles from `nt0`
template <int K, int NT>
__device__ __forceinline__ void mma_chunk(const unsigned short* Wb, int lr, int kq, int C2, const bf16x8& bh, const bf16x8& bm, const bf16x8& bl,
                                          f32x4 (&acc)[NT]) {
  constexpr int LDB = K + 8, PL = 96 * LDB;
  const unsigned short* wa = Wb + lr * LDB + 32 * C2 + 8 * kq;
  // product term OUTER, output tile inner: consecutive MFMAs write different accumulators (six back-to-back MFMAs into one accumulator
  // wait for each other's result: the first form of this loop spent half its matrix time in that chain)
  bf16x8 wh[NT], wm[NT], wl[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) wl[nt] = *reinterpret_cast<const bf16x8*>(wa + 16 * nt * LDB + 2 * PL);
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) wh[nt] = *reinterpret_cast<const bf16x8*>(wa + 16 * nt * LDB);
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) wm[nt] = *reinterpret_cast<const bf16x8*>(wa + 16 * nt * LDB + PL);
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma16_bf16(wl[nt], bh, acc[nt]);
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma16_bf16(wh[nt], bl, acc[nt]);
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma16_bf16(wm[nt], bm, acc[nt]);
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma16_bf16(wm[nt], bh, acc[nt]);
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma16_bf16(wh[nt], bm, acc[nt]);
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma16_bf16(wh[nt], bh, acc[nt]);
}

// ---------------------------------------------------------------------------------- k_gemm_rowreg in mode 2
// Same block / wave / tile walk, prologues and epilogues as gemm.hip k_gemm_rowreg (PRO_NONE / PRO_LN, RR_EPI 1 .. 5); one LDS tenant
// more: 96 x (K + 8) x 3 halves of weight planes (59.9 KB at K = 96: two blocks per CU; 115 KB at K = 192: one).
template <int K, int PRO, int EPI>
__global__ __launch_bounds__(256, K <= 96 ? 2 : 1) void k_gemm_rowreg_x3(const float* __restrict__ x, int ldx, const float* __restrict__ w,
                                                                         float* __restrict__ y, int ldy, int M, int N, ProArgs p, EpiArgs e) {
  constexpr int KC = K / 16, K2 = K / 32, LDB = K + 8, BN = 96, NT = 6, KV = K / 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  unsigned short* Wb = reinterpret_cast<unsigned short*>(smem);      // [3][96][LDB]
  float* pb = smem + 3 * BN * LDB / 2;                               // [96] bias (b' under PRO_LN), [96] rowsum(W')
  float* scr = pb + 2 * BN;                                          // PRO_LN: [96][KV][2] partial sums of the fold
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, kq = lane >> 4;
  const int n_blk = blockIdx.y * BN;
  const int tiles = M / 16;
  const int stride = gridDim.x * 4;
  int tile = (blockIdx.x * 4 + wave) * (EPI == 4 ? 2 : 1);

  constexpr int WL = (BN * KV + 255) / 256;
  float4 wv[WL];
#pragma unroll
  for (int u = 0; u < WL; ++u) {
    const int i = min(tid + u * 256, BN * KV - 1);
    wv[u] = *reinterpret_cast<const float4*>(w + (size_t)(n_blk + i / KV) * K + (i % KV) * 4);
  }
  f32x4 xr[KC];
  auto load_rows = [&](int t_) {
    const size_t m = (size_t)(t_ < tiles ? t_ : tiles - 1) * 16 + lr;      // clamped, never predicated
#pragma unroll
    for (int c = 0; c < KC; ++c) xr[c] = *reinterpret_cast<const f32x4*>(x + m * ldx + 16 * c + 4 * kq);
  };
  load_rows(tile);
#pragma unroll
  for (int u = 0; u < WL; ++u) {
    const int i = tid + u * 256;
    if (i < BN * KV) {
      const int r = i / KV, c4 = (i % KV) * 4;
      float4 v = wv[u];
      if (PRO == PRO_LN) {        // the fold of k_gemm_rowreg: gamma into the weights, W beta and rowsum(W') in the same order
        const float4 gm = *reinterpret_cast<const float4*>(p.ln_w + c4), bt = *reinterpret_cast<const float4*>(p.ln_b + c4);
        const float4 wb = make_float4(v.x * bt.x, v.y * bt.y, v.z * bt.z, v.w * bt.w);
        v = make_float4(v.x * gm.x, v.y * gm.y, v.z * gm.z, v.w * gm.w);
        scr[(r * KV + c4 / 4) * 2] = (v.x + v.y) + (v.z + v.w);
        scr[(r * KV + c4 / 4) * 2 + 1] = (wb.x + wb.y) + (wb.z + wb.w);
      }
      stage_w4<K>(Wb, r, c4, v);
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
    float mean = 0.f, rstd = 1.f;
    if (PRO == PRO_LN) {                   // two-pass row statistics over the 4 kq partners of the row (like nn.LayerNorm), on the fp32 rows
      float s0 = 0.f, s1 = 0.f;
#pragma unroll
      for (int c = 0; c < KC; ++c) { s0 += xr[c][0] + xr[c][1]; s1 += xr[c][2] + xr[c][3]; }
      float s_ = s0 + s1;
      s_ += xshfl<16>(s_); s_ += xshfl<32>(s_);
      mean = s_ * (1.0f / K);
      float q0 = 0.f, q1 = 0.f, q2 = 0.f, q3 = 0.f;
#pragma unroll
      for (int c = 0; c < KC; ++c) {
        const float d0 = xr[c][0] - mean, d1 = xr[c][1] - mean, d2 = xr[c][2] - mean, d3 = xr[c][3] - mean;
        q0 = fmaf(d0, d0, q0); q1 = fmaf(d1, d1, q1); q2 = fmaf(d2, d2, q2); q3 = fmaf(d3, d3, q3);
      }
      float q = (q0 + q1) + (q2 + q3);
      q += xshfl<16>(q); q += xshfl<32>(q);
      rstd = 1.0f / sqrtf(q * (1.0f / K) + p.eps);
    }
    bf16x8 bh[K2], bm[K2], bl[K2];
#pragma unroll
    for (int c2 = 0; c2 < K2; ++c2) split_rows8(xr[2 * c2], xr[2 * c2 + 1], bh[c2], bm[c2], bl[c2]);
    // the next tile's rows are requested NOW (the fp32 copies of this tile are dead): they fly during this tile's MFMAs and epilogue
    __builtin_amdgcn_sched_barrier(0);
    load_rows(next_tile);
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c2 = 0; c2 < K2; ++c2) {
      mma_chunk<K, NT>(Wb, lr, kq, c2, bh[c2], bm[c2], bl[c2], acc);
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

// ---------------------------------------------------------------------------------- k_sk_mlp_in in mode 2
// The first product (proj_head, K = 32: a quarter of the MFMAs) stays on v_mfma_f32_16x16x4_f32 with the fp32 kernel's instructions --
// x1 is bitwise the fp32 kernel's (and k_gemm_rowreg<CG, PRO_SKSEL, 3>'s) -- and its fp32 weights keep their 13.8 KB; fc1 (K = 96) runs on
// the planes: 59.9 + 13.8 + 1.2 KB = two blocks per CU like the fp32 kernel at OCC = 2.
template <int C, int CG, bool SAVE>
__global__ __launch_bounds__(256, 2) void k_sk_mlp_in_x3(const float* __restrict__ cat, const float* __restrict__ sel, int rows_per_image,
                                                          const float* __restrict__ w_head, const float* __restrict__ b_head,
                                                          const float* __restrict__ feats, const float* __restrict__ shortcut, float* __restrict__ x1,
                                                          const float* __restrict__ ln_w, const float* __restrict__ ln_b, float eps,
                                                          const float* __restrict__ w_fc1, const float* __restrict__ b_fc1, float* __restrict__ y,
                                                          int M, int N, float* __restrict__ v_out, float* __restrict__ n2_out, float p_row,
                                                          unsigned long long seed_row) {
  constexpr int KC = C / 16, K2 = C / 32, KV = C / 4, LDB = C + 8, BN = 96, NT = 6, G = C / CG, HC = CG / 16, LDH = CG + PAD;
  static_assert(C == 96 && G == 3, "built for dim 96, three window groups");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  unsigned short* Wb = reinterpret_cast<unsigned short*>(smem);      // [3][96][LDB] fc1 rows of this column group, gamma folded in (eval)
  float* pb = smem + 3 * BN * LDB / 2;                               // [96] b' = b + W beta, [96] rowsum(W'), [96] b_head
  float* Wh = pb + 3 * BN;                                           // [96][LDH] proj_head, fp32 (the fold's scratch lives here first)
  float* scr = Wh;                                                   // [96][KV][2] partial sums of rowsum(W') and W beta: 18.4 KB > Wh's 13.8
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
    const size_t m = (size_t)(t_ < tiles ? t_ : tiles - 1) * 16 + lr;
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int c = 0; c < HC; ++c) xr[g][c] = *reinterpret_cast<const f32x4*>(cat + m * C + g * CG + 16 * c + 4 * kq);
  };
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
      if (!SAVE) {        // eval: LayerNorm folded into the weights (gamma) and the bias (W beta), the sums of k_sk_mlp_in in its order
        const float4 gm = *reinterpret_cast<const float4*>(ln_w + c4), bt = *reinterpret_cast<const float4*>(ln_b + c4);
        const float4 wb = make_float4(v.x * bt.x, v.y * bt.y, v.z * bt.z, v.w * bt.w);
        v = make_float4(v.x * gm.x, v.y * gm.y, v.z * gm.z, v.w * gm.w);
        scr[(r * KV + c4 / 4) * 2] = (v.x + v.y) + (v.z + v.w);
        scr[(r * KV + c4 / 4) * 2 + 1] = (wb.x + wb.y) + (wb.z + wb.w);
      }
      stage_w4<C>(Wb, r, c4, v);
    }
  }
  __syncthreads();
  if (tid < BN) {
    float bb = b_fc1 ? b_fc1[n_blk + tid] : 0.f, cw = 0.f;
    if (!SAVE)
      for (int k = 0; k < KV; ++k) { cw += scr[(tid * KV + k) * 2]; bb += scr[(tid * KV + k) * 2 + 1]; }      // fixed order
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
    // ---- x1 = proj_head(sel) + b_head + feats + shortcut   (fp32 MFMAs: the fp32 kernel's x1, bit for bit)
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
    if (SAVE) {          // training forward: the normalised row feeds the MFMAs with the unfolded weights (k_sk_mlp_in)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const f32x4 gm = *reinterpret_cast<const f32x4*>(ln_w + 16 * nt + 4 * kq), bt = *reinterpret_cast<const f32x4*>(ln_b + 16 * nt + 4 * kq);
#pragma unroll
        for (int r = 0; r < 4; ++r) x1r[nt][r] = (x1r[nt][r] - mean) * rstd * gm[r] + bt[r];
        if (write_x1) *reinterpret_cast<f32x4*>(n2_out + roff + 16 * nt) = x1r[nt];
      }
    }
    // ---- y = rstd * (W' x1 - mean * rowsum(W')) + b'   (the accumulator tiles 2 C2, 2 C2 + 1 of x1 ARE the lane's channels of chunk C2)
    f32x4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c2 = 0; c2 < K2; ++c2) {
      bf16x8 bh, bm, bl;
      split_rows8(x1r[2 * c2], x1r[2 * c2 + 1], bh, bm, bl);
      mma_chunk<C, NT>(Wb, lr, kq, c2, bh, bm, bl, acc);
      __builtin_amdgcn_sched_barrier(0);
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
int launch_rr(const float* x, int ldx, const float* w, float* y, int ldy, int M, int N, const ProArgs& p, const EpiArgs& e, int gx, hipStream_t st) {
  const size_t smem = (size_t)3 * 96 * (K + 8) * 2 + (size_t)(2 * 96 + (PRO == PRO_LN ? 96 * (K / 4) * 2 : 0)) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_rowreg_x3<K, PRO, EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_set = true;
  }
  hipLaunchKernelGGL((k_gemm_rowreg_x3<K, PRO, EPI>), dim3(gx, N / 96), dim3(256), smem, st, x, ldx, w, y, ldy, M, N, p, e);
  return 0;
}

template <int K, int PRO>
int launch_rr_epi(int epi, const float* x, int ldx, const float* w, float* y, int ldy, int M, int N, const ProArgs& p, const EpiArgs& e, int gx,
                  hipStream_t st) {
  switch (epi) {
    case 1: return launch_rr<K, PRO, 1>(x, ldx, w, y, ldy, M, N, p, e, gx, st);
    case 2: return launch_rr<K, PRO, 2>(x, ldx, w, y, ldy, M, N, p, e, gx, st);
    case 3: return launch_rr<K, PRO, 3>(x, ldx, w, y, ldy, M, N, p, e, gx, st);
    case 5: return launch_rr<K, PRO, 5>(x, ldx, w, y, ldy, M, N, p, e, gx, st);
    case 4:
      if constexpr (PRO == PRO_NONE) return launch_rr<K, PRO, 4>(x, ldx, w, y, ldy, M, N, p, e, gx, st);
      return -1;
    default: return -1;
  }
}
}  // namespace

namespace dpmn_gemm {
int x3_launch_rowreg(int K, int pro, int epi, const float* x, int ldx, const float* w, float* y, int ldy, int M, int N, const ProArgs& p,
                     const EpiArgs& e, int gx, hipStream_t st) {
  if (K == 96 && pro == PRO_NONE) return launch_rr_epi<96, PRO_NONE>(epi, x, ldx, w, y, ldy, M, N, p, e, gx, st);
  if (K == 96 && pro == PRO_LN) return launch_rr_epi<96, PRO_LN>(epi, x, ldx, w, y, ldy, M, N, p, e, gx, st);
  if (K == 192 && pro == PRO_NONE) return launch_rr_epi<192, PRO_NONE>(epi, x, ldx, w, y, ldy, M, N, p, e, gx, st);
  if (K == 192 && pro == PRO_LN) return launch_rr_epi<192, PRO_LN>(epi, x, ldx, w, y, ldy, M, N, p, e, gx, st);
  return -1;
}

int x3_launch_sk_mlp_in(const float* cat, const float* sel, int rows_per_image, const float* w_head, const float* b_head, const float* feats,
                        const float* shortcut, float* x1, const float* ln_w, const float* ln_b, float eps, const float* w_fc1, const float* b_fc1,
                        float* y, int M, int N, float* v_out, float* n2_out, float p_row, unsigned long long seed_row, int gx, hipStream_t st) {
  constexpr int Cc = 96, CG = 32;
  const size_t smem = (size_t)3 * 96 * (Cc + 8) * 2 + (size_t)(3 * 96 + 96 * (Cc / 4) * 2) * sizeof(float);      // planes + pb + the fold's scratch (> proj_head)
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sk_mlp_in_x3<Cc, CG, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sk_mlp_in_x3<Cc, CG, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_set = true;
  }
  if (v_out)
    hipLaunchKernelGGL((k_sk_mlp_in_x3<Cc, CG, true>), dim3(gx, N / 96), dim3(256), smem, st, cat, sel, rows_per_image, w_head, b_head, feats, shortcut,
                       x1, ln_w, ln_b, eps, w_fc1, b_fc1, y, M, N, v_out, n2_out, p_row, seed_row);
  else
    hipLaunchKernelGGL((k_sk_mlp_in_x3<Cc, CG, false>), dim3(gx, N / 96), dim3(256), smem, st, cat, sel, rows_per_image, w_head, b_head, feats, shortcut,
                       x1, ln_w, ln_b, eps, w_fc1, b_fc1, y, M, N, v_out, n2_out, p_row, seed_row);
  return 0;
}
}  // namespace dpmn_gemm
