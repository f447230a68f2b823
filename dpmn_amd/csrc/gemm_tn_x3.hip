// "f32 via bf16x3" form of k_gemm_tn_reg (backward.hip): dW[N][K] (+ db[N]) partials of every nn.Linear weight gradient, the reduction
// over the token rows on six v_mfma_f32_16x16x32_bf16 per 16 x 16 tile and 32 rows (common.h x3_split2t: exact three-term split).
//
// Same tiling as the fp32 kernel -- 96 x 96 block tile, 2 x 2 waves of 48 x 48, MFMA tile i of a wave = the columns {3 lr + i}, no LDS,
// no barrier, both operands straight from global memory with 12-byte buffer loads (16 lanes = 192 contiguous bytes of one row) -- but a
// step covers 32 rows: the bf16 MFMA contracts k = 8 kq .. 8 kq + 7 per lane, and WHICH row of the step stands at k is free as long as
// both operands agree, so lane (lr, kq) takes the rows {4 t + kq, t = 0 .. 7}: load t of a step reads four consecutive rows across the
// kq groups exactly like a step of the fp32 kernel, and the pair (t, t + 1) of one column is one split (11 vector instructions -> the
// three dwords of bf16 pairs the A / B operand registers want).  Per step and lane: 16 loads, 24 splits (264 VALU), 54 MFMAs.
// Rows past the block's range lie beyond the buffer's num_records (zeros: h = m = l = 0), columns past N / K likewise.
#include <cstdlib>
#include "gemm_tn.h"

namespace {
typedef float f32x3 __attribute__((ext_vector_type(3)));
typedef int i32x3_ __attribute__((ext_vector_type(3)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct Planes { u32x4 h, m, l; };      // one operand column's 8 rows as bf16 pairs

__device__ __forceinline__ Planes split_col(const f32x3 (&v)[8], int i) {
  Planes p;
  unsigned h, m, l;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    x3_split2t(v[2 * q][i], v[2 * q + 1][i], h, m, l);
    p.h[q] = h; p.m[q] = m; p.l[q] = l;
  }
  return p;
}

template <int D>      // D: 32-row steps in flight
__device__ __forceinline__ void tn_reg_x3_body(const float* __restrict__ dy, const float* __restrict__ x, int M, int N, int K, int rows_per_block,
                                               const float* __restrict__ db, float* __restrict__ part, int gx, int gy, int gz, int lid) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave & 1, wk = wave >> 1;
  const int lr = lane & 15, kq = lane >> 4;
  // XCD c takes the row splits z = c, c + 8, ... and runs their tiles back to back (backward.hip tn_reg_body)
  int bx = lid % gx, by = (lid / gx) % gy, bz = lid / (gx * gy);
  if ((gz & 7) == 0 && gx * gy > 1) {
    const int tiles = gx * gy;
    const int c = lid & 7, j = lid >> 3;
    const int zq = j / tiles, t = j - zq * tiles;
    bz = c + 8 * zq; by = t / gx; bx = t - by * gx;
  }
  const int n_w = bx * 96 + wn * 48, k_w = by * 96 + wk * 48;
  const int m_lo = bz * rows_per_block;
  const int nrows = min(M, m_lo + rows_per_block) - m_lo;
  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dy) + (size_t)m_lo * N, 0, nrows * N * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x) + (size_t)m_lo * K, 0, nrows * K * 4, 0x00020000);
  const int offy = n_w + 3 * lr < N ? (kq * N + n_w + 3 * lr) * 4 : (int)0x80000000;
  const int offx = k_w + 3 * lr < K ? (kq * K + k_w + 3 * lr) * 4 : (int)0x80000000;
  const int nsteps = (nrows + 31) >> 5;
  f32x4 acc[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const bool with_db = db != nullptr && by == 0 && wk == 0;     // wave-uniform
  f32x3 sb = (f32x3){0.f, 0.f, 0.f};
  f32x3 a[D][8], b[D][8];
  auto load = [&](int d, int step) {       // steps past the range: beyond num_records -> zeros
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      a[d][t] = __builtin_bit_cast(f32x3, __builtin_amdgcn_raw_buffer_load_b96(yrs, offy, (step * 32 + 4 * t) * N * 4, 0));
      b[d][t] = __builtin_bit_cast(f32x3, __builtin_amdgcn_raw_buffer_load_b96(xrs, offx, (step * 32 + 4 * t) * K * 4, 0));
    }
  };
#pragma unroll
  for (int d = 0; d < D; ++d) load(d, d);
  for (int s0 = 0; s0 < nsteps; s0 += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      Planes pb[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) pb[j] = split_col(b[d], j);
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const Planes pa = split_col(a[d], i);
        const bf16x8 ah = __builtin_bit_cast(bf16x8, pa.h), am = __builtin_bit_cast(bf16x8, pa.m), al = __builtin_bit_cast(bf16x8, pa.l);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const bf16x8 bh = __builtin_bit_cast(bf16x8, pb[j].h), bm = __builtin_bit_cast(bf16x8, pb[j].m), bl = __builtin_bit_cast(bf16x8, pb[j].l);
          f32x4 c = acc[i][j];
          c = mfma16_bf16(al, bh, c);
          c = mfma16_bf16(ah, bl, c);
          c = mfma16_bf16(am, bm, c);
          c = mfma16_bf16(am, bh, c);
          c = mfma16_bf16(ah, bm, c);
          c = mfma16_bf16(ah, bh, c);
          acc[i][j] = c;
        }
      }
      if (with_db) {
#pragma unroll
        for (int t = 0; t < 8; ++t) sb += a[d][t];
      }
      load(d, s0 + D + d);
    }
  }
  float* pz = part + (size_t)bz * ((size_t)N * K + N);
  const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(pz, 0, N * K * 4, 0x00020000);
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n_w + 3 * (4 * kq + r) + i;
      const int off = (n < N && k_w + 3 * lr < K) ? (n * K + k_w + 3 * lr) * 4 : (int)0x80000000;
      __builtin_amdgcn_raw_buffer_store_b96(__builtin_bit_cast(i32x3_, (f32x3){acc[i][0][r], acc[i][1][r], acc[i][2][r]}), prs, off, 0, 0);
    }
  if (with_db) {       // lane (lr, kq) summed the rows = kq (mod 4) of columns n_w + 3 lr + i (fp32 adds: the bias gradient needs no product)
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      float v = sb[i];
      v += xshfl<16>(v);
      v += xshfl<32>(v);
      const int n = n_w + 3 * lr + i;
      if (kq == 0 && n < N) pz[(size_t)N * K + n] = v;
    }
  }
}

template <int D>
__global__ __launch_bounds__(256, 2) void k_gemm_tn_reg_x3(const float* __restrict__ dy, const float* __restrict__ x, int M, int N, int K,
                                                            int rows_per_block, const float* __restrict__ db, float* __restrict__ part) {
  tn_reg_x3_body<D>(dy, x, M, N, K, rows_per_block, db, part, gridDim.x, gridDim.y, gridDim.z,
                    blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z));
}

template <int D>      // the grouped launch (backward.hip k_gemm_tn_reg_multi)
__global__ __launch_bounds__(256, 2) void k_gemm_tn_reg_x3_multi(dpmn_gemm::TnGroup g) {
  int i = 0;
#pragma unroll
  for (int j = 1; j < 8; ++j) i += (j < g.n && (int)blockIdx.x >= g.first[j]) ? 1 : 0;
  const dpmn_gemm::TnItem& t = g.it[i];
  tn_reg_x3_body<D>(t.dy, t.x, t.M, t.N, t.K, t.rows, t.db, t.part, t.gx, t.gy, t.gz, (int)blockIdx.x - g.first[i]);
}
}  // namespace

namespace dpmn_gemm {
static int x3_tn_depth() {
  static const int depth = getenv("DPMN_X3_TN_DEPTH") ? atoi(getenv("DPMN_X3_TN_DEPTH")) : 1;
  return depth;
}
int x3_launch_tn_multi(const TnGroup& g, hipStream_t st) {
  if (x3_tn_depth() == 2) hipLaunchKernelGGL(k_gemm_tn_reg_x3_multi<2>, dim3(g.first[g.n]), dim3(256), 0, st, g);
  else hipLaunchKernelGGL(k_gemm_tn_reg_x3_multi<1>, dim3(g.first[g.n]), dim3(256), 0, st, g);
  return 0;
}
int x3_launch_tn(const float* dy, const float* x, int M, int N, int K, int rows, const float* db, float* part, dim3 grid, hipStream_t st) {
  // (one 32-row step in flight, three blocks per CU: 24.0 us per launch on average over a training step against 25.3 us with two
  //  steps in flight at two blocks per CU and 27.0 us for the fp32 kernel)
  if (x3_tn_depth() == 2) hipLaunchKernelGGL(k_gemm_tn_reg_x3<2>, grid, dim3(256), 0, st, dy, x, M, N, K, rows, db, part);
  else hipLaunchKernelGGL(k_gemm_tn_reg_x3<1>, grid, dim3(256), 0, st, dy, x, M, N, K, rows, db, part);
  return 0;
}
}  // namespace dpmn_gemm
