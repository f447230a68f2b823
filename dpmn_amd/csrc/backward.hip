// Backward building blocks of the DPMN training step (interfaces/super_resolution.py:140-278; autograd of
// model/pgrm.py, model/cmm.py, loss/image_loss.py in the reference).
//   * k_gemm_tn        dW[N][K] += dY[M][N]^T . X[M][K]   -- every nn.Linear weight gradient (reduction over tokens)
//   * k_colsum         db[N]    += sum_m dY[m][n]
//   * k_ln_bwd         LayerNorm backward (dx, dgamma, dbeta) from the saved pre-norm input
//   * k_act_bwd        dpre = dy * act'(pre)  (GELU / ReLU / LeakyReLU / mish)
//   * k_image_loss_*   ImageLoss = MSE + L1 of gradient-magnitude maps (loss/image_loss.py:15-43), forward + backward
// Data-gradients of linears / convs reuse the forward GEMM / implicit-GEMM kernels with transposed weights.
#include <cstdlib>
#include <vector>
#include "common.h"

#include "gemm_tn.h"

namespace {

// ---------------------------------------------------------------------------------- dW += dY^T X
// MFMA roles: D[n][k] += sum_m A[n][m] B[m][k] with A = dY^T, B = X; both operands are read from [m][.] LDS tiles
// with ds_read_b32 (row = 4*kq + s of a 16-row chunk, column = l & 15).
// Block 256 threads: 96 (n) x 96 (k) output tile, waves 2 x 2 -> 48 x 48 each (3 x 3 MFMA tiles); grid.z splits M;
// partial results are accumulated into dW with fp32 atomics (gradient accumulation semantics: caller zeroes).
__global__ __launch_bounds__(256) void k_gemm_tn(const float* __restrict__ dy, int ldy, const float* __restrict__ x, int ldx,
                                                  float* __restrict__ dw, int ldw, int M, int N, int K, int rows_per_block,
                                                  float* __restrict__ db, float* __restrict__ part) {
  constexpr int BT = 96, BMc = 32, LD = BT + 4;
  __shared__ __attribute__((aligned(16))) float Ys[2][BMc * LD];
  __shared__ __attribute__((aligned(16))) float Xs[2][BMc * LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n_blk = blockIdx.x * BT, k_blk = blockIdx.y * BT;
  const int m_lo = blockIdx.z * rows_per_block;
  const int m_hi = min(M, m_lo + rows_per_block);
  // loaders: 32 rows x 24 float4 per operand = 768 float4 -> 3 per thread
  float4 yr[3], xr[3];
  auto gload = [&](int m0) {
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      const int i = tid + p * 256;
      const int r = i / 24, c = (i % 24) * 4;
      const int m = m0 + r;
      yr[p] = (m < m_hi && n_blk + c < N) ? *reinterpret_cast<const float4*>(dy + (size_t)m * ldy + n_blk + c) : make_float4(0.f, 0.f, 0.f, 0.f);
      xr[p] = (m < m_hi && k_blk + c < K) ? *reinterpret_cast<const float4*>(x + (size_t)m * ldx + k_blk + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      const int i = tid + p * 256;
      const int r = i / 24, c = (i % 24) * 4;
      *reinterpret_cast<float4*>(&Ys[buf][r * LD + c]) = yr[p];
      *reinterpret_cast<float4*>(&Xs[buf][r * LD + c]) = xr[p];
    }
  };
  const int wn = wave & 1, wk = wave >> 1;
  const int lr = lane & 15, kq = lane >> 4;
  f32x4 acc[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // bias gradient db[n] = sum_m dY[m][n] rides along as dY^T . 1 on the waves of the first k tile (wave-uniform branch)
  const bool with_db = db != nullptr && blockIdx.y == 0 && wk == 0;
  f32x4 accb[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) accb[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (m_lo < m_hi) { gload(m_lo); sstore(0); }
  __syncthreads();
  int buf = 0;
  for (int m0 = m_lo; m0 < m_hi; m0 += BMc) {
    if (m0 + BMc < m_hi) gload(m0 + BMc);
#pragma unroll
    for (int mc = 0; mc < BMc; mc += 16) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int row = mc + kq * 4 + s;
        float a[3], b[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) a[i] = Ys[buf][row * LD + wn * 48 + i * 16 + lr];
#pragma unroll
        for (int j = 0; j < 3; ++j) b[j] = Xs[buf][row * LD + wk * 48 + j * 16 + lr];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j) acc[i][j] = mfma16(a[i], b[j], acc[i][j]);
        if (with_db) {
#pragma unroll
          for (int i = 0; i < 3; ++i) accb[i] = mfma16(a[i], 1.0f, accb[i]);
        }
      }
    }
    if (m0 + BMc < m_hi) sstore(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
  // lane holds D[n = .. + kq*4 + r][k = .. + lr]
  if (part) {     // deterministic path: this split's tile goes to part[z][n][k] (+ part[z][N*K + n] for db), summed by k_tn_reduce
    float* pz = part + (size_t)blockIdx.z * ((size_t)N * K + N);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int k = k_blk + wk * 48 + j * 16 + lr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int n = n_blk + wn * 48 + i * 16 + kq * 4 + r;
          if (n < N && k < K) pz[(size_t)n * K + k] = acc[i][j][r];
        }
      }
    if (with_db && lr == 0) {
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int n = n_blk + wn * 48 + i * 16 + kq * 4 + r;
          if (n < N) pz[(size_t)N * K + n] = accb[i][r];
        }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int k = k_blk + wk * 48 + j * 16 + lr;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n_blk + wn * 48 + i * 16 + kq * 4 + r;
        if (n < N && k < K) atomicAdd(dw + (size_t)n * ldw + k, acc[i][j][r]);
      }
    }
  if (with_db && lr == 0) {       // every column of accb holds the same sum
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n_blk + wn * 48 + i * 16 + kq * 4 + r;
        if (n < N) atomicAdd(db + n, accb[i][r]);
      }
  }
}

// ---------------------------------------------------------------------------------- dW += dY^T X, operands straight into registers
// The same 96 x 96 block tile / 2 x 2 waves of 48 x 48, but NO LDS and NO barrier: both MFMA operands of a reduction over rows are
// row-major global data already -- lane (lr, kq) of a 4-row step needs dY[m0 + kq][n] for its A values and X[m0 + kq][k] for its B
// values -- and which 16 columns form an MFMA tile is free.  Tile i of a wave takes the columns {3 lr + i}: a lane's three A
// values are 12 consecutive bytes of one dY row (one buffer_load_dwordx3, 16 lanes = 192 contiguous bytes), likewise X.  So a
// step is 2 loads + 9 MFMAs per lane, the loads of the next D steps are in flight (rotating register sets, the compiler's own
// vmcnt accounting; rows past the block's range and columns past N / K are beyond num_records: the range check returns 0, no
// branches), and a block's only synchronisation is its end.  Result lane (lr, kq) holds dW[n_w + 3 (4 kq + r) + i][k_w + 3 lr + j].
// (k_gemm_tn staged 32-row tiles through LDS with predicated loads and fed every MFMA operand with its own ds_read_b32:
//  34.5 us per launch on average at M = 24576 -- 13 TFLOP/s, 0.55 TB/s -- against ~5-12 us of HBM time.)
typedef float f32x3 __attribute__((ext_vector_type(3)));
// (gx, gy, gz): the product's own grid -- N tiles, K tiles, row splits -- and lid the block's linear id in it (x fastest): the kernel of
// one product passes its launch grid, the grouped kernel (k_gemm_tn_reg_multi) the descriptor's
template <int D>
__device__ __forceinline__ void tn_reg_body(const float* __restrict__ dy, const float* __restrict__ x, int M, int N, int K, int rows_per_block,
                                            const float* __restrict__ db, float* __restrict__ part, int gx, int gy, int gz, int lid) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave & 1, wk = wave >> 1;
  const int lr = lane & 15, kq = lane >> 4;
  // the tiles of one row split read the same dY / X rows: XCD c (workgroups are dealt round-robin by linear id) takes the splits
  // z = c, c + 8, ... and runs their tiles back to back, so the shared rows cross the fabric once
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
  const int nsteps = (nrows + 3) >> 2;
  f32x4 acc[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // bias gradient db[n] = sum_m dY[m][n]: the lane already holds the dY values of its rows -- three vector adds per step and
  // one cross-kq shuffle at the end (as dY^T . 1 on the matrix pipe it cost 3 more MFMAs per step on half the waves: +35 %)
  const bool with_db = db != nullptr && by == 0 && wk == 0;     // wave-uniform
  f32x3 sb = (f32x3){0.f, 0.f, 0.f};
  f32x3 a[D], b[D];
  auto load = [&](int d, int step) {       // steps past the range: beyond num_records -> zeros
    a[d] = __builtin_bit_cast(f32x3, __builtin_amdgcn_raw_buffer_load_b96(yrs, offy, step * 16 * N, 0));
    b[d] = __builtin_bit_cast(f32x3, __builtin_amdgcn_raw_buffer_load_b96(xrs, offx, step * 16 * K, 0));
  };
#pragma unroll
  for (int d = 0; d < D; ++d) load(d, d);
  for (int s0 = 0; s0 < nsteps; s0 += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = mfma16(a[d][i], b[d][j], acc[i][j]);
      if (with_db) sb += a[d];
      load(d, s0 + D + d);
    }
  }
  float* pz = part + (size_t)bz * ((size_t)N * K + N);
  typedef int i32x3_ __attribute__((ext_vector_type(3)));
  const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(pz, 0, N * K * 4, 0x00020000);
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n_w + 3 * (4 * kq + r) + i;
      // the lane's three k are consecutive: one 12-byte store, 16 lanes = 192 contiguous bytes of row n (K % 48 == 0: all or none)
      const int off = (n < N && k_w + 3 * lr < K) ? (n * K + k_w + 3 * lr) * 4 : (int)0x80000000;
      __builtin_amdgcn_raw_buffer_store_b96(__builtin_bit_cast(i32x3_, (f32x3){acc[i][0][r], acc[i][1][r], acc[i][2][r]}), prs, off, 0, 0);
    }
  if (with_db) {       // lane (lr, kq) summed the rows = kq (mod 4) of columns n_w + 3 lr + i
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
__global__ __launch_bounds__(256) void k_gemm_tn_reg(const float* __restrict__ dy, const float* __restrict__ x, int M, int N, int K,
                                                      int rows_per_block, const float* __restrict__ db, float* __restrict__ part) {
  tn_reg_body<D>(dy, x, M, N, K, rows_per_block, db, part, gridDim.x, gridDim.y, gridDim.z,
                 blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z));
}

// Up to 8 products in ONE launch (the Linear weight gradients of a Swin block whose operands are live together, dpmn_gemm_tn_group_f32):
// a block finds its product by its linear id and runs the body above on the product's own grid -- the partials are the bits the
// per-product launches write.  A product alone fills the chip with ONE short block per CU (one wave per SIMD: every load latency is
// exposed); grouped, the CUs hold blocks of several products and run them back to back without launch gaps.
template <int D>
__global__ __launch_bounds__(256) void k_gemm_tn_reg_multi(dpmn_gemm::TnGroup g) {
  int i = 0;
#pragma unroll
  for (int j = 1; j < 8; ++j) i += (j < g.n && (int)blockIdx.x >= g.first[j]) ? 1 : 0;
  const dpmn_gemm::TnItem& t = g.it[i];
  tn_reg_body<D>(t.dy, t.x, t.M, t.N, t.K, t.rows, t.db, t.part, t.gx, t.gy, t.gz, (int)blockIdx.x - g.first[i]);
}

// dw[e] += sum_z part[z][e] for e < N*K ; db[n] += sum_z part[z][N*K + n].  Block = 64 elements x 4 split groups.
__global__ __launch_bounds__(256) void k_tn_reduce(const float* __restrict__ part, float* __restrict__ dw, float* __restrict__ db,
                                                    int NK, int N, int splits) {
  __shared__ float red[4][64];
  const int e = blockIdx.x * 64 + (threadIdx.x & 63), zg = threadIdx.x >> 6;
  const int tot = NK + (db ? N : 0);
  const size_t zs = (size_t)NK + N;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (e < tot) {
    int z = zg;
    for (; z + 12 < splits; z += 16) {
      s0 += part[(size_t)z * zs + e]; s1 += part[(size_t)(z + 4) * zs + e];
      s2 += part[(size_t)(z + 8) * zs + e]; s3 += part[(size_t)(z + 12) * zs + e];
    }
    for (; z < splits; z += 4) s0 += part[(size_t)z * zs + e];
  }
  red[zg][threadIdx.x & 63] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (zg == 0 && e < tot) {
    const int c = threadIdx.x;
    const float v = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
    if (e < NK) dw[e] += v; else db[e - NK] += v;
  }
}

// up to 16 pending reductions in ONE launch (the weight gradients of a whole PGRM backward: 14 separate k_tn_reduce launches of
// ~6 us each were paid at the launch rate of the two-stream backward).  Same per-element arithmetic as k_tn_reduce.
struct TnMulti {
  dpmn_tn_pending d[16];
  int first_block[17];
  int n;
  unsigned vec;       // bit i: descriptor i goes through the 16-byte path (rows, destinations and lengths are multiples of 4 floats)
};
__global__ __launch_bounds__(256) void k_tn_reduce_multi(TnMulti m) {
  __shared__ float4 red[4][64];
  int i = 0;
  while (i + 1 < m.n && (int)blockIdx.x >= m.first_block[i + 1]) ++i;
  const dpmn_tn_pending& d = m.d[i];
  const float* part = d.part;
  const int NK = d.NK, N = d.N, splits = d.splits;
  const int lane = threadIdx.x & 63, zg = threadIdx.x >> 6;
  const int tot = NK + (d.db ? N : 0);
  const size_t zs = (size_t)NK + N;
  if ((m.vec >> i) & 1u) {
    // four consecutive elements per thread: the same per-element order of additions as the scalar path below (split z goes to
    // accumulator (z / 4) % 4 of split group z % 4), 16 bytes per load and eight loads in flight per thread
    const int e = (((int)blockIdx.x - m.first_block[i]) * 64 + lane) * 4;
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
    auto add = [](float4& a_, const float4& b_) { a_.x += b_.x; a_.y += b_.y; a_.z += b_.z; a_.w += b_.w; };
    if (e < tot) {
      const float* p = part + e;
      int z = zg;
      for (; z + 28 < splits; z += 32) {
        const float4 a0 = *reinterpret_cast<const float4*>(p + (size_t)z * zs), a1 = *reinterpret_cast<const float4*>(p + (size_t)(z + 4) * zs);
        const float4 a2 = *reinterpret_cast<const float4*>(p + (size_t)(z + 8) * zs), a3 = *reinterpret_cast<const float4*>(p + (size_t)(z + 12) * zs);
        const float4 a4 = *reinterpret_cast<const float4*>(p + (size_t)(z + 16) * zs), a5 = *reinterpret_cast<const float4*>(p + (size_t)(z + 20) * zs);
        const float4 a6 = *reinterpret_cast<const float4*>(p + (size_t)(z + 24) * zs), a7 = *reinterpret_cast<const float4*>(p + (size_t)(z + 28) * zs);
        add(s0, a0); add(s1, a1); add(s2, a2); add(s3, a3);
        add(s0, a4); add(s1, a5); add(s2, a6); add(s3, a7);
      }
      for (; z + 12 < splits; z += 16) {
        add(s0, *reinterpret_cast<const float4*>(p + (size_t)z * zs)); add(s1, *reinterpret_cast<const float4*>(p + (size_t)(z + 4) * zs));
        add(s2, *reinterpret_cast<const float4*>(p + (size_t)(z + 8) * zs)); add(s3, *reinterpret_cast<const float4*>(p + (size_t)(z + 12) * zs));
      }
      for (; z < splits; z += 4) add(s0, *reinterpret_cast<const float4*>(p + (size_t)z * zs));
    }
    red[zg][lane] = make_float4((s0.x + s1.x) + (s2.x + s3.x), (s0.y + s1.y) + (s2.y + s3.y), (s0.z + s1.z) + (s2.z + s3.z), (s0.w + s1.w) + (s2.w + s3.w));
    __syncthreads();
    if (zg == 0 && e < tot) {
      const float4 r0 = red[0][lane], r1 = red[1][lane], r2 = red[2][lane], r3 = red[3][lane];
      float4* dst = reinterpret_cast<float4*>(e < NK ? d.dw + e : d.db + (e - NK));
      float4 o = *dst;
      o.x += (r0.x + r1.x) + (r2.x + r3.x); o.y += (r0.y + r1.y) + (r2.y + r3.y);
      o.z += (r0.z + r1.z) + (r2.z + r3.z); o.w += (r0.w + r1.w) + (r2.w + r3.w);
      *dst = o;
    }
    return;
  }
  float* reds = reinterpret_cast<float*>(&red[0][0]);      // [4][64] floats
  const int e = ((int)blockIdx.x - m.first_block[i]) * 64 + lane;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (e < tot) {
    int z = zg;
    for (; z + 12 < splits; z += 16) {
      s0 += part[(size_t)z * zs + e]; s1 += part[(size_t)(z + 4) * zs + e];
      s2 += part[(size_t)(z + 8) * zs + e]; s3 += part[(size_t)(z + 12) * zs + e];
    }
    for (; z < splits; z += 4) s0 += part[(size_t)z * zs + e];
  }
  reds[zg * 64 + lane] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (zg == 0 && e < tot) {
    const int c = lane;
    const float v = (reds[c] + reds[64 + c]) + (reds[128 + c] + reds[192 + c]);
    if (e < NK) d.dw[e] += v; else d.db[e - NK] += v;
  }
}

// ---- deferred ordered reductions.  Every "partial rows added in order" finish (k_tn_reduce) of a backward pass can be postponed and
// run as ONE multi-descriptor launch at its end (dpmn_reduce_defer_*): the branch streams' backward is a chain of short kernels and
// every tiny serial launch in it costs wall time.  The caller guarantees that the partial rows stay untouched until the flush (it
// hands out workspace slices from an arena it does not reuse before).  Per host thread: autograd runs a backward node on one thread.
struct DeferCtx {
  bool on = false;
  std::vector<dpmn_tn_pending> v;
};
thread_local DeferCtx g_defer;
// true: the reduction was queued (the caller must not launch it)
static bool reduce_deferred(const float* part, float* dw, float* db, int NK, int N, int rows) {
  if (!g_defer.on) return false;
  g_defer.v.push_back(dpmn_tn_pending{const_cast<float*>(part), dw, db, NK, N, rows});
  return true;
}

// db[n] += sum_m dy[m][n].  Block = 256 threads over a (rows_per_block x N) slab: thread -> (row lane = tid / cols4,
// float4 column = tid % cols4); LDS reduction over the row lanes, one atomic per column per block.
__global__ __launch_bounds__(256) void k_colsum(const float* __restrict__ dy, int ldy, float* __restrict__ db, long M, int N,
                                                 int rows_per_block, float* __restrict__ part = nullptr) {
  __shared__ float4 red[256];
  const int cols4 = (N + 3) / 4;                       // N % 4 == 0 for every caller
  const int lanes = 256 / cols4 > 0 ? 256 / cols4 : 1; // row lanes per column group
  const int cg = threadIdx.x % cols4, rl = threadIdx.x / cols4;
  const long m_lo = (long)blockIdx.x * rows_per_block;
  const long m_hi = m_lo + rows_per_block < M ? m_lo + rows_per_block : M;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (rl < lanes && cg * 4 < N && cols4 <= 256) {
    long m = m_lo + rl;
    for (; m + 3L * lanes < m_hi; m += 4L * lanes) {      // four rows in flight per thread: the loop was one load latency per row
      const float4 v0 = *reinterpret_cast<const float4*>(dy + m * ldy + cg * 4);
      const float4 v1 = *reinterpret_cast<const float4*>(dy + (m + lanes) * ldy + cg * 4);
      const float4 v2 = *reinterpret_cast<const float4*>(dy + (m + 2L * lanes) * ldy + cg * 4);
      const float4 v3 = *reinterpret_cast<const float4*>(dy + (m + 3L * lanes) * ldy + cg * 4);
      s.x += (v0.x + v1.x) + (v2.x + v3.x); s.y += (v0.y + v1.y) + (v2.y + v3.y);
      s.z += (v0.z + v1.z) + (v2.z + v3.z); s.w += (v0.w + v1.w) + (v2.w + v3.w);
    }
    for (; m < m_hi; m += lanes) {
      const float4 v = *reinterpret_cast<const float4*>(dy + m * ldy + cg * 4);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  }
  red[threadIdx.x] = s;
  __syncthreads();
  if (rl == 0 && cg * 4 < N) {
    float4 a = red[cg];
    for (int r = 1; r < lanes; ++r) { const float4 v = red[r * cols4 + cg]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
    if (part) {      // deterministic form: this block's column sums go to part[block][N], summed in block order by k_tn_reduce
      *reinterpret_cast<float4*>(part + (size_t)blockIdx.x * N + cg * 4) = a;
      return;
    }
    atomicAdd(db + cg * 4, a.x); atomicAdd(db + cg * 4 + 1, a.y); atomicAdd(db + cg * 4 + 2, a.z); atomicAdd(db + cg * 4 + 3, a.w);
  }
}

// ---------------------------------------------------------------------------------- LayerNorm backward
// x: pre-norm input (M,C); dy: grad wrt LN output; dx (M,C) written (or accumulated); dgamma/dbeta accumulated.
template <int C>
__global__ __launch_bounds__(256) void k_ln_bwd(const float* __restrict__ x, const float* __restrict__ dy,
                                                 const float* __restrict__ gamma, float eps, float* __restrict__ dx,
                                                 int accumulate_dx, float* __restrict__ dgamma, float* __restrict__ dbeta, long M,
                                                 float* __restrict__ part = nullptr) {
  constexpr int PER = C / 32;
  __shared__ float red_g[8][C], red_b[8][C];
  const int sub = threadIdx.x >> 5, t = threadIdx.x & 31;   // 8 rows per block pass, 32 threads per row
  float gam[PER], ag[PER], ab[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) { gam[i] = gamma[t + 32 * i]; ag[i] = 0.f; ab[i] = 0.f; }
  // R rows per 32-thread group are in flight at once: the loop is a chain of dependent cross-lane reductions, so the
  // memory latency of a single row per iteration (measured 1.3 TB/s) has to be covered by independent rows
  constexpr int R = 4;
  const long stride = (long)gridDim.x * 8;
  for (long row0 = (long)blockIdx.x * 8 + sub; row0 < M; row0 += stride * R) {
    float xv[R][PER], dv[R][PER], s[R];
#pragma unroll
    for (int u = 0; u < R; ++u) {
      const long row = row0 + u * stride;
      s[u] = 0.f;
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        xv[u][i] = row < M ? x[row * C + t + 32 * i] : 0.f;
        dv[u][i] = row < M ? dy[row * C + t + 32 * i] : 0.f;
        s[u] += xv[u][i];
      }
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1)
#pragma unroll
      for (int u = 0; u < R; ++u) s[u] += xshfl_v(s[u], o);
    float q[R];
#pragma unroll
    for (int u = 0; u < R; ++u) {
      s[u] *= (1.0f / C);      // mean
      q[u] = 0.f;
#pragma unroll
      for (int i = 0; i < PER; ++i) { const float d = xv[u][i] - s[u]; q[u] += d * d; }
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1)
#pragma unroll
      for (int u = 0; u < R; ++u) q[u] += xshfl_v(q[u], o);
    float s1[R], s2[R];
#pragma unroll
    for (int u = 0; u < R; ++u) {
      q[u] = 1.0f / sqrtf(q[u] * (1.0f / C) + eps);   // rstd
      s1[u] = 0.f; s2[u] = 0.f;
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        const float xh = (xv[u][i] - s[u]) * q[u];
        const float dgi = dv[u][i] * gam[i];
        s1[u] += dgi;
        s2[u] += dgi * xh;
        ag[i] += dv[u][i] * xh;      // rows past M carry dv = 0
        ab[i] += dv[u][i];
        xv[u][i] = xh; dv[u][i] = dgi;
      }
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1)
#pragma unroll
      for (int u = 0; u < R; ++u) { s1[u] += xshfl_v(s1[u], o); s2[u] += xshfl_v(s2[u], o); }
#pragma unroll
    for (int u = 0; u < R; ++u) {
      const long row = row0 + u * stride;
      if (row >= M) continue;
      const float m1 = s1[u] * (1.0f / C), m2 = s2[u] * (1.0f / C);
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        const float g = q[u] * (dv[u][i] - m1 - xv[u][i] * m2);
        const long o = row * C + t + 32 * i;
        dx[o] = accumulate_dx ? dx[o] + g : g;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < PER; ++i) { red_g[sub][t + 32 * i] = ag[i]; red_b[sub][t + 32 * i] = ab[i]; }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float g = 0.f, b = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) { g += red_g[r][c]; b += red_b[r][c]; }
    if (part) { part[(size_t)blockIdx.x * 2 * C + c] = g; part[(size_t)blockIdx.x * 2 * C + C + c] = b; continue; }   // summed in block order by k_tn_reduce
    atomicAdd(dgamma + c, g);
    atomicAdd(dbeta + c, b);
  }
}

// LayerNorm backward, vector form: 8 threads per row (C/32 float4 each, the row reductions are 3 xor-shuffles), 32 rows per
// block pass, R rows per thread group in flight; every load is unconditional (rows past M are clamped and masked), so the
// loads of a pass are issued back to back instead of one `s_waitcnt vmcnt(0)` per predicated load.
struct LnDrop {       // optional masked second output of the LayerNorm backward (k_ln_bwd_v4)
  float* out2 = nullptr;
  float p_elem = 0.f;
  unsigned long long seed_elem = 0;
  float p_row = 0.f;
  unsigned long long seed_row = 0;
  long row_len = 1;
};
template <int C>
__global__ __launch_bounds__(256) void k_ln_bwd_v4(const float* __restrict__ x, const float* __restrict__ dy,
                                                    const float* __restrict__ gamma, float eps, float* __restrict__ dx,
                                                    int accumulate_dx, float* __restrict__ dgamma, float* __restrict__ dbeta, long M,
                                                 float* __restrict__ part = nullptr, LnDrop dr = LnDrop{}) {
  // dr.out2 != null: a second output out2 = dx_final * m_elem * m_row, the masks of dpmn_dropout_f32(dx, n = M C, row_len, p_elem,
  // seed_elem, p_row, seed_row) -- the masked copy of the gradient that the NEXT Linear's backward wants (Mlp.fc2 behind Dropout +
  // DropPath, SKConv behind DropPath: pgrm.py:329-330), written here instead of by a dropout launch re-reading dx
  constexpr int V = C / 32;                 // float4 per thread
  __shared__ float red_g[32][C + 4], red_b[32][C + 4];
  const int sub = threadIdx.x >> 3, t = threadIdx.x & 7;      // 32 row groups per block, 8 threads per row
  float4 gam[V], ag[V], ab[V];
#pragma unroll
  for (int i = 0; i < V; ++i) {
    gam[i] = *reinterpret_cast<const float4*>(gamma + (t + 8 * i) * 4);
    ag[i] = make_float4(0.f, 0.f, 0.f, 0.f); ab[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  constexpr int R = C > 96 ? 1 : 2;      // rows per thread group in flight (C = 192: 6 float4 per stream and row already)
  const long stride = (long)gridDim.x * 32;
  for (long row0 = (long)blockIdx.x * 32 + sub; row0 < M; row0 += stride * R) {
    float4 xv[R][V], dv[R][V], dxo[R][V];
    float okf[R], s[R];
#pragma unroll
    for (int u = 0; u < R; ++u) {
      const long row = row0 + u * stride;
      okf[u] = row < M ? 1.f : 0.f;
      const long rc = row < M ? row : M - 1;
      s[u] = 0.f;
#pragma unroll
      for (int i = 0; i < V; ++i) {
        const long o = rc * C + (t + 8 * i) * 4;
        xv[u][i] = *reinterpret_cast<const float4*>(x + o);
        dv[u][i] = *reinterpret_cast<const float4*>(dy + o);
        if (accumulate_dx) dxo[u][i] = *reinterpret_cast<const float4*>(dx + o);
        s[u] += (xv[u][i].x + xv[u][i].y) + (xv[u][i].z + xv[u][i].w);
      }
    }
#pragma unroll
    for (int o = 4; o > 0; o >>= 1)
#pragma unroll
      for (int u = 0; u < R; ++u) s[u] += xshfl_v(s[u], o);
    float q[R];
#pragma unroll
    for (int u = 0; u < R; ++u) {
      s[u] *= (1.0f / C);
      q[u] = 0.f;
#pragma unroll
      for (int i = 0; i < V; ++i) {
        const float a = xv[u][i].x - s[u], b = xv[u][i].y - s[u], c = xv[u][i].z - s[u], d = xv[u][i].w - s[u];
        q[u] += (a * a + b * b) + (c * c + d * d);
      }
    }
#pragma unroll
    for (int o = 4; o > 0; o >>= 1)
#pragma unroll
      for (int u = 0; u < R; ++u) q[u] += xshfl_v(q[u], o);
    float s1[R], s2[R];
#pragma unroll
    for (int u = 0; u < R; ++u) {
      q[u] = 1.0f / sqrtf(q[u] * (1.0f / C) + eps);
      s1[u] = 0.f; s2[u] = 0.f;
#pragma unroll
      for (int i = 0; i < V; ++i) {
        float* xp = reinterpret_cast<float*>(&xv[u][i]);
        float* dp = reinterpret_cast<float*>(&dv[u][i]);
        const float* gp = reinterpret_cast<const float*>(&gam[i]);
        float* agp = reinterpret_cast<float*>(&ag[i]);
        float* abp = reinterpret_cast<float*>(&ab[i]);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float xh = (xp[c] - s[u]) * q[u];
          const float dvm = dp[c] * okf[u];          // rows past M contribute nothing
          const float dgi = dvm * gp[c];
          s1[u] += dgi; s2[u] += dgi * xh;
          agp[c] += dvm * xh; abp[c] += dvm;
          xp[c] = xh; dp[c] = dgi;
        }
      }
    }
#pragma unroll
    for (int o = 4; o > 0; o >>= 1)
#pragma unroll
      for (int u = 0; u < R; ++u) { s1[u] += xshfl_v(s1[u], o); s2[u] += xshfl_v(s2[u], o); }
#pragma unroll
    for (int u = 0; u < R; ++u) {
      const long row = row0 + u * stride;
      if (row >= M) continue;
      const float m1 = s1[u] * (1.0f / C), m2 = s2[u] * (1.0f / C);
#pragma unroll
      for (int i = 0; i < V; ++i) {
        float4 gq;
        gq.x = q[u] * (dv[u][i].x - m1 - xv[u][i].x * m2); gq.y = q[u] * (dv[u][i].y - m1 - xv[u][i].y * m2);
        gq.z = q[u] * (dv[u][i].z - m1 - xv[u][i].z * m2); gq.w = q[u] * (dv[u][i].w - m1 - xv[u][i].w * m2);
        if (accumulate_dx) { gq.x += dxo[u][i].x; gq.y += dxo[u][i].y; gq.z += dxo[u][i].z; gq.w += dxo[u][i].w; }
        *reinterpret_cast<float4*>(dx + row * C + (t + 8 * i) * 4) = gq;
        if (dr.out2) {
          const unsigned long long idx = (unsigned long long)(row * C + (t + 8 * i) * 4);
          float o[4] = {gq.x, gq.y, gq.z, gq.w};
          if (dr.p_elem > 0.f) {
            const float ik = 1.0f / (1.0f - dr.p_elem);
            const unsigned long long z0 = drop_z0(dr.seed_elem, idx);
#pragma unroll
            for (int c = 0; c < 4; ++c) o[c] *= drop_scale_z(z0 + (unsigned long long)c * DROP_PHI, dr.p_elem, ik);
          }
          if (dr.p_row > 0.f) {
            const float mr = drop_scale(dr.seed_row, idx / (unsigned long long)dr.row_len, dr.p_row, 1.0f / (1.0f - dr.p_row));
#pragma unroll
            for (int c = 0; c < 4; ++c) o[c] *= mr;
          }
          *reinterpret_cast<float4*>(dr.out2 + row * C + (t + 8 * i) * 4) = make_float4(o[0], o[1], o[2], o[3]);
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < V; ++i) {
    *reinterpret_cast<float4*>(&red_g[sub][(t + 8 * i) * 4]) = ag[i];
    *reinterpret_cast<float4*>(&red_b[sub][(t + 8 * i) * 4]) = ab[i];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float g = 0.f, b = 0.f;
#pragma unroll
    for (int r = 0; r < 32; ++r) { g += red_g[r][c]; b += red_b[r][c]; }
    if (part) { part[(size_t)blockIdx.x * 2 * C + c] = g; part[(size_t)blockIdx.x * 2 * C + C + c] = b; continue; }
    atomicAdd(dgamma + c, g);
    atomicAdd(dbeta + c, b);
  }
}

// ---------------------------------------------------------------------------------- activation backward
__device__ __forceinline__ float act_grad(float pre, int act, float slope) {
  switch (act) {
    case ACT_GELU: {
      const float cdf = 0.5f * (1.0f + erf_as(pre * 0.70710678118654752440f));
      return cdf + pre * 0.3989422804014327f * __expf(-0.5f * pre * pre);
    }
    case ACT_RELU: return pre > 0.f ? 1.f : 0.f;
    case ACT_LEAKY02: return pre > 0.f ? 1.f : 0.2f;
    case ACT_LEAKY001: return pre > 0.f ? 1.f : 0.01f;
    case ACT_PRELU: return pre > 0.f ? 1.f : slope;
    case ACT_MISH: {
      const float sp = softplus_t(pre), th = tanhf(sp);
      return th + pre * (1.f - th * th) * sigmoid_f(pre);
    }
    case ACT_TANH: { const float th = tanhf(pre); return 1.f - th * th; }
    case ACT_SIGMOID: { const float sg = sigmoid_f(pre); return sg * (1.f - sg); }
    default: return 1.f;
  }
}
__global__ void k_act_bwd(const float* __restrict__ dy, const float* __restrict__ pre, float* __restrict__ dpre, int act,
                          float slope, long n4) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const float4 d = reinterpret_cast<const float4*>(dy)[i];
  const float4 p = reinterpret_cast<const float4*>(pre)[i];
  reinterpret_cast<float4*>(dpre)[i] = make_float4(d.x * act_grad(p.x, act, slope), d.y * act_grad(p.y, act, slope),
                                                   d.z * act_grad(p.z, act, slope), d.w * act_grad(p.w, act, slope));
}

// ---------------------------------------------------------------------------------- ImageLoss
// out/tgt: NCHW with per-image strides (tgt may be a (B,4,H,W) tensor read as its first C channels).
// part[2*blk] = sum (o-t)^2 , part[2*blk+1] = sum |G(o)-G(t)| over channels < 3; optional U,V planes for backward.
__device__ __forceinline__ float at(const float* p, int y, int x, int H, int W) {
  return (y >= 0 && y < H && x >= 0 && x < W) ? p[y * W + x] : 0.f;
}
__global__ __launch_bounds__(256) void k_image_loss_fwd(const float* __restrict__ o, long o_stride, const float* __restrict__ t,
                                                         long t_stride, float* __restrict__ part, float* __restrict__ U,
                                                         float* __restrict__ V, int C, int H, int W, long total) {
  __shared__ float red[2][4];
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  float se = 0.f, l1 = 0.f;
  if (idx < total) {
    const int x = idx % W, y = (idx / W) % H, c = (idx / ((long)W * H)) % C;
    const long n = idx / ((long)W * H * C);
    const float* op = o + n * o_stride + (size_t)c * H * W;
    const float* tp = t + n * t_stride + (size_t)c * H * W;
    const float d = op[y * W + x] - tp[y * W + x];
    se = d * d;
    if (c < 3) {
      const float gxo = (at(op, y, x + 1, H, W) - at(op, y, x - 1, H, W)) * 0.5f;
      const float gyo = (at(op, y - 1, x, H, W) - at(op, y + 1, x, H, W)) * 0.5f;
      const float gxt = (at(tp, y, x + 1, H, W) - at(tp, y, x - 1, H, W)) * 0.5f;
      const float gyt = (at(tp, y - 1, x, H, W) - at(tp, y + 1, x, H, W)) * 0.5f;
      const float Go = sqrtf(gxo * gxo + gyo * gyo + 1e-6f), Gt = sqrtf(gxt * gxt + gyt * gyt + 1e-6f);
      const float df = Go - Gt;
      l1 = fabsf(df);
      if (U) {
        const float sgn = df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f);
        const long u = (n * 3 + c) * (long)H * W + y * W + x;
        U[u] = sgn * gxo / (2.f * Go);
        V[u] = sgn * gyo / (2.f * Go);
      }
    }
  }
  se = wave_sum(se); l1 = wave_sum(l1);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = se; red[1][threadIdx.x >> 6] = l1; }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[2 * blockIdx.x] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    part[2 * blockIdx.x + 1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  }
}
__global__ void k_image_loss_final(const float* __restrict__ part, int nblocks, float w_mse, float inv_n_mse, float w_grad,
                                   float inv_n_grad, float* __restrict__ loss) {
  double se = 0.0, l1 = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += 64) { se += part[2 * i]; l1 += part[2 * i + 1]; }
  for (int o = 32; o > 0; o >>= 1) { se += xshfl_v(se, o); l1 += xshfl_v(l1, o); }
  if (threadIdx.x == 0) loss[0] = (float)(w_mse * se * inv_n_mse + w_grad * l1 * inv_n_grad);
}
// grad_out = gscale * ( w_mse*2(o-t)/N + w_grad/N3 * (U[x-1] - U[x+1] + V[y+1] - V[y-1]) )   (c < 3 for the second term)
__global__ void k_image_loss_bwd(const float* __restrict__ o, long o_stride, const float* __restrict__ t, long t_stride,
                                 const float* __restrict__ U, const float* __restrict__ V, const float* __restrict__ gscale,
                                 float c_mse, float c_grad, float* __restrict__ grad, int accumulate, int C, int H, int W,
                                 long total) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int x = idx % W, y = (idx / W) % H, c = (idx / ((long)W * H)) % C;
  const long n = idx / ((long)W * H * C);
  const float gs = gscale[0];
  float g = c_mse * (o[n * o_stride + (size_t)c * H * W + y * W + x] - t[n * t_stride + (size_t)c * H * W + y * W + x]);
  if (c < 3 && U) {
    const float* up = U + (n * 3 + c) * (long)H * W;
    const float* vp = V + (n * 3 + c) * (long)H * W;
    g += c_grad * (at(up, y, x - 1, H, W) - at(up, y, x + 1, H, W) + at(vp, y + 1, x, H, W) - at(vp, y - 1, x, H, W));
  }
  g *= gs;
  grad[idx] = accumulate ? grad[idx] + g : g;
}

}  // namespace

extern "C" {

static void tn_plan(int M, int N, int K, int* splits_out, int* rows_out) {
  const int tiles = cdiv(N, 96) * cdiv(K, 96);
  static const int want = getenv("DPMN_TN_BLOCKS") ? atoi(getenv("DPMN_TN_BLOCKS")) : 256;       // experiment knob
  // (one block per CU and product: since the products of a Swin block share launches (dpmn_gemm_tn_group_f32) the CUs hold blocks of
  //  several products anyway, and fewer splits are fewer partial slabs to write and re-read -- 512 blocks for the 4-tile shapes fc1 / fc2
  //  were 45.0 against 48.2 us per product launched alone, but 23.56 against 23.12 ms per training step grouped; 128: 23.22, 64: 24.14)
  int splits = cdiv(want, tiles);
  int rows = cdiv(cdiv(M, splits), 32) * 32;      // (multiples of 32: the LDS kernel's chunk; of 4: an MFMA step of the register kernel)
  if (rows < 32) rows = 32;
  *splits_out = cdiv(M, rows);
  *rows_out = rows;
}

// split plan + launch of the partial-sum kernel; defer != nullptr: no reduce launch, the caller gets the descriptor of the pending one
static int gemm_tn_impl(const float* dy, const float* x, float* dw, float* db, int M, int N, int K, float* ws, size_t ws_bytes,
                        dpmn_tn_pending* defer, dpmn_stream_t stream) {
  DPMN_REQUIRE(dy && x && dw && M > 0 && N % 4 == 0 && K % 4 == 0, "gemm_tn: bad arguments (N, K multiples of 4)");
  int splits, rows;
  tn_plan(M, N, K, &splits, &rows);
  dim3 grid(cdiv(N, 96), cdiv(K, 96), splits);
  // with a workspace the splits are reduced by a second kernel (deterministic, no same-address atomic pile-up);
  // without one they are accumulated with fp32 atomics
  const size_t need = (size_t)splits * ((size_t)N * K + N) * sizeof(float);
  float* part = (ws && ws_bytes >= need && (splits > 1 || defer)) ? ws : nullptr;      // (deferred: also a single split goes through the reduce)
  // operands straight from global memory into the MFMA registers when a wave's 48 columns tile N and K (every Linear of the
  // PGRM block: 96 / 192 / 384) and the byte offsets fit the buffer instructions
  static const int reg_on = getenv("DPMN_TN_REG") ? atoi(getenv("DPMN_TN_REG")) : 1;
  ProfScope prof(PT_GEMM_TN, as_stream(stream), 2.0 * M * (double)N * K, 4.0 * ((double)M * N + (double)M * K + (double)splits * ((double)N * K + N)));
  const bool reg_ok = reg_on && part && N % 48 == 0 && K % 48 == 0 && (size_t)rows * (N > K ? N : K) * 4 < (1ull << 31);
  if (reg_ok && x3_on(64))       // mode 2: the same partials on six bf16 MFMAs per tile (gemm_tn_x3.hip)
    dpmn_gemm::x3_launch_tn(dy, x, M, N, K, rows, db, part, grid, as_stream(stream));
  else if (reg_ok)
    hipLaunchKernelGGL(k_gemm_tn_reg<8>, grid, dim3(256), 0, as_stream(stream), dy, x, M, N, K, rows, db, part);
  else
    hipLaunchKernelGGL(k_gemm_tn, grid, dim3(256), 0, as_stream(stream), dy, N, x, K, dw, K, M, N, K, rows, db, part);
  DPMN_CHECK_LAUNCH();
  if (defer) {
    DPMN_REQUIRE(part, "gemm_tn_partial: the workspace must hold the split partials (dpmn_gemm_tn_partial_bytes)");
    *defer = dpmn_tn_pending{part, dw, db, N * K, N, splits};
    return DPMN_OK;
  }
  if (part && !reduce_deferred(part, dw, db, N * K, N, splits)) {
    const int tot = N * K + (db ? N : 0);
    hipLaunchKernelGGL(k_tn_reduce, dim3(cdiv(tot, 64)), dim3(256), 0, as_stream(stream), part, dw, db, N * K, N, splits);
    DPMN_CHECK_LAUNCH();
  }
  return DPMN_OK;
}

int dpmn_gemm_tn_f32(const float* dy, const float* x, float* dw, float* db, int M, int N, int K, float* ws, size_t ws_bytes,
                     dpmn_stream_t stream) {
  return gemm_tn_impl(dy, x, dw, db, M, N, K, ws, ws_bytes, nullptr, stream);
}

// n Linear weight gradients whose operands are live together: the bits of n dpmn_gemm_tn_f32 calls, with every product that takes the
// operands-in-registers kernel and has its workspace in ONE launch (k_gemm_tn_reg_multi; up to 8 per launch)
int dpmn_gemm_tn_group_f32(const dpmn_tn_item* items, int n, dpmn_stream_t stream) {
  DPMN_REQUIRE(items && n >= 0, "gemm_tn_group: bad arguments");
  static const int reg_on = getenv("DPMN_TN_REG") ? atoi(getenv("DPMN_TN_REG")) : 1;
  static const int group_on = getenv("DPMN_TN_GROUP") ? atoi(getenv("DPMN_TN_GROUP")) : 1;
  dpmn_gemm::TnGroup g{};
  int idx[8], splits_of[8];
  double flops = 0.0, bytes = 0.0;
  auto flush = [&]() -> int {
    if (g.n == 0) return DPMN_OK;
    {
      ProfScope prof(PT_GEMM_TN, as_stream(stream), flops, bytes);
      if (x3_on(64)) dpmn_gemm::x3_launch_tn_multi(g, as_stream(stream));
      else hipLaunchKernelGGL(k_gemm_tn_reg_multi<8>, dim3(g.first[g.n]), dim3(256), 0, as_stream(stream), g);
      DPMN_CHECK_LAUNCH();
    }
    for (int j = 0; j < g.n; ++j) {
      const dpmn_tn_item& t = items[idx[j]];
      if (!reduce_deferred(g.it[j].part, t.dw, t.db, t.N * t.K, t.N, splits_of[j])) {
        const int tot = t.N * t.K + (t.db ? t.N : 0);
        hipLaunchKernelGGL(k_tn_reduce, dim3(cdiv(tot, 64)), dim3(256), 0, as_stream(stream), g.it[j].part, t.dw, t.db, t.N * t.K, t.N, splits_of[j]);
        DPMN_CHECK_LAUNCH();
      }
    }
    g = dpmn_gemm::TnGroup{};
    flops = bytes = 0.0;
    return DPMN_OK;
  };
  for (int i = 0; i < n; ++i) {
    const dpmn_tn_item& t = items[i];
    DPMN_REQUIRE(t.dy && t.x && t.dw && t.M > 0 && t.N % 4 == 0 && t.K % 4 == 0, "gemm_tn_group: bad item (N, K multiples of 4)");
    int splits, rows;
    tn_plan(t.M, t.N, t.K, &splits, &rows);
    const size_t need = (size_t)splits * ((size_t)t.N * t.K + t.N) * sizeof(float);
    const bool reg_ok = group_on && reg_on && t.ws && t.ws_bytes >= need && splits > 1 && t.N % 48 == 0 && t.K % 48 == 0 &&
                        (size_t)rows * (t.N > t.K ? t.N : t.K) * 4 < (1ull << 31);
    if (!reg_ok) {
      const int rc = gemm_tn_impl(t.dy, t.x, t.dw, t.db, t.M, t.N, t.K, t.ws, t.ws_bytes, nullptr, stream);
      if (rc != DPMN_OK) return rc;
      continue;
    }
    // two sums into one tensor must not share a launch's reduce queue position: the deferred queue keeps call order, nothing to do here
    const int j = g.n;
    g.it[j] = dpmn_gemm::TnItem{t.dy, t.x, t.ws, t.db, t.M, t.N, t.K, rows, cdiv(t.N, 96), cdiv(t.K, 96), splits};
    g.first[j + 1] = g.first[j] + g.it[j].gx * g.it[j].gy * splits;
    idx[j] = i;
    splits_of[j] = splits;
    flops += 2.0 * t.M * (double)t.N * t.K;
    bytes += 4.0 * ((double)t.M * t.N + (double)t.M * t.K + (double)splits * ((double)t.N * t.K + t.N));
    if (++g.n == 8) {
      const int rc = flush();
      if (rc != DPMN_OK) return rc;
    }
  }
  return flush();
}

size_t dpmn_gemm_tn_partial_bytes(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  int splits, rows;
  tn_plan(M, N, K, &splits, &rows);
  return ((size_t)splits * ((size_t)N * K + N) * sizeof(float) + 255) / 256 * 256;
}

int dpmn_gemm_tn_partial_f32(const float* dy, const float* x, float* dw, float* db, int M, int N, int K, float* ws, size_t ws_bytes,
                             dpmn_tn_pending* pending, dpmn_stream_t stream) {
  DPMN_REQUIRE(pending && ws, "gemm_tn_partial: null pointer");
  return gemm_tn_impl(dy, x, dw, db, M, N, K, ws, ws_bytes, pending, stream);
}

int dpmn_tn_reduce_multi_f32(const dpmn_tn_pending* pending, int n, dpmn_stream_t stream) {
  DPMN_REQUIRE(pending && n >= 0, "tn_reduce_multi: bad arguments");
  static const bool tn_vec = !(getenv("DPMN_TN_REDUCE_VEC") && atoi(getenv("DPMN_TN_REDUCE_VEC")) == 0);
  int i0 = 0;
  while (i0 < n) {
    // one launch = up to 16 descriptors with pairwise DIFFERENT destinations (two sums into one tensor inside a launch would race:
    // the second one goes to the next launch, i.e. behind the first in stream order)
    TnMulti m{};
    int nb = 0, cnt = 0;
    for (; i0 + cnt < n && cnt < 16; ++cnt) {
      const dpmn_tn_pending& d = pending[i0 + cnt];
      bool clash = false;
      for (int j = 0; j < cnt && !clash; ++j)
        clash = m.d[j].dw == d.dw || (d.db && (m.d[j].db == d.db || m.d[j].dw == d.db)) || (m.d[j].db && m.d[j].db == d.dw);
      if (clash) break;
      m.d[cnt] = d;
      m.first_block[cnt] = nb;
      const bool vec = tn_vec && d.NK % 4 == 0 && d.N % 4 == 0 && ((uintptr_t)d.part & 15) == 0 && ((uintptr_t)d.dw & 15) == 0 &&
                       (!d.db || ((uintptr_t)d.db & 15) == 0);
      if (vec) m.vec |= 1u << cnt;
      nb += cdiv(d.NK + (d.db ? d.N : 0), vec ? 256 : 64);
    }
    m.n = cnt;
    m.first_block[cnt] = nb;
    double rbytes = 0.0;
    for (int j = 0; j < cnt; ++j) rbytes += 4.0 * ((double)m.d[j].splits + 2.0) * (m.d[j].NK + (m.d[j].db ? m.d[j].N : 0));
    ProfScope prof(PT_TN_REDUCE, as_stream(stream), 0.0, rbytes);
    hipLaunchKernelGGL(k_tn_reduce_multi, dim3(nb), dim3(256), 0, as_stream(stream), m);
    DPMN_CHECK_LAUNCH();
    i0 += cnt;
  }
  return DPMN_OK;
}

int dpmn_reduce_defer_begin(void) {
  g_defer.on = true;
  g_defer.v.clear();
  return DPMN_OK;
}

int dpmn_reduce_defer_enable(int on) {       // pause / resume queueing without touching the queue (a reduction whose result is read at once)
  g_defer.on = on != 0;
  return DPMN_OK;
}

int dpmn_reduce_defer_push(const dpmn_tn_pending* p) {
  DPMN_REQUIRE(p && p->part && p->dw && p->NK > 0 && p->splits > 0, "reduce_defer_push: bad descriptor");
  g_defer.v.push_back(*p);
  return DPMN_OK;
}

int dpmn_reduce_defer_pending(void) { return (int)g_defer.v.size(); }

int dpmn_reduce_defer_flush(int end, dpmn_stream_t stream) {
  int rc = DPMN_OK;
  if (!g_defer.v.empty()) rc = dpmn_tn_reduce_multi_f32(g_defer.v.data(), (int)g_defer.v.size(), stream);
  g_defer.v.clear();
  if (end) g_defer.on = false;
  return rc;
}

// dw[e] += sum_z part[z][e] (e < NK), db[n] += sum_z part[z][NK + n]: rows of NK + N floats added in row order (k_tn_reduce) -- the
// finish step of every "per-block / per-image partials instead of atomics" gradient in backward_pgrm.hip
int dpmn_rows_reduce_f32(const float* part, float* dw, float* db, int NK, int N, int rows, dpmn_stream_t stream) {
  DPMN_REQUIRE(part && dw && NK > 0 && N >= 0 && rows > 0 && (db || N == 0), "rows_reduce: bad arguments");
  if (reduce_deferred(part, dw, db, NK, N, rows)) return DPMN_OK;
  hipLaunchKernelGGL(k_tn_reduce, dim3(cdiv(NK + N, 64)), dim3(256), 0, as_stream(stream), part, dw, db, NK, N, rows);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

int dpmn_colsum_f32(const float* dy, float* db, long M, int N, dpmn_stream_t stream) {
  DPMN_REQUIRE(dy && db && M > 0 && N > 0 && N % 4 == 0 && N <= 1024, "colsum: N must be a multiple of 4 (<= 1024)");
  const int rows = 256;
  hipLaunchKernelGGL(k_colsum, dim3((unsigned)((M + rows - 1) / rows)), dim3(256), 0, as_stream(stream), dy, N, db, M, N, rows,
                     static_cast<float*>(nullptr));
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

// the same without atomics: per-block partial sums in ws (ceil(M / 256) * N floats), added to db in block order -- bitwise reproducible
int dpmn_colsum_det_f32(const float* dy, float* db, long M, int N, float* ws, size_t ws_bytes, dpmn_stream_t stream) {
  DPMN_REQUIRE(dy && db && ws && M > 0 && N > 0 && N % 4 == 0 && N <= 1024, "colsum_det: N must be a multiple of 4 (<= 1024)");
  const int rows = 256;
  const long nb = (M + rows - 1) / rows;
  if ((size_t)nb * N * sizeof(float) > ws_bytes) return dpmn_set_error(DPMN_ERR_WORKSPACE, "colsum_det: workspace too small");
  hipLaunchKernelGGL(k_colsum, dim3((unsigned)nb), dim3(256), 0, as_stream(stream), dy, N, db, M, N, rows, ws);
  DPMN_CHECK_LAUNCH();
  // (the block-order sum of the partial rows is k_tn_reduce's job: 64 columns x 4 split groups per block, fixed order)
  if (reduce_deferred(ws, db, nullptr, N, 0, (int)nb)) return DPMN_OK;
  hipLaunchKernelGGL(k_tn_reduce, dim3(cdiv(N, 64)), dim3(256), 0, as_stream(stream), ws, db, static_cast<float*>(nullptr), N, 0, (int)nb);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

static int layernorm_bwd_impl(const float* x, const float* dy, const float* gamma, float eps, float* dx, int accumulate_dx,
                              float* dgamma, float* dbeta, long M, int C, float* part, size_t part_bytes, dpmn_stream_t stream,
                              LnDrop dr = LnDrop{}) {
  DPMN_REQUIRE(x && dy && gamma && dx && dgamma && dbeta && M > 0, "layernorm_bwd: bad arguments");
  DPMN_REQUIRE(!dr.out2 || ((C == 96 || C == 192) && dr.row_len > 0 && dr.p_elem >= 0.f && dr.p_elem < 1.f && dr.p_row >= 0.f && dr.p_row < 1.f),
               "layernorm_bwd: the masked second output needs C = 96 / 192 and drop rates in [0, 1)");
  DPMN_REQUIRE(!part || part_bytes >= (size_t)512 * 2 * C * sizeof(float), "layernorm_bwd_det: workspace of 512 * 2 C floats");
  // every block ends with 2*C same-address atomics (dgamma, dbeta), which serialise: few, fat blocks (4 rows in flight per
  // 32-thread group).  In-pipeline sweep at M = 49152: 256 blocks 46.6 us, 512: 36.9, 1024: 41.9, 2048: 59.9
  static const long cap = getenv("DPMN_LNB_BLOCKS") ? atol(getenv("DPMN_LNB_BLOCKS")) : 512;
  const unsigned blocks = (unsigned)(M / 8 < cap ? (M + 7) / 8 : cap);
  unsigned nblk = blocks;
  static const int v4 = getenv("DPMN_LNB_V4") ? atoi(getenv("DPMN_LNB_V4")) : 1;
  ProfScope prof(PT_LN_BWD, as_stream(stream), 0.0, 4.0 * (accumulate_dx ? 4 : 3) * (double)M * C);
  if ((C == 96 || C == 192) && (v4 || dr.out2)) {
    // in-pipeline sweep of the vector kernel at M = 49152: 256 blocks 20.2 us, 384: 21.2, 512: 24.3, 768: 27.6, 1024: 33.3
    // (the scalar kernel it replaces: 36.8 us) -- one block per CU, the same-address dgamma / dbeta atomics set the slope
    static const long cap4 = getenv("DPMN_LNB_BLOCKS") ? atol(getenv("DPMN_LNB_BLOCKS")) : 256;
    const unsigned b4 = (unsigned)(M / 32 < cap4 ? (M + 31) / 32 : cap4);
    if (C == 96) hipLaunchKernelGGL((k_ln_bwd_v4<96>), dim3(b4), dim3(256), 0, as_stream(stream), x, dy, gamma, eps, dx, accumulate_dx, dgamma, dbeta, M, part, dr);
    else hipLaunchKernelGGL((k_ln_bwd_v4<192>), dim3(b4), dim3(256), 0, as_stream(stream), x, dy, gamma, eps, dx, accumulate_dx, dgamma, dbeta, M, part, dr);
    nblk = b4;
  } else if (C == 96)
    hipLaunchKernelGGL((k_ln_bwd<96>), dim3(blocks), dim3(256), 0, as_stream(stream), x, dy, gamma, eps, dx, accumulate_dx, dgamma, dbeta, M, part);
  else if (C == 192)
    hipLaunchKernelGGL((k_ln_bwd<192>), dim3(blocks), dim3(256), 0, as_stream(stream), x, dy, gamma, eps, dx, accumulate_dx, dgamma, dbeta, M, part);
  else if (C == 64)
    hipLaunchKernelGGL((k_ln_bwd<64>), dim3(blocks), dim3(256), 0, as_stream(stream), x, dy, gamma, eps, dx, accumulate_dx, dgamma, dbeta, M, part);
  else
    return dpmn_set_error(DPMN_ERR_ARG, "layernorm_bwd: C must be 64, 96 or 192");
  DPMN_CHECK_LAUNCH();
  if (part) {      // the blocks' [dgamma | dbeta] rows, added in block order (no atomics: bitwise reproducible)
    // one launch: k_tn_reduce's [N * K | N] row layout with N * K = C (-> dgamma) and N = C (-> dbeta)
    if (reduce_deferred(part, dgamma, dbeta, C, C, (int)nblk)) return DPMN_OK;
    hipLaunchKernelGGL(k_tn_reduce, dim3(cdiv(2 * C, 64)), dim3(256), 0, as_stream(stream), part, dgamma, dbeta, C, C, (int)nblk);
    DPMN_CHECK_LAUNCH();
  }
  return DPMN_OK;
}

int dpmn_layernorm_bwd_f32(const float* x, const float* dy, const float* gamma, float eps, float* dx, int accumulate_dx,
                           float* dgamma, float* dbeta, long M, int C, dpmn_stream_t stream) {
  return layernorm_bwd_impl(x, dy, gamma, eps, dx, accumulate_dx, dgamma, dbeta, M, C, nullptr, 0, stream);
}

int dpmn_layernorm_bwd_det_f32(const float* x, const float* dy, const float* gamma, float eps, float* dx, int accumulate_dx,
                               float* dgamma, float* dbeta, long M, int C, float* ws, size_t ws_bytes, dpmn_stream_t stream) {
  DPMN_REQUIRE(ws, "layernorm_bwd_det: null workspace");
  return layernorm_bwd_impl(x, dy, gamma, eps, dx, accumulate_dx, dgamma, dbeta, M, C, ws, ws_bytes, stream);
}

int dpmn_layernorm_bwd_det_drop_f32(const float* x, const float* dy, const float* gamma, float eps, float* dx, int accumulate_dx,
                                    float* dgamma, float* dbeta, long M, int C, float* ws, size_t ws_bytes, float* masked_out,
                                    float p_elem, unsigned long long seed_elem, float p_row, unsigned long long seed_row, long row_len,
                                    dpmn_stream_t stream) {
  DPMN_REQUIRE(ws && masked_out, "layernorm_bwd_det_drop: null pointer");
  LnDrop dr;
  dr.out2 = masked_out; dr.p_elem = p_elem; dr.seed_elem = seed_elem; dr.p_row = p_row; dr.seed_row = seed_row; dr.row_len = row_len;
  return layernorm_bwd_impl(x, dy, gamma, eps, dx, accumulate_dx, dgamma, dbeta, M, C, ws, ws_bytes, stream, dr);
}

int dpmn_act_bwd_f32(const float* dy, const float* pre, float* dpre, int act, float slope, long n, dpmn_stream_t stream) {
  DPMN_REQUIRE(dy && pre && dpre && n > 0 && n % 4 == 0, "act_bwd: bad arguments (n multiple of 4)");
  const long n4 = n / 4;
  hipLaunchKernelGGL(k_act_bwd, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, as_stream(stream), dy, pre, dpre, act, slope, n4);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

size_t dpmn_image_loss_workspace_bytes(int B, int C, int H, int W) {
  const long total = (long)B * C * H * W;
  return (size_t)((total + 255) / 256) * 2 * sizeof(float);
}

int dpmn_image_loss_fwd_f32(const float* out, long out_stride, const float* tgt, long tgt_stride, float w_mse, float w_grad,
                            int gradient, float* loss, float* U, float* V, void* workspace, int B, int C, int H, int W,
                            dpmn_stream_t stream) {
  DPMN_REQUIRE(out && tgt && loss && workspace && B > 0 && C >= 1, "image_loss_fwd: bad arguments");
  const long total = (long)B * C * H * W;
  const int nb = (int)((total + 255) / 256);
  hipLaunchKernelGGL(k_image_loss_fwd, dim3(nb), dim3(256), 0, as_stream(stream), out, out_stride, tgt, tgt_stride,
                     static_cast<float*>(workspace), gradient ? U : nullptr, gradient ? V : nullptr, C, H, W, total);
  DPMN_CHECK_LAUNCH();
  const int c3 = C < 3 ? C : 3;
  hipLaunchKernelGGL(k_image_loss_final, dim3(1), dim3(64), 0, as_stream(stream), static_cast<const float*>(workspace), nb, w_mse,
                     1.0f / (float)total, gradient ? w_grad : 0.f, 1.0f / (float)((long)B * c3 * H * W), loss);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

int dpmn_image_loss_bwd_f32(const float* out, long out_stride, const float* tgt, long tgt_stride, const float* U,
                            const float* V, const float* grad_scale, float w_mse, float w_grad, int gradient, float* grad_out,
                            int accumulate, int B, int C, int H, int W, dpmn_stream_t stream) {
  DPMN_REQUIRE(out && tgt && grad_scale && grad_out && B > 0, "image_loss_bwd: bad arguments");
  DPMN_REQUIRE(!gradient || (U && V), "image_loss_bwd: U/V planes from the forward pass are required");
  const long total = (long)B * C * H * W;
  const int c3 = C < 3 ? C : 3;
  hipLaunchKernelGGL(k_image_loss_bwd, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), out, out_stride,
                     tgt, tgt_stride, gradient ? U : nullptr, gradient ? V : nullptr, grad_scale, w_mse * 2.0f / (float)total,
                     gradient ? w_grad / (float)((long)B * c3 * H * W) : 0.f, grad_out, accumulate, C, H, W, total);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

}  // extern "C"
