// Convolution weight gradients and train-mode BatchNorm plumbing for the NHWC implicit-GEMM convs (conv.hip).
//   k_conv_wgrad      dW[co][k] += sum_pixels dY[pix][co] * pro(in)[pix @ tap(k)][ci(k)]   (packed or parameter layout)
//   k_bn_finalize     per-channel (sum, sumsq) -> (scale, shift) for the consumers' affine-on-load + running stats
//   k_affine_act_bwd  G (+)= dA * act'(scale*r + shift)          (consumer-side activation backward)
//   k_bn_bwd_*        BatchNorm backward through batch statistics: dgamma, dbeta, d(raw conv output)
//   k_se_gate_bwd     backward of the CMM channel gate (cmm.py:135-147)
// Data gradients are ordinary convolutions of dY with re-packed weights and run through k_conv_igemm / k_conv_halo.
#include <cstdlib>
#include "conv_wgrad.h"

namespace {

using dpmn_conv::WgArgs;

// exact m / d for 0 <= m < 2^24 through one float multiply and a +-1 fix-up
__device__ __forceinline__ void fdivmod(int m, int d, float inv, int& q, int& r) {
  q = (int)((float)m * inv);
  r = m - q * d;
  if (r < 0) { --q; r += d; }
  if (r >= d) { ++q; r -= d; }
}

// Weight gradient of the implicit-GEMM conv: dW[co][k] = sum_pixels dY[pix][co] * pro(in)[pix @ tap(k)][ci(k)].
// Block = BN_ (co) x BKT (k) tile of dW over its pixel range, staged 32 pixels at a time as [pixel][channel] LDS
// tiles.  The contraction index of the 16x16x4 MFMA is the pixel; each lane's co / k indices are interleaved
// (co = 4*i + ti, k = NJ*j + tj) so that ONE ds_read_b128 along the channel axis feeds 4 MFMA tiles.  The 4 waves split
// the k columns, every wave holds all BN_ rows: no cross-wave reduction, one atomic per tile element per block.
// P2: Hp and Wp are powers of two (every conv of the CMM / PSN trunks) and the input transform is none / ReLU / LeakyReLU(0.2):
// the pixel decode is two shifts and two masks instead of reciprocal divisions and wrap loops, dY comes through raw buffer loads
// (rows past the block's range / channels past Cout: offset bit 31 -> the range check returns 0, no masks), the activation is one
// v_max per element instead of a switch.  The generic path spent several hundred vector instructions and ~40 branches per
// 128-MFMA chunk on that -- paid in matrix time ("What bounds these fp32 kernels").
template <int BN_, int BKT, bool P2 = false>
__global__ __launch_bounds__(256) void k_conv_wgrad(WgArgs a) {
  constexpr int BMc = 32, NI = BN_ / 16, NJ = BKT / 64, LDY = BN_ + 4, LDX = BKT + 4;
  // two LDS stages for the 128 x 128 tile only (107 -> 101 us): the other shapes lose more to the lower occupancy than they gain
  constexpr int WG_STAGES = (BN_ == 128 && BKT == 128) ? 2 : 1;
  constexpr int YC4 = BN_ / 4, XC4 = BKT / 4;             // float4 columns
  constexpr int YRS = 256 / YC4, XRS = 256 / XC4;         // row stride between a thread's loads
  constexpr int YP = (BMc + YRS - 1) / YRS, XP = BMc / XRS;
  // two LDS stages: chunk i+1 is stored while (other waves still run) the MFMAs of chunk i -- ONE barrier per chunk instead of two
  __shared__ __attribute__((aligned(16))) float Ys2[WG_STAGES][BMc * LDY];
  __shared__ __attribute__((aligned(16))) float Xs2[WG_STAGES][BMc * LDX];
  float* Ys = Ys2[0];
  float* Xs = Xs2[0];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // XCD-aware decode: workgroups are dealt round-robin to the 8 XCDs; keep the tiles of one pixel range (which share
  // dY and the input rows) on one XCD's L2.
  int vid = blockIdx.x;
  const int total = a.gx * a.gy * a.gz;
  if ((total & 7) == 0) vid = (vid & 7) * (total >> 3) + (vid >> 3);
  const int n_blk = (vid % a.gx) * BN_, k_blk = ((vid / a.gx) % a.gy) * BKT;
  const int bz = vid / (a.gx * a.gy);
  const int M = a.B * a.Hp * a.Wp, HW = a.Hp * a.Wp;
  const int m_lo = bz * a.pix_per_block;
  const int m_hi = min(M, m_lo + a.pix_per_block);
  const int c01 = a.cseg[0] + a.cseg[1];
  // ---- X loader: fixed k column per thread
  const int xrow0 = tid / XC4, xc = (tid % XC4) * 4;
  const int kcol = k_blk + xc;
  const int tap = kcol / a.cin, cch = kcol - tap * a.cin;
  const int ky = tap / a.KW, kx = tap - ky * a.KW;
  int seg = 0, cl = cch;
  if (cch >= c01) { seg = 2; cl = cch - c01; }
  else if (cch >= a.cseg[0]) { seg = 1; cl = cch - a.cseg[0]; }
  // (selects, not a.in[seg]: a runtime index would push the argument struct into scratch)
  const float* src = seg == 0 ? a.in[0] : (seg == 1 ? a.in[1] : a.in[2]);
  const int cs = seg == 0 ? a.cseg[0] : (seg == 1 ? a.cseg[1] : a.cseg[2]);
  const float* scp = seg == 0 ? a.in_scale[0] : (seg == 1 ? a.in_scale[1] : a.in_scale[2]);
  const float* shp = seg == 0 ? a.in_shift[0] : (seg == 1 ? a.in_shift[1] : a.in_shift[2]);
  const bool kvalid = kcol < a.K;
  float4 s4 = make_float4(1.f, 1.f, 1.f, 1.f), h4 = make_float4(0.f, 0.f, 0.f, 0.f);
  const bool aff = scp != nullptr;
  if (aff && kvalid) { s4 = *reinterpret_cast<const float4*>(scp + cl); h4 = *reinterpret_cast<const float4*>(shp + cl); }
  const int iy_off = ky * a.dil_y - a.pad_y, ix_off = kx * a.dil_x - a.pad_x;
  // ---- Y loader
  const int yrow0 = tid / YC4, yc = (tid % YC4) * 4;
  const int yn = n_blk + yc;
  float4 yr[YP], xr[XP];
  unsigned xmask = 0;      // bit p: xr[p] holds an in-image value
  // Branch-free loads: every lane reads a clamped, always-valid address and out-of-range elements are zeroed through the
  // masks in sstore.  (A predicated load leaves a PHI at the join; its register copies come with an s_waitcnt vmcnt(0) that
  // lands BEFORE the MFMA block and serialises load latency with compute.)
  const float* src_v = kvalid ? src : a.in[0];
  const int cs_v = kvalid ? cs : a.cseg[0], cl_v = kvalid ? cl : 0;
  const int yn_v = yn < a.Cout ? yn : 0;
  unsigned ymask = 0;
  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy), 0, a.B * a.Hout * a.Wout * a.Cout * 4, 0x00020000);
  const int wmask = a.Wp - 1, hmask = a.Hp - 1;
  auto gload = [&](int m0) {
    if constexpr (P2) {
#pragma unroll
      for (int p = 0; p < YP; ++p) {
        const int row = yrow0 + p * YRS, m = m0 + row;
        const int px = m & wmask, py = (m >> a.lgW) & hmask, b = m >> a.lgHW;
        const bool ok = (YP * YRS == BMc || row < BMc) && m < m_hi && yn < a.Cout;
        const int pix = __mul24(__mul24(b, a.Hout) + py * a.ostep + a.ooy, a.Wout) + px * a.ostep + a.oox;
        const unsigned off = ok ? (unsigned)(__mul24(pix, a.Cout) + yn) * 4u : 0x80000000u;
        yr[p] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(yrs, (int)off, 0, 0));
      }
      xmask = 0;
#pragma unroll
      for (int p = 0; p < XP; ++p) {
        const int m = m0 + xrow0 + p * XRS;
        const int px = m & wmask, py = (m >> a.lgW) & hmask, b = m >> a.lgHW;
        const int iy = py * a.stride + iy_off, ix = px * a.stride + ix_off;
        const bool ok = kvalid && m < m_hi && (unsigned)iy < (unsigned)a.Hin && (unsigned)ix < (unsigned)a.Win;
        const int pix = __mul24(__mul24(b, a.Hin) + iy, a.Win) + ix;
        const int off = ok ? __mul24(pix, cs_v) + cl_v : 0;
        xr[p] = *reinterpret_cast<const float4*>(src_v + off);
        xmask |= (ok ? 1u : 0u) << p;
      }
      return;
    }
    {
      int b, rr, py, px;
      fdivmod(min(m0 + yrow0, M - 1), HW, a.inv_hw, b, rr);
      fdivmod(rr, a.Wp, a.inv_w, py, px);
      ymask = 0;
#pragma unroll
      for (int p = 0; p < YP; ++p) {
        const int row = yrow0 + p * YRS, m = m0 + row;
        const bool ok = row < BMc && m < m_hi && yn < a.Cout;
        const int bc = min(b, a.B - 1);
        const int oy = py * a.ostep + a.ooy, ox = px * a.ostep + a.oox;
        yr[p] = *reinterpret_cast<const float4*>(a.dy + (((size_t)bc * a.Hout + oy) * a.Wout + ox) * a.Cout + yn_v);
        ymask |= (ok ? 1u : 0u) << p;
        px += YRS;                                     // advance the pixel by the row stride without dividing
        while (px >= a.Wp) { px -= a.Wp; if (++py >= a.Hp) { py = 0; ++b; } }
      }
    }
    {
      int b, rr, py, px;
      fdivmod(min(m0 + xrow0, M - 1), HW, a.inv_hw, b, rr);
      fdivmod(rr, a.Wp, a.inv_w, py, px);
      xmask = 0;
#pragma unroll
      for (int p = 0; p < XP; ++p) {
        const int m = m0 + xrow0 + p * XRS;
        const int iy = py * a.stride + iy_off, ix = px * a.stride + ix_off;
        const bool ok = kvalid && m < m_hi && iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win;
        const int iyc = min(max(iy, 0), a.Hin - 1), ixc = min(max(ix, 0), a.Win - 1), bc = min(b, a.B - 1);
        xr[p] = *reinterpret_cast<const float4*>(src_v + (((size_t)bc * a.Hin + iyc) * a.Win + ixc) * cs_v + cl_v);
        xmask |= (ok ? 1u : 0u) << p;      // transformed (affine + activation) in sstore, once the MFMAs of this chunk are issued
        px += XRS;
        while (px >= a.Wp) { px -= a.Wp; if (++py >= a.Hp) { py = 0; ++b; } }
      }
    }
  };
  auto sstore = [&]() {
    if constexpr (P2) {
#pragma unroll
      for (int p = 0; p < YP; ++p) {
        const int row = yrow0 + p * YRS;
        if (YP * YRS == BMc || row < BMc) *reinterpret_cast<float4*>(&Ys[row * LDY + yc]) = yr[p];
      }
      const float sl = a.pro_act == ACT_LEAKY02 ? 0.2f : 0.0f;
#pragma unroll
      for (int p = 0; p < XP; ++p) {
        float4 xv = xr[p];
        if (aff) { xv.x = xv.x * s4.x + h4.x; xv.y = xv.y * s4.y + h4.y; xv.z = xv.z * s4.z + h4.z; xv.w = xv.w * s4.w + h4.w; }
        if (a.pro_act != ACT_NONE) {
          xv.x = fmaxf(xv.x, sl * xv.x); xv.y = fmaxf(xv.y, sl * xv.y); xv.z = fmaxf(xv.z, sl * xv.z); xv.w = fmaxf(xv.w, sl * xv.w);
        }
        if (!((xmask >> p) & 1u)) xv = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(&Xs[(xrow0 + p * XRS) * LDX + xc]) = xv;
      }
      return;
    }
#pragma unroll
    for (int p = 0; p < YP; ++p) {
      const int row = yrow0 + p * YRS;
      if (row < BMc) *reinterpret_cast<float4*>(&Ys[row * LDY + yc]) = ((ymask >> p) & 1u) ? yr[p] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int p = 0; p < XP; ++p) {
      float4 xv = make_float4(0.f, 0.f, 0.f, 0.f);
      if ((xmask >> p) & 1u) {
        xv = xr[p];
        if (aff) { xv.x = xv.x * s4.x + h4.x; xv.y = xv.y * s4.y + h4.y; xv.z = xv.z * s4.z + h4.z; xv.w = xv.w * s4.w + h4.w; }
        float v4[4] = {xv.x, xv.y, xv.z, xv.w};
        apply_act4(v4, a.pro_act, 0.f);
        xv = make_float4(v4[0], v4[1], v4[2], v4[3]);
      }
      *reinterpret_cast<float4*>(&Xs[(xrow0 + p * XRS) * LDX + xc]) = xv;
    }
  };
  const int lr = lane & 15, kq = lane >> 4;
  f32x4 acc[NI][NJ];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (m_lo < m_hi) { gload(m_lo); sstore(); }
  __syncthreads();
  int stage = 0;
  for (int m0 = m_lo; m0 < m_hi; m0 += BMc) {
#ifdef WG_NOLOAD
    const bool more = false;
#else
    const bool more = m0 + BMc < m_hi;
#endif
    if (more) gload(m0 + BMc);
    __builtin_amdgcn_sched_barrier(0);   // keep the loads' consumers (transform + LDS store) behind the MFMA block:
                                         // the scheduler otherwise hoists them, and their vmcnt(0), in front of it
#ifndef WG_NOMFMA
#pragma unroll
    for (int ks = 0; ks < BMc / 4; ++ks) {
      const int row = ks * 4 + kq;
      float av[NI], bv[NJ];
      if constexpr (NI == 1) av[0] = Ys[row * LDY + lr];
      else {
#pragma unroll
        for (int h = 0; h < NI / 4; ++h) {
          const float4 t4 = *reinterpret_cast<const float4*>(&Ys[row * LDY + h * 64 + lr * 4]);
          av[h * 4 + 0] = t4.x; av[h * 4 + 1] = t4.y; av[h * 4 + 2] = t4.z; av[h * 4 + 3] = t4.w;
        }
      }
      if constexpr (NJ == 4) {
        const float4 t4 = *reinterpret_cast<const float4*>(&Xs[row * LDX + wave * 64 + lr * 4]);
        bv[0] = t4.x; bv[1] = t4.y; bv[2] = t4.z; bv[3] = t4.w;
      } else {
        const float2 t2 = *reinterpret_cast<const float2*>(&Xs[row * LDX + wave * 32 + lr * 2]);
        bv[0] = t2.x; bv[1] = t2.y;
      }
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = mfma16(av[i], bv[j], acc[i][j]);
    }
#endif
    __builtin_amdgcn_sched_barrier(0);
    if (WG_STAGES == 2) {
      // sstore writes the OTHER stage (its last readers passed the barrier of the previous chunk); the MFMA loop above read `stage`
      stage ^= 1;
      Ys = Ys2[stage]; Xs = Xs2[stage];
      if (more) sstore();
      __syncthreads();
    } else {
      __syncthreads();
      if (more) sstore();
      __syncthreads();
    }
  }
  if (a.excl) {
    // the block owns tile (n_blk, k_blk) of slot bz: a lane's NJ consecutive k values go out as one vector store, 16 lanes
    // cover a contiguous 16*NJ-float run of one co row.  k in [K, Kp) holds zeros (the X loader zero-fills k >= K).
    float* slotp = a.dw + (long)bz * a.slot_stride;
    const int k0 = k_blk + wave * (NJ * 16) + lr * NJ;
    if (k0 < (int)a.s_co) {
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ii = kq * 4 + r;
          const int n = n_blk + (NI == 1 ? ii : (i >> 2) * 64 + ii * 4 + (i & 3));
          if (n >= a.Cout) continue;
          float* dst = slotp + (long)n * a.s_co + k0;
          if constexpr (NJ == 4) *reinterpret_cast<float4*>(dst) = make_float4(acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]);
          else *reinterpret_cast<float2*>(dst) = make_float2(acc[i][0][r], acc[i][1][r]);
        }
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int k = k_blk + wave * (NJ * 16) + lr * NJ + j;
    if (k >= a.K) continue;
    const int tp = k / a.cin, ci = k - tp * a.cin;
    if (ci >= a.ci_lim) continue;
    const int ty = tp / a.KW, tx = tp - ty * a.KW;
    float* dst = a.dw + (a.nslots > 1 ? (long)(bz % a.nslots) * a.slot_stride : 0L) + a.base + (long)ci * a.s_ci + (long)ty * a.s_ky +
                 (long)tx * a.s_kx;
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ii = kq * 4 + r;
        const int n = n_blk + (NI == 1 ? ii : (i >> 2) * 64 + ii * 4 + (i & 3));
#ifdef WG_NOATOMIC
        if (n < a.co_lim && acc[i][j][r] == 1.2345f) dst[(long)n * a.s_co] = 1.f;
#else
        if (n < a.co_lim) atomicAdd(dst + (long)n * a.s_co, acc[i][j][r]);
#endif
      }
  }
}

// ---------------------------------------------------------------------------------- parameter layout <-> packed (Cout, Kp)
// Both directions are a transpose between the parameter's strided layout (w[base + co*s_co + ci*s_ci + ky*s_ky + kx*s_kx]) and
// the packed rows wp[co][(ky*KW+kx)*cin + ci]: whichever side a thread-per-element kernel walks contiguously, the other side is
// a 36..64-byte-strided gather / scatter (measured: 0.9 ms for one step's 240 MB of packs).  Here a block moves one
// (co_t x ci_t x taps) tile through LDS: it walks the STRIDED side in that side's own memory order (order 0: tap fastest, then
// ci, then co = nn.Conv2d; order 1: tap, then co, then ci = nn.ConvTranspose2d and the (N,K)->(K,N) transposes) and the packed
// side ci-fastest, so both sides move in contiguous runs.
constexpr int PK_LDS = 4608;      // floats: co_t * taps * (ci_t + 1) <= PK_LDS
struct PackDesc {
  const float* w;
  float* wp;
  long s_co, s_ci, s_ky, s_kx, base, n_elems;
  int Cout, Kp, K, cin, KW, co_lim, ci_lim, co_t, ci_t, order, nci, pad_;
};
static_assert(sizeof(PackDesc) == 112, "PackDesc is mirrored by dpmn_amd/model/packing.py (struct format <2Q6q12i)");

// tile shape for a (Cout, cin, taps) pack; shared by the host launchers and (through dpmn_conv_pack_tile_shape) by PackCache
static void pack_tile_shape(int Cout, int cin, int taps, long s_co, long s_ci, int* co_t, int* ci_t, int* order) {
  const long aco = s_co < 0 ? -s_co : s_co, aci = s_ci < 0 ? -s_ci : s_ci;
  if (aco < aci) {            // co is the faster axis of the strided side
    *order = 1;
    int ct = cin < 32 ? cin : 32;
    while (taps * (ct + 1) > PK_LDS && ct > 1) --ct;
    int co = PK_LDS / (taps * (ct + 1));
    co = co > 64 ? 64 : (co < 1 ? 1 : co);
    *ci_t = ct; *co_t = co < Cout ? co : Cout;
  } else {
    *order = 0;
    int ct = cin < 128 ? cin : 128;
    while (taps * (ct + 1) > PK_LDS && ct > 1) --ct;
    int co = (PK_LDS / 2) / (taps * (ct + 1));
    co = co < 1 ? 1 : co;
    *ci_t = ct; *co_t = co < Cout ? co : Cout;
  }
  // small weights: a full-size tile would leave a handful of blocks (a 96 -> 384 linear: 17), and the slotted unpack reads
  // up to 32 copies per element -- trade tile size for blocks until the chip has work
  auto blocks = [&]() { return ((Cout + *co_t - 1) / *co_t) * ((cin + *ci_t - 1) / *ci_t); };
  while (blocks() < 512 && *co_t > 1) *co_t = (*co_t + 1) / 2;
  while (blocks() < 512 && *ci_t > 32) *ci_t = (*ci_t + 1) / 2;
}
static int pack_tile_blocks(const PackDesc& d) { return ((d.Cout + d.co_t - 1) / d.co_t) * d.nci; }

__device__ __forceinline__ int pk_div(int e, float rd) { return (int)(((float)e + 0.5f) * rd); }     // exact for e, e/d < 2^12

// UNPACK = false: strided -> packed (zero fill outside co_lim / ci_lim / K).  UNPACK = true: sum of `nslots` packed copies
// -> += into the strided layout, optionally clearing the packed copies.
template <bool UNPACK>
__device__ __forceinline__ void pack_tile(const PackDesc& d, int blk, float* lds, int nslots, int clear) {
  const int co_blk = blk / d.nci, ci_blk = blk - co_blk * d.nci;
  const int co0 = co_blk * d.co_t, ci0 = ci_blk * d.ci_t;
  const int cot = min(d.co_t, d.Cout - co0), cit = min(d.ci_t, d.cin - ci0);
  const int taps = d.K / d.cin;
  const int n = cot * cit * taps, ldc = d.ci_t + 1;
  const float r_taps = 1.0f / (float)taps, r_cit = 1.0f / (float)cit, r_cot = 1.0f / (float)cot, r_kw = 1.0f / (float)d.KW;
  float* wp = d.wp;
  auto strided_side = [&](int e, bool store) {
    const int q = pk_div(e, r_taps), tap = e - q * taps;
    int co_l, ci_l;
    if (d.order == 0) { co_l = pk_div(q, r_cit); ci_l = q - co_l * cit; }
    else { ci_l = pk_div(q, r_cot); co_l = q - ci_l * cot; }
    const int co = co0 + co_l, ci = ci0 + ci_l;
    const int ty = pk_div(tap, r_kw), tx = tap - ty * d.KW;
    const bool in = co < d.co_lim && ci < d.ci_lim;
    const long addr = d.base + co * d.s_co + ci * d.s_ci + ty * d.s_ky + tx * d.s_kx;
    float* cell = lds + (co_l * taps + tap) * ldc + ci_l;
    if (!store) *cell = in ? d.w[addr] : 0.f;
    else if (in) const_cast<float*>(d.w)[addr] += *cell;
  };
  auto packed_side = [&](int e, bool store) {
    const int q = pk_div(e, r_cit), ci_l = e - q * cit;
    const int co_l = pk_div(q, r_taps), tap = q - co_l * taps;
    const long idx = (long)(co0 + co_l) * d.Kp + tap * d.cin + ci0 + ci_l;
    float* cell = lds + (co_l * taps + tap) * ldc + ci_l;
    if (store) wp[idx] = *cell;
    else {
      float v = 0.f;
#pragma unroll 8
      for (int sl = 0; sl < nslots; ++sl) {
        v += wp[sl * d.n_elems + idx];
        if (clear) wp[sl * d.n_elems + idx] = 0.f;
      }
      *cell = v;
    }
  };
  // many copies of a small tile (the exclusive-slot weight gradients of huge-M convs: up to 256 copies, tiles shrunk to a few
  // dozen elements for parallelism): SG threads per element each sum every SG-th copy -- the chain of dependent loads per
  // thread is what bounds this kernel -- and the partial sums meet in LDS
  int SG = 1;
  if (UNPACK && !clear && cot * taps * ldc <= PK_LDS - 256)
    while (SG < 16 && SG * 2 * n <= (int)blockDim.x && SG * 2 <= nslots) SG *= 2;
  if (UNPACK && SG > 1) {
    float* part = lds + PK_LDS - 256;            // [SG][n], SG * n <= 256; the tile cells live below (n <= 128 here)
    const int e = threadIdx.x % n, g = threadIdx.x / n;
    if (g < SG) {
      const int q = pk_div(e, r_cit), ci_l = e - q * cit;
      const int co_l = pk_div(q, r_taps), tap = q - co_l * taps;
      const long idx = (long)(co0 + co_l) * d.Kp + tap * d.cin + ci0 + ci_l;
      float v = 0.f;
#pragma unroll 8
      for (int sl = g; sl < nslots; sl += SG) v += wp[sl * d.n_elems + idx];
      part[g * n + e] = v;
    }
    __syncthreads();
    if (g == 0) {
      const int q = pk_div(e, r_cit), ci_l = e - q * cit;
      const int co_l = pk_div(q, r_taps), tap = q - co_l * taps;
      float v = 0.f;
      for (int h = 0; h < SG; ++h) v += part[h * n + e];
      lds[(co_l * taps + tap) * ldc + ci_l] = v;
    }
  } else {
    for (int e = threadIdx.x; e < n; e += blockDim.x) { if (UNPACK) packed_side(e, false); else strided_side(e, false); }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < n; e += blockDim.x) { if (UNPACK) strided_side(e, true); else packed_side(e, true); }
  const int tail = d.Kp - d.K;      // K padding of the packed rows: zero (pack) / cleared (unpack)
  if (ci_blk == 0 && tail > 0 && (!UNPACK || clear)) {
    const float r_tail = 1.0f / (float)tail;
    for (int e = threadIdx.x; e < cot * tail; e += blockDim.x) {
      const int r = pk_div(e, r_tail), c = e - r * tail;
      for (int sl = 0; sl < (UNPACK ? nslots : 1); ++sl) wp[sl * d.n_elems + (long)(co0 + r) * d.Kp + d.K + c] = 0.f;
    }
  }
}

__global__ __launch_bounds__(256) void k_wgrad_unpack(PackDesc d, int clear, int nslots) {
  __shared__ float lds[PK_LDS];
  pack_tile<true>(d, blockIdx.x, lds, nslots, clear);
}

__global__ __launch_bounds__(256) void k_conv_pack(PackDesc d) {
  __shared__ float lds[PK_LDS];
  pack_tile<false>(d, blockIdx.x, lds, 1, 0);
}

// All weight packs of a training step in ONE launch: block b finds its descriptor by binary search in the block prefix and moves
// one tile of it.  (The step re-packs ~135 weights -- forward, data-gradient and transposed variants -- because the parameters
// changed; one launch each was 1.0 ms of 7 us launches.)
__global__ __launch_bounds__(256) void k_conv_pack_multi(const PackDesc* __restrict__ descs, const int* __restrict__ block_prefix, int n_desc) {
  __shared__ float lds[PK_LDS];
  int lo = 0, hi = n_desc - 1;                       // largest d with block_prefix[d] <= blockIdx.x
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (block_prefix[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const PackDesc d = descs[lo];
  pack_tile<false>(d, (int)blockIdx.x - block_prefix[lo], lds, 1, 0);
}

// The weight-gradient unpacks of a whole backward pass (the CMM's ~40 convs) in ONE launch, like k_conv_pack_multi: the slot count of a
// descriptor travels in its last field (exclusive-slot workspaces: nothing to clear).  Destinations of different descriptors are
// disjoint (own weights; the four phases of a ConvTranspose2d(4,2,1) write interleaved elements of one tensor).
__global__ __launch_bounds__(256) void k_wgrad_unpack_multi(const PackDesc* __restrict__ descs, const int* __restrict__ block_prefix, int n_desc) {
  __shared__ float lds[PK_LDS];
  int lo = 0, hi = n_desc - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (block_prefix[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const PackDesc d = descs[lo];
  pack_tile<true>(d, (int)blockIdx.x - block_prefix[lo], lds, d.pad_ > 1 ? d.pad_ : 1, 0);
}

// ---------------------------------------------------------------------------------- train-mode BatchNorm
// stats (32,2,C) DOUBLES = slotted (sum, sumsq) over `count` values per channel (accumulated by the conv epilogue with fp64 atomics)
// clear: zero the slots after reading them (a persistent statistics buffer then needs no memset per conv); nbt: the module's
// num_batches_tracked counter, incremented here instead of by a separate one-element kernel
__global__ void k_bn_finalize(float* __restrict__ stats, const float* __restrict__ gamma, const float* __restrict__ beta,
                              float count, float eps, float momentum, float* __restrict__ scale, float* __restrict__ shift,
                              float* __restrict__ mean_out, float* __restrict__ rstd_out, float* __restrict__ running_mean,
                              float* __restrict__ running_var, int C, long long* __restrict__ nbt, int clear) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c == 0 && nbt) *nbt += 1;
  if (c >= C) return;
  double* sd = reinterpret_cast<double*>(stats);      // (32, 2, C) doubles (conv.hip STAT_SLOTS)
  double s1 = 0.0, s2 = 0.0;
  for (int s0 = 0; s0 < 32; s0 += 8) {   // STAT_SLOTS; eight slots' loads in flight, then their clears (a store between two loads of
    double a[8], q[8];                     // the same array orders them: 64 dependent latencies on the CMM forward's main chain)
#pragma unroll
    for (int u = 0; u < 8; ++u) { a[u] = sd[(size_t)(s0 + u) * 2 * C + c]; q[u] = sd[(size_t)(s0 + u) * 2 * C + C + c]; }
#pragma unroll
    for (int u = 0; u < 8; ++u) { s1 += a[u]; s2 += q[u]; }
    if (clear) {
#pragma unroll
      for (int u = 0; u < 8; ++u) { sd[(size_t)(s0 + u) * 2 * C + c] = 0.0; sd[(size_t)(s0 + u) * 2 * C + C + c] = 0.0; }
    }
  }
  const double mean_d = s1 / (double)count;
  const float mean = (float)mean_d;
  float var = (float)(s2 / (double)count - mean_d * mean_d);   // biased variance used for normalisation
  var = var > 0.f ? var : 0.f;
  const float rstd = 1.0f / sqrtf(var + eps);
  const float s = gamma[c] * rstd;
  scale[c] = s;
  shift[c] = beta[c] - mean * s;
  mean_out[c] = mean;
  rstd_out[c] = rstd;
  if (running_mean) {
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * var * (count / (count - 1.f));   // unbiased
  }
}
// G (+)= dA * act'(scale*r + shift) ; r, dA, G: (pixels, C) NHWC with optional strides for G
__global__ void k_affine_act_bwd(const float* __restrict__ dA, const float* __restrict__ r, const float* __restrict__ scale,
                                 const float* __restrict__ shift, int act, float* __restrict__ G, int accumulate, long pixels, int C) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= pixels * C) return;
  const int c = idx % C;
  const float z = scale ? r[idx] * scale[c] + shift[c] : r[idx];
  float g = dA[idx];
  switch (act) {
    case ACT_RELU: g = z > 0.f ? g : 0.f; break;
    case ACT_LEAKY02: g = z > 0.f ? g : 0.2f * g; break;
    case ACT_LEAKY001: g = z > 0.f ? g : 0.01f * g; break;
    default: break;
  }
  G[idx] = accumulate ? G[idx] + g : g;
}
// four channels of one pixel per thread (C % 4 == 0): no per-element modulo, 16-byte loads / stores
__global__ void k_affine_act_bwd_v4(const float* __restrict__ dA, const float* __restrict__ r, const float* __restrict__ scale,
                                    const float* __restrict__ shift, int act, float* __restrict__ G, int accumulate, long quads, int Cq) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= quads) return;
  const int cq = (int)(idx % Cq);
  float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
  if (scale) { sc = *reinterpret_cast<const float4*>(scale + cq * 4); sh = *reinterpret_cast<const float4*>(shift + cq * 4); }
  const float slope = act == ACT_RELU ? 0.f : (act == ACT_LEAKY02 ? 0.2f : (act == ACT_LEAKY001 ? 0.01f : 1.f));
  const float4 x = *reinterpret_cast<const float4*>(r + idx * 4);
  float4 g = *reinterpret_cast<const float4*>(dA + idx * 4);
  g.x = (x.x * sc.x + sh.x) > 0.f ? g.x : slope * g.x; g.y = (x.y * sc.y + sh.y) > 0.f ? g.y : slope * g.y;
  g.z = (x.z * sc.z + sh.z) > 0.f ? g.z : slope * g.z; g.w = (x.w * sc.w + sh.w) > 0.f ? g.w : slope * g.w;
  if (accumulate) {
    const float4 o = *reinterpret_cast<const float4*>(G + idx * 4);
    g.x = o.x + g.x; g.y = o.y + g.y; g.z = o.z + g.z; g.w = o.w + g.w;
  }
  *reinterpret_cast<float4*>(G + idx * 4) = g;
}
// The same with the BatchNorm-backward reduction of the PRODUCER folded in: the call that completes G (the last consumer's) has
// the final value of every element in registers, so sum G and sum G * xhat per channel are accumulated here -- block-reduced,
// then one fp64 atomic per channel and block (order-independent after the final rounding, like the forward statistics) -- and
// dpmn_bn_bwd_f32's own pass over G and r (k_bn_bwd_reduce: 33 launches, 1.2 ms per step) disappears.  One float4 (4 channels
// of one pixel) per thread and iteration; C % 4 == 0, C / 4 divides 256.
__global__ __launch_bounds__(256) void k_affine_act_bwd_stats(const float* __restrict__ dA, const float* __restrict__ r,
                                                               const float* __restrict__ scale, const float* __restrict__ shift, int act,
                                                               float* __restrict__ G, int accumulate, long pixels, int C,
                                                               const float* __restrict__ mean, const float* __restrict__ rstd,
                                                               double* __restrict__ sums, int pix_per_block) {
  __shared__ float red[256][8];
  const int Cq = C >> 2, PL = 256 / Cq;
  const int cq = threadIdx.x % Cq, pl = threadIdx.x / Cq;
  const long p0 = (long)blockIdx.x * pix_per_block;
  const long p1 = p0 + pix_per_block < pixels ? p0 + pix_per_block : pixels;
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  if (pl < PL) {
    const float4 mu = *reinterpret_cast<const float4*>(mean + cq * 4), rs = *reinterpret_cast<const float4*>(rstd + cq * 4);
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (scale) { sc = *reinterpret_cast<const float4*>(scale + cq * 4); sh = *reinterpret_cast<const float4*>(shift + cq * 4); }
    const float slope = act == ACT_RELU ? 0.f : (act == ACT_LEAKY02 ? 0.2f : (act == ACT_LEAKY001 ? 0.01f : 1.f));
    // (-ffp-contract=off: z = r * scale + shift is a multiply then an add, as in k_affine_act_bwd and the forward's on-load affine)
    auto one = [&](const float4& x, float4 g, const float4& o, long e) {
      g.x = (x.x * sc.x + sh.x) > 0.f ? g.x : slope * g.x; g.y = (x.y * sc.y + sh.y) > 0.f ? g.y : slope * g.y;
      g.z = (x.z * sc.z + sh.z) > 0.f ? g.z : slope * g.z; g.w = (x.w * sc.w + sh.w) > 0.f ? g.w : slope * g.w;
      if (accumulate) { g.x = o.x + g.x; g.y = o.y + g.y; g.z = o.z + g.z; g.w = o.w + g.w; }
      *reinterpret_cast<float4*>(G + e) = g;
      s1[0] += g.x; s1[1] += g.y; s1[2] += g.z; s1[3] += g.w;
      s2[0] += g.x * (x.x - mu.x) * rs.x; s2[1] += g.y * (x.y - mu.y) * rs.y;
      s2[2] += g.z * (x.z - mu.z) * rs.z; s2[3] += g.w * (x.w - mu.w) * rs.w;
    };
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    long p = p0 + pl;
    for (; p + 3 * PL < p1; p += 4 * PL) {      // 4 pixels per thread in flight: 8 (12) independent float4 loads
      float4 x[4], g[4], o[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long e = (p + u * PL) * C + cq * 4;
        x[u] = *reinterpret_cast<const float4*>(r + e);
        g[u] = *reinterpret_cast<const float4*>(dA + e);
        o[u] = accumulate ? *reinterpret_cast<const float4*>(G + e) : z4;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) one(x[u], g[u], o[u], (p + u * PL) * C + cq * 4);
    }
    for (; p < p1; p += PL) {
      const long e = p * C + cq * 4;
      one(*reinterpret_cast<const float4*>(r + e), *reinterpret_cast<const float4*>(dA + e),
          accumulate ? *reinterpret_cast<const float4*>(G + e) : z4, e);
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) { red[threadIdx.x][q] = s1[q]; red[threadIdx.x][4 + q] = s2[q]; }
  __syncthreads();
  if (pl == 0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float a = 0.f, b2 = 0.f;
      for (int l = 0; l < PL; ++l) { a += red[l * Cq + cq][q]; b2 += red[l * Cq + cq][4 + q]; }
      atomicAdd(sums + cq * 4 + q, (double)a);
      atomicAdd(sums + C + cq * 4 + q, (double)b2);
    }
  }
}
// sums (2,C): sum G, sum G * xhat
__global__ __launch_bounds__(256) void k_bn_bwd_reduce(const float* __restrict__ G, const float* __restrict__ r,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd,
                                                        float* __restrict__ sums, long pixels, int C, int pix_per_block) {
  // thread = (pixel lane, channel quad): float4 loads, 256 / (C/4) pixels in flight per block, LDS reduction over the pixel
  // lanes, then ONE atomic per channel per block (the launcher keeps the block count low: same-address atomics serialise)
  __shared__ float red[256][8];
  const int Cq = C >> 2;
  const int PL = 256 / Cq;                    // C/4 divides 256 for every BatchNorm width on the path (64 ... 512 channels)
  const int cq = threadIdx.x % Cq, pl = threadIdx.x / Cq;
  const long p0 = (long)blockIdx.x * pix_per_block;
  const long p1 = p0 + pix_per_block < pixels ? p0 + pix_per_block : pixels;
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  if (pl < PL) {
    const float4 mu = *reinterpret_cast<const float4*>(mean + cq * 4), rs = *reinterpret_cast<const float4*>(rstd + cq * 4);
    // 4 pixels per thread in flight (8 independent float4 loads), then the tail
    long p = p0 + pl;
    for (; p + 3 * PL < p1; p += 4 * PL) {
      float4 g[4], x[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        g[u] = *reinterpret_cast<const float4*>(G + (p + u * PL) * C + cq * 4);
        x[u] = *reinterpret_cast<const float4*>(r + (p + u * PL) * C + cq * 4);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        s1[0] += g[u].x; s1[1] += g[u].y; s1[2] += g[u].z; s1[3] += g[u].w;
        s2[0] += g[u].x * (x[u].x - mu.x) * rs.x; s2[1] += g[u].y * (x[u].y - mu.y) * rs.y;
        s2[2] += g[u].z * (x[u].z - mu.z) * rs.z; s2[3] += g[u].w * (x[u].w - mu.w) * rs.w;
      }
    }
    for (; p < p1; p += PL) {
      const float4 g = *reinterpret_cast<const float4*>(G + p * C + cq * 4);
      const float4 x = *reinterpret_cast<const float4*>(r + p * C + cq * 4);
      s1[0] += g.x; s1[1] += g.y; s1[2] += g.z; s1[3] += g.w;
      s2[0] += g.x * (x.x - mu.x) * rs.x; s2[1] += g.y * (x.y - mu.y) * rs.y;
      s2[2] += g.z * (x.z - mu.z) * rs.z; s2[3] += g.w * (x.w - mu.w) * rs.w;
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) { red[threadIdx.x][q] = s1[q]; red[threadIdx.x][4 + q] = s2[q]; }
  __syncthreads();
  if (pl == 0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float a = 0.f, b2 = 0.f;
      for (int l = 0; l < PL; ++l) { a += red[l * Cq + cq][q]; b2 += red[l * Cq + cq][4 + q]; }
      atomicAdd(sums + cq * 4 + q, a);
      atomicAdd(sums + C + cq * 4 + q, b2);
    }
  }
}
// dr = gamma*rstd*(G - sums0/count - xhat*sums1/count) ; dgamma += sums1 ; dbeta += sums0 (done once by block 0)
template <typename ST>      // sums as fp32 (k_bn_bwd_reduce) or fp64 (k_affine_act_bwd_stats): converted on load, no separate pass
__global__ void k_bn_bwd_apply(const float* __restrict__ G, const float* __restrict__ r, const float* __restrict__ gamma,
                               const float* __restrict__ mean, const float* __restrict__ rstd, const ST* __restrict__ sums,
                               float count, float* __restrict__ dr, float* __restrict__ dgamma, float* __restrict__ dbeta,
                               long pixels, int C) {
  // one float4 (4 channels of one pixel) per thread; C % 4 == 0
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < C && dgamma) { atomicAdd(dgamma + idx, (float)sums[C + idx]); atomicAdd(dbeta + idx, (float)sums[idx]); }
  const long e = idx * 4;
  if (e >= pixels * C) return;
  const int c = (int)(e % C);
  const float4 g4 = *reinterpret_cast<const float4*>(G + e), r4 = *reinterpret_cast<const float4*>(r + e);
  const float4 ga = *reinterpret_cast<const float4*>(gamma + c), mu = *reinterpret_cast<const float4*>(mean + c);
  const float4 rs = *reinterpret_cast<const float4*>(rstd + c);
  const float4 a0 = make_float4((float)sums[c], (float)sums[c + 1], (float)sums[c + 2], (float)sums[c + 3]);
  const float4 a1 = make_float4((float)sums[C + c], (float)sums[C + c + 1], (float)sums[C + c + 2], (float)sums[C + c + 3]);
  float4 o;
  o.x = ga.x * rs.x * (g4.x - a0.x / count - (r4.x - mu.x) * rs.x * a1.x / count);
  o.y = ga.y * rs.y * (g4.y - a0.y / count - (r4.y - mu.y) * rs.y * a1.y / count);
  o.z = ga.z * rs.z * (g4.z - a0.z / count - (r4.z - mu.z) * rs.z * a1.z / count);
  o.w = ga.w * rs.w * (g4.w - a0.w / count - (r4.w - mu.w) * rs.w * a1.w / count);
  *reinterpret_cast<float4*>(dr + e) = o;
}

// ---------------------------------------------------------------------------------- CMM channel gate backward
// g = x * (1 + w), w = sigmoid(fc2(relu(fc1(mean_p x)))).
// Pass 1, one workgroup per image: recompute the gate, form dx and leave (S, relu(h), dlogit, dh) in ws[b].
// Pass 2, one thread per weight element: reduce the per-image outer products over the batch (no atomics).
// pass 1a: S[b][c] = mean_p x, Dw[b][c] = sum_p dg * x ; block = (image, 64 channels), 4 pixel lanes per channel
__global__ __launch_bounds__(256) void k_se_gate_bwd_stats(const float* __restrict__ x, const float* __restrict__ dg,
                                                            float* __restrict__ S, float* __restrict__ Dw, int P, int C) {
  __shared__ float red[4][64][2];
  const int b = blockIdx.x, c = blockIdx.y * 64 + (threadIdx.x & 63), pl = threadIdx.x >> 6;
  const float* xb = x + (size_t)b * P * C;
  const float* gb = dg + (size_t)b * P * C;
  float s = 0.f, dw = 0.f;
  if (c < C)
    for (int p = pl; p < P; p += 4) { const float xv = xb[(size_t)p * C + c]; s += xv; dw += gb[(size_t)p * C + c] * xv; }
  red[pl][threadIdx.x & 63][0] = s; red[pl][threadIdx.x & 63][1] = dw;
  __syncthreads();
  if (pl == 0 && c < C) {
    const int l = threadIdx.x;
    S[(size_t)b * C + c] = (red[0][l][0] + red[1][l][0] + red[2][l][0] + red[3][l][0]) / (float)P;
    Dw[(size_t)b * C + c] = red[0][l][1] + red[1][l][1] + red[2][l][1] + red[3][l][1];
  }
}
// gate elementwise: w = sigmoid(logit); dlog = Dw * w * (1 - w); onepw = 1 + w        (B x C)
__global__ void k_se_gate_elem(const float* __restrict__ logit, const float* __restrict__ Dw, float* __restrict__ dlog,
                               float* __restrict__ onepw, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float w = sigmoid_f(logit[i]);
  dlog[i] = Dw[i] * w * (1.f - w);
  onepw[i] = 1.f + w;
}
__global__ void k_relu_mask(const float* __restrict__ hrelu, float* __restrict__ dh, long n) {   // dh *= (h > 0)
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && !(hrelu[i] > 0.f)) dh[i] = 0.f;
}
__global__ void k_transpose(const float* __restrict__ a, float* __restrict__ at, int R, int Cc) {   // a (R, Cc) -> at (Cc, R)
  __shared__ float tile[32][33];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int j = ty; j < 32; j += 8)
    if (by + j < R && bx + tx < Cc) tile[j][tx] = a[(size_t)(by + j) * Cc + bx + tx];
  __syncthreads();
  for (int j = ty; j < 32; j += 8)
    if (bx + j < Cc && by + tx < R) at[(size_t)(bx + j) * R + by + tx] = tile[tx][j];
}
// pass 1c: dx = dg * (1 + gate) + dS / P, elementwise over (image, pixel, channel)
__global__ void k_se_gate_bwd_dx(const float* __restrict__ dg, const float* __restrict__ onepw, const float* __restrict__ dS,
                                 float* __restrict__ dx, int P, int C, float inv_p, long total4) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total4) return;
  const long e = i * 4;
  const int c = (int)(e % C);
  const long b = e / ((long)P * C);
  const float4 g = *reinterpret_cast<const float4*>(dg + e);
  const float4 w1 = *reinterpret_cast<const float4*>(onepw + b * C + c), ds = *reinterpret_cast<const float4*>(dS + b * C + c);
  *reinterpret_cast<float4*>(dx + e) = make_float4(g.x * w1.x + ds.x * inv_p, g.y * w1.y + ds.y * inv_p, g.z * w1.z + ds.z * inv_p,
                                                   g.w * w1.w + ds.w * inv_p);
}

// ---------------------------------------------------------------------------------- DistillModule tail + optimizer
// f = relu(scale*r + shift) on NHWC (pixels, C) ; distill_module.py:21-27
__global__ void k_affine_act_fwd(const float* __restrict__ r, const float* __restrict__ scale, const float* __restrict__ shift,
                                 int act, float* __restrict__ y, long pixels, int C) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= pixels * C) return;
  const int c = idx % C;
  const float z = scale ? r[idx] * scale[c] + shift[c] : r[idx];
  y[idx] = apply_act(z, act, 0.f);
}
// L1 between two feature maps: part[blk] = sum |a-b| ; optional gradient kernel below
__global__ __launch_bounds__(256) void k_l1_partial(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ part,
                                                     long n) {
  __shared__ float red[4];
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  float v = i < n ? fabsf(a[i] - b[i]) : 0.f;
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void k_sum_final(const float* __restrict__ part, int n, float mul, float* __restrict__ out) {
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 64) s += part[i];
  for (int o = 32; o > 0; o >>= 1) s += xshfl_v(s, o);
  if (threadIdx.x == 0) out[0] = (float)(s * mul);
}
// da = gs*c*sign(a-b) (+ extra) ; db = -gs*c*sign(a-b)
__global__ void k_l1_bwd(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ gscale, float c,
                         const float* __restrict__ extra_a, float* __restrict__ da, float* __restrict__ db, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float d = a[i] - b[i];
  const float g = gscale[0] * c * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
  da[i] = g + (extra_a ? extra_a[i] : 0.f);
  db[i] = -g;
}
// sum of squares of a flat buffer -> part[blk]; final: out[0] = sum
__global__ __launch_bounds__(256) void k_sumsq_partial(const float* __restrict__ x, float* __restrict__ part, long n) {
  __shared__ float red[4];
  float s = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) { const float v = x[i]; s += v * v; }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
// clip_grad_norm_(max_norm) + Adam (torch.optim.Adam semantics, no weight decay / amsgrad); normsq[0] = ||g||^2
__global__ void k_adam_clip(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            const float* __restrict__ normsq, float max_norm, float lr, float b1, float b2, float eps, float bc1,
                            float bc2_sqrt, long n, const float* __restrict__ step_dev) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (step_dev) {      // step count lives on the device (hipGraph replay: host scalars are frozen at capture time)
    const float t = step_dev[0];
    bc1 = 1.f - powf(b1, t);
    bc2_sqrt = sqrtf(1.f - powf(b2, t));
  }
  float coef = 1.f;
  if (max_norm > 0.f) { coef = max_norm / (sqrtf(normsq[0]) + 1e-6f); coef = coef < 1.f ? coef : 1.f; }
  const float gi = g[i] * coef;
  const float mi = b1 * m[i] + (1.f - b1) * gi;
  const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
  m[i] = mi; v[i] = vi;
  p[i] -= (lr / bc1) * mi / (sqrtf(vi) / bc2_sqrt + eps);
}

}  // namespace

extern "C" {

// excl_slots: 0 = atomic accumulation (nslots copies); > 0 = exclusive mode, must equal the split count of the plan;
// plan_only != nullptr: only report the split count exclusive mode would use
static int launch_wgrad(const dpmn_conv_desc* d, const float* dy, float* dw, int co_lim, int ci_lim, long s_co, long s_ci,
                        long s_ky, long s_kx, long base, int nslots, long slot_stride, dpmn_stream_t stream, int excl_slots = 0,
                        int* plan_only = nullptr) {
  DPMN_REQUIRE(d && d->in[0] && (plan_only || (dy && dw)), "conv2d_wgrad: null pointer");
  WgArgs a{};
  int cin = 0;
  for (int s = 0; s < 3; ++s) {
    a.in[s] = d->in[s]; a.in_scale[s] = d->in_scale[s]; a.in_shift[s] = d->in_shift[s]; a.cseg[s] = d->in[s] ? d->cseg[s] : 0;
    DPMN_REQUIRE(a.cseg[s] % 4 == 0, "conv2d_wgrad: segment channel counts must be multiples of 4");
    cin += a.cseg[s];
  }
  a.cin = cin; a.B = d->B; a.Hin = d->Hin; a.Win = d->Win; a.KH = d->KH; a.KW = d->KW; a.stride = d->stride;
  a.dil_y = d->dil_y; a.dil_x = d->dil_x; a.pad_y = d->pad_y; a.pad_x = d->pad_x; a.Hp = d->Hp; a.Wp = d->Wp;
  a.Hout = d->Hout; a.Wout = d->Wout; a.ostep = d->ostep; a.ooy = d->ooy; a.oox = d->oox; a.pro_act = d->pro_act;
  a.dy = dy; a.Cout = d->Cout; a.dw = dw;
  a.K = d->KH * d->KW * cin;
  a.s_co = s_co; a.s_ci = s_ci; a.s_ky = s_ky; a.s_kx = s_kx; a.base = base;
  a.nslots = nslots > 1 ? nslots : 1; a.slot_stride = slot_stride;
  a.co_lim = co_lim < a.Cout ? co_lim : a.Cout; a.ci_lim = ci_lim < cin ? ci_lim : cin;
  const long Ml = (long)a.B * a.Hp * a.Wp;
  DPMN_REQUIRE(Ml > 0 && Ml < (1L << 24), "conv2d_wgrad: pixel count must be below 2^24");
  DPMN_REQUIRE(a.Cout % 4 == 0, "conv2d_wgrad: dy must carry a multiple of 4 channels (pad the output gradient)");
  const int M = (int)Ml;
  a.inv_hw = 1.0f / (float)(a.Hp * a.Wp); a.inv_w = 1.0f / (float)a.Wp;
  const int bn = a.Cout <= 16 ? 16 : (a.Cout <= 64 ? 64 : 128);
  int bk = bn == 128 ? 128 : 256;
  if (bn == 64 && cdiv(a.K, 128) * 128 < cdiv(a.K, 256) * 256) bk = 128;   // less padding in the last k tile
  a.gx = cdiv(a.Cout, bn); a.gy = cdiv(a.K, bk);
  const int tiles = a.gx * a.gy;
  // every block ends with one atomic per VALID tile element: keep >= 1 pixel of MFMA work per 32 of them (512 pixels for a
  // full 128x128 tile, down to 64 for the 4 x 72 tiles of the DistillModule convs, which are latency-bound and want blocks)
  const int tile_elems = (a.Cout < bn ? a.Cout : bn) * (a.K < bk ? a.K : bk);
  int min_ppb = tile_elems / 32 / a.nslots;      // slotted destinations divide the same-address pile-up
  min_ppb = min_ppb < 64 ? 64 : (min_ppb > 512 ? 512 : min_ppb);
  if (a.nslots == 1 && min_ppb < 512) min_ppb = 512;
  int splits = cdiv(2048, tiles);
  int ppb = cdiv(cdiv(M, splits), 32) * 32;
  if (ppb < min_ppb) ppb = min_ppb;
  splits = cdiv(M, ppb);
  if (excl_slots > 0 || plan_only) {
    // no atomic epilogue to amortise: split for parallelism (~768 blocks), bounded by the slot round trip (every split writes
    // and the unpack reads one Cout x Kp slot: <= 32 MB in flight, but never fewer than 2 splits of a large M)
    static const int want_blocks = getenv("DPMN_WG_BLOCKS") ? atoi(getenv("DPMN_WG_BLOCKS")) : 768;
    static const long cap_bytes = (long)(getenv("DPMN_WG_CAP_MB") ? atoi(getenv("DPMN_WG_CAP_MB")) : 32) << 20;
    const long slot_bytes = (long)a.Cout * s_co * 4;
    long sp = cdiv(want_blocks, tiles);
    const long by_cap = cap_bytes / slot_bytes < 2 ? 2 : cap_bytes / slot_bytes;
    if (sp > by_cap) sp = by_cap;
    if (sp > M / 64) sp = M / 64;
    if (sp > 256) sp = 256;          // the unpack sums the copies: 256 is what its slot-parallel reduction is sized for
    if (sp < 1) sp = 1;
    ppb = cdiv(cdiv(M, (int)sp), 32) * 32;
    splits = cdiv(M, ppb);
    // a few hundred gradient values over 10^5 pixels (the 4-channel DistillModule convs): the MFMA tile is mostly padding and
    // the launch is latency-bound either way; 32 slotted atomic copies measured faster there (60 vs 95 us) -> report 0
    if (plan_only) { *plan_only = (bn == 16 && a.K <= 128) ? 0 : splits; return DPMN_OK; }
    DPMN_REQUIRE(excl_slots == splits, "conv2d_wgrad_excl: slots must be the count dpmn_conv2d_wgrad_excl_slots reports");
    a.excl = 1; a.nslots = splits;
  }
  a.pix_per_block = ppb; a.gz = splits;
  const dim3 grid((unsigned)(tiles * splits));
  // 2 M Cout K FLOPs; bytes: the input and dY read once, one (Cout, Kp) partial tile set written per split (exclusive slots)
  ProfScope prof(PT_CONV_WGRAD, as_stream(stream), 2.0 * M * (double)a.Cout * a.K,
                 4.0 * ((double)a.B * a.Hin * a.Win * cin + (double)a.B * a.Hout * a.Wout * a.Cout + (double)(a.excl ? splits : 1) * a.Cout * a.K));
  static const int p2_on = getenv("DPMN_WG_P2") ? atoi(getenv("DPMN_WG_P2")) : 1;
  auto pow2 = [](int v) { return v > 0 && (v & (v - 1)) == 0; };
  bool p2 = p2_on && pow2(a.Hp) && pow2(a.Wp) && (a.pro_act == ACT_NONE || a.pro_act == ACT_RELU || a.pro_act == ACT_LEAKY02) &&
            (size_t)a.B * a.Hout * a.Wout * a.Cout * 4 < (1ull << 31);
  for (int s = 0; s < 3; ++s) p2 = p2 && (size_t)a.B * a.Hin * a.Win * (a.cseg[s] > 0 ? a.cseg[s] : 1) * 4 < (1ull << 31);
  if (p2) {
    for (a.lgW = 0; (1 << a.lgW) < a.Wp; ++a.lgW) {}
    for (a.lgHW = 0; (1 << a.lgHW) < a.Hp * a.Wp; ++a.lgHW) {}
    if (x3_on(16) && bn != 16 && dpmn_conv::x3_wgrad_ok(a, bn, bk)) (void)dpmn_conv::x3_launch_wgrad(a, bn, bk, grid, as_stream(stream));
    else if (bn == 16) hipLaunchKernelGGL((k_conv_wgrad<16, 256, true>), grid, dim3(256), 0, as_stream(stream), a);
    else if (bn == 64 && bk == 128) hipLaunchKernelGGL((k_conv_wgrad<64, 128, true>), grid, dim3(256), 0, as_stream(stream), a);
    else if (bn == 64) hipLaunchKernelGGL((k_conv_wgrad<64, 256, true>), grid, dim3(256), 0, as_stream(stream), a);
    else hipLaunchKernelGGL((k_conv_wgrad<128, 128, true>), grid, dim3(256), 0, as_stream(stream), a);
  } else
  if (bn == 16) hipLaunchKernelGGL((k_conv_wgrad<16, 256>), grid, dim3(256), 0, as_stream(stream), a);
  else if (bn == 64 && bk == 128) hipLaunchKernelGGL((k_conv_wgrad<64, 128>), grid, dim3(256), 0, as_stream(stream), a);
  else if (bn == 64) hipLaunchKernelGGL((k_conv_wgrad<64, 256>), grid, dim3(256), 0, as_stream(stream), a);
  else hipLaunchKernelGGL((k_conv_wgrad<128, 128>), grid, dim3(256), 0, as_stream(stream), a);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

int dpmn_conv2d_wgrad_f32(const dpmn_conv_desc* d, const float* dy, float* dwp, int nslots, dpmn_stream_t stream) {
  DPMN_REQUIRE(d, "conv2d_wgrad: null descriptor");
  int cin = 0;
  for (int s = 0; s < 3; ++s) cin += d->in[s] ? d->cseg[s] : 0;
  const long Kp = ((long)(d->KH * d->KW * cin + 31) / 32) * 32;
  return launch_wgrad(d, dy, dwp, d->Cout, cin, Kp, 1, (long)d->KW * cin, cin, 0, nslots, (long)d->Cout * Kp, stream);
}

int dpmn_conv2d_wgrad_excl_slots(const dpmn_conv_desc* d, int* slots) {
  DPMN_REQUIRE(d && slots, "conv2d_wgrad_excl_slots: null pointer");
  int cin = 0;
  for (int s = 0; s < 3; ++s) cin += d->in[s] ? d->cseg[s] : 0;
  const long Kp = ((long)(d->KH * d->KW * cin + 31) / 32) * 32;
  return launch_wgrad(d, nullptr, nullptr, d->Cout, cin, Kp, 1, (long)d->KW * cin, cin, 0, 1, (long)d->Cout * Kp, nullptr, 0, slots);
}

int dpmn_conv2d_wgrad_excl_f32(const dpmn_conv_desc* d, const float* dy, float* dwp, int slots, dpmn_stream_t stream) {
  DPMN_REQUIRE(d && slots > 0, "conv2d_wgrad_excl: bad arguments");
  int cin = 0;
  for (int s = 0; s < 3; ++s) cin += d->in[s] ? d->cseg[s] : 0;
  const long Kp = ((long)(d->KH * d->KW * cin + 31) / 32) * 32;
  return launch_wgrad(d, dy, dwp, d->Cout, cin, Kp, 1, (long)d->KW * cin, cin, 0, slots, (long)d->Cout * Kp, stream, slots);
}

static PackDesc make_pack_desc(const float* w, float* wp, int Cout, int cin, int KH, int KW, int co_lim, int ci_lim, long s_co, long s_ci,
                               long s_ky, long s_kx, long base) {
  PackDesc d{};
  d.w = w; d.wp = wp; d.s_co = s_co; d.s_ci = s_ci; d.s_ky = s_ky; d.s_kx = s_kx; d.base = base;
  d.K = KH * KW * cin; d.Kp = (d.K + 31) / 32 * 32; d.n_elems = (long)Cout * d.Kp;
  d.Cout = Cout; d.cin = cin; d.KW = KW; d.co_lim = co_lim < Cout ? co_lim : Cout; d.ci_lim = ci_lim < cin ? ci_lim : cin;
  pack_tile_shape(Cout, cin, KH * KW, s_co, s_ci, &d.co_t, &d.ci_t, &d.order);
  d.nci = (cin + d.ci_t - 1) / d.ci_t;
  return d;
}

int dpmn_conv_pack_f32(const float* w, float* wp, int Cout, int cin, int KH, int KW, int co_lim, int ci_lim, long s_co, long s_ci,
                       long s_ky, long s_kx, long base, dpmn_stream_t stream) {
  DPMN_REQUIRE(w && wp && Cout > 0 && cin > 0 && KH > 0 && KW > 0 && KH * KW <= 2304, "conv_pack: bad arguments");
  const PackDesc d = make_pack_desc(w, wp, Cout, cin, KH, KW, co_lim, ci_lim, s_co, s_ci, s_ky, s_kx, base);
  ProfScope prof(PT_CONV_PACK, as_stream(stream), 0.0, 4.0 * ((double)d.n_elems + (double)d.co_lim * d.ci_lim * KH * KW));
  hipLaunchKernelGGL(k_conv_pack, dim3((unsigned)pack_tile_blocks(d)), dim3(256), 0, as_stream(stream), d);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

int dpmn_conv_pack_tile_shape(int Cout, int cin, int taps, long s_co, long s_ci, int* shape3) {
  DPMN_REQUIRE(shape3 && Cout > 0 && cin > 0 && taps > 0 && taps <= 2304, "conv_pack_tile_shape: bad arguments");
  pack_tile_shape(Cout, cin, taps, s_co, s_ci, &shape3[0], &shape3[1], &shape3[2]);
  return DPMN_OK;
}

int dpmn_conv_pack_multi_f32(const void* descs, const int* block_prefix, int n_desc, int n_blocks, dpmn_stream_t stream) {
  DPMN_REQUIRE(descs && block_prefix && n_desc > 0 && n_blocks > 0, "conv_pack_multi: bad arguments");
  ProfScope prof(PT_CONV_PACK, as_stream(stream), 0.0, g_dpmn_prof_hint_bytes);      // (descriptors live on the device: dpmn_profile_hint_bytes)
  hipLaunchKernelGGL(k_conv_pack_multi, dim3((unsigned)n_blocks), dim3(256), 0, as_stream(stream), reinterpret_cast<const PackDesc*>(descs),
                     block_prefix, n_desc);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

int dpmn_conv2d_wgrad_unpack_f32(float* dwp, float* dw, int Cout, int cin, int KH, int KW, int co_lim, int ci_lim, long s_co,
                                 long s_ci, long s_ky, long s_kx, long base, int clear, int nslots, dpmn_stream_t stream) {
  DPMN_REQUIRE(dwp && dw && Cout > 0 && cin > 0 && KH > 0 && KW > 0 && KH * KW <= 2304, "conv2d_wgrad_unpack: bad arguments");
  const PackDesc d = make_pack_desc(dw, dwp, Cout, cin, KH, KW, co_lim, ci_lim, s_co, s_ci, s_ky, s_kx, base);
  ProfScope prof(PT_WGRAD_UNPACK, as_stream(stream), 0.0, 4.0 * ((double)(nslots > 1 ? nslots : 1) * (clear ? 2 : 1) * d.n_elems + (double)d.co_lim * d.ci_lim * KH * KW));
  hipLaunchKernelGGL(k_wgrad_unpack, dim3((unsigned)pack_tile_blocks(d)), dim3(256), 0, as_stream(stream), d, clear, nslots > 1 ? nslots : 1);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

int dpmn_conv2d_wgrad_unpack_multi_f32(const void* descs, const int* block_prefix, int n_desc, int n_blocks, dpmn_stream_t stream) {
  DPMN_REQUIRE(descs && block_prefix && n_desc > 0 && n_blocks > 0, "conv2d_wgrad_unpack_multi: bad arguments");
  ProfScope prof(PT_WGRAD_UNPACK, as_stream(stream), 0.0, g_dpmn_prof_hint_bytes);
  hipLaunchKernelGGL(k_wgrad_unpack_multi, dim3((unsigned)n_blocks), dim3(256), 0, as_stream(stream), reinterpret_cast<const PackDesc*>(descs),
                     block_prefix, n_desc);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

int dpmn_conv2d_wgrad_strided_f32(const dpmn_conv_desc* d, const float* dy, float* dw, int co_lim, int ci_lim, long s_co,
                                  long s_ci, long s_ky, long s_kx, long base, dpmn_stream_t stream) {
  return launch_wgrad(d, dy, dw, co_lim, ci_lim, s_co, s_ci, s_ky, s_kx, base, 1, 0, stream);
}

int dpmn_bn_finalize_f32(float* stats, const float* gamma, const float* beta, float count, float eps, float momentum,
                         float* scale, float* shift, float* mean, float* rstd, float* running_mean, float* running_var, int C,
                         long long* num_batches_tracked, int clear_stats, dpmn_stream_t stream) {
  DPMN_REQUIRE(stats && gamma && beta && scale && shift && mean && rstd && C > 0 && count > 1.f, "bn_finalize: bad arguments");
  hipLaunchKernelGGL(k_bn_finalize, dim3(cdiv(C, 128)), dim3(128), 0, as_stream(stream), stats, gamma, beta, count, eps, momentum,
                     scale, shift, mean, rstd, running_mean, running_var, C, num_batches_tracked, clear_stats);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

int dpmn_affine_act_bwd_f32(const float* dA, const float* r, const float* scale, const float* shift, int act, float* G,
                            int accumulate, long pixels, int C, dpmn_stream_t stream) {
  DPMN_REQUIRE(dA && r && G && pixels > 0 && C > 0, "affine_act_bwd: bad arguments");
  const long total = pixels * C;
  const bool lin = act == ACT_NONE || act == ACT_RELU || act == ACT_LEAKY02 || act == ACT_LEAKY001;      // (the only ones the old kernel handles too)
  ProfScope prof(PT_AFFINE_ACT_BWD, as_stream(stream), 0.0, 4.0 * (accumulate ? 4 : 3) * (double)total);
  if (C % 4 == 0 && lin)
    hipLaunchKernelGGL(k_affine_act_bwd_v4, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, as_stream(stream), dA, r, scale, shift,
                       act, G, accumulate, total / 4, C / 4);
  else
  hipLaunchKernelGGL(k_affine_act_bwd, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), dA, r, scale, shift,
                     act, G, accumulate, pixels, C);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

int dpmn_affine_act_bwd_stats_f32(const float* dA, const float* r, const float* scale, const float* shift, int act, float* G,
                                  int accumulate, long pixels, int C, const float* mean, const float* rstd, double* sums,
                                  dpmn_stream_t stream) {
  DPMN_REQUIRE(dA && r && G && mean && rstd && sums && pixels > 0, "affine_act_bwd_stats: bad arguments");
  DPMN_REQUIRE(C % 4 == 0 && C / 4 <= 256 && 256 % (C / 4) == 0, "affine_act_bwd_stats: C/4 must divide 256");
  DPMN_REQUIRE(act == ACT_NONE || act == ACT_RELU || act == ACT_LEAKY02 || act == ACT_LEAKY001, "affine_act_bwd_stats: piecewise-linear activations");
  static const long aab_blocks = getenv("DPMN_AAB_BLOCKS") ? atol(getenv("DPMN_AAB_BLOCKS")) : 512;      // (2048: 28.1 us per launch on average, 512: 25.0 -- same-address fp64 atomics)
  int ppb = (int)((pixels + aab_blocks - 1) / aab_blocks);       // <= 2048 blocks: one fp64 atomic pair per channel and block
  const int pl = 256 / (C / 4);
  if (ppb < 8 * pl) ppb = 8 * pl;                // two 4-pixel rounds per thread at least
  ProfScope prof(PT_AFFINE_ACT_BWD, as_stream(stream), 0.0, 4.0 * (accumulate ? 4 : 3) * (double)pixels * C);
  hipLaunchKernelGGL(k_affine_act_bwd_stats, dim3((unsigned)((pixels + ppb - 1) / ppb)), dim3(256), 0, as_stream(stream), dA, r, scale,
                     shift, act, G, accumulate, pixels, C, mean, rstd, sums, ppb);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

int dpmn_bn_bwd_apply_f32(const float* G, const float* r, const float* gamma, const float* mean, const float* rstd, const double* sums,
                          float* sums_ws, float* dr, float* dgamma, float* dbeta, long pixels, int C, dpmn_stream_t stream) {
  DPMN_REQUIRE(G && r && gamma && mean && rstd && sums && sums_ws && dr && dgamma && dbeta && pixels > 1 && C % 4 == 0, "bn_bwd_apply: bad arguments");
  (void)sums_ws;      // (the fp64 sums are converted on load)
  long total = pixels * C / 4;
  if (total < C) total = C;
  hipLaunchKernelGGL(k_bn_bwd_apply<double>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), G, r, gamma, mean, rstd,
                     sums, (float)pixels, dr, dgamma, dbeta, pixels, C);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

int dpmn_bn_bwd_f32(const float* G, const float* r, const float* gamma, const float* mean, const float* rstd, float* sums_ws,
                    float* dr, float* dgamma, float* dbeta, long pixels, int C, dpmn_stream_t stream) {
  DPMN_REQUIRE(G && r && gamma && mean && rstd && sums_ws && dr && dgamma && dbeta && pixels > 1, "bn_bwd: bad arguments");
  (void)hipMemsetAsync(sums_ws, 0, (size_t)2 * C * sizeof(float), as_stream(stream));
  DPMN_REQUIRE(C % 4 == 0 && C / 4 <= 256 && 256 % (C / 4) == 0, "bn_bwd: C/4 must divide 256");
  int ppb = (int)((pixels + 511) / 512);       // <= 512 blocks
  if (ppb < 64) ppb = 64;
  hipLaunchKernelGGL(k_bn_bwd_reduce, dim3((unsigned)((pixels + ppb - 1) / ppb)), dim3(256), 0, as_stream(stream), G, r, mean, rstd,
                     sums_ws, pixels, C, ppb);
  DPMN_CHECK_LAUNCH();
  long total = pixels * C / 4;                 // one float4 per thread; at least C threads for the dgamma / dbeta adds
  if (total < C) total = C;
  hipLaunchKernelGGL(k_bn_bwd_apply<float>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), G, r, gamma, mean, rstd,
                     sums_ws, (float)pixels, dr, dgamma, dbeta, pixels, C);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

int dpmn_se_gate_bwd_f32(const float* x, const float* dg, const float* fc1_w, const float* fc1_b, const float* fc2_w,
                         const float* fc2_b, float* dx, float* dfc1_w, float* dfc1_b, float* dfc2_w, float* dfc2_b, float* ws,
                         int B, int P, int C, int Cmid, dpmn_stream_t stream) {
  DPMN_REQUIRE(x && dg && fc1_w && fc1_b && fc2_w && fc2_b && dx && dfc1_w && dfc1_b && dfc2_w && dfc2_b && ws && B > 0 &&
                   C % 32 == 0 && Cmid % 32 == 0,
               "se_gate_bwd: bad arguments (ws: B*(5C+2Cmid) + 2*C*Cmid floats; C, Cmid multiples of 32)");
  // g = x * (1 + w), w = sigmoid(fc2(relu(fc1(mean_p x)))).  The pooled vectors of the whole batch go through the two FC
  // layers (forward and backward) as four small (B x .) GEMMs; the weight gradients are two dY^T.X GEMMs over the batch.
  hipStream_t st = as_stream(stream);
  const size_t BC = (size_t)B * C, BM = (size_t)B * Cmid;
  float* S = ws; float* Dw = S + BC; float* logit = Dw + BC; float* dlog = logit + BC; float* dS = dlog + BC;   // Dw doubles as 1+w
  float* hrelu = dS + BC; float* dh = hrelu + BM;
  float* w1t = dh + BM;                     // fc1_w^T (C, Cmid)
  float* w2t = w1t + (size_t)C * Cmid;      // fc2_w^T (Cmid, C)
  hipLaunchKernelGGL(k_se_gate_bwd_stats, dim3(B, cdiv(C, 64)), dim3(256), 0, st, x, dg, S, Dw, P, C);
  DPMN_CHECK_LAUNCH();
  int rc;
  if ((rc = dpmn_linear_f32(S, fc1_w, fc1_b, nullptr, nullptr, hrelu, B, Cmid, C, ACT_RELU, 0.f, stream)) != DPMN_OK) return rc;
  if ((rc = dpmn_linear_f32(hrelu, fc2_w, fc2_b, nullptr, nullptr, logit, B, C, Cmid, ACT_NONE, 0.f, stream)) != DPMN_OK) return rc;
  hipLaunchKernelGGL(k_se_gate_elem, dim3((unsigned)((BC + 255) / 256)), dim3(256), 0, st, logit, Dw, dlog, Dw, (long)BC);
  DPMN_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_transpose, dim3(cdiv(C, 32), cdiv(Cmid, 32)), dim3(256), 0, st, fc1_w, w1t, Cmid, C);   // (Cmid,C) -> (C,Cmid)
  hipLaunchKernelGGL(k_transpose, dim3(cdiv(Cmid, 32), cdiv(C, 32)), dim3(256), 0, st, fc2_w, w2t, C, Cmid);   // (C,Cmid) -> (Cmid,C)
  DPMN_CHECK_LAUNCH();
  if ((rc = dpmn_linear_f32(dlog, w2t, nullptr, nullptr, nullptr, dh, B, Cmid, C, ACT_NONE, 0.f, stream)) != DPMN_OK) return rc;
  hipLaunchKernelGGL(k_relu_mask, dim3((unsigned)((BM + 255) / 256)), dim3(256), 0, st, hrelu, dh, (long)BM);
  DPMN_CHECK_LAUNCH();
  if ((rc = dpmn_linear_f32(dh, w1t, nullptr, nullptr, nullptr, dS, B, C, Cmid, ACT_NONE, 0.f, stream)) != DPMN_OK) return rc;
  const long total4 = (long)B * P * C / 4;
  hipLaunchKernelGGL(k_se_gate_bwd_dx, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, st, dg, Dw, dS, dx, P, C, 1.0f / (float)P, total4);
  DPMN_CHECK_LAUNCH();
  if ((rc = dpmn_gemm_tn_f32(dlog, hrelu, dfc2_w, dfc2_b, B, C, Cmid, nullptr, 0, stream)) != DPMN_OK) return rc;
  return dpmn_gemm_tn_f32(dh, S, dfc1_w, dfc1_b, B, Cmid, C, nullptr, 0, stream);
}

int dpmn_affine_act_fwd_f32(const float* r, const float* scale, const float* shift, int act, float* y, long pixels, int C,
                            dpmn_stream_t stream) {
  DPMN_REQUIRE(r && y && pixels > 0 && C > 0, "affine_act_fwd: bad arguments");
  const long total = pixels * C;
  hipLaunchKernelGGL(k_affine_act_fwd, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), r, scale, shift, act, y, pixels, C);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

int dpmn_l1_loss_fwd_f32(const float* a, const float* b, float inv_count, float* loss, float* part_ws, long n, dpmn_stream_t stream) {
  DPMN_REQUIRE(a && b && loss && part_ws && n > 0, "l1_loss_fwd: bad arguments (part_ws: ceil(n/256) floats)");
  const int nb = (int)((n + 255) / 256);
  hipLaunchKernelGGL(k_l1_partial, dim3(nb), dim3(256), 0, as_stream(stream), a, b, part_ws, n);
  DPMN_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_sum_final, dim3(1), dim3(64), 0, as_stream(stream), part_ws, nb, inv_count, loss);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

int dpmn_l1_loss_bwd_f32(const float* a, const float* b, const float* grad_scale, float inv_count, const float* extra_a, float* da,
                         float* db, long n, dpmn_stream_t stream) {
  DPMN_REQUIRE(a && b && grad_scale && da && db && n > 0, "l1_loss_bwd: bad arguments");
  hipLaunchKernelGGL(k_l1_bwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), a, b, grad_scale, inv_count, extra_a, da, db, n);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

int dpmn_sumsq_f32(const float* x, float* out, float* part_ws, long n, dpmn_stream_t stream) {
  DPMN_REQUIRE(x && out && part_ws && n > 0, "sumsq: bad arguments (part_ws: 1024 floats)");
  const int nb = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
  hipLaunchKernelGGL(k_sumsq_partial, dim3(nb), dim3(256), 0, as_stream(stream), x, part_ws, n);
  DPMN_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_sum_final, dim3(1), dim3(64), 0, as_stream(stream), part_ws, nb, 1.0f, out);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

int dpmn_adam_clip_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, const float* grad_normsq, float max_norm,
                       float lr, float beta1, float beta2, float eps, int step, const float* step_dev, long n,
                       dpmn_stream_t stream) {
  DPMN_REQUIRE(param && grad && exp_avg && exp_avg_sq && n > 0 && (step >= 1 || step_dev), "adam_clip: bad arguments");
  if (step < 1) step = 1;
  DPMN_REQUIRE(max_norm <= 0.f || grad_normsq, "adam_clip: grad_normsq required when clipping");
  const float bc1 = 1.f - powf(beta1, (float)step);
  const float bc2s = sqrtf(1.f - powf(beta2, (float)step));
  hipLaunchKernelGGL(k_adam_clip, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), param, grad, exp_avg, exp_avg_sq,
                     grad_normsq, max_norm, lr, beta1, beta2, eps, bc1, bc2s, n, step_dev);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

}  // extern "C"
