// "f32 via bf16x3" weight gradient of the implicit-GEMM convs (dpmn_set_compute_dtype(2); common.h x3_split2t):
//   dW[co][k] = sum_pixels dY[pix][co] * pro(in)[pix @ tap(k)][ci(k)]       (autograd of cmm.py:38-77, the conv stacks of tsrn.py / tatt.py)
// The 128 (co) x 128 (k) tile of k_conv_wgrad's power-of-two fast path (conv_bwd.hip) with the products on v_mfma_f32_16x16x32_bf16.
// The contraction index is the PIXEL and both operands are channel-contiguous in memory (NHWC), while the bf16 MFMA wants 8
// consecutive contraction indices per lane: a 32-pixel chunk is staged as pixel PAIRS -- dword (p, c) = (bf16 v[2p][c], bf16 v[2p+1][c]),
// three planes per operand -- by threads that hold the same channel quad of two adjacent pixels (one ds_write_b128 per plane and
// pair), and a lane reads pairs 4 kq .. 4 kq + 3 of its channel quad: one ds_read_b128 feeds FOUR tiles (co = 4 lr + ti, the fp32
// kernel's interleave), the register "transpose" (pair x tile -> tile x pair) is free.  dY rows: 128 dwords (4 pair rows = 0 mod 64
// banks: the 16 lanes of a ds_read_b128 group hold 16 different channel quads); X rows: 136 dwords (ds_read_b64, 4 pair rows = 32 mod 64).
// One LDS stage (50.7 KB at 128 x 128, 61 KB at 64 x 256), two barriers per chunk, 96 MFMAs per wave and chunk; epilogues (exclusive
// slots / atomics) as k_conv_wgrad.
#include <cstdlib>
#include "conv_wgrad.h"

namespace {
using dpmn_conv::WgArgs;

// <BN_, BKT>: 128 x 128 (NI = 8 co tiles, NJ = 2 k tiles per wave), 64 x 256 and 64 x 128 (the layers with <= 64 output channels)
template <int BN_, int BKT>
__global__ __launch_bounds__(256, 2) void k_conv_wgrad_x3(WgArgs a) {
  constexpr int BMc = 32, NP = BMc / 2, NI = BN_ / 16, NJ = BKT / 64;
  constexpr int LDY = BN_, LDX = NJ == 2 ? BKT + 8 : BKT;      // ds_read_b128 rows: 4 pair rows = 0 mod 64 banks; ds_read_b64 rows: = 32 mod 64
  constexpr int YC4 = BN_ / 4, XC4 = BKT / 4;                  // float4 columns; a thread holds one column of a pixel PAIR per pass
  constexpr int YPR = 256 / YC4, XPR = 256 / XC4, YPASS = NP / YPR, XPASS = NP / XPR;      // pair rows per pass, passes
  static_assert(YPASS >= 1 && XPASS >= 1 && (NJ == 2 || NJ == 4) && NI % 4 == 0, "tile shapes of k_conv_wgrad");
  constexpr int YPL = NP * LDY, XPL = NP * LDX;                 // one plane (dwords)
  __shared__ __attribute__((aligned(16))) unsigned Yp[3 * YPL];
  __shared__ __attribute__((aligned(16))) unsigned Xp[3 * XPL];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int vid = blockIdx.x;
  const int total = a.gx * a.gy * a.gz;
  if ((total & 7) == 0) vid = (vid & 7) * (total >> 3) + (vid >> 3);      // the tiles of one pixel range on one XCD (as k_conv_wgrad)
  const int n_blk = (vid % a.gx) * BN_, k_blk = ((vid / a.gx) % a.gy) * BKT;
  const int bz = vid / (a.gx * a.gy);
  const int M = a.B * a.Hp * a.Wp;
  const int m_lo = bz * a.pix_per_block;
  const int m_hi = min(M, m_lo + a.pix_per_block);
  const int c01 = a.cseg[0] + a.cseg[1];
  // loaders: thread (q, c) holds channel quad c of the pixel pairs q, q + PR, ... (pixels 2 q, 2 q + 1 of each)
  const int yq = tid / YC4, yc4 = (tid % YC4) * 4;
  const int xq = tid / XC4, xc4 = (tid % XC4) * 4;
  const int kcol = k_blk + xc4;
  const int tap = kcol / a.cin, cch = kcol - tap * a.cin;
  const int ky = tap / a.KW, kx = tap - ky * a.KW;
  int seg = 0, cl = cch;
  if (cch >= c01) { seg = 2; cl = cch - c01; }
  else if (cch >= a.cseg[0]) { seg = 1; cl = cch - a.cseg[0]; }
  const float* src = seg == 0 ? a.in[0] : (seg == 1 ? a.in[1] : a.in[2]);
  const int cs = seg == 0 ? a.cseg[0] : (seg == 1 ? a.cseg[1] : a.cseg[2]);
  const float* scp = seg == 0 ? a.in_scale[0] : (seg == 1 ? a.in_scale[1] : a.in_scale[2]);
  const float* shp = seg == 0 ? a.in_shift[0] : (seg == 1 ? a.in_shift[1] : a.in_shift[2]);
  const bool kvalid = kcol < a.K;
  float4 s4 = make_float4(1.f, 1.f, 1.f, 1.f), h4 = make_float4(0.f, 0.f, 0.f, 0.f);
  const bool aff = scp != nullptr;
  if (aff && kvalid) { s4 = *reinterpret_cast<const float4*>(scp + cl); h4 = *reinterpret_cast<const float4*>(shp + cl); }
  const int iy_off = ky * a.dil_y - a.pad_y, ix_off = kx * a.dil_x - a.pad_x;
  const int yn = n_blk + yc4;
  const float* src_v = kvalid ? src : a.in[0];
  const int cs_v = kvalid ? cs : a.cseg[0], cl_v = kvalid ? cl : 0;
  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy), 0, a.B * a.Hout * a.Wout * a.Cout * 4, 0x00020000);
  const int wmask = a.Wp - 1, hmask = a.Hp - 1;
  float4 yr[2 * YPASS], xr[2 * XPASS];
  unsigned xmask = 0;
  auto gload = [&](int m0) {
    xmask = 0;
#pragma unroll
    for (int p = 0; p < 2 * YPASS; ++p) {
      const int m = m0 + 2 * (yq + (p >> 1) * YPR) + (p & 1);
      const int px = m & wmask, py = (m >> a.lgW) & hmask, b = m >> a.lgHW;
      const bool oky = m < m_hi && yn < a.Cout;
      const int pixy = __mul24(__mul24(b, a.Hout) + py * a.ostep + a.ooy, a.Wout) + px * a.ostep + a.oox;
      const unsigned off = oky ? (unsigned)(__mul24(pixy, a.Cout) + yn) * 4u : 0x80000000u;      // beyond num_records: reads 0
      yr[p] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(yrs, (int)off, 0, 0));
    }
#pragma unroll
    for (int p = 0; p < 2 * XPASS; ++p) {
      const int m = m0 + 2 * (xq + (p >> 1) * XPR) + (p & 1);
      const int px = m & wmask, py = (m >> a.lgW) & hmask, b = m >> a.lgHW;
      const int iy = py * a.stride + iy_off, ix = px * a.stride + ix_off;
      const bool okx = kvalid && m < m_hi && (unsigned)iy < (unsigned)a.Hin && (unsigned)ix < (unsigned)a.Win;
      const int pixx = __mul24(__mul24(b, a.Hin) + iy, a.Win) + ix;
      xr[p] = *reinterpret_cast<const float4*>(src_v + (okx ? __mul24(pixx, cs_v) + cl_v : 0));      // clamped, masked in sstore
      xmask |= (okx ? 1u : 0u) << p;
    }
  };
  auto put_pair = [&](unsigned* base, int plane, int ld, int pair, int c4, const float4& lo, const float4& hi) {
    uint4 h, m, l;
    x3_split2t(lo.x, hi.x, h.x, m.x, l.x);
    x3_split2t(lo.y, hi.y, h.y, m.y, l.y);
    x3_split2t(lo.z, hi.z, h.z, m.z, l.z);
    x3_split2t(lo.w, hi.w, h.w, m.w, l.w);
    unsigned* d_ = base + pair * ld + c4;
    *reinterpret_cast<uint4*>(d_) = h;
    *reinterpret_cast<uint4*>(d_ + plane) = m;
    *reinterpret_cast<uint4*>(d_ + 2 * plane) = l;
  };
  auto sstore = [&]() {
#pragma unroll
    for (int p = 0; p < YPASS; ++p) put_pair(Yp, YPL, LDY, yq + p * YPR, yc4, yr[2 * p], yr[2 * p + 1]);
    const float sl = a.pro_act == ACT_LEAKY02 ? 0.2f : 0.0f;
#pragma unroll
    for (int p = 0; p < 2 * XPASS; ++p) {      // the transform of k_conv_wgrad's fast path, same expressions
      float4 xv = xr[p];
      if (aff) { xv.x = xv.x * s4.x + h4.x; xv.y = xv.y * s4.y + h4.y; xv.z = xv.z * s4.z + h4.z; xv.w = xv.w * s4.w + h4.w; }
      if (a.pro_act != ACT_NONE) {
        xv.x = fmaxf(xv.x, sl * xv.x); xv.y = fmaxf(xv.y, sl * xv.y); xv.z = fmaxf(xv.z, sl * xv.z); xv.w = fmaxf(xv.w, sl * xv.w);
      }
      if (!((xmask >> p) & 1u)) xv = make_float4(0.f, 0.f, 0.f, 0.f);
      xr[p] = xv;
    }
#pragma unroll
    for (int p = 0; p < XPASS; ++p) put_pair(Xp, XPL, LDX, xq + p * XPR, xc4, xr[2 * p], xr[2 * p + 1]);
  };
  const int lr = lane & 15, kq = lane >> 4;
  f32x4 acc[NI][NJ];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (m_lo < m_hi) { gload(m_lo); sstore(); }
  __syncthreads();
  typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
  const unsigned* ya = Yp + (4 * kq) * LDY + lr * 4;
  const unsigned* xa = Xp + (4 * kq) * LDX + wave * (NJ * 16) + lr * NJ;
  for (int m0 = m_lo; m0 < m_hi; m0 += BMc) {
    const bool more = m0 + BMc < m_hi;
    if (more) gload(m0 + BMc);
    __builtin_amdgcn_sched_barrier(0);
    // B operand (input columns): all three planes
    bf16x8 bf[3][NJ];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
      if constexpr (NJ == 2) {
        uint2 t_[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) t_[t] = *reinterpret_cast<const uint2*>(xa + pl * XPL + t * LDX);
        bf[pl][0] = __builtin_bit_cast(bf16x8, (u32x4_){t_[0].x, t_[1].x, t_[2].x, t_[3].x});
        bf[pl][1] = __builtin_bit_cast(bf16x8, (u32x4_){t_[0].y, t_[1].y, t_[2].y, t_[3].y});
      } else {
        uint4 t_[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) t_[t] = *reinterpret_cast<const uint4*>(xa + pl * XPL + t * LDX);
        bf[pl][0] = __builtin_bit_cast(bf16x8, (u32x4_){t_[0].x, t_[1].x, t_[2].x, t_[3].x});
        bf[pl][1] = __builtin_bit_cast(bf16x8, (u32x4_){t_[0].y, t_[1].y, t_[2].y, t_[3].y});
        bf[pl][2] = __builtin_bit_cast(bf16x8, (u32x4_){t_[0].z, t_[1].z, t_[2].z, t_[3].z});
        bf[pl][3] = __builtin_bit_cast(bf16x8, (u32x4_){t_[0].w, t_[1].w, t_[2].w, t_[3].w});
      }
    }
    // A operand (dY rows): one plane at a time, four tiles per ds_read_b128 column
    auto load_a = [&](int pl, bf16x8 (&af)[NI]) {
#pragma unroll
      for (int h = 0; h < NI / 4; ++h) {
        uint4 t_[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) t_[t] = *reinterpret_cast<const uint4*>(ya + pl * YPL + t * LDY + h * 64);
        af[h * 4 + 0] = __builtin_bit_cast(bf16x8, (u32x4_){t_[0].x, t_[1].x, t_[2].x, t_[3].x});
        af[h * 4 + 1] = __builtin_bit_cast(bf16x8, (u32x4_){t_[0].y, t_[1].y, t_[2].y, t_[3].y});
        af[h * 4 + 2] = __builtin_bit_cast(bf16x8, (u32x4_){t_[0].z, t_[1].z, t_[2].z, t_[3].z});
        af[h * 4 + 3] = __builtin_bit_cast(bf16x8, (u32x4_){t_[0].w, t_[1].w, t_[2].w, t_[3].w});
      }
    };
#define WG_X3_TERM(AF, PB)                                                  \
    _Pragma("unroll") for (int i = 0; i < NI; ++i)                          \
      _Pragma("unroll") for (int j = 0; j < NJ; ++j) acc[i][j] = mfma16_bf16(AF[i], bf[PB][j], acc[i][j]);
    bf16x8 a0[NI], a1[NI];
    load_a(0, a0);
    load_a(1, a1);
    WG_X3_TERM(a0, 2) WG_X3_TERM(a0, 1) WG_X3_TERM(a0, 0)
    load_a(2, a0);
    WG_X3_TERM(a1, 1) WG_X3_TERM(a1, 0)
    WG_X3_TERM(a0, 0)
#undef WG_X3_TERM
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    if (more) sstore();
    __syncthreads();
  }
  if (a.excl) {
    float* slotp = a.dw + (long)bz * a.slot_stride;
    const int k0 = k_blk + wave * (NJ * 16) + lr * NJ;
    if (k0 < (int)a.s_co) {
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ii = kq * 4 + r;
          const int n = n_blk + (i >> 2) * 64 + ii * 4 + (i & 3);
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
        const int n = n_blk + (i >> 2) * 64 + ii * 4 + (i & 3);
        if (n < a.co_lim) atomicAdd(dst + (long)n * a.s_co, acc[i][j][r]);
      }
  }
}

}  // namespace

namespace dpmn_conv {
bool x3_wgrad_ok(const WgArgs& a, int bn, int bk) {      // (the caller established the power-of-two fast path)
  static const int t64 = getenv("DPMN_X3_WGRAD64") ? atoi(getenv("DPMN_X3_WGRAD64")) : 1;
  return a.pix_per_block % 32 == 0 && ((bn == 128 && bk == 128) || (t64 && bn == 64 && (bk == 128 || bk == 256)));
}
int x3_launch_wgrad(const WgArgs& a, int bn, int bk, dim3 grid, hipStream_t st) {
  if (bn == 128) hipLaunchKernelGGL((k_conv_wgrad_x3<128, 128>), grid, dim3(256), 0, st, a);
  else if (bk == 256) hipLaunchKernelGGL((k_conv_wgrad_x3<64, 256>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((k_conv_wgrad_x3<64, 128>), grid, dim3(256), 0, st, a);
  return 0;
}
}  // namespace dpmn_conv
