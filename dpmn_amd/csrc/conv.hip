// NHWC implicit-GEMM convolution on fp32 MFMA (v_mfma_f32_16x16x4_f32) for the CMM U-Net
// (cmm.py:38-77, 120-161) and the TSRN/TATT/TBSRN conv stacks (tsrn.py:26-40,83-110; tatt.py:596-636).
//
//   out[b, oy, ox, co] = epi( bias[co] + sum_{tap, ci} W[co][tap][ci] * pro(in)[b, iy(tap), ix(tap), ci] )
//
//   M = B*Hp*Wp output pixels of one "phase" grid, N = Cout, K = KH*KW*Cin (Cin contiguous: NHWC).
//   * up to 3 channel-concatenated input segments (decoder skip concats, cmm.py:150-158) are read in
//     place -- the concat is never materialised;
//   * prologue on load: per-channel affine (train-mode BatchNorm of the producer) then activation
//     (LeakyReLU 0.2 / ReLU that opens every Encode/DecodeBlock) -- zero padding stays zero;
//   * stride-2 ConvTranspose2d(4,2,1) runs as 4 output phases, each a 2x2 conv (dil = -1,
//     pad = -phase) over pre-packed per-phase weights; ConvTranspose2d(3,1,1) is a flipped conv;
//   * epilogue: bias, activation, residual add, optional per-channel sum / sum-of-squares for
//     train-mode BatchNorm statistics, NHWC or NCHW store, optional PixelShuffle(2) store.
// MFMA roles as in gemm.hip: "A" = weight rows (co), "B" = pixels, so a lane owns 4 consecutive co.
#include "conv_body.h"

namespace {

template <int BM, int BN, int WM, int WN, bool UNI = false, int BKT = 32, bool SIMPLE = false, bool AFF = false, bool BF = false, bool M32 = false>
__global__ __launch_bounds__(256) void k_conv_igemm(ConvArgs a) {
  conv_igemm_body<BM, BN, WM, WN, UNI, BKT, SIMPLE, AFF, BF, M32>(a);
}
// split-K with the reduction inside the launch (XRED above)
template <int BM, int BN, int WM, int WN, bool AFF>
__global__ __launch_bounds__(256, 2) void k_conv_igemm_xr(ConvArgs a) {      // two resident blocks per CU (<= 256 registers), as the fixed-split kernel
  conv_igemm_body<BM, BN, WM, WN, true, 32, true, AFF, false, false, true>(a);
}
// the SIMPLE path with 16-deep chunks (36.9 KB of LDS) AND a register budget for three waves per SIMD (<= 168 registers): three
// resident blocks per CU instead of two -- the 16-deep switch alone (DPMN_CONV_BK16) stayed at two because of its 200 registers
template <int BM, int BN, int WM, int WN, bool AFF>
__global__ __launch_bounds__(256, 3) void k_conv_igemm_o3(ConvArgs a) {
  conv_igemm_body<BM, BN, WM, WN, true, 16, true, AFF, false, false>(a);
}

// sum the split-K partials and run the epilogue.  Block = 64 channel-quads x 4 row lanes, 64 rows per block, so the
// BatchNorm statistics are reduced over 64 rows in registers / LDS before one (slotted) atomic per channel.
__global__ __launch_bounds__(256) void k_conv_splitk_reduce(ConvArgs a, int rows, int cq_lanes) {
  // cq_lanes (64 / 32 / 16: the channel quads of a row, capped at 64) x 256 / cq_lanes row lanes: with the fixed 64 x 4 mapping half
  // of every block sat idle on the 128-channel layers (32 quads), three quarters on 64 channels
  __shared__ float red[256][8];
  const PhaseSel ph = conv_select_phase(a, blockIdx.z * a.ksplit);
  const int M = a.B * a.Hp * a.Wp;
  const int n4 = a.npad / 4;
  const int cl = threadIdx.x & (cq_lanes - 1), rl = threadIdx.x / cq_lanes, RL = 256 / cq_lanes;
  const int cq = blockIdx.x * cq_lanes + cl;
  const int n = cq * 4;
  const int m_lo = blockIdx.y * rows;
  float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};
  const size_t zs = (size_t)M * a.npad;
  if (cq < n4) {
    const int m_hi = min(M, m_lo + rows);
    auto sum_row = [&](int m, float (&v)[4]) {
      v[0] = v[1] = v[2] = v[3] = 0.f;
      const float* pp = ph.partial + (size_t)m * a.npad + n;
      int z = 0;
      for (; z + 4 <= a.ksplit; z += 4) {     // 4 independent loads in flight
        const float4 p0 = *reinterpret_cast<const float4*>(pp + (size_t)z * zs);
        const float4 p1 = *reinterpret_cast<const float4*>(pp + (size_t)(z + 1) * zs);
        const float4 p2 = *reinterpret_cast<const float4*>(pp + (size_t)(z + 2) * zs);
        const float4 p3 = *reinterpret_cast<const float4*>(pp + (size_t)(z + 3) * zs);
        v[0] += (p0.x + p1.x) + (p2.x + p3.x); v[1] += (p0.y + p1.y) + (p2.y + p3.y);
        v[2] += (p0.z + p1.z) + (p2.z + p3.z); v[3] += (p0.w + p1.w) + (p2.w + p3.w);
      }
      for (; z < a.ksplit; ++z) {
        const float4 p = *reinterpret_cast<const float4*>(pp + (size_t)z * zs);
        v[0] += p.x; v[1] += p.y; v[2] += p.z; v[3] += p.w;
      }
    };
    // two rows per iteration: a layer split in two has only two loads per row -- the second row's are in flight under the first's
    // epilogue (same per-row arithmetic, same order of the statistics sums: row m, then row m + RL)
    int m = m_lo + rl;
    for (; m + RL < m_hi; m += 2 * RL) {
      float v0[4], v1[4];
      sum_row(m, v0);
      sum_row(m + RL, v1);
      conv_store(a, m, n, v0, ssum, ssq, ph.ooy, ph.oox);
      conv_store(a, m + RL, n, v1, ssum, ssq, ph.ooy, ph.oox);
    }
    if (m < m_hi) {
      float v0[4];
      sum_row(m, v0);
      conv_store(a, m, n, v0, ssum, ssq, ph.ooy, ph.oox);
    }
  }
  if (a.stats) {
#pragma unroll
    for (int r = 0; r < 4; ++r) { red[threadIdx.x][r] = ssum[r]; red[threadIdx.x][4 + r] = ssq[r]; }
    __syncthreads();
    if (rl == 0 && cq < n4) {
      double* st = reinterpret_cast<double*>(a.stats) + (size_t)(blockIdx.y % STAT_SLOTS) * 2 * a.Cout;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (n + r < a.Cout) {
          float s_ = 0.f, q_ = 0.f;
          for (int l = 0; l < RL; ++l) { s_ += red[l * cq_lanes + cl][r]; q_ += red[l * cq_lanes + cl][4 + r]; }      // fixed order
          atomicAdd(st + n + r, (double)s_);
          atomicAdd(st + a.Cout + n + r, (double)q_);
        }
    }
  }
}

// ---------------------------------------------------------------------------------- stream-K implicit GEMM
// The split-K layers of the CMM (deep levels: 8 ... 384 output tiles against 256 CUs, K = 512 ... 13824) as ONE persistent launch:
// the (tile, k chunk) steps of the whole layer -- T tiles x nk 32-deep chunks -- are cut into G equal contiguous ranges, one per
// workgroup (G = the resident capacity, 2 per CU), so every CU does the same number of MFMAs whatever T is (the fixed-split
// grid gave 384 whole tiles to 256 CUs, or 768 short blocks in 1.5 rounds of 512 slots).  A range that covers a tile's whole K
// runs the epilogue directly.  Otherwise the accumulators go to slot (block + tile) of the workspace in their register
// layout (16-byte lane-consecutive stores), the block bumps the tile's arrival counter, and the block that arrives LAST reads all
// contributions back IN BLOCK ORDER (its own included: the sum is the same whoever arrives last -- bitwise reproducible) and runs
// the epilogue.  No reduce launch; partial tiles <= G + T - 1 per launch instead of S T.  Nothing waits on another workgroup.
// Logical block ids are XCD-contiguous (workgroups are dealt to the 8 XCDs round-robin): one XCD walks consecutive tiles, which
// share a weight column tile (order 0: row tile fastest) or an input row tile (order 1), as the wlocal mapping above does.
struct SkArgs {
  int tiles_m, tiles_n, nk, order;
  long total;                 // T * nk chunk steps
  float* partial;             // (G + T) slots of BM * BN floats
  unsigned* cnt;              // T arrival counters: zero on entry, zero again on exit
};

template <int BM, int BN, int WM, int WN, bool AFF>
__global__ __launch_bounds__(256) void k_conv_igemm_sk(ConvArgs a, SkArgs sk) {
  constexpr int MT = BM / WM / 16, NT = BN / WN / 16;
  constexpr int RPP = 32;                                 // 8 threads per 32-float tile row, 32 rows per pass
  constexpr int APASS = BM / RPP, BPASS = BN / RPP;
  constexpr int BK = 32, LDK = BK + PAD;
  static_assert(WM * WN == 4 && BM % RPP == 0 && BN % RPP == 0, "4 waves; whole staging passes");
  __shared__ __attribute__((aligned(16))) float Xs[2][BM * LDK];
  __shared__ __attribute__((aligned(16))) float Ws[2][BN * LDK];
  __shared__ int s_fix[2];
  typedef int i32x4_ __attribute__((ext_vector_type(4)));

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int M = a.B * a.Hp * a.Wp;
  const int G = (int)gridDim.x, pb_ = (int)blockIdx.x;
  const int L = (G & 7) == 0 ? (pb_ & 7) * (G >> 3) + (pb_ >> 3) : pb_;
  long c = sk.total * L / G;
  const long c_end = sk.total * (L + 1) / G;
  const int lrow = tid >> 3, lcol = (tid & 7) * 4;
  const int HWp = a.Hp * a.Wp;
  const float inv_hw = 1.0f / (float)HWp, inv_w = 1.0f / (float)a.Wp;
  const int c01 = a.cseg[0] + a.cseg[1];
  const int ktaps = a.KH * a.KW;
  const int minoff = (a.dil_y < 0 ? (a.KH - 1) * a.dil_y : 0) * a.Win + (a.dil_x < 0 ? (a.KW - 1) * a.dil_x : 0);
  const int wm = wave % WM, wn = wave / WM;
  const int lr = lane & 15, kq = lane >> 4;
  const int nph = a.nphase > 1 ? a.nphase : 1;

  while (c < c_end) {
    const int t = (int)(c / sk.nk);
    const int kt0 = (int)(c - (long)t * sk.nk);
    const int kt1 = min(sk.nk, kt0 + (int)(c_end - c));
    c += kt1 - kt0;
    int bx, by, phs;
    if (sk.order == 0) { bx = t % sk.tiles_m; const int r = t / sk.tiles_m; by = r % sk.tiles_n; phs = r / sk.tiles_n; }
    else { by = t % sk.tiles_n; const int r = t / sk.tiles_n; phs = r % nph; bx = r / nph; }
    const int m_blk = bx * BM, n_blk = by * BN;
    int pad_y = a.pad_y, pad_x = a.pad_x, ooy = a.ooy, oox = a.oox;
    const float* wbase = a.w + (conv_group_of(a, m_blk) ? a.wgs : 0L);
    if (a.nphase > 1) { pad_y = -(phs >> 1); pad_x = -(phs & 1); ooy = phs >> 1; oox = phs & 1; wbase += (size_t)phs * a.wps; }

    // ---- per-tile state of the SIMPLE load path (see k_conv_igemm)
    int pix[APASS];
    unsigned nok[APASS];
#pragma unroll
    for (int p = 0; p < APASS; ++p) {
      const int m = m_blk + lrow + p * RPP;
      int py = -(1 << 20), px = -(1 << 20);
      pix[p] = 0;
      if (m < M) {
        int b = (int)((float)m * inv_hw), r = m - b * HWp;
        if (r < 0) { --b; r += HWp; }
        if (r >= HWp) { ++b; r -= HWp; }
        int y = (int)((float)r * inv_w), x = r - y * a.Wp;
        if (x < 0) { --y; x += a.Wp; }
        if (x >= a.Wp) { ++y; x -= a.Wp; }
        py = y * a.stride - pad_y;
        px = x * a.stride - pad_x;
        pix[p] = (b * a.Hin + py) * a.Win + px;
      }
      unsigned colm = 0, okb = 0;
      for (int kx = 0; kx < a.KW; ++kx) colm |= ((unsigned)(px + kx * a.dil_x) < (unsigned)a.Win ? 1u : 0u) << kx;
      for (int ky = 0; ky < a.KH; ++ky)
        if ((unsigned)(py + ky * a.dil_y) < (unsigned)a.Hin) okb |= colm << (ky * a.KW);
      nok[p] = ~okb;
    }
    const int padoff = pad_y * a.Win + pad_x;
    const int baseshift = padoff - minoff;
    int voff[APASS], wofs[BPASS];
    unsigned inv[APASS];
#pragma unroll
    for (int p = 0; p < BPASS; ++p) wofs[p] = ((n_blk + lrow + p * RPP) * a.Kp + lcol) * 4;
    const __amdgpu_buffer_rsrc_t s_wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wbase), 0, a.Cout * a.Kp * 4, 0x00020000);
    __amdgpu_buffer_rsrc_t s_xrs, s_scrs, s_shrs;
    int s_seg = -1, s_segstart = 0;
    int u_tap = 0, u_c0 = 0, u_ky = 0, u_kx = 0, u_k0 = -2;
    float4 xr[APASS], wr[BPASS];
    float4 s4r, h4r;
    auto gload = [&](int kt_) {
      if (kt_ == u_k0 + 1) {
        ++u_tap;
        if (++u_kx == a.KW) { u_kx = 0; ++u_ky; }
        if (u_tap == ktaps) { u_tap = 0; u_ky = 0; u_c0 += BK; }
      } else if (kt_ != u_k0) {
        const int cch = kt_ / ktaps;
        u_tap = kt_ - cch * ktaps; u_c0 = cch * BK;
        u_ky = u_tap / a.KW; u_kx = u_tap - u_ky * a.KW;
      }
      u_k0 = kt_;
      const int k0 = u_tap * a.cin + u_c0;
      const int seg = u_c0 >= c01 ? 2 : (u_c0 >= a.cseg[0] ? 1 : 0);
      const int cs = seg == 2 ? a.cseg[2] : (seg == 1 ? a.cseg[1] : a.cseg[0]);
      if (seg != s_seg) {
        s_seg = seg;
        s_segstart = seg == 2 ? c01 : (seg == 1 ? a.cseg[0] : 0);
        const float* src = seg == 2 ? a.in[2] : (seg == 1 ? a.in[1] : a.in[0]);
        s_xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src) - (ptrdiff_t)baseshift * cs, 0,
                                                  (a.B * a.Hin * a.Win + baseshift) * cs * 4, 0x00020000);
#pragma unroll
        for (int p = 0; p < APASS; ++p) voff[p] = (__mul24(pix[p] + padoff, cs) + lcol) * 4;
        if (AFF) {
          const float* sc = seg == 2 ? a.in_scale[2] : (seg == 1 ? a.in_scale[1] : a.in_scale[0]);
          const float* sf = seg == 2 ? a.in_shift[2] : (seg == 1 ? a.in_shift[1] : a.in_shift[0]);
          s_scrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sc), 0, cs * 4, 0x00020000);
          s_shrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sf), 0, cs * 4, 0x00020000);
        }
      }
      const int soff = __builtin_amdgcn_readfirstlane(((u_ky * a.dil_y * a.Win + u_kx * a.dil_x - minoff) * cs + u_c0 - s_segstart) * 4);
      const int sh = 31 - min(u_tap, 31);
#pragma unroll
      for (int p = 0; p < APASS; ++p) {
        const unsigned oob = (nok[p] << sh) & 0x80000000u;
        if (AFF) inv[p] = oob;
        xr[p] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(s_xrs, (int)(oob | (unsigned)voff[p]), soff, 0));
      }
      if (AFF) {
        const int coff = __builtin_amdgcn_readfirstlane((u_c0 - s_segstart) * 4);
        s4r = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(s_scrs, lcol * 4, coff, 0));
        h4r = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(s_shrs, lcol * 4, coff, 0));
      }
#pragma unroll
      for (int p = 0; p < BPASS; ++p)
        wr[p] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(s_wrs, wofs[p], k0 * 4, 0));
    };
    auto sstore = [&](int buf) {
      if (AFF) {
#pragma unroll
        for (int p = 0; p < APASS; ++p) {
          xr[p].x = xr[p].x * s4r.x + h4r.x; xr[p].y = xr[p].y * s4r.y + h4r.y;
          xr[p].z = xr[p].z * s4r.z + h4r.z; xr[p].w = xr[p].w * s4r.w + h4r.w;
        }
      }
      if (a.pro_act != ACT_NONE) {
        const float sl = a.pro_act == ACT_LEAKY02 ? 0.2f : 0.0f;
#pragma unroll
        for (int p = 0; p < APASS; ++p) {
          xr[p].x = vmax_raw(xr[p].x, sl * xr[p].x); xr[p].y = vmax_raw(xr[p].y, sl * xr[p].y);
          xr[p].z = vmax_raw(xr[p].z, sl * xr[p].z); xr[p].w = vmax_raw(xr[p].w, sl * xr[p].w);
        }
      }
      if (AFF) {
#pragma unroll
        for (int p = 0; p < APASS; ++p)
          if (inv[p]) xr[p] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int p = 0; p < APASS; ++p) *reinterpret_cast<float4*>(&Xs[buf][(lrow + p * RPP) * LDK + lcol]) = xr[p];
#pragma unroll
      for (int p = 0; p < BPASS; ++p) *reinterpret_cast<float4*>(&Ws[buf][(lrow + p * RPP) * LDK + lcol]) = wr[p];
    };

    f32x4 acc[NT][MT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int j = 0; j < MT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    gload(kt0);
    sstore(0);
    __syncthreads();
    for (int kt = kt0; kt < kt1; ++kt) {
      const int buf = (kt - kt0) & 1;
      gload(min(kt + 1, kt1 - 1));
      __builtin_amdgcn_sched_barrier(0);
      const float* xa = &Xs[buf][0] + (wm * (MT * 16) + lr) * LDK + kq * 4;
      const float* wa = &Ws[buf][0] + (wn * (NT * 16) + lr) * LDK + kq * 4;
#pragma unroll
      for (int kc = 0; kc < BK; kc += 16) {
        f32x4 xf[MT], wf[NT];
#pragma unroll
        for (int j = 0; j < MT; ++j) xf[j] = *reinterpret_cast<const f32x4*>(xa + j * 16 * LDK + kc);
#pragma unroll
        for (int i = 0; i < NT; ++i) wf[i] = *reinterpret_cast<const f32x4*>(wa + i * 16 * LDK + kc);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int j = 0; j < MT; ++j) acc[i][j] = mfma16(wf[i][s], xf[j][s], acc[i][j]);
      }
      sstore(buf ^ 1);
      __syncthreads();
    }

    // ---- partial tile: hand over / collect
    if (kt0 != 0 || kt1 != sk.nk) {
      // The 8 XCDs have separate L2s.  A device-scope fence would write back and invalidate the WHOLE L2 of the XCD per workgroup
      // and partial tile (measured: every layer 1.5-2x slower -- the weight / input working set is refetched each time);
      // instead only the partial tiles themselves move with system-scope cache policy (sc0 sc1: stores write through, loads
      // miss), ordered by plain vmcnt waits around the device-scope arrival counter.
      constexpr int QN = NT * MT;
      constexpr int POL = 0x11;              // cache policy of the buffer instructions: sc0 | sc1 (gfx940+ encoding of the aux operand)
      const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(sk.partial, 0, (G + sk.tiles_m * sk.tiles_n * nph) * (QN * 4096), 0x00020000);
      {
        const int so = __builtin_amdgcn_readfirstlane((L + t) * (QN * 4096));
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
          for (int j = 0; j < MT; ++j)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4_, acc[i][j]), prs, tid * 16 + (i * MT + j) * 4096, so, POL);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // written through before this wave reaches the barrier
      __syncthreads();
      if (tid == 0) {
        const long x0 = (long)t * sk.nk;
        const int Lf = (int)(((x0 + 1) * G - 1) / sk.total), Ll = (int)(((x0 + sk.nk) * G - 1) / sk.total);
        const unsigned old = __hip_atomic_fetch_add(sk.cnt + t, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool last = old == (unsigned)(Ll - Lf);
        if (last) __hip_atomic_store(sk.cnt + t, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // nobody else touches it before the next launch
        s_fix[0] = last ? Lf : -1;
        s_fix[1] = Ll;
      }
      __syncthreads();
      const int Lf = s_fix[0], Ll = s_fix[1];
      __syncthreads();                       // (s_fix is rewritten by the next partial tile of this block)
      if (Lf < 0) continue;
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      for (int Lc = Lf; Lc <= Ll; ++Lc) {
        const int so = __builtin_amdgcn_readfirstlane((Lc + t) * (QN * 4096));
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
          for (int j = 0; j < MT; ++j)
            acc[i][j] += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(prs, tid * 16 + (i * MT + j) * 4096, so, POL));
      }
    }

    // ---- epilogue: lane holds out[pixel m = .. + (l&15)][co = .. + (l>>4)*4 + r]
    float ssum[NT][4], ssq[NT][4];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) { ssum[i][r] = 0.f; ssq[i][r] = 0.f; }
#pragma unroll
    for (int j = 0; j < MT; ++j) {
      const int m = m_blk + wm * (MT * 16) + j * 16 + lr;
      if (m >= M) continue;
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        const int n = n_blk + wn * (NT * 16) + i * 16 + kq * 4;
        if (n >= a.Cout) continue;
        float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
        conv_store(a, m, n, v, ssum[i], ssq[i], ooy, oox);
      }
    }
    if (a.stats) {
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        const int n = n_blk + wn * (NT * 16) + i * 16 + kq * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float s = ssum[i][r], q = ssq[i][r];
          s += xshfl<1>(s); s += xshfl<2>(s); s += xshfl<4>(s); s += xshfl<8>(s);
          q += xshfl<1>(q); q += xshfl<2>(q); q += xshfl<4>(q); q += xshfl<8>(q);
          if (lr == 0 && n + r < a.Cout) {
            double* st = reinterpret_cast<double*>(a.stats) + (size_t)(bx % STAT_SLOTS) * 2 * a.Cout;
            atomicAdd(st + n + r, (double)(s));
            atomicAdd(st + a.Cout + n + r, (double)(q));
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------- halo conv for Cout <= 4
// A 16x16x4 MFMA with 4 output channels wastes 12 of its 16 feature rows.  Here the feature axis carries
// n = 4*co + dx (4 channels x 4 horizontal sub-taps): with kx = 4q + dx,
//     out[co][y][x] = sum_dx U_dx[co][y][x + dx],    U_dx[co][y][x'] = sum_{ky,q,c} W[co][ky][4q+dx][c] * X[y+ky-p][x'+4q-p][c],
// and all four U_dx read the SAME input pixel, so they share one B operand: a KS x KS conv costs KS*ceil(KS/4) MFMA taps per
// 16 pixels instead of KS*KS (27 vs 81 at 9x9, 3 vs 9 at 3x3).  The dx-shifted sum is 3 intra-row lane shuffles at the end;
// a 16-column tile therefore yields 13 finished output columns (tiles advance by 13).  Weights are read from the ordinary
// packed (Cout, Kp) layout with a different address map -- no extra pack.  Staging: halo tile per 32-channel chunk, weights
// per group of 3 taps (one ky row at 9x9, everything at 3x3), double-buffered.
template <int KS, int TH>
__global__ __launch_bounds__(256) void k_conv_halo_c4(ConvArgs a) {
  constexpr int TW = 16, VW = 13, HH = TH + KS - 1, HW_ = TW + KS - 1, NPX = HH * HW_;
  constexpr int Q = (KS + 3) / 4, MR = TH / 4, NG = (KS * Q) / 3;     // tap groups of 3 per chunk
  static_assert((KS * Q) % 3 == 0, "taps must come in groups of 3");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* halo = smem;                       // [NPX][LDK]
  float* Wt = smem + NPX * LDK;             // [2][3 taps][16][LDK]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_x = (a.Win + VW - 1) / VW, tiles_y = a.Hin / TH;
  const int b = blockIdx.x / (tiles_x * tiles_y), trem = blockIdx.x % (tiles_x * tiles_y);
  const int ty0 = (trem / tiles_x) * TH, tx0 = (trem % tiles_x) * VW;
  const int padk = (KS - 1) / 2;
  const int c01 = a.cseg[0] + a.cseg[1];
  const int nchunks = a.cin / BK;

  // The halo tile of the NEXT 32-channel chunk is requested into registers before this chunk's MFMAs and written to LDS behind
  // them (the weights were already double-buffered; the halo was loaded between two barriers, its global latency exposed once per
  // chunk: 6 chunks x ~2 us against 0.7 us of MFMAs per chunk on the 192 -> 3 output conv).  Affine / activation at the commit.
  constexpr int HV = (NPX * 8 + 255) / 256;
  float4 hraw[HV];
  unsigned hvalid = 0;
  auto chunk_seg = [&](int chunk, int& seg, int& cl0) {
    const int c0 = chunk * BK;
    seg = 0; cl0 = c0;
    if (c0 >= c01) { seg = 2; cl0 = c0 - c01; }
    else if (c0 >= a.cseg[0]) { seg = 1; cl0 = c0 - a.cseg[0]; }
  };
  auto issue_halo = [&](int chunk) {
    int seg, cl0;
    chunk_seg(chunk, seg, cl0);
    const float* src = a.in[seg];
    const int cs = a.cseg[seg];
    hvalid = 0;
#pragma unroll
    for (int v = 0; v < HV; ++v) {
      const int i = tid + v * 256;
      const int px = i >> 3, c4 = (i & 7) * 4;
      const int iy = ty0 + px / HW_ - padk, ix = tx0 + px % HW_ - padk;
      const bool ok = i < NPX * 8 && iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win;
      hraw[v] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ok) {
        hraw[v] = *reinterpret_cast<const float4*>(src + (((size_t)b * a.Hin + iy) * a.Win + ix) * cs + cl0 + c4);
        hvalid |= 1u << v;
      }
    }
  };
  auto commit_halo = [&](int chunk) {
    int seg, cl0;
    chunk_seg(chunk, seg, cl0);
    const float* sc = a.in_scale[seg];
    const float* sh = a.in_shift[seg];
#pragma unroll
    for (int v = 0; v < HV; ++v) {
      const int i = tid + v * 256;
      if (i < NPX * 8) {
        float4 val = hraw[v];
        const int px = i >> 3, c4 = (i & 7) * 4;
        if ((hvalid >> v) & 1u) {
          if (sc) {
            const float4 s4 = *reinterpret_cast<const float4*>(sc + cl0 + c4);
            const float4 h4 = *reinterpret_cast<const float4*>(sh + cl0 + c4);
            val.x = val.x * s4.x + h4.x; val.y = val.y * s4.y + h4.y; val.z = val.z * s4.z + h4.z; val.w = val.w * s4.w + h4.w;
          }
          if (a.pro_act != ACT_NONE) {
            val.x = apply_act(val.x, a.pro_act, 0.f); val.y = apply_act(val.y, a.pro_act, 0.f);
            val.z = apply_act(val.z, a.pro_act, 0.f); val.w = apply_act(val.w, a.pro_act, 0.f);
          }
        }
        *reinterpret_cast<float4*>(halo + px * LDK + c4) = val;
      }
    }
  };
  // group g of chunk: taps t = 3g .. 3g+2 ; tap -> (ky, q) = (t / Q, t % Q) ; LDS row n = 4*co + dx
  float4 wraw[2];
  auto issue_w = [&](int chunk, int g) {
#pragma unroll
    for (int v = 0; v < 2; ++v) {
      const int i = tid + v * 256;                 // 3 taps x 16 rows x 8 float4 = 384
      float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i < 384) {
        const int tl = i >> 7, n = (i >> 3) & 15, c4 = (i & 7) * 4;
        const int tp = 3 * g + tl, ky = tp / Q, q = tp - ky * Q;
        const int co = n >> 2, kx = 4 * q + (n & 3);
        if (co < a.Cout && kx < KS)
          val = *reinterpret_cast<const float4*>(a.w + (size_t)co * a.Kp + (size_t)(ky * KS + kx) * a.cin + chunk * BK + c4);
      }
      wraw[v] = val;
    }
  };
  auto commit_w = [&](int buf) {
#pragma unroll
    for (int v = 0; v < 2; ++v) {
      const int i = tid + v * 256;
      if (i < 384) *reinterpret_cast<float4*>(Wt + (size_t)buf * 48 * LDK + (i >> 3) * LDK + (i & 7) * 4) = wraw[v];
    }
  };

  const int lr = lane & 15, kq = lane >> 4;
  f32x4 acc[MR];
#pragma unroll
  for (int j = 0; j < MR; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  static_assert(HV <= 32, "halo validity mask");
  issue_w(0, 0);
  commit_w(0);
  issue_halo(0);
  commit_halo(0);
  int wb = 0;
  for (int chunk = 0; chunk < nchunks; ++chunk) {
    __syncthreads();                 // this chunk's halo (and the first weight group) visible
    if (chunk + 1 < nchunks) issue_halo(chunk + 1);
    for (int g = 0; g < NG; ++g) {
      const bool lastg = g == NG - 1, more = chunk + 1 < nchunks;
      if (!lastg) issue_w(chunk, g + 1);
      else if (more) issue_w(chunk + 1, 0);
#pragma unroll
      for (int tl = 0; tl < 3; ++tl) {
        const int tp = 3 * g + tl, ky = tp / Q, q = tp - ky * Q;
        const float* hp = halo + ((MR * wave + ky) * HW_ + lr + 4 * q) * LDK + kq * 4;
        const float* wp = Wt + (size_t)wb * 48 * LDK + (tl * 16 + lr) * LDK + kq * 4;
#pragma unroll
        for (int kc = 0; kc < BK; kc += 16) {
          const f32x4 wf = *reinterpret_cast<const f32x4*>(wp + kc);
          f32x4 xf[MR];
#pragma unroll
          for (int j = 0; j < MR; ++j) xf[j] = *reinterpret_cast<const f32x4*>(hp + j * HW_ * LDK + kc);
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
            for (int j = 0; j < MR; ++j) acc[j] = mfma16(wf[s4], xf[j][s4], acc[j]);
        }
      }
      if (!lastg || more) commit_w(wb ^ 1);
      __syncthreads();
      wb ^= 1;
    }
    // (the barrier that closed the last tap group: every wave is done with this chunk's halo)
    if (chunk + 1 < nchunks) commit_halo(chunk + 1);
  }
  // lane (lr, kq = co): acc[j][r] = U_r[co][row j][x' = tx0 + lr].  out[x] = sum_r U_r[x + r]: shift within the 16-lane row.
  float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < MR; ++j) {
    float o = acc[j][0];
#pragma unroll
    for (int r = 1; r < 4; ++r) o += __shfl(acc[j][r], (lane & 48) | ((lr + r) & 15), 64);
    // gather the 4 channels of pixel lr into the kq = 0 lane, then the common epilogue (bias / act / stats / layout)
    float v[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] = __shfl(o, lr + 16 * c, 64);
    const int oy = ty0 + MR * wave + j, ox = tx0 + lr;
    if (kq == 0 && lr < VW && ox < a.Win) conv_store(a, (b * a.Hin + oy) * a.Win + ox, 0, v, ssum, ssq, a.ooy, a.oox);
  }
  if (a.stats) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float s_ = ssum[r], q_ = ssq[r];     // only kq == 0 lanes hold non-zero partials
      s_ += xshfl<1>(s_); s_ += xshfl<2>(s_); s_ += xshfl<4>(s_); s_ += xshfl<8>(s_);
      q_ += xshfl<1>(q_); q_ += xshfl<2>(q_); q_ += xshfl<4>(q_); q_ += xshfl<8>(q_);
      if (lane == 0 && r < a.Cout) {
        double* st = reinterpret_cast<double*>(a.stats) + (size_t)(blockIdx.x % STAT_SLOTS) * 2 * a.Cout;
        atomicAdd(st + r, (double)(s_)); atomicAdd(st + a.Cout + r, (double)(q_));
      }
    }
  }
}

// ---------------------------------------------------------------------------------- direct conv for <= 16 x <= 16 channels
// The DistillModule convs (4/8 -> 4 channels over 196608 pixels, distill_module.py:9-12), their data gradients and the
// 12 -> 12 tail convs of the PGRM: a few hundred MACs per pixel.  On the MFMA tiles these are 1-6 % utilised and latency /
// atomic bound (25 us each; 95 us with BatchNorm statistics: 1536 blocks x 4 waves of same-address atomics).  Here one thread
// owns one output pixel and all its channels; the weights are wave-uniform, so hipcc reads them with scalar loads and feeds
// them to the FMAs as SGPR operands -- no LDS, no MFMA.  Prologue (affine + activation, zero outside the image) and epilogue
// (conv_store) are the implicit-GEMM path's; the statistics are reduced over the block before ONE atomic per channel.
template <int NG>      // channel quads of the output (Cout <= 4 NG)
__global__ __launch_bounds__(256) void k_conv_direct(ConvArgs a) {
  __shared__ float red[4][NG * 8];
  const int M = a.B * a.Hp * a.Wp;
  const int m = blockIdx.x * 256 + threadIdx.x;
  const bool live = m < M;
  const int mm = live ? m : M - 1;
  const int b = mm / (a.Hp * a.Wp), rr = mm - b * (a.Hp * a.Wp);
  const int py = rr / a.Wp, px = rr - py * a.Wp;
  const int iy0 = py * a.stride - a.pad_y, ix0 = px * a.stride - a.pad_x;
  float acc[NG][4];
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[g][r] = 0.f;
  int cbase = 0;
  for (int seg = 0; seg < 3; ++seg) {
    const int cs = a.cseg[seg];
    if (cs == 0) continue;
    const float* src = a.in[seg];
    const float* sc = a.in_scale[seg];
    const float* sh = a.in_shift[seg];
    for (int ky = 0; ky < a.KH; ++ky) {
      const int iy = iy0 + ky * a.dil_y;
      for (int kx = 0; kx < a.KW; ++kx) {
        const int ix = ix0 + kx * a.dil_x;
        const bool ok = live && iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win;
        const int iyc = min(max(iy, 0), a.Hin - 1), ixc = min(max(ix, 0), a.Win - 1);
        const float* xp = src + ((size_t)(b * a.Hin + iyc) * a.Win + ixc) * cs;
        const int k0 = (ky * a.KW + kx) * a.cin + cbase;
        for (int c4 = 0; c4 < cs; c4 += 4) {
          float4 x = *reinterpret_cast<const float4*>(xp + c4);
          if (sc) {
            const float4 s4 = *reinterpret_cast<const float4*>(sc + c4), h4 = *reinterpret_cast<const float4*>(sh + c4);
            x.x = x.x * s4.x + h4.x; x.y = x.y * s4.y + h4.y; x.z = x.z * s4.z + h4.z; x.w = x.w * s4.w + h4.w;
          }
          if (a.pro_act != ACT_NONE) {
            float v4[4] = {x.x, x.y, x.z, x.w};
            apply_act4(v4, a.pro_act, 0.f);
            x = make_float4(v4[0], v4[1], v4[2], v4[3]);
          }
          if (!ok) x = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
          for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int co = min(g * 4 + r, a.Cout - 1);                                   // (uniform: scalar loads)
              const float4 w4 = *reinterpret_cast<const float4*>(a.w + (size_t)co * a.Kp + k0 + c4);
              acc[g][r] += x.x * w4.x + x.y * w4.y + x.z * w4.z + x.w * w4.w;
            }
        }
      }
    }
    cbase += cs;
  }
  float ssum[NG][4], ssq[NG][4];
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int r = 0; r < 4; ++r) { ssum[g][r] = 0.f; ssq[g][r] = 0.f; }
  if (live) {
#pragma unroll
    for (int g = 0; g < NG; ++g)
      if (g * 4 < a.Cout) conv_store(a, m, g * 4, acc[g], ssum[g], ssq[g], a.ooy, a.oox);
  }
  if (a.stats) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float s_ = ssum[g][r], q = ssq[g][r];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { s_ += xshfl_v(s_, o); q += xshfl_v(q, o); }
        if (lane == 0) { red[wave][g * 8 + r] = s_; red[wave][g * 8 + 4 + r] = q; }
      }
    __syncthreads();
    const int t = threadIdx.x;
    if (t < NG * 8) {
      const int g = t >> 3, r = t & 3, sq = (t >> 2) & 1, n = g * 4 + r;
      if (n < a.Cout) {
        double* st = reinterpret_cast<double*>(a.stats) + (size_t)(blockIdx.x % STAT_SLOTS) * 2 * a.Cout;
        atomicAdd(st + (sq ? a.Cout : 0) + n, (double)(red[0][t] + red[1][t] + red[2][t] + red[3][t]));
      }
    }
  }
}


template <int KS, int TH>
int launch_halo_c4(const ConvArgs& a, hipStream_t st) {
  constexpr int NPX = (TH + KS - 1) * (16 + KS - 1);
  const size_t smem = (size_t)(NPX + 2 * 48) * LDK * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_halo_c4<KS, TH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_set = true;
  }
  dim3 grid(a.B * (a.Hin / TH) * cdiv(a.Win, 13));
  ProfScope prof(PT_CONV_HALO_C4, st, conv_flops(a), conv_bytes(a));
  hipLaunchKernelGGL((k_conv_halo_c4<KS, TH>), grid, dim3(256), smem, st, a);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

template <int KS, int BN, int TH, bool BF = false>
int launch_halo_th(const ConvArgs& a, hipStream_t st, bool x3 = false) {
  if constexpr (!BF && KS == 3 && BN == 64) {
    if (g_dpmn_bf16) return launch_halo_th<KS, BN, TH, true>(a, st);
    if (x3) {
      ProfScope prof(PT_CONV_HALO, st, conv_flops(a), conv_bytes(a));
      if (dpmn_conv::x3_launch_halo(KS, BN, TH, a, dim3(a.B * (a.Hin / TH) * (a.Win / 16), cdiv(a.Cout, BN)), st) != 0)
        return dpmn_set_error(DPMN_ERR_LAUNCH, "conv2d: bf16x3 halo launch failed");
      DPMN_CHECK_LAUNCH();
      return DPMN_OK;
    }
  }
  constexpr int NPX = (TH + KS - 1) * (16 + KS - 1);
  constexpr int TPS = (KS == 3 && !BF && TH == 4) ? DPMN_HALO_TPS : 1;      // as in the kernel
  const size_t smem = (size_t)(NPX + 2 * TPS * BN) * (BF ? (BK + 8) / 2 : LDK) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_halo<KS, BN, TH, BF>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_set = true;
  }
  dim3 grid(a.B * (a.Hin / TH) * (a.Win / 16), cdiv(a.Cout, BN));
  ProfScope prof(PT_CONV_HALO, st, conv_flops(a), conv_bytes(a));
  hipLaunchKernelGGL((k_conv_halo<KS, BN, TH, BF>), grid, dim3(256), smem, st, a);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

template <int KS, int BN>
int launch_halo(const ConvArgs& a, hipStream_t st) {
  // 8x16-pixel tiles amortise the weight staging best, but a map with fewer than ~3 tiles per CU leaves the CUs unevenly
  // loaded (384 tiles on 256 CUs = 1 or 2 per CU): halve the tile there
  static const int force = getenv("DPMN_HALO_TH") ? atoi(getenv("DPMN_HALO_TH")) : 0;
  const long blocks8 = (long)a.B * (a.Hin / 8) * (a.Win / 16) * cdiv(a.Cout, BN);
  const bool small = force ? force == 4 : blocks8 < 768;
  // mode 2 ("f32 via bf16x3", conv_x3.hip): 8-row tiles only -- with one output row per wave the split of a tap's weight slice costs
  // more vector time than the tap has MFMA time (measured: the 4-row variant is SLOWER than the fp32 kernel) -- and only where
  // those tiles fill the chip (DPMN_X3_HALO_MIN blocks); the other layers keep the fp32 kernel
  // (the rule is per IMAGE -- 8-row tiles x 64-channel blocks of one image, 8 = the 384 blocks of the B = 48 forward -- so that a sample
  //  meets the same kernel family whatever batch it travels in: the bf16x3 and the fp32 kernel differ by fp32-class round-off, which
  //  a mask threshold downstream -- toMask, util.py:27-35 -- can turn into a flipped pixel)
  static const int x3_min_img = getenv("DPMN_X3_HALO_MIN") ? atoi(getenv("DPMN_X3_HALO_MIN")) : 8;
  const long x3_min = (long)x3_min_img * a.B;
  static const int x3_th4 = getenv("DPMN_X3_HALO_TH4") ? atoi(getenv("DPMN_X3_HALO_TH4")) : 0;
  if constexpr (KS == 3 && BN == 64) {
    // 16 x 16-pixel tiles on eight waves (k_conv_halo_x3w): the small maps, where 8-row tiles do not fill the chip and the 4-row
    // variant is slower than fp32 -- and, by default, every layer the 8-row x3 kernel would take (half the weight-split work per MFMA)
    static const int x3_w16 = getenv("DPMN_X3_HALO16") ? atoi(getenv("DPMN_X3_HALO16")) : 0;      // 0 off (default: measured 55.8 vs 54.4 us forward, 79.6 vs 74.3 us training step per launch), 1 only below x3_min, 2 wherever it applies
    static const int x3_w16_min = getenv("DPMN_X3_HALO16_MIN") ? atoi(getenv("DPMN_X3_HALO16_MIN")) : 96;
    const long blocks16 = (long)a.B * (a.Hin / 16) * (a.Win / 16) * cdiv(a.Cout, BN);
    if (x3_on(2) && x3_w16 && a.Hin % 16 == 0 && blocks16 >= x3_w16_min && (x3_w16 >= 2 || blocks8 < x3_min)) {
      ProfScope prof(PT_CONV_HALO, st, conv_flops(a), conv_bytes(a));
      if (dpmn_conv::x3_launch_halo(KS, BN, 16, a, dim3(a.B * (a.Hin / 16) * (a.Win / 16), cdiv(a.Cout, BN)), st) != 0)
        return dpmn_set_error(DPMN_ERR_LAUNCH, "conv2d: bf16x3 halo launch failed");
      DPMN_CHECK_LAUNCH();
      return DPMN_OK;
    }
  }
  if (x3_on(2) && KS == 3 && BN == 64 && blocks8 >= x3_min) return launch_halo_th<KS, BN, 8>(a, st, true);
  if (KS == 3 && small) return launch_halo_th<KS, BN, 4>(a, st, x3_on(2) && x3_th4);
  return launch_halo_th<KS, BN, 8>(a, st);
}

// stream-K launch (k_conv_igemm_sk): returns -1 when the layer does not qualify (the caller falls through to the fixed-split path)
template <int BM, int BN, int WM, int WN>
int launch_conv_sk(const ConvArgs& a, float* ws, size_t ws_bytes, unsigned* cnt, int cnt_len, hipStream_t st) {
  static const int sk_on = getenv("DPMN_CONV_SK") ? atoi(getenv("DPMN_CONV_SK")) : 1;
  if (!sk_on || !ws || !cnt || g_dpmn_bf16) return -1;
  if ((long)a.B * a.Hp * a.Wp >= (1L << 24) || (a.groups == 2 && a.m_per_group % BM != 0)) return -1;
  // the SIMPLE load path (see launch_conv): 32-channel chunks in one segment, <= 31 taps, act(0) = 0 prologue, all-or-none affine
  bool simple = a.cin % 32 == 0 && (size_t)a.Cout * a.Kp * 4 < (1ull << 31) && a.KH * a.KW <= 31 &&
                (a.pro_act == ACT_NONE || a.pro_act == ACT_RELU || a.pro_act == ACT_LEAKY02);
  int n_seg = 0, n_aff = 0;
  for (int i = 0; i < 3; ++i) {
    if (a.cseg[i] > 0) { ++n_seg; n_aff += a.in_scale[i] != nullptr; }
    simple = simple && a.cseg[i] % 32 == 0 &&
             ((size_t)a.B * a.Hin * a.Win + (size_t)(abs(a.pad_y) + a.KH * abs(a.dil_y) + 2) * a.Win) * a.cseg[i] * 4 < (1ull << 31);
  }
  if (!simple || (n_aff != 0 && n_aff != n_seg)) return -1;
  static int capacity = 0;      // resident workgroups of the device (2 per CU by LDS and registers)
  if (!capacity) {
    int dev = 0, cus = 0, per_cu = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(&k_conv_igemm_sk<128, 128, 2, 2, false>), 256, 0);
    capacity = (cus > 0 ? cus : 256) * (per_cu > 0 ? per_cu : 2);
    if (getenv("DPMN_SK_BLOCKS")) capacity = atoi(getenv("DPMN_SK_BLOCKS"));      // experiment / test knob
  }
  const int M = a.B * a.Hp * a.Wp;
  const int nph = a.nphase > 1 ? a.nphase : 1;
  SkArgs sk{};
  sk.tiles_m = cdiv(M, BM); sk.tiles_n = cdiv(a.Cout, BN); sk.nk = a.Kp / BK;
  const int T = sk.tiles_m * sk.tiles_n * nph;
  sk.total = (long)T * sk.nk;
  int G = capacity;
  if (T >= 2 * G || T > cnt_len) return -1;      // enough whole tiles to balance the CUs without sharing any
  if (sk.total < 4L * G) G = (int)(sk.total / 4);      // at least 4 chunks per block
  if (G >= 8) G &= ~7;
  if (G < 1) G = 1;
  if ((size_t)(G + T) * BM * BN * sizeof(float) > ws_bytes) return -1;
  sk.partial = ws; sk.cnt = cnt;
  // consecutive tiles run on one XCD: let them share what costs more to re-read (as the wlocal rule of the fixed-split path)
  const double w_reread = (double)a.Cout * a.Kp * nph * (a.groups > 1 ? 2 : 1) * (sk.tiles_m > 8 ? 8 : sk.tiles_m);
  const double x_reread = (double)a.B * a.Hin * a.Win * a.cin * (sk.tiles_n > 8 ? 8 : sk.tiles_n);
  static const int force_order = getenv("DPMN_SK_ORDER") ? atoi(getenv("DPMN_SK_ORDER")) : -1;
  sk.order = force_order >= 0 ? force_order : (w_reread > x_reread ? 0 : 1);
  ProfScope prof(PT_CONV_IGEMM_SK, st, conv_flops(a), conv_bytes(a));
  if (n_aff) hipLaunchKernelGGL((k_conv_igemm_sk<BM, BN, WM, WN, true>), dim3(G), dim3(256), 0, st, a, sk);
  else hipLaunchKernelGGL((k_conv_igemm_sk<BM, BN, WM, WN, false>), dim3(G), dim3(256), 0, st, a, sk);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

// group size of the two-level in-launch reduction (XRED): minimise the chain (group size + number of groups) a tile's last
// blocks walk; and the arrival words, one line apart, must fit the counter array
static bool xred_fits(int tiles, int S, int cnt_len, ConvArgs& a) {
  static const int force_g = getenv("DPMN_XRED_GROUP") ? atoi(getenv("DPMN_XRED_GROUP")) : 0;
  static const int cstride = getenv("DPMN_XRED_CSTRIDE") ? atoi(getenv("DPMN_XRED_CSTRIDE")) : 32;
  int best = S, best_cost = S;         // one group: a chain of S slots
  for (int g = S - 1; g >= 2; --g)
    if (g + cdiv(S, g) + 1 < best_cost) { best = g; best_cost = g + cdiv(S, g) + 1; }      // (+1: the group sum is stored once more)
  if (force_g > 0) best = force_g < S ? force_g : S;
  a.xr_group = best;
  a.xr_cstride = cstride > 0 ? cstride : 1;
  while (a.xr_cstride > 1 && (long)tiles * (cdiv(S, best) + 1) * a.xr_cstride > cnt_len) a.xr_cstride >>= 1;
  return (long)tiles * (cdiv(S, best) + 1) * a.xr_cstride <= cnt_len;
}

template <int BM, int BN, int WM, int WN>
int launch_conv(ConvArgs a, float* ws, size_t ws_bytes, hipStream_t st, unsigned* cnt = nullptr, int cnt_len = 0) {
  if ((long)a.B * a.Hp * a.Wp >= (1L << 24))
    return dpmn_set_error(DPMN_ERR_ARG, "conv2d: the implicit-GEMM path decodes pixel indices in fp32 (B*Hp*Wp must be below 2^24)");
  if (a.groups == 2 && a.m_per_group % BM != 0)
    return dpmn_set_error(DPMN_ERR_ARG, "conv2d: groups = 2 needs the pixels of one half to fill whole row tiles (B/2*Hp*Wp % 128 == 0)");
  const int M = a.B * a.Hp * a.Wp;
  const int nph = a.nphase > 1 ? a.nphase : 1;
  const int tiles = cdiv(M, BM) * cdiv(a.Cout, BN) * nph;
  const int nk = a.Kp / BK;
  int S = 1;
  static const int force_s = getenv("DPMN_CONV_S") ? atoi(getenv("DPMN_CONV_S")) : 0;      // experiment knob: fixed split count
  static const int target = getenv("DPMN_CONV_TARGET") ? atoi(getenv("DPMN_CONV_TARGET")) : 768;
  if (ws && tiles < 384 && nk >= 16) {
    S = force_s > 0 ? force_s : cdiv(target, tiles);
    if (S > nk / 8) S = nk / 8;
    if (S > 64) S = 64;
    static const size_t cap_mb = getenv("DPMN_SPLITK_CAP_MB") ? (size_t)atoi(getenv("DPMN_SPLITK_CAP_MB")) : 32;
    const size_t cap = ws_bytes < (cap_mb << 20) ? ws_bytes : (cap_mb << 20);   // keep the partial-sum round trip small
    while (S > 1 && (size_t)S * nph * M * a.npad * sizeof(float) > cap) --S;
    const int cps = cdiv(nk, S);
    S = cdiv(nk, cps);   // no empty splits
  }
  a.ksplit = S;
  a.partial = S > 1 ? ws : nullptr;
  dim3 grid(cdiv(M, BM), cdiv(a.Cout, BN), S * nph);
  {
    // weight-local XCD mapping when re-reading the weights per row tile costs more than re-reading the input per column tile
    static const int wlocal_on = getenv("DPMN_CONV_WLOCAL") ? atoi(getenv("DPMN_CONV_WLOCAL")) : 1;
    const double w_reread = (double)a.Cout * a.Kp * nph * (a.groups > 1 ? 2 : 1) * (grid.x > 8 ? 8 : grid.x);
    const double x_reread = (double)a.B * a.Hin * a.Win * a.cin * (grid.y > 8 ? 8 : grid.y);
    a.wlocal = wlocal_on && ((grid.y * grid.z) % 8 == 0) && grid.x > 1 && w_reread > x_reread;
  }
  // segment-uniform chunks (all channel counts multiples of 32) and 32-bit byte offsets: the buffer-load instantiation
  static const int uni_on = getenv("DPMN_CONV_UNI") ? atoi(getenv("DPMN_CONV_UNI")) : 1;
  bool uni = uni_on && a.cin % 32 == 0 && (size_t)a.Cout * a.Kp * 4 < (1ull << 31);
  for (int i = 0; i < 3; ++i)
    uni = uni && a.cseg[i] % 32 == 0 && (size_t)a.B * a.Hin * a.Win * a.cseg[i] * 4 < (1ull << 31);
  {
    // split-K launches additionally move S partial-sum slabs (written here, read by the reduce kernel): not algorithmic
    ProfScope prof(BN == 128 ? PT_CONV_IGEMM_128 : (BN == 64 ? PT_CONV_IGEMM_64 : PT_CONV_IGEMM_NARROW), st, conv_flops(a), conv_bytes(a));
    static const int bk16 = getenv("DPMN_CONV_BK16") ? atoi(getenv("DPMN_CONV_BK16")) : 0;
    static const int simple_on = getenv("DPMN_CONV_SIMPLE") ? atoi(getenv("DPMN_CONV_SIMPLE")) : 1;
    static const int m32_on = getenv("DPMN_CONV_M32") ? atoi(getenv("DPMN_CONV_M32")) : 0;
    static const int xred_env = getenv("DPMN_CONV_XRED") ? atoi(getenv("DPMN_CONV_XRED")) : 0;      // split-K reduced in the launch through one XCD's L2
    const int xred_on = g_xred_enabled >= 0 ? g_xred_enabled : xred_env;
    static const int o3_on = getenv("DPMN_CONV_O3") ? atoi(getenv("DPMN_CONV_O3")) : 0;          // 16-deep chunks + 3 waves per SIMD      // 32x32x2 MFMAs on the 128 x 128 tile
    bool simple = simple_on && uni && a.KH * a.KW <= 31 && (a.pro_act == ACT_NONE || a.pro_act == ACT_RELU || a.pro_act == ACT_LEAKY02);
    int n_seg = 0, n_aff = 0;
    for (int i = 0; i < 3; ++i) {    // + the shift of the buffer base must keep the byte range below 2^31
      if (a.cseg[i] > 0) { ++n_seg; n_aff += a.in_scale[i] != nullptr; }
      simple = simple &&
               ((size_t)a.B * a.Hin * a.Win + (size_t)(abs(a.pad_y) + a.KH * abs(a.dil_y) + 2) * a.Win) * a.cseg[i] * 4 < (1ull << 31);
    }
    simple = simple && (n_aff == 0 || n_aff == n_seg);       // mixed segments: the general UNI path
  static const int x3_t64 = getenv("DPMN_X3_TILE64") ? atoi(getenv("DPMN_X3_TILE64")) : 0;      // the 64 x 64 tile in mode 2: measured 93 vs 88 us (forward), 65 vs 60 (training step) -- off
  if (x3_on(1) && simple && ((BM == BN && (BM == 128 || (BM == 64 && x3_t64))) || (BM == 128 && BN == 64))) {
    if (dpmn_conv::x3_launch_igemm(BM == BN ? BM : 12864, n_aff != 0, a, grid, st) != 0) return dpmn_set_error(DPMN_ERR_LAUNCH, "conv2d: bf16x3 launch failed");
  } else
  if (g_dpmn_bf16 && simple && BM == BN && (BM == 128 || BM == 64)) {
    if (n_aff) hipLaunchKernelGGL((k_conv_igemm<BM, BN, WM, WN, true, 32, true, true, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((k_conv_igemm<BM, BN, WM, WN, true, 32, true, false, true>), grid, dim3(256), 0, st, a);
  } else
  if (uni && BM == 128 && BN == 128 && bk16) hipLaunchKernelGGL((k_conv_igemm<BM, BN, WM, WN, true, 16>), grid, dim3(256), 0, st, a);
  else if (simple && o3_on && BM == 128 && BN == 128) {
    if constexpr (BM == 128 && BN == 128) {
      if (n_aff) hipLaunchKernelGGL((k_conv_igemm_o3<BM, BN, WM, WN, true>), grid, dim3(256), 0, st, a);
      else hipLaunchKernelGGL((k_conv_igemm_o3<BM, BN, WM, WN, false>), grid, dim3(256), 0, st, a);
    }
  }
  else if (simple && m32_on && BM / WM == 64 && BN / WN == 64) {
    if constexpr (BM / WM == 64 && BN / WN == 64) {
      if (n_aff) hipLaunchKernelGGL((k_conv_igemm<BM, BN, WM, WN, true, 32, true, true, false, true>), grid, dim3(256), 0, st, a);
      else hipLaunchKernelGGL((k_conv_igemm<BM, BN, WM, WN, true, 32, true, false, false, true>), grid, dim3(256), 0, st, a);
    }
  }
  else if (simple && S > 1 && BM == 128 && BN == 128 && xred_on && cnt && xred_fits(tiles, S, cnt_len, a) &&
           (size_t)tiles * S * BM * BN * sizeof(float) <= ws_bytes) {
    // split-K reduced inside the launch through one XCD's L2 (XRED in conv_igemm_body): 1-D grid, XCD c owns a contiguous run of
    // tiles.  Tile order = the one that re-reads fewer bytes: every XCD fetches the weight column tiles and the input row tiles
    // its run touches once (counted exactly: T <= 4096 tiles)
    if constexpr (BM == 128 && BN == 128) {
      const int tm = cdiv(M, BM), tn = cdiv(a.Cout, BN), t8 = cdiv(tiles, 8);
      double cost[2];
      for (int order = 0; order < 2; ++order) {
        double wsets = 0, xsets = 0;
        for (int c = 0; c < 8; ++c) {
          const int t0 = c * t8, t1 = (c + 1) * t8 < tiles ? (c + 1) * t8 : tiles;
          if (t0 >= t1) break;
          if (order == 0) {      // t = (ph * tn + by) * tm + bx
            const int r0 = t0 / tm, r1 = (t1 - 1) / tm;
            wsets += (a.groups > 1 && r0 == r1) ? ((t0 % tm) * BM < a.m_per_group && ((t1 - 1) % tm) * BM >= a.m_per_group ? 2 : 1)
                                                : (r1 - r0 + 1) * (a.groups > 1 ? 2 : 1);
            xsets += r1 > r0 ? tm : t1 - t0;
          } else {                // t = (bx * nph + ph) * tn + by
            const int per = tn * nph, r0 = t0 / per, r1 = (t1 - 1) / per;
            xsets += r1 - r0 + 1;
            wsets += r1 > r0 ? per : t1 - t0;
          }
        }
        cost[order] = wsets * BN * (double)a.Kp + xsets * BM * (double)a.stride * a.stride * a.cin;
      }
      static const int force_order = getenv("DPMN_XRED_ORDER") ? atoi(getenv("DPMN_XRED_ORDER")) : -1;
      static const int ablate = getenv("DPMN_XRED_ABLATE") ? atoi(getenv("DPMN_XRED_ABLATE")) : 0;
      a.xr_cnt = cnt; a.xr_tm = tm; a.xr_tn = tn; a.xr_t8 = t8; a.xr_force_redo = g_xred_force_recompute; a.xr_ablate = ablate;      // (xr_group / xr_cstride: xred_fits)
      a.xr_order = force_order >= 0 ? force_order : (cost[0] <= cost[1] ? 0 : 1);
      a.wlocal = 0;
      const dim3 g1(8 * t8 * S);
      if (n_aff) hipLaunchKernelGGL((k_conv_igemm_xr<BM, BN, WM, WN, true>), g1, dim3(256), 0, st, a);
      else hipLaunchKernelGGL((k_conv_igemm_xr<BM, BN, WM, WN, false>), g1, dim3(256), 0, st, a);
      DPMN_CHECK_LAUNCH();
      return DPMN_OK;
    }
  }
  else if (simple && n_aff) hipLaunchKernelGGL((k_conv_igemm<BM, BN, WM, WN, true, 32, true, true>), grid, dim3(256), 0, st, a);
  else if (simple) hipLaunchKernelGGL((k_conv_igemm<BM, BN, WM, WN, true, 32, true>), grid, dim3(256), 0, st, a);
  else if (uni) hipLaunchKernelGGL((k_conv_igemm<BM, BN, WM, WN, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((k_conv_igemm<BM, BN, WM, WN>), grid, dim3(256), 0, st, a);
  }
  DPMN_CHECK_LAUNCH();
  if (S > 1) {
    ProfScope prof(PT_CONV_SPLITK_REDUCE, st, 0.0, 4.0 * (double)(S + 1) * nph * M * a.npad);
    // rows per block: 64 amortises the BatchNorm-statistics atomics on big outputs; small outputs need the parallelism
    const int n4 = a.npad / 4;
    const int cql = n4 >= 64 ? 64 : (n4 >= 32 ? 32 : 16);
    const int cb = cdiv(n4, cql);
    int rows = cb * cdiv(M, 64) >= 1024 ? 64 : (cb * cdiv(M, 16) >= 1024 ? 16 : 4);
    if (rows < 256 / cql) rows = 256 / cql;      // at least one row per row lane
    hipLaunchKernelGGL(k_conv_splitk_reduce, dim3(cb, cdiv(M, rows), nph), dim3(256), 0, st, a, rows, cql);
    DPMN_CHECK_LAUNCH();
  }
  return DPMN_OK;
}

// NCHW (B, C, H, W) -> NHWC (B, H, W, Cp) with zero-filled channels C..Cp-1
__global__ void k_nchw_to_nhwc(const float* __restrict__ in, float* __restrict__ out, int B, int C, int H, int W, int Cp) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)B * H * W;
  if (idx >= total) return;
  const int b = idx / ((long)H * W);
  const long hw = idx % ((long)H * W);
  for (int c = 0; c < Cp; ++c) out[idx * Cp + c] = c < C ? in[((size_t)b * C + c) * H * W + hw] : 0.f;
}

// NHWC (B, H, W, C) -> NCHW
__global__ void k_nhwc_to_nchw(const float* __restrict__ in, float* __restrict__ out, int B, int C, int H, int W) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)B * C * H * W;
  if (idx >= total) return;
  const int x = idx % W, y = (idx / W) % H, c = (idx / ((long)W * H)) % C, b = idx / ((long)W * H * C);
  out[idx] = in[(((size_t)b * H + y) * W + x) * C + c];
}

}  // namespace

extern "C" {

int dpmn_conv2d_nhwc_f32(const dpmn_conv_desc* d, dpmn_stream_t stream) {
  DPMN_REQUIRE(d && d->in[0] && d->w && d->out, "conv2d: null pointer");
  ConvArgs a{};
  int cin = 0;
  for (int s = 0; s < 3; ++s) {
    a.in[s] = d->in[s]; a.in_scale[s] = d->in_scale[s]; a.in_shift[s] = d->in_shift[s]; a.cseg[s] = d->in[s] ? d->cseg[s] : 0;
    DPMN_REQUIRE(a.cseg[s] % 4 == 0, "conv2d: segment channel counts must be multiples of 4");
    DPMN_REQUIRE((d->in_scale[s] == nullptr) == (d->in_shift[s] == nullptr), "conv2d: scale and shift go together");
    cin += a.cseg[s];
  }
  a.cin = cin;
  a.B = d->B; a.Hin = d->Hin; a.Win = d->Win;
  a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.dil_y = d->dil_y; a.dil_x = d->dil_x; a.pad_y = d->pad_y; a.pad_x = d->pad_x;
  a.Hp = d->Hp; a.Wp = d->Wp; a.Hout = d->Hout; a.Wout = d->Wout; a.ostep = d->ostep; a.ooy = d->ooy; a.oox = d->oox;
  a.pro_act = d->pro_act; a.w = d->w; a.bias = d->bias; a.Cout = d->Cout; a.epi_act = d->epi_act; a.slope = d->slope;
  a.res = d->res; a.out = d->out; a.out_ld = d->out_ld > 0 ? d->out_ld : d->Cout; a.out_coff = d->out_coff;
  a.out_nchw = d->out_nchw; a.pixel_shuffle = d->pixel_shuffle; a.stats = d->stats;
  a.Kp = ((d->KH * d->KW * cin + 31) / 32) * 32;
  a.nphase = d->nphase == 4 ? 4 : 1; a.wps = d->w_phase_stride;
  a.groups = d->groups == 2 ? 2 : 1; a.wgs = d->w_group_stride; a.m_per_group = a.groups == 2 ? d->B / 2 * d->Hp * d->Wp : 0;
  DPMN_REQUIRE(d->groups == 0 || d->groups == 1 || (d->groups == 2 && d->B % 2 == 0 && d->w_group_stride > 0 && !d->stats),
               "conv2d: groups = 2 splits an even batch into halves with their own weights (w_group_stride) and bias (+Cout)");
  DPMN_REQUIRE(d->nphase == 0 || d->nphase == 1 || (d->nphase == 4 && d->KH == 2 && d->KW == 2 && d->dil_y == -1 && d->dil_x == -1 &&
                                                    d->ostep == 2 && d->w_phase_stride >= (long)d->Cout * a.Kp),
               "conv2d: nphase = 4 is the fused ConvTranspose2d(4,2,1) launch (k 2, dil -1, ostep 2, 4 packed phase weights)");
  DPMN_REQUIRE(d->B > 0 && d->Hp > 0 && d->Wp > 0 && d->Cout > 0, "conv2d: empty shape");
  DPMN_REQUIRE(!(d->pixel_shuffle && (d->Cout % 4 != 0 || d->out_nchw || d->res)), "conv2d: bad pixel-shuffle epilogue");
  DPMN_REQUIRE(d->out_nchw || (a.out_ld % 4 == 0 && a.out_coff % 4 == 0) || d->Cout < 4 || d->pixel_shuffle,
               "conv2d: NHWC output needs 16-byte aligned channel rows");
  hipStream_t st = as_stream(stream);
  const int M = a.B * a.Hp * a.Wp;
  a.npad = (a.Cout + 3) / 4 * 4;
  float* ws = d->splitk_ws;
  const size_t wsb = d->splitk_ws_bytes;
  // stride-1 "same" 3x3 / 9x9 convs on 8x16-tileable maps: input tile resident in LDS
  const bool halo_ok = a.stride == 1 && a.dil_y == 1 && a.dil_x == 1 && a.KH == a.KW && (a.KH == 3 || a.KH == 9) &&
                       a.pad_y == (a.KH - 1) / 2 && a.pad_x == a.pad_y && a.Hin % 8 == 0 && a.Win % 16 == 0 && a.ostep == 1 &&
                       a.Hp == a.Hin && a.Wp == a.Win && cin % 32 == 0 && a.cseg[0] % 32 == 0 && a.cseg[1] % 32 == 0 &&
                       a.cseg[2] % 32 == 0 && M >= 1024 &&
                       (size_t)a.B * a.Hin * a.Win * (size_t)cin * 4 < (1ull << 31);   // 32-bit buffer-load offsets
  // too few 8x16-pixel tiles to fill 256 CUs (deep decoder levels with 3-segment inputs): split-K implicit GEMM instead
  const bool halo_starved = (M / 128) * cdiv(a.Cout, 64) < 256 && a.Cout >= 128 && ws != nullptr;
  static const bool direct_on = !(getenv("DPMN_CONV_DIRECT") && atoi(getenv("DPMN_CONV_DIRECT")) == 0);
  // (measured: 8 -> 4 with statistics 95 -> 20 us, 4 -> 4 25 -> 12 us; 12 -> 12 over 49152 pixels 22 -> 37 us -- too few,
  //  too heavy threads -- so the rule is cin * Cout <= 64)
  if (direct_on && cin <= 16 && a.Cout <= 16 && cin * a.Cout <= 64 && a.nphase == 1 && a.groups == 1 && !a.pixel_shuffle && M >= 4096) {
    ProfScope prof(PT_CONV_IGEMM_NARROW, st, conv_flops(a), conv_bytes(a));
    const dim3 grid(cdiv(M, 256));
    if (a.Cout <= 4) hipLaunchKernelGGL(k_conv_direct<1>, grid, dim3(256), 0, st, a);
    else if (a.Cout <= 8) hipLaunchKernelGGL(k_conv_direct<2>, grid, dim3(256), 0, st, a);
    else if (a.Cout <= 12) hipLaunchKernelGGL(k_conv_direct<3>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(k_conv_direct<4>, grid, dim3(256), 0, st, a);
    DPMN_CHECK_LAUNCH();
    return DPMN_OK;
  }
  static const bool c4_on = !(getenv("DPMN_CONV_C4") && atoi(getenv("DPMN_CONV_C4")) == 0);
  if (halo_ok && a.Cout <= 4 && c4_on && !a.pixel_shuffle && !a.res && a.groups == 1)
    return a.KH == 3 ? launch_halo_c4<3, 8>(a, st) : launch_halo_c4<9, 8>(a, st);   // (3x3: 4- and 16-row tiles measured no better)
  if (halo_ok && !halo_starved) {
    // (128 output channels per block -- one halo staging instead of two, 64 MFMAs per tap and barrier -- measured 212-220 us
    //  against 143-148 us on the 64 -> 128 / 64 -> 256 convs: rejected, switch kept for the record)
    static const int bn128 = getenv("DPMN_HALO_BN128") ? atoi(getenv("DPMN_HALO_BN128")) : 0;
    if (a.KH == 3 && bn128 && a.Cout % 128 == 0) return launch_halo<3, 128>(a, st);
    if (a.KH == 3) return a.Cout <= 16 ? launch_halo<3, 16>(a, st) : launch_halo<3, 64>(a, st);
    return a.Cout <= 16 ? launch_halo<9, 16>(a, st) : launch_halo<9, 64>(a, st);
  }
  // groups = 2 on the implicit-GEMM tiles needs the pixels of one half to fill whole row tiles; otherwise one launch per half
  auto split_groups = [&]() -> int {
    DPMN_REQUIRE(!a.out_nchw && !a.pixel_shuffle && !a.res, "conv2d: groups = 2 with partial row tiles needs a plain NHWC output");
    for (int g = 0; g < 2; ++g) {
      dpmn_conv_desc h = *d;
      h.B = d->B / 2; h.groups = 1;
      for (int s = 0; s < 3; ++s)
        if (h.in[s]) h.in[s] += (size_t)g * h.B * d->Hin * d->Win * d->cseg[s];
      h.w += (size_t)g * d->w_group_stride;
      if (h.bias) h.bias += (size_t)g * d->Cout;
      h.out += (size_t)g * h.B * d->Hout * d->Wout * a.out_ld;
      const int e = dpmn_conv2d_nhwc_f32(&h, stream);
      if (e != DPMN_OK) return e;
    }
    return DPMN_OK;
  };
  if (a.groups == 2 && !(a.Cout >= 128 && M >= 128) && a.m_per_group % (a.Cout <= 32 ? 128 : 64) != 0) return split_groups();
  if (a.Cout <= 16) return launch_conv<128, 16, 4, 1>(a, ws, wsb, st);
  if (a.Cout <= 32) return launch_conv<128, 32, 4, 1>(a, ws, wsb, st);
  // 128x128 tiles halve the L2->LDS bytes per FLOP of the 64x64 tile (which is L2-bound); small-M convs regain
  // parallelism through split-K (deep CMM levels: K = 2304..13824)
  static const int force_tile = getenv("DPMN_CONV_TILE") ? atoi(getenv("DPMN_CONV_TILE")) : 0;          // experiment knob
  if (force_tile == 64) return launch_conv<64, 64, 2, 2>(a, ws, wsb, st);
  if (a.Cout >= 128 && M >= 128) {
    // stream-K (one persistent launch, no reduce kernel) where the fixed-split path would split K or leave CUs idle; 64-pixel
    // row tiles when 128-pixel ones would be partly empty (the 1x4 bottleneck maps: M = 192 per branch / phase)
    const int mg = a.groups == 2 ? a.m_per_group : M;
    const bool rows64 = mg % 128 != 0 && mg % 64 == 0 && M <= 1024;
    // DPMN_CONV_SK: 0 never, 1 (default) the 64-row layers only, 2 every layer that qualifies (measured: the deep-K layers lose
    // to the fixed split + parallel reduce launch -- a tile shared by 5 ... 21 workgroups is collected by ONE of them)
    static const int sk_mode = getenv("DPMN_CONV_SK") ? atoi(getenv("DPMN_CONV_SK")) : 1;
    static const int rows64_fixed = getenv("DPMN_ROWS64_FIXED") ? atoi(getenv("DPMN_ROWS64_FIXED")) : 0;
    if (rows64 && rows64_fixed && (a.groups != 2 || mg % 64 == 0)) return launch_conv<64, 128, 1, 4>(a, ws, wsb, st);
    const int r = rows64 ? launch_conv_sk<64, 128, 1, 4>(a, ws, wsb, d->arrive_cnt, d->arrive_cnt_len, st)
                         : (sk_mode >= 2 ? launch_conv_sk<128, 128, 2, 2>(a, ws, wsb, d->arrive_cnt, d->arrive_cnt_len, st) : -1);
    if (r >= 0) return r;
    if (a.groups == 2 && mg % 128 != 0) return split_groups();
    {
      // 64-pixel row tiles WITHOUT a K split where the 128 x 128 tiling would split K only to fill the CUs (no partial-sum slabs, no
      // reduce launch): DPMN_CONV_ALT64 = minimum number of 64 x 128 tiles (0 = off)
      static const int alt64 = getenv("DPMN_CONV_ALT64") ? atoi(getenv("DPMN_CONV_ALT64")) : 0;
      const int nph = a.nphase > 1 ? a.nphase : 1;
      const int t128 = cdiv(M, 128) * cdiv(a.Cout, 128) * nph, t64 = cdiv(M, 64) * cdiv(a.Cout, 128) * nph;
      if (alt64 > 0 && ws && t128 < 384 && a.Kp / 32 >= 16 && t64 >= alt64 && (a.groups != 2 || mg % 64 == 0))
        return launch_conv<64, 128, 1, 4>(a, nullptr, 0, st);
    }
    return launch_conv<128, 128, 2, 2>(a, ws, wsb, st, d->arrive_cnt, d->arrive_cnt_len);
  }
  // mode 2 ("f32 via bf16x3"): 128-pixel row tiles -- the operand split of a 64 x 64 tile costs as many vector instructions as the
  // tile has MFMA cycles (conv_x3.hip); the fp32 kernels keep the 64 x 64 tile (more blocks for the small maps)
  static const int x3_rows128 = getenv("DPMN_X3_ROWS128") ? atoi(getenv("DPMN_X3_ROWS128")) : 0;      // measured: 73.7 vs 60.1 us (fp32 64 x 64) over the 18 launches of the training step -- off
  if (x3_on(1) && x3_rows128 && M >= 128 * 256 && (a.groups != 2 || a.m_per_group % 128 == 0)) return launch_conv<128, 64, 2, 2>(a, ws, wsb, st);
  return launch_conv<64, 64, 2, 2>(a, ws, wsb, st);
}

int dpmn_xred_fallbacks(unsigned* count_out, int reset) {
  DPMN_REQUIRE(count_out, "xred_fallbacks: null pointer");
  unsigned v = 0;
  if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_xred_fallbacks), sizeof(v)) != hipSuccess) return dpmn_set_error(DPMN_ERR_RUNTIME, "xred_fallbacks: copy failed");
  *count_out = v;
  if (reset) {
    v = 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_xred_fallbacks), &v, sizeof(v)) != hipSuccess) return dpmn_set_error(DPMN_ERR_RUNTIME, "xred_fallbacks: reset failed");
  }
  return DPMN_OK;
}

int dpmn_xred_enable(int on) {
  g_xred_enabled = on < 0 ? -1 : (on ? 1 : 0);
  return DPMN_OK;
}

int dpmn_xred_test_force_recompute(int on) {
  g_xred_force_recompute = on ? 1 : 0;
  return DPMN_OK;
}

int dpmn_nchw_to_nhwc_f32(const float* in, float* out, int B, int C, int H, int W, int Cpad, dpmn_stream_t stream) {
  DPMN_REQUIRE(in && out && Cpad >= C, "nchw_to_nhwc: bad arguments");
  const long total = (long)B * H * W;
  hipLaunchKernelGGL(k_nchw_to_nhwc, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), in, out, B, C, H, W, Cpad);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

int dpmn_nhwc_to_nchw_f32(const float* in, float* out, int B, int C, int H, int W, dpmn_stream_t stream) {
  DPMN_REQUIRE(in && out, "nhwc_to_nchw: bad arguments");
  const long total = (long)B * C * H * W;
  hipLaunchKernelGGL(k_nhwc_to_nchw, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), in, out, B, C, H, W);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

}  // extern "C"
