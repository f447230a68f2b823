// Shared between conv.hip and conv_x3.hip: the argument block of the conv kernels, the implicit-GEMM body (conv_igemm_body), the
// halo-tile kernel (k_conv_halo) and the algorithmic FLOP / byte counts.  Everything but ConvArgs lives in an anonymous namespace:
// each translation unit instantiates the variants it launches (conv_x3.hip: the bf16x3 ones -- a change there recompiles in
// seconds instead of the two minutes of conv.hip).
#pragma once
#include <cstdlib>
#include <type_traits>
#include "common.h"

namespace dpmn_conv {
struct ConvArgs {
  const float* in[3];
  const float* in_scale[3];   // per-channel affine on load (or null)
  const float* in_shift[3];
  int cseg[3];                // channels per segment (multiples of 4); unused segments 0
  int cin;                    // sum of cseg (padded channel count used in the weight pack)
  int B, Hin, Win;
  int KH, KW, stride, dil_y, dil_x, pad_y, pad_x;   // iy = oy'*stride + ky*dil_y - pad_y
  int Hp, Wp;                 // phase-grid size (number of output pixels computed per image = Hp*Wp)
  int Hout, Wout, ostep, ooy, oox;                  // oy = oy'*ostep + ooy
  int pro_act;                // activation applied to the loaded input (after affine)
  const float* w;             // packed (Cout, Kp) with Kp = roundup(KH*KW*cin, 32), zero padded
  int Kp;
  const float* bias;          // (Cout) or null
  int Cout;
  int epi_act;
  float slope;
  const float* res;           // residual, same layout as the output, or null
  float* out;
  int out_ld, out_coff;       // NHWC: channel stride of the output buffer and channel offset
  int out_nchw;               // 1: store NCHW (B, Cout, Hout, Wout)
  int pixel_shuffle;          // 1: PixelShuffle(2) store: NHWC (B, 2*Hout, 2*Wout, Cout/4)
  float* stats;               // (2, Cout) running sum / sum of squares of the pre-activation output, or null
  float* partial;             // split-K scratch (ksplit, M, Npad) or null
  int ksplit;                 // number of K splits (gridDim.z)
  int npad;                   // Cout rounded up to 4
  int nphase;                 // 4: the phases of ConvTranspose2d(4,2,1) in one launch (gridDim.z = nphase * ksplit):
  long wps;                   //    phase p = 2*py + px uses w + p*wps, pad = -(py,px), output offset (py,px)
  int groups;                 // 2: the batch holds two independent halves (the CMM's twin encoder branches, cmm.py:86-99): pixels
  int m_per_group;            //    m >= m_per_group use w + wgs and bias + Cout -- one launch, twice the tiles, half the split-K
  long wgs;
  int wlocal;                 // implicit GEMM: 1 = blocks that share a weight slice (same n tile, same k split) are dealt to ONE XCD
  // XRED (in-L2 split-K reduction, see conv_igemm_body): 1-D grid of 8 * xr_t8 * ksplit workgroups
  unsigned* xr_cnt;           // one arrival word per tile: zero on entry, zero again on exit
  int xr_tm, xr_tn;           // row / column tiles
  int xr_t8;                  // tiles per XCD (ceil(T / 8)): XCD c owns the tiles [c * xr_t8, (c + 1) * xr_t8)
  int xr_order;               // 0: row tile fastest (consecutive tiles share a weight column tile), 1: column tile fastest
  int xr_force_redo;          // test hook: treat every tile as misplaced (the recompute path)
  int xr_group;               // splits per first-level group
  int xr_cstride;             // words between two arrival words
  int xr_ablate;              // timing experiments only (DPMN_XRED_ABLATE): 1 no collect loads, 2 no wait for the partial stores, 4 no atomics / barriers
};
// conv_x3.hip ("f32 via bf16x3" instantiations, dpmn_set_compute_dtype(2)); both return 0 or -1 = no such variant (caller falls through)
int x3_launch_igemm(int tile, bool aff, const ConvArgs& a, dim3 grid, hipStream_t st);      // tile: 128 (128 x 128), 12864 (128 x 64) or 64 (64 x 64), SIMPLE path
int x3_launch_halo(int ks, int bn, int th, const ConvArgs& a, dim3 grid, hipStream_t st);  // 3 x 3, 64 output channels per block
}  // namespace dpmn_conv

namespace {

constexpr int PAD = 4, BK = 32, LDK = BK + PAD;
constexpr int STAT_SLOTS = 32;   // BatchNorm statistics are accumulated into (STAT_SLOTS, 2, Cout) DOUBLES and summed by bn_finalize.
// fp64 atomics: a block's fp32 partial sum is exact in fp64 and the fp64 additions of <= a few thousand partials lose nothing a
// final rounding to fp32 can see, so the statistics -- hence the whole training forward -- no longer depend on the order in which
// the blocks arrive (fp32 atomics made two runs of the same step differ by 1e-7 ... 5e-4 downstream)

using dpmn_conv::ConvArgs;
int g_xred_enabled = -1;                     // -1: DPMN_CONV_XRED (default 0: measured slower than the reduce launch); dpmn_xred_enable
int g_xred_force_recompute = 0;              // test hook (dpmn_xred_test_force_recompute): every tile takes the recompute path
__device__ unsigned g_xred_fallbacks = 0;      // tiles that took the recompute path (contributors on different XCDs): diagnostics
__device__ __forceinline__ int conv_group_of(const ConvArgs& a, int m) { return (a.groups > 1 && m >= a.m_per_group) ? 1 : 0; }

// phase-fused launch: the phase-dependent arguments of this workgroup (the kernel argument struct itself stays
// read-only -- writing to it would spill it to scratch)
struct PhaseSel {
  int pad_y, pad_x, ooy, oox, zsplit;
  const float* w;
  float* partial;
};
__device__ __forceinline__ PhaseSel conv_select_phase(const ConvArgs& a, int z, int m_first = 0) {
  PhaseSel s{a.pad_y, a.pad_x, a.ooy, a.oox, z, a.w + (conv_group_of(a, m_first) ? a.wgs : 0L), a.partial};
  if (a.nphase > 1) {
    const int ph = z / a.ksplit;
    const int phy = ph >> 1, phx = ph & 1;
    s.pad_y = -phy; s.pad_x = -phx; s.ooy = phy; s.oox = phx;
    s.w += (size_t)ph * a.wps;
    if (a.partial) s.partial = a.partial + (size_t)ph * a.ksplit * a.B * a.Hp * a.Wp * a.npad;
    s.zsplit = z - ph * a.ksplit;
  }
  return s;
}

// bias + stats + activation + residual + store of 4 consecutive output channels of one pixel
__device__ __forceinline__ void conv_store(const ConvArgs& a, int m, int n, float (&v)[4], float (&ssum)[4], float (&ssq)[4], int ooy,
                                           int oox) {
  const int b = m / (a.Hp * a.Wp), rr = m % (a.Hp * a.Wp);
  const int oy = (rr / a.Wp) * a.ostep + ooy, ox = (rr % a.Wp) * a.ostep + oox;
  const float* bias = a.bias ? a.bias + (conv_group_of(a, m) ? a.Cout : 0) : nullptr;
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] += ((bias && n + r < a.Cout) ? bias[n + r] : 0.f);
  if (a.stats) {
#pragma unroll
    for (int r = 0; r < 4; ++r) { ssum[r] += v[r]; ssq[r] += v[r] * v[r]; }
  }
  apply_act4(v, a.epi_act, a.slope);
  const bool post = a.epi_act == ACT_RELU_POST_RES;      // ReLU after the residual add (apply_act4 leaves this code alone)
  if (a.out_nchw) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (n + r < a.Cout) {
        const size_t o = (((size_t)b * a.Cout + n + r) * a.Hout + oy) * a.Wout + ox;
        const float t = v[r] + (a.res ? a.res[o] : 0.f);
        a.out[o] = post ? fmaxf(t, 0.f) : t;
      }
  } else if (a.pixel_shuffle) {
    // out[b, 2*oy+dy, 2*ox+dx, c] = conv[b, oy, ox, c*4 + dy*2 + dx]; the lane's 4 channels are one c
    const int c = n >> 2, Co = a.Cout >> 2;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int dy = r >> 1, dx = r & 1;
      a.out[(((size_t)b * 2 * a.Hout + 2 * oy + dy) * 2 * a.Wout + 2 * ox + dx) * Co + c] = v[r];
    }
  } else {
    const size_t o = (((size_t)b * a.Hout + oy) * a.Wout + ox) * a.out_ld + a.out_coff + n;
    if (n + 3 < a.Cout) {
      float4 q = make_float4(v[0], v[1], v[2], v[3]);
      if (a.res) {
        const float4 rs = *reinterpret_cast<const float4*>(a.res + o);
        q.x += rs.x; q.y += rs.y; q.z += rs.z; q.w += rs.w;
      }
      if (post) { q.x = fmaxf(q.x, 0.f); q.y = fmaxf(q.y, 0.f); q.z = fmaxf(q.z, 0.f); q.w = fmaxf(q.w, 0.f); }
      *reinterpret_cast<float4*>(a.out + o) = q;
    } else {
      for (int r = 0; r < 4; ++r)
        if (n + r < a.Cout) {
          const float t = v[r] + (a.res ? a.res[o + r] : 0.f);
          a.out[o + r] = post ? fmaxf(t, 0.f) : t;
        }
    }
  }
}

// UNI: cin and every input segment are multiples of 32, so one 32-wide k-chunk lies in ONE segment and ONE tap for the whole
// block.  The chunk is then decoded once, on scalars, and every tile load is a raw buffer load whose hardware range check
// returns 0 for the lanes that fall outside the image (offset 0x80000000) or past Cout -- no predicated loads, i.e. no
// branches whose joins make hipcc drain vmcnt(0) in front of the MFMA block.
// BKT = k-chunk: 32, or 16 for the 128x128 tile (36.9 KB of LDS instead of 73.7: three resident blocks per CU instead of two, the
// same trade the pointwise GEMM makes -- fewer MFMAs per barrier, but a third block to run while two wait)
__device__ __forceinline__ float vmax_raw(float x, float y) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
  return r;
}

// SIMPLE (implies UNI): additionally the input affine on none (AFF = false) or on all (AFF = true: the training forward,
// BatchNorm applied on load) of the segments, input activation in {none, ReLU, LeakyReLU(0.2)} and
// at most 31 taps.  The in-image test of a (pixel row, tap) pair is then precomputed ONCE per block into a per-row tap
// bitmask, the per-chunk address of a row is `voff[row] | bit31-if-outside` (2 vector instructions) with the tap / channel
// part of the address in the buffer load's scalar offset, and the store to LDS needs no validity mask (act(0) = 0).
// BF (with SIMPLE): the operands are rounded to bf16 on the way into LDS (80-byte rows: conflict-free ds_read_b128) and one
// v_mfma_f32_16x16x32_bf16 per tile pair replaces the eight fp32 MFMAs of a 32-deep chunk; accumulation and epilogue stay fp32.
// M32 (with SIMPLE, fp32, 64 x 64 wave tiles): v_mfma_f32_32x32x2_f32 instead of 16x16x4 -- the same FLOPs per cycle and the same
// LDS words per FLOP (a ds_read_b128 still feeds 4 MFMAs: lane (l & 31, l >> 5) holds k = 4 (l >> 5) + s of an 8-deep step), but
// half the MFMA instructions, each with a 64-cycle shadow: the per-chunk vector work (prologue activation, LDS staging) costs
// less matrix time (profiles/r03e_ubench_mfma_valu.txt).  D layout: register v of lane l = out[co = 8 (v / 4) + 4 (l >> 5) +
// v % 4][pixel = l & 31] -- again 4 consecutive output channels per lane and register quad.
typedef float f32x16 __attribute__((ext_vector_type(16)));
// XRED (with SIMPLE, 16x16x4 MFMAs): the split-K reduction inside the launch, through the L2 of ONE XCD.  Workgroups are dealt to
// the 8 XCDs round-robin by linear id; the grid is 1-D and block L = 8 j + c is the j-th block of XCD c, which works on tile
// c * xr_t8 + j / S, k split j % S: all S splits of a tile run on the same XCD, next to each other in time.  A block stores its
// accumulators (register layout) to its slot of the workspace with PLAIN stores -- after s_waitcnt vmcnt(0) they are in that XCD's
// L2 -- and bumps the tile's arrival word; the block that arrives LAST reads all S slots back with device-scope loads (sc1: miss in
// the vector L1, hit in L2), adds them IN SPLIT ORDER (bitwise reproducible whoever arrives last) and runs the epilogue.  No reduce
// launch, no cross-XCD visibility protocol, nothing ever waits on another workgroup.
// The placement is an observed property of the dispatcher, not an architectural guarantee, so it is CHECKED: every block adds
// (1, x, x^2) of its hardware XCC id x to the arrival word; the last block takes the fast path only if all S ids equal its own
// (sum x = S m and sum x^2 = S m^2).  Otherwise it recomputes the S splits itself, in order, with the running sum parked in its
// own slot -- the same additions in the same order, so even a misplaced tile is bitwise equal (counted in g_xred_fallbacks).
// X3 (with SIMPLE, 32-deep chunks; dpmn_set_compute_dtype(2), common.h x3_split2t): both operands are split into three bf16 planes on
// the way into LDS (80-byte rows per plane, the BF layout three times) and a tile pair costs six v_mfma_f32_16x16x32_bf16 per chunk
// instead of eight v_mfma_f32_16x16x4_f32 -- 96 x 16 cycles against 128 x 32 per wave and chunk on the 128 x 128 tile.  The
// accumulator layout is that of the fp32 path, so everything behind the k loop (split-K partials, epilogues, statistics) is shared.
// 128 x 128: ONE LDS buffer (61 KB, two blocks per CU as the fp32 kernel): the split + store of chunk k + 1 sits between two
// barriers while the other resident block multiplies; 64 x 64: both buffers (61 KB).
template <int BM, int BN, int WM, int WN, bool UNI = false, int BKT = 32, bool SIMPLE = false, bool AFF = false, bool BF = false, bool M32 = false,
          bool XRED = false, int X3 = 0>      // X3: 0 off, 1 = 80-byte LDS rows, 2 = 96-byte rows
__device__ __forceinline__ void conv_igemm_body(const ConvArgs& a) {
  static_assert(!X3 || (SIMPLE && !BF && !M32 && !XRED && BKT == 32), "the bf16x3 variant exists for the SIMPLE path with 32-deep chunks");
  static_assert(!XRED || (SIMPLE && !M32 && BKT == 32), "the in-L2 reduction exists for the SIMPLE path");
  static_assert(!M32 || (SIMPLE && !BF && BKT == 32 && BM / WM == 64 && BN / WN == 64), "the 32x32x2 variant: fp32 SIMPLE path, 64 x 64 wave tiles");
  static_assert(!BF || (SIMPLE && BKT == 32), "the bf16 variant exists for the SIMPLE path with 32-deep chunks");
  static_assert(!SIMPLE || UNI, "SIMPLE is a refinement of the UNI path");
  static_assert(!AFF || SIMPLE, "AFF is a variant of the SIMPLE path");
#ifndef DPMN_IGEMM_FENCE
#define DPMN_IGEMM_FENCE 1
#endif
  constexpr bool SCHED_FENCE = DPMN_IGEMM_FENCE;
#ifndef DPMN_IGEMM_ABLATE
#define DPMN_IGEMM_ABLATE 0      // timing experiments only (tools/build_variants.sh): 1 no global loads, 2 no LDS stores, 4 no barrier, 8 no LDS reads
#endif
  constexpr int ABL = SIMPLE ? DPMN_IGEMM_ABLATE : 0;
  constexpr int MT = BM / WM / 16, NT = BN / WN / 16;
  constexpr int TPR = BKT / 4, RPP = 256 / TPR;          // threads per tile row, tile rows per pass
  constexpr int APASS = BM / RPP, BPASS = (BN + RPP - 1) / RPP;
  constexpr int BK = BKT, LDK = BKT + PAD;
  // bf16 row stride.  BF: 32 + 8 elements = 80 bytes.  X3: 32 + 16 = 96 bytes -- a ds_read_b128 is served in groups of 16 lanes
  // ({0-3, 12-15, 20-27}, ...: MI355X_MICROARCH.md, LDS): with 80-byte rows three of a group's sixteen 16-byte reads share a bank
  // quad with another lane (quad = 5 lr + kq mod 16) and every group takes two LDS cycles -- SQ_LDS_BANK_CONFLICT = 50 % of the
  // active LDS cycles (profiles/r06_pmc_sq.txt) -- while 96-byte rows (quad = 6 lr + kq mod 16) give sixteen different quads for
  // any row base.  The x3 kernels move 1.5 x the LDS bytes of the fp32 kernel in 0.375 x its MFMA time: the conflicts are not free there.
  constexpr int LDKB = X3 == 2 ? BKT + 16 : BKT + 8;
  constexpr bool B16 = BF || X3;                              // bf16 rows in LDS
  constexpr int NPL = X3 ? 3 : 1;                             // operand planes
  constexpr int NBUF = (X3 && BM * BN > 64 * 64) ? 1 : 2;
  constexpr int XPL = BM * LDKB, WPL = BN * LDKB;             // one bf16 plane (halves)
  typedef typename std::conditional<B16, unsigned short, float>::type lds_t;
  __shared__ __attribute__((aligned(16))) lds_t Xs[NBUF][NPL * BM * (B16 ? LDKB : LDK)];
  __shared__ __attribute__((aligned(16))) lds_t Ws[NBUF][NPL * BN * (B16 ? LDKB : LDK)];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int M = a.B * a.Hp * a.Wp;
  // Workgroups are dealt to the 8 XCDs round-robin by linear id, each XCD with its own L2.  With the natural order the row tiles
  // that share one (n tile, k split) weight slice land on 8 different XCDs and every one of them pulls the slice from HBM: the
  // deep CMM levels (12-48 row tiles against 17-28 MB of weights) fetched 4x their compulsory bytes.  wlocal: XCD c owns the
  // slices nz = c, c + 8, ... and walks their row tiles back to back.  nz counts the k split FASTEST (nz = z + Z n): the n tiles
  // of one k split read the same input slice, and with Z a multiple of 8 they all sit on one XCD (in general on
  // min(n tiles, 8 / gcd(Z, 8)) of them).
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  int xr_tile = 0;
  if constexpr (XRED) {
    const int L = blockIdx.x, c = L & 7, j = L >> 3;
    const int tl = j / a.ksplit, z = j - tl * a.ksplit;
    const int nph_ = a.nphase > 1 ? a.nphase : 1;
    xr_tile = c * a.xr_t8 + tl;
    if (tl >= a.xr_t8 || xr_tile >= a.xr_tm * a.xr_tn * nph_) return;
    int phs;
    if (a.xr_order == 0) { bx = xr_tile % a.xr_tm; const int r = xr_tile / a.xr_tm; by = r % a.xr_tn; phs = r / a.xr_tn; }
    else { by = xr_tile % a.xr_tn; const int r = xr_tile / a.xr_tn; phs = r % nph_; bx = r / nph_; }
    bz = phs * a.ksplit + z;
  } else
  if (a.wlocal) {
    const int L = bx + gridDim.x * (by + gridDim.y * bz);
    const int c = L & 7, j = L >> 3;
    const int jm = j / (int)gridDim.x;
    bx = j - jm * (int)gridDim.x;
    const int nz = c + 8 * jm;
    by = nz / (int)gridDim.z;
    bz = nz - by * (int)gridDim.z;
  }
  const int m_blk = bx * BM, n_blk = by * BN;
  const PhaseSel ph = conv_select_phase(a, bz, m_blk);      // (a tile never straddles the two groups: checked at launch)
  const int zsplit = ph.zsplit;
  const int lrow = tid / TPR, lcol = (tid % TPR) * 4;

  // per-thread pixel rows of the A tile
  // (m -> image, row, column through one float multiply and a +-1 fix-up each, exact below 2^24 pixels -- checked at launch:
  //  eight 32-bit integer divisions per thread were a third of the block prologue, paid by every short split-K block)
  int pb[APASS], py[APASS], px[APASS];
  const int HWp = a.Hp * a.Wp;
  const float inv_hw = 1.0f / (float)HWp, inv_w = 1.0f / (float)a.Wp;
#pragma unroll
  for (int p = 0; p < APASS; ++p) {
    const int m = m_blk + lrow + p * RPP;
    if (m < M) {
      int b = (int)((float)m * inv_hw), r = m - b * HWp;
      if (r < 0) { --b; r += HWp; }
      if (r >= HWp) { ++b; r -= HWp; }
      int y = (int)((float)r * inv_w), x = r - y * a.Wp;
      if (x < 0) { --y; x += a.Wp; }
      if (x >= a.Wp) { ++y; x -= a.Wp; }
      pb[p] = b;
      py[p] = y * a.stride - ph.pad_y;
      px[p] = x * a.stride - ph.pad_x;
    } else {
      pb[p] = -1; py[p] = 0; px[p] = 0;
    }
  }
  const int c01 = a.cseg[0] + a.cseg[1];
  const int ktaps = a.KH * a.KW;

  // gload only ISSUES the global loads (raw values + the 2 affine vectors); the on-load transform -- BatchNorm affine and
  // input activation, which must leave out-of-image taps at exactly 0 -- runs in sstore, after the MFMAs of the current
  // chunk, so the loads are in flight during the matrix work instead of being waited for one by one.
  // (Tried and measured slower: clamped always-valid addresses instead of the predicated loads, per-tile
  // amdgpu_waves_per_eu caps.)
  float4 xr[APASS], wr[BPASS];
  float4 s4r, h4r;
  unsigned vmask = 0;        // bit p: xr[p] came from inside the image
  bool has_aff = false;
  auto gload = [&](int k0) {
    const int k = k0 + lcol;
    const int tap = k / a.cin, c = k - tap * a.cin;
    const int ky = tap / a.KW, kx = tap - ky * a.KW;
    const int dy = ky * a.dil_y, dx = kx * a.dil_x;
    // segment of channel c
    int seg = 0, cl = c;
    if (c >= c01) { seg = 2; cl = c - c01; }
    else if (c >= a.cseg[0]) { seg = 1; cl = c - a.cseg[0]; }
    const float* src = a.in[seg];
    const int cs = a.cseg[seg];
    const float* sc = a.in_scale[seg];
    const float* sh = a.in_shift[seg];
    const bool tap_ok = tap < ktaps;
    has_aff = sc != nullptr && tap_ok;
    if (has_aff) { s4r = *reinterpret_cast<const float4*>(sc + cl); h4r = *reinterpret_cast<const float4*>(sh + cl); }
    vmask = 0;
#pragma unroll
    for (int p = 0; p < APASS; ++p) {
      const int iy = py[p] + dy, ix = px[p] + dx;
      const bool ok = pb[p] >= 0 && tap_ok && iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win;
      xr[p] = ok ? *reinterpret_cast<const float4*>(src + (((size_t)pb[p] * a.Hin + iy) * a.Win + ix) * cs + cl)
                 : make_float4(0.f, 0.f, 0.f, 0.f);
      vmask |= (ok ? 1u : 0u) << p;
    }
#pragma unroll
    for (int p = 0; p < BPASS; ++p) {
      const int r = lrow + p * RPP;
      const int n = n_blk + r;
      wr[p] = (r < BN && n < a.Cout) ? *reinterpret_cast<const float4*>(ph.w + (size_t)n * a.Kp + k0 + lcol)
                                     : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  typedef int i32x4_ __attribute__((ext_vector_type(4)));
  // UNI path: everything that does not depend on the chunk is hoisted -- per pass the pixel's linear index and a row/column
  // pair in which rows beyond M are parked far outside the image (their range test then fails like an out-of-image tap);
  // per chunk the tap / segment decode runs on scalars, and a pass costs two adds, two unsigned compares, one 24-bit
  // multiply-add and the buffer load (the matrix pipe shares its issue port with the vector ALU: address arithmetic in
  // the loop is paid in MFMA time).
  int pix[APASS];
#pragma unroll
  for (int p = 0; p < APASS; ++p) {
    pix[p] = pb[p] >= 0 ? (pb[p] * a.Hin + py[p]) * a.Win + px[p] : 0;
    if (pb[p] < 0) { py[p] = -(1 << 20); px[p] = -(1 << 20); }
  }
  // chunk decode state (scalars): advanced by one 32-channel chunk per call instead of two integer divisions per chunk
  int u_tap = 0, u_c0 = 0, u_ky = 0, u_kx = 0, u_k0 = -2;      // (-2: the first call always decodes)
  int wofs[BPASS];
#pragma unroll
  for (int p = 0; p < BPASS; ++p) wofs[p] = ((n_blk + lrow + p * RPP) * a.Kp + lcol) * 4;      // + k0 * 4 through the scalar offset
  auto gload_uni = [&](int k0) {
    if (k0 == u_k0 + BK) {                 // the next chunk (the common case)
      u_c0 += BK;
      if (u_c0 >= a.cin) { u_c0 = 0; ++u_tap; if (++u_kx == a.KW) { u_kx = 0; ++u_ky; } }
    } else if (k0 != u_k0) {               // first chunk of a split (or the clamped re-read of the last one: unchanged)
      u_tap = k0 / a.cin; u_c0 = k0 - u_tap * a.cin;
      u_ky = u_tap / a.KW; u_kx = u_tap - u_ky * a.KW;
    }
    u_k0 = k0;
    const int tap = u_tap, c0 = u_c0, ky = u_ky, kx = u_kx;      // wave-uniform
    const bool tap_ok = tap < ktaps;
    const int dy = tap_ok ? ky * a.dil_y : (1 << 21), dx = kx * a.dil_x;      // K-padding chunk: every range test fails
    const int seg = c0 >= c01 ? 2 : (c0 >= a.cseg[0] ? 1 : 0);
    const int cl = c0 - (seg == 2 ? c01 : (seg == 1 ? a.cseg[0] : 0)) + lcol;
    const float* src = a.in[seg];
    const int cs = a.cseg[seg];
    const float* sc = a.in_scale[seg];
    const float* sh = a.in_shift[seg];
    has_aff = sc != nullptr && tap_ok;
    if (has_aff) { s4r = *reinterpret_cast<const float4*>(sc + cl); h4r = *reinterpret_cast<const float4*>(sh + cl); }
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, a.B * a.Hin * a.Win * cs * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ph.w), 0, a.Cout * a.Kp * 4, 0x00020000);
    const int tapoff = dy * a.Win + dx;                          // scalar
    vmask = 0;
#pragma unroll
    for (int p = 0; p < APASS; ++p) {
      const bool ok = (unsigned)(py[p] + dy) < (unsigned)a.Hin && (unsigned)(px[p] + dx) < (unsigned)a.Win;
      // branch-free: bit 31 set = beyond num_records (< 2^31, checked by the launcher) whatever the low bits are
      const unsigned off = (unsigned)((__mul24(pix[p] + tapoff, cs) + cl) * 4) | (ok ? 0u : 0x80000000u);
      xr[p] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)off, 0, 0));
      vmask |= (ok ? 1u : 0u) << p;
    }
#pragma unroll
    for (int p = 0; p < BPASS; ++p) {
      const int r = lrow + p * RPP;
      if (BN % RPP == 0 || r < BN)       // rows past Cout are beyond num_records: the range check returns 0
        wr[p] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(wrs, wofs[p], k0 * 4, 0));
    }
  };
  // ---- SIMPLE path state
  unsigned nok[APASS];       // bit t: tap t of this pixel row reads outside the image (or the row is beyond M)
  int voff[APASS];           // byte offset of (pixel row, channel lcol) from the pad-shifted base, for the current segment
  int s_seg = -1, s_segstart = 0;
  __amdgpu_buffer_rsrc_t s_xrs, s_scrs, s_shrs;
  unsigned inv[APASS];       // AFF: bit 31 set = this chunk's tap is outside the image for the row (the affine must leave 0 there)
  // address = base + (pix + tapoff) * cs * 4 is split into a per-row vector part (pix + padoff >= 0) and a per-chunk scalar
  // part (tapoff - minoff >= 0; minoff < 0 for the reversed taps, dil -1, of the transposed-conv phases) over a base
  // shifted down by (padoff - minoff) pixels -- never dereferenced there, valid lanes land inside the tensor
  const int padoff = ph.pad_y * a.Win + ph.pad_x;
  const int minoff = (a.dil_y < 0 ? (a.KH - 1) * a.dil_y : 0) * a.Win + (a.dil_x < 0 ? (a.KW - 1) * a.dil_x : 0);
  const int baseshift = padoff - minoff;
  if (SIMPLE) {
#pragma unroll
    for (int p = 0; p < APASS; ++p) {
      unsigned colm = 0, okb = 0;
      for (int kx = 0; kx < a.KW; ++kx) colm |= ((unsigned)(px[p] + kx * a.dil_x) < (unsigned)a.Win ? 1u : 0u) << kx;
      for (int ky = 0; ky < a.KH; ++ky)
        if ((unsigned)(py[p] + ky * a.dil_y) < (unsigned)a.Hin) okb |= colm << (ky * a.KW);
      nok[p] = ~okb;                                          // parked rows fail every test; bits >= KH*KW stay set
    }
  }
  const __amdgpu_buffer_rsrc_t s_wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ph.w), 0, a.Cout * a.Kp * 4, 0x00020000);
  // The chunks are VISITED channel-chunk-major, tap-minor (step kt = cchunk * taps + tap; the packed layout stays tap-major, the
  // weight column of a step is k0 = tap * cin + cchunk * 32): the taps of one 32-channel chunk re-read the same pixels a few
  // steps apart, while they are still in L2.  In tap-major order a block swept ALL channels of a tap (393 KB per block on the
  // 768-channel decoder convs, x 64 resident blocks per XCD against 4 MB of L2) before coming back to the same pixels for the
  // next tap, and the input went over the fabric once per tap.  A k split is a contiguous range of steps = a channel range.
  auto gload_simple = [&](int kt_) {
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
    if (seg != s_seg) {                                       // wave-uniform, once per segment change
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
    // scalar, >= 0 (readfirstlane: keeps it in an SGPR -- a VGPR soffset makes hipcc emit a waterfall loop around every load)
    const int soff = __builtin_amdgcn_readfirstlane(((u_ky * a.dil_y * a.Win + u_kx * a.dil_x - minoff) * cs + u_c0 - s_segstart) * 4);
    const int sh = 31 - min(u_tap, 31);
    if (ABL & 1) {
#pragma unroll
      for (int p = 0; p < APASS; ++p) xr[p] = make_float4(1.f, 1.f, 1.f, (float)(soff + sh));
#pragma unroll
      for (int p = 0; p < BPASS; ++p) wr[p] = make_float4(1.f, 1.f, 1.f, 1.f);
      return;
    }
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
    for (int p = 0; p < BPASS; ++p) {
      const int r = lrow + p * RPP;
      if (BN % RPP == 0 || r < BN)
        wr[p] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(s_wrs, wofs[p], k0 * 4, 0));
    }
  };
  auto sstore_simple = [&](int buf) {
    if (AFF) {          // same expression as the general path (mul, then add: -ffp-contract=off)
#pragma unroll
      for (int p = 0; p < APASS; ++p) {
        xr[p].x = xr[p].x * s4r.x + h4r.x; xr[p].y = xr[p].y * s4r.y + h4r.y;
        xr[p].z = xr[p].z * s4r.z + h4r.z; xr[p].w = xr[p].w * s4r.w + h4r.w;
      }
    }
    // one v_max per element (fmaxf would first canonicalise both operands); ReLU = slope 0, none = skipped
    if (a.pro_act != ACT_NONE) {
      const float sl = a.pro_act == ACT_LEAKY02 ? 0.2f : 0.0f;
#pragma unroll
      for (int p = 0; p < APASS; ++p) {
        xr[p].x = vmax_raw(xr[p].x, sl * xr[p].x); xr[p].y = vmax_raw(xr[p].y, sl * xr[p].y);
        xr[p].z = vmax_raw(xr[p].z, sl * xr[p].z); xr[p].w = vmax_raw(xr[p].w, sl * xr[p].w);
      }
    }
    if ((ABL & 2) && xr[0].w != 12345.f) return;
    if (AFF) {          // act(shift) is not 0: out-of-image taps are zeroed explicitly
#pragma unroll
      for (int p = 0; p < APASS; ++p)
        if (inv[p]) xr[p] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if constexpr (X3) {
#pragma unroll
      for (int p = 0; p < APASS; ++p) {
        uint2 h, m, l;
        x3_split4t(xr[p], h, m, l);
        lds_t* d_ = &Xs[buf][(lrow + p * RPP) * LDKB + lcol];
        *reinterpret_cast<uint2*>(d_) = h;
        *reinterpret_cast<uint2*>(d_ + XPL) = m;
        *reinterpret_cast<uint2*>(d_ + 2 * XPL) = l;
      }
#pragma unroll
      for (int p = 0; p < BPASS; ++p)
        if (BN % RPP == 0 || lrow + p * RPP < BN) {
          uint2 h, m, l;
          x3_split4t(wr[p], h, m, l);
          lds_t* d_ = &Ws[buf][(lrow + p * RPP) * LDKB + lcol];
          *reinterpret_cast<uint2*>(d_) = h;
          *reinterpret_cast<uint2*>(d_ + WPL) = m;
          *reinterpret_cast<uint2*>(d_ + 2 * WPL) = l;
        }
    } else if constexpr (BF) {
#pragma unroll
      for (int p = 0; p < APASS; ++p)
        *reinterpret_cast<uint2*>(&Xs[buf][(lrow + p * RPP) * LDKB + lcol]) = pack_bf16x4(xr[p].x, xr[p].y, xr[p].z, xr[p].w);
#pragma unroll
      for (int p = 0; p < BPASS; ++p)
        if (BN % RPP == 0 || lrow + p * RPP < BN)
          *reinterpret_cast<uint2*>(&Ws[buf][(lrow + p * RPP) * LDKB + lcol]) = pack_bf16x4(wr[p].x, wr[p].y, wr[p].z, wr[p].w);
    } else {
#pragma unroll
      for (int p = 0; p < APASS; ++p) *reinterpret_cast<float4*>(&Xs[buf][(lrow + p * RPP) * LDK + lcol]) = xr[p];
#pragma unroll
      for (int p = 0; p < BPASS; ++p)
        if (BN % RPP == 0 || lrow + p * RPP < BN) *reinterpret_cast<float4*>(&Ws[buf][(lrow + p * RPP) * LDK + lcol]) = wr[p];
    }
  };
  auto sstore = [&](int buf) {
    if constexpr (!B16) {
#pragma unroll
    for (int p = 0; p < APASS; ++p) {
      float4 v = xr[p];
      if (!has_aff && (a.pro_act == ACT_LEAKY02 || a.pro_act == ACT_RELU)) {
        // the usual case (eval: BatchNorm folded into the producer): out-of-image lanes already hold 0 and act(0) = 0, so no
        // mask; LeakyReLU(0.2) = max(x, 0.2 x), ReLU = max(x, 0)
        const float sl = a.pro_act == ACT_LEAKY02 ? 0.2f : 0.0f;
        v.x = fmaxf(v.x, sl * v.x); v.y = fmaxf(v.y, sl * v.y); v.z = fmaxf(v.z, sl * v.z); v.w = fmaxf(v.w, sl * v.w);
      } else if ((vmask >> p) & 1u) {
        if (has_aff) { v.x = v.x * s4r.x + h4r.x; v.y = v.y * s4r.y + h4r.y; v.z = v.z * s4r.z + h4r.z; v.w = v.w * s4r.w + h4r.w; }
        if (a.pro_act != ACT_NONE) {
          v.x = apply_act(v.x, a.pro_act, 0.f); v.y = apply_act(v.y, a.pro_act, 0.f);
          v.z = apply_act(v.z, a.pro_act, 0.f); v.w = apply_act(v.w, a.pro_act, 0.f);
        }
      }
      *reinterpret_cast<float4*>(&Xs[buf][(lrow + p * RPP) * LDK + lcol]) = v;
    }
#pragma unroll
    for (int p = 0; p < BPASS; ++p)
      if (lrow + p * RPP < BN) *reinterpret_cast<float4*>(&Ws[buf][(lrow + p * RPP) * LDK + lcol]) = wr[p];
    }
  };

  const int wm = wave % WM, wn = wave / WM;
  const int lr = lane & 15, kq = lane >> 4;
  f32x4 acc[NT][MT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x16 acc32[2][2];
  if constexpr (M32) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc32[i][j][v] = 0.f;
  }
  const int l31 = lane & 31, lh = lane >> 5;

  const int nk_all = a.Kp / BK;
  const int cps = (nk_all + a.ksplit - 1) / a.ksplit;          // chunks per split
  // XRED: the pass below runs once for this block's own split; a last-arriving block whose contributors were NOT all on its XCD
  // runs it again for every split (redo), see the hand-over behind the loop
  int zcur = zsplit;
  int mode = 0;                // 0: this block's own split; 1 / 2: recomputing its group / every split (XRED)
  for (;;) {
  const int kt0 = zcur * cps;
  const int nk = min(nk_all, kt0 + cps);
  if constexpr (XRED) {
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int j = 0; j < MT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  if (kt0 < nk) {
    if (SIMPLE) { gload_simple(kt0); sstore_simple(0); }
    else {
      if (UNI) gload_uni(kt0 * BK); else gload(kt0 * BK);
      sstore(0);
    }
  }
  __syncthreads();
#ifndef DPMN_X3_DBG
#define DPMN_X3_DBG 0      // debugging builds (tools/build_variants.sh): 1 = an extra barrier + LDS fence at the top of every chunk, 4 = no MFMAs behind the barrier
#endif
  for (int kt = kt0; kt < nk; ++kt) {
    const int buf = NBUF == 2 ? (kt - kt0) & 1 : 0;
    if (X3 && (DPMN_X3_DBG & 1)) { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __syncthreads(); }
    if (SIMPLE) gload_simple(min(kt + 1, nk - 1));
    else if (UNI) gload_uni(min(kt + 1, nk - 1) * BK);      // unconditional: the refill past the end re-reads the last chunk
    else if (kt + 1 < nk) gload((kt + 1) * BK);
    // hipcc otherwise sinks the buffer loads deep into the MFMA block (the last ones ~100 MFMAs down): they must be in
    // flight for the WHOLE block to cover HBM / L2 latency before sstore waits for them
    if (UNI && SCHED_FENCE) __builtin_amdgcn_sched_barrier(0);
    if constexpr (X3) {
      const lds_t* xa = &Xs[buf][(wm * (MT * 16) + lr) * LDKB + kq * 8];
      const lds_t* wa = &Ws[buf][(wn * (NT * 16) + lr) * LDKB + kq * 8];
      bf16x8 w0[NT], w1[NT], w2[NT], xp[MT], xq[MT];
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        w0[i] = *reinterpret_cast<const bf16x8*>(wa + i * 16 * LDKB);
        w1[i] = *reinterpret_cast<const bf16x8*>(wa + WPL + i * 16 * LDKB);
        w2[i] = *reinterpret_cast<const bf16x8*>(wa + 2 * WPL + i * 16 * LDKB);
      }
#pragma unroll
      for (int j = 0; j < MT; ++j) xp[j] = *reinterpret_cast<const bf16x8*>(xa + j * 16 * LDKB);
#pragma unroll
      for (int j = 0; j < MT; ++j) xq[j] = *reinterpret_cast<const bf16x8*>(xa + XPL + j * 16 * LDKB);
      // six terms: the high plane of x against all three weight planes (smallest first), then the middle, then the low plane
#define IG_X3_TERM(XF, WF)                                                  \
      _Pragma("unroll") for (int i = 0; i < NT; ++i)                        \
        _Pragma("unroll") for (int j = 0; j < MT; ++j) acc[i][j] = mfma16_bf16(WF[i], XF[j], acc[i][j]);
      IG_X3_TERM(xp, w2) IG_X3_TERM(xp, w1) IG_X3_TERM(xp, w0)
#pragma unroll
      for (int j = 0; j < MT; ++j) xp[j] = *reinterpret_cast<const bf16x8*>(xa + 2 * XPL + j * 16 * LDKB);
      IG_X3_TERM(xq, w1) IG_X3_TERM(xq, w0)
      IG_X3_TERM(xp, w0)
#undef IG_X3_TERM
    } else if constexpr (BF) {
      const lds_t* xa = &Xs[buf][(wm * (MT * 16) + lr) * LDKB + kq * 8];
      const lds_t* wa = &Ws[buf][(wn * (NT * 16) + lr) * LDKB + kq * 8];
      bf16x8 xf[MT], wf[NT];
#pragma unroll
      for (int j = 0; j < MT; ++j) xf[j] = *reinterpret_cast<const bf16x8*>(xa + j * 16 * LDKB);
#pragma unroll
      for (int i = 0; i < NT; ++i) wf[i] = *reinterpret_cast<const bf16x8*>(wa + i * 16 * LDKB);
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = mfma16_bf16(wf[i], xf[j], acc[i][j]);
    } else if constexpr (M32) {
      const float* xa = reinterpret_cast<const float*>(&Xs[buf][0]) + (wm * 64 + l31) * LDK + lh * 4;
      const float* wa = reinterpret_cast<const float*>(&Ws[buf][0]) + (wn * 64 + l31) * LDK + lh * 4;
#pragma unroll
      for (int kc = 0; kc < BK; kc += 8) {
        f32x4 xf[2], wf[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) xf[j] = *reinterpret_cast<const f32x4*>(xa + j * 32 * LDK + kc);
#pragma unroll
        for (int i = 0; i < 2; ++i) wf[i] = *reinterpret_cast<const f32x4*>(wa + i * 32 * LDK + kc);
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc32[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[i][s_], xf[j][s_], acc32[i][j], 0, 0, 0);
      }
    } else {
    const float* xa = reinterpret_cast<const float*>(&Xs[buf][0]) + (wm * (MT * 16) + lr) * LDK + kq * 4;
    const float* wa = reinterpret_cast<const float*>(&Ws[buf][0]) + (wn * (NT * 16) + lr) * LDK + kq * 4;
#pragma unroll
    for (int kc = 0; kc < BK; kc += 16) {
      f32x4 xf[MT], wf[NT];
#pragma unroll
      for (int j = 0; j < MT; ++j)
        if (ABL & 8) xf[j] = (f32x4){1.f, (float)kt, 1.f, 1.f}; else xf[j] = *reinterpret_cast<const f32x4*>(xa + j * 16 * LDK + kc);
#pragma unroll
      for (int i = 0; i < NT; ++i)
        if (ABL & 8) wf[i] = (f32x4){1.f, (float)kt, 1.f, 1.f}; else wf[i] = *reinterpret_cast<const f32x4*>(wa + i * 16 * LDK + kc);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
          for (int j = 0; j < MT; ++j) acc[i][j] = mfma16(wf[i][s], xf[j][s], acc[i][j]);
    }
    }
    if (X3 && (DPMN_X3_DBG & 4)) __builtin_amdgcn_sched_barrier(0);
    if (NBUF == 1) __syncthreads();                           // single buffer: every wave has read chunk kt before it is overwritten
    if (X3 && (DPMN_X3_DBG & 4)) __builtin_amdgcn_sched_barrier(0);
    if (SIMPLE) sstore_simple(NBUF == 2 ? buf ^ 1 : 0);
    else if (UNI || kt + 1 < nk) sstore(buf ^ 1);
    if (!(ABL & 4)) __syncthreads();
  }
  if constexpr (!XRED) break;
  else {
    if (a.ksplit <= 1) break;
    constexpr int QN = NT * MT, HQ = QN / 2;
    static_assert(QN % 2 == 0, "the collect moves half slots");
    typedef int i32x4x_ __attribute__((ext_vector_type(4)));
    const int S = a.ksplit, GS = a.xr_group, ngrp = (S + GS - 1) / GS;
    const int g_lo = (zsplit / GS) * GS, g_hi = min(S, g_lo + GS);
    const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(a.partial, 0, a.xr_tm * a.xr_tn * (a.nphase > 1 ? a.nphase : 1) * S * (QN * 4096), 0x00020000);
    constexpr int SC1 = 0x10;               // cache policy of the loads: device scope (miss in the vector L1, served by this XCD's L2)
    auto slot_of = [&](int z_) { return __builtin_amdgcn_readfirstlane((xr_tile * S + z_) * (QN * 4096)); };
    auto st_slot = [&](int z_) {            // accumulators -> slot z_ of this tile, complete (in L2) on return
      const int so = slot_of(z_);
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4x_, acc[i][j]), prs, tid * 16 + (i * MT + j) * 4096, so, 0);
      if (!(a.xr_ablate & 2)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    auto add_slot = [&](int z_) {           // accumulators = slot z_ + accumulators
      const int so = slot_of(z_);
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) {
          const f32x4 t_ = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(prs, tid * 16 + (i * MT + j) * 4096, so, SC1));
          acc[i][j] = t_ + acc[i][j];
        }
    };
    int* s_flag = reinterpret_cast<int*>(&Xs[0][0]);      // (the tiles are dead: every wave passed the loop's last barrier)
    // arrival at a word that expects `target` workgroups.  Returns 0: others still to come (this block is done); 1: last, and
    // every contributor ran on this block's XCD (their partial tiles are in the L2 this block reads); 2: last, but not so
    auto arrive = [&](unsigned* w, int target) -> int {
      __syncthreads();                      // every wave's stores are complete (st_slot waited)
      if (tid == 0 && (a.xr_ablate & 8)) s_flag[0] = (zcur == (target == ngrp ? S - 1 : g_hi - 1)) ? 1 : 0;      // timing only: no atomic
      else if (tid == 0) {
        const unsigned x = (a.xr_ablate & 16) ? 0u : (__builtin_amdgcn_s_getreg(20 | (3 << 11)) & 15u);      // hwreg(HW_REG_XCC_ID, 0, 4)
        const unsigned add = 1u | (x << 7) | ((x * x) << 16);
        const unsigned old = (a.xr_ablate & 32) ? __hip_atomic_fetch_add(w, add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
                                                : __hip_atomic_fetch_add(w, add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int f = 0;
        if ((old & 127u) == (unsigned)(target - 1)) {
          __hip_atomic_store(w, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // nobody touches it before the next launch
          const unsigned sx = ((old >> 7) & 511u) + x, sxx = (old >> 16) + x * x;   // sum x = n m and sum x^2 = n m^2 <=> all x = m
          f = (sx == (unsigned)target * x && sxx == (unsigned)target * x * x && !a.xr_force_redo) ? 1 : 2;
          if (f == 2) atomicAdd(&g_xred_fallbacks, 1u);
        }
        s_flag[0] = f;
      }
      __syncthreads();
      const int f = s_flag[0];
      __syncthreads();
      return f;
    };
    // accumulators = slot z0 + slot (z0 + step) + ... (n slots, in this order, from zero); half slots in flight two deep
    auto collect = [&](int z0, int n, int step) {
      f32x4 t0[HQ], t1[HQ];
      auto ldhalf = [&](f32x4 (&t_)[HQ], int u_) {      // unit u = 2 i + h of slot z0 + i step
        const int uu = min(u_, 2 * n - 1);
        const int so = __builtin_amdgcn_readfirstlane((xr_tile * S + z0 + (uu >> 1) * step) * (QN * 4096) + (uu & 1) * (HQ * 4096));
#pragma unroll
        for (int q = 0; q < HQ; ++q) t_[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(prs, tid * 16 + q * 4096, so, SC1));
      };
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      ldhalf(t0, 0);
      for (int i_ = 0; i_ < n; ++i_) {
        ldhalf(t1, 2 * i_ + 1);
#pragma unroll
        for (int q = 0; q < HQ; ++q) acc[q / MT][q % MT] += t0[q];
        ldhalf(t0, 2 * i_ + 2);
#pragma unroll
        for (int q = 0; q < HQ; ++q) acc[(HQ + q) / MT][(HQ + q) % MT] += t1[q];
      }
    };
    // Two levels: the S splits are cut into groups of GS consecutive ones.  The last block of a GROUP adds the group's partial
    // tiles (in split order); with more than one group it stores the group sum over the group's first slot and arrives at the
    // tile's word, where the last group adds the group sums in group order -- the groups are collected by different CUs in
    // parallel (one CU reads a 64 KB slot in ~0.5 us: 32 splits in one chain cost 17 us at the end of the launch).
    // Word layout (stride xr_cstride words each: atomics on one line serialise): tile * (ngrp + 1) + group, the tile's word last.
    unsigned* words = a.xr_cnt + (size_t)xr_tile * (ngrp + 1) * a.xr_cstride;
    bool to_l2 = false;
    if (mode == 0) {
      st_slot(zsplit);
      if (a.xr_ablate & 4) { if (zsplit != S - 1) return; break; }
      const int f = arrive(words + (zsplit / GS) * a.xr_cstride, g_hi - g_lo);
      if (f == 0) return;
      if (f == 2) { mode = 1; zcur = g_lo; continue; }
      if (!(a.xr_ablate & 1)) collect(g_lo, g_hi - g_lo, 1);
      to_l2 = true;
    } else {
      // recompute (a contributor ran on another XCD): acc = split zcur; the same additions in the same order as the collect
      // path, with the running group sum parked in slot B and (mode 2) the running sum of the groups in slot A -- every slot of
      // the tile is dead by now, and a thread reads back only words it wrote itself
      const int zlo = mode == 1 ? g_lo : (zcur / GS) * GS, zhi = min(S, zlo + GS);
      const int slotB = mode == 1 ? zsplit : 1, slotA = 0;
      if (zcur > zlo) add_slot(slotB);
      if (zcur + 1 < zhi) { st_slot(slotB); ++zcur; continue; }
      if (mode == 1) to_l2 = true;
      else {
        if (zlo > 0) add_slot(slotA);
        if (zhi < S) { st_slot(slotA); zcur = zhi; continue; }
        break;
      }
    }
    if (to_l2) {
      if (ngrp == 1) break;
      st_slot(g_lo);                         // the group's first slot now holds the group sum (only this block read the group's slots)
      const int f = arrive(words + ngrp * a.xr_cstride, ngrp);
      if (f == 0) return;
      if (f == 2) { mode = 2; zcur = 0; continue; }
      if (!(a.xr_ablate & 1)) collect(0, ngrp, GS);
    }
    break;
  }
  }

  if constexpr (M32) {
    // lane holds out[pixel m = .. + j * 32 + (l & 31)][co = .. + i * 32 + 8 g + 4 (l >> 5) + r], r = register 4 g + r of tile (i, j)
    float ssum[8][4], ssq[8][4];
#pragma unroll
    for (int q = 0; q < 8; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) { ssum[q][r] = 0.f; ssq[q][r] = 0.f; }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int m = m_blk + wm * 64 + j * 32 + l31;
      if (m >= M) continue;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = n_blk + wn * 64 + i * 32 + g * 8 + lh * 4;
          if (a.ksplit > 1) {
            if (n < a.npad)
              *reinterpret_cast<float4*>(ph.partial + ((size_t)zsplit * M + m) * a.npad + n) =
                  make_float4(acc32[i][j][4 * g], acc32[i][j][4 * g + 1], acc32[i][j][4 * g + 2], acc32[i][j][4 * g + 3]);
          } else if (n < a.Cout) {
            float v[4] = {acc32[i][j][4 * g], acc32[i][j][4 * g + 1], acc32[i][j][4 * g + 2], acc32[i][j][4 * g + 3]};
            conv_store(a, m, n, v, ssum[i * 4 + g], ssq[i * 4 + g], ph.ooy, ph.oox);
          }
        }
    }
    if (a.stats && a.ksplit <= 1) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int n = n_blk + wn * 64 + (q >> 2) * 32 + (q & 3) * 8 + lh * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float s_ = ssum[q][r], q_ = ssq[q][r];
#pragma unroll
          for (int o = 1; o < 32; o <<= 1) { s_ += xshfl_v(s_, o); q_ += xshfl_v(q_, o); }
          if (l31 == 0 && n + r < a.Cout) {
            double* st = reinterpret_cast<double*>(a.stats) + (size_t)(bx % STAT_SLOTS) * 2 * a.Cout;
            atomicAdd(st + n + r, (double)(s_));
            atomicAdd(st + a.Cout + n + r, (double)(q_));
          }
        }
      }
    }
    return;
  }
  // ---- epilogue: lane holds out[pixel m = .. + (l&15)][co = .. + (l>>4)*4 + r]
  if (!XRED && a.ksplit > 1) {
#pragma unroll
    for (int j = 0; j < MT; ++j) {
      const int m = m_blk + wm * (MT * 16) + j * 16 + lr;
      if (m >= M) continue;
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        const int n = n_blk + wn * (NT * 16) + i * 16 + kq * 4;
        if (n >= a.npad) continue;
        *reinterpret_cast<float4*>(ph.partial + ((size_t)zsplit * M + m) * a.npad + n) =
            make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
      }
    }
    return;
  }
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
      conv_store(a, m, n, v, ssum[i], ssq[i], ph.ooy, ph.oox);
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
          double* st = reinterpret_cast<double*>(a.stats) + (size_t)(bx % STAT_SLOTS) * 2 * a.Cout;   // slotted: spreads same-address atomics
          atomicAdd(st + n + r, (double)(s));
          atomicAdd(st + a.Cout + n + r, (double)(q));
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------- halo-tile direct conv
// Stride-1 "same" KxK convs (3x3 of the CMM / PSN trunks, 9x9 output conv) with the INPUT tile resident in LDS:
// a block owns 8x16 output pixels of one image x BN output channels.  Per 32-channel chunk the (8+K-1)x(16+K-1)
// halo tile is fetched once (prologue affine/activation applied on the way in) and every tap reads its shifted
// window straight from LDS, so activations cross L2->LDS once instead of K*K times (the im2col redundancy that
// made k_conv_igemm L2-bound); only the (BN x 32) weight slice of each (chunk, tap) is streamed, double-buffered.
// One output row of the tile = one 16-pixel MFMA column block; waves 4(m: 2 rows each) x 1(n).
// BF: bf16 operands (halo tile and weight slices rounded on the way into LDS, 80-byte rows), one v_mfma_f32_16x16x32_bf16 per tap
// and tile pair instead of eight fp32 MFMAs; fp32 accumulation and epilogue.
// X3: "f32 via bf16x3" (common.h x3_split2t): halo tile and weight slices as three bf16 planes each, six bf16 MFMAs per tap and tile pair.
// NW waves per block (4, or 8 for the 16-row tile of mode 2: the weight slice of a tap is split once per BLOCK, so twice the pixels per
// block halve the split work per MFMA; LDS for one block per CU, its eight waves = the two blocks of four it replaces) and R96: 96-byte
// LDS rows (conflict-free ds_read_b128 at any pixel base, conv_igemm_body) where one block per CU leaves the room.
template <int KS, int BN, int TH, bool BF = false, bool X3 = false, int NW = 4, bool R96 = false>    // TH x 16 output pixels per block: TH / NW rows per wave
__device__ __forceinline__ void conv_halo_body(const ConvArgs& a) {
  static_assert(!(BF && X3), "one operand format");
  constexpr int NTH = NW * 64, RPS = NTH / 8;            // threads; pixel rows (8 float4 each) staged per pass
  static_assert(!R96 || X3, "96-byte rows: the bf16x3 variant");
  constexpr int LDH = R96 ? (BK + 16) / 2 : (BF || X3) ? (BK + 8) / 2 : LDK;   // LDS row stride in FLOAT units (bf16 rows: 40 halves = 20 floats = 80 bytes; 96-byte rows -- conflict-free
                                                         // ds_read_b128, conv_igemm_body -- cost the second resident block here: 77.6 vs 54.4 us per launch, measured)
  constexpr int NPL = X3 ? 3 : 1;                        // operand planes
  constexpr int TW = 16, HH = TH + KS - 1, HW_ = TW + KS - 1, NPX = HH * HW_;
  constexpr int NT = BN / 16, T = KS * KS, MR = TH / NW;
  constexpr bool PREFETCH = false;
  static_assert(!(BF || X3) || !PREFETCH, "bf16 variants: direct staging only");   // halo staged directly into ONE LDS buffer: 3 blocks per CU hide the staging latency
                                     // (measured: tatt 3x3 60.6 -> 55.3 us, en2b 118 -> 84 us vs the register-prefetch variant;
                                     //  weights straight from L1/L2 to registers instead of LDS measured 76 / 146 us: rejected)
  constexpr int HBUF = PREFETCH ? 2 : 1;
  constexpr int HV = PREFETCH ? (NPX * 8 + NTH - 1) / NTH : 1;   // halo float4 per thread held in registers
  constexpr int WV = (BN * 8 + NTH - 1) / NTH;               // weight float4 per thread and tap
#ifndef DPMN_HALO_TPS
#define DPMN_HALO_TPS 1                                   // 3: the three taps of a kernel row share one weight stage and ONE barrier
#endif
  constexpr int TPS = (KS == 3 && !BF && !X3 && TH == 4) ? DPMN_HALO_TPS : 1;      // taps per weight stage
  static_assert(T % TPS == 0, "whole stages");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* halo = smem;                                    // [HBUF][NPL][NPX][LDK]
  float* Wt = smem + HBUF * NPL * NPX * LDH;             // [2][TPS][NPL][BN][LDK]
  constexpr int HPL = NPX * LDH, WPLH = BN * LDH;        // plane strides (float units)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_x = a.Win / TW, tiles_y = a.Hin / TH;
  const int b = blockIdx.x / (tiles_x * tiles_y), trem = blockIdx.x % (tiles_x * tiles_y);
  const int ty0 = (trem / tiles_x) * TH, tx0 = (trem % tiles_x) * TW;
  const int n_blk = blockIdx.y * BN;
  const int padk = (KS - 1) / 2;
  const int c01 = a.cseg[0] + a.cseg[1];
  const int nchunks = a.cin / BK;

  float4 hraw[HV], wraw[TPS * WV];
  // one staged float4 (4 consecutive k of one row) -> LDS, fp32 or rounded to bf16
  auto put4 = [&](float* base, int row, int c4, const float4& v, int plane) {
    if constexpr (X3) {
      uint2 h, m, l;
      x3_split4t(v, h, m, l);
      unsigned short* d_ = reinterpret_cast<unsigned short*>(base + row * LDH) + c4;
      *reinterpret_cast<uint2*>(d_) = h;
      *reinterpret_cast<uint2*>(d_ + 2 * plane) = m;
      *reinterpret_cast<uint2*>(d_ + 4 * plane) = l;
    } else if constexpr (BF) *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(base + row * LDH) + c4) = pack_bf16x4(v.x, v.y, v.z, v.w);
    else *reinterpret_cast<float4*>(base + row * LDH + c4) = v;
  };
  // fetch (and transform) halo element i of channel chunk `chunk`
  auto halo_elem = [&](int chunk, int i) -> float4 {
    const int c0 = chunk * BK;
    int seg = 0, cl0 = c0;
    if (c0 >= c01) { seg = 2; cl0 = c0 - c01; }
    else if (c0 >= a.cseg[0]) { seg = 1; cl0 = c0 - a.cseg[0]; }
    const float* src = a.in[seg];
    const int cs = a.cseg[seg];
    const float* sc = a.in_scale[seg];
    const float* sh = a.in_shift[seg];
    float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
    const int px = i >> 3, c4 = (i & 7) * 4;
    const int iy = ty0 + px / HW_ - padk, ix = tx0 + px % HW_ - padk;
    if (iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win) {
      val = *reinterpret_cast<const float4*>(src + (((size_t)b * a.Hin + iy) * a.Win + ix) * cs + cl0 + c4);
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
    return val;
  };
  // All of a thread's halo elements are fetched first, as raw buffer loads (the range check returns 0 for pixels outside the
  // image: no predicated load, no branch, so the HVD loads are in flight together instead of one wait per element), then
  // transformed and stored.  256 % 8 == 0: every element of a thread has the same channel quad, hence one affine pair.
  constexpr int HVD = (NPX * 8 + NTH - 1) / NTH;
  constexpr bool HALO_BUF = TH == 4 || NW == 8;      // (8-row tiles, 6-12 loads per thread: measured slower than the per-element loop)
  auto stage_halo_direct = [&](int chunk) {
    if constexpr (!HALO_BUF) {
      for (int i = tid; i < NPX * 8; i += NTH)
        put4(halo, i >> 3, (i & 7) * 4, halo_elem(chunk, i), HPL);
      return;
    }
    const int c0 = chunk * BK;                                   // wave-uniform chunk decode
    const int seg = c0 >= c01 ? 2 : (c0 >= a.cseg[0] ? 1 : 0);
    const int cl = c0 - (seg == 2 ? c01 : (seg == 1 ? a.cseg[0] : 0)) + (tid & 7) * 4;
    const float* src = a.in[seg];
    const int cs = a.cseg[seg];
    const float* sc = a.in_scale[seg];
    const float* sh = a.in_shift[seg];
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, a.B * a.Hin * a.Win * cs * 4, 0x00020000);
    float4 raw[HVD];
    unsigned inb = 0;
#pragma unroll
    for (int v = 0; v < HVD; ++v) {
      const int px = (tid >> 3) + v * RPS;
      const int iy = ty0 + px / HW_ - padk, ix = tx0 + px % HW_ - padk;
      const bool ok = px < NPX && iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win;
      const unsigned off = (unsigned)((((b * a.Hin + iy) * a.Win + ix) * cs + cl) * 4) | (ok ? 0u : 0x80000000u);
      raw[v] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0));
      inb |= (ok ? 1u : 0u) << v;
    }
    float4 s4 = make_float4(1.f, 1.f, 1.f, 1.f), h4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (sc) { s4 = *reinterpret_cast<const float4*>(sc + cl); h4 = *reinterpret_cast<const float4*>(sh + cl); }
#pragma unroll
    for (int v = 0; v < HVD; ++v) {
      const int px = (tid >> 3) + v * RPS;
      float4 val = raw[v];
      if (sc) { val.x = val.x * s4.x + h4.x; val.y = val.y * s4.y + h4.y; val.z = val.z * s4.z + h4.z; val.w = val.w * s4.w + h4.w; }
      if (a.pro_act != ACT_NONE) {
        val.x = apply_act(val.x, a.pro_act, 0.f); val.y = apply_act(val.y, a.pro_act, 0.f);
        val.z = apply_act(val.z, a.pro_act, 0.f); val.w = apply_act(val.w, a.pro_act, 0.f);
      }
      if (!((inb >> v) & 1u)) val = make_float4(0.f, 0.f, 0.f, 0.f);      // padding stays exactly 0 after the transform
      if (HVD * RPS == NPX || px < NPX) put4(halo, px, (tid & 7) * 4, val, HPL);
    }
  };
  auto issue_halo = [&](int chunk) {
#pragma unroll
    for (int v = 0; v < HV; ++v) {
      const int i = tid + v * NTH;
      hraw[v] = (i < NPX * 8) ? halo_elem(chunk, i) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto commit_halo = [&](int buf) {
#pragma unroll
    for (int v = 0; v < HV; ++v) {
      const int i = tid + v * NTH;
      if (i < NPX * 8) put4(halo + (size_t)buf * NPX * LDH, i >> 3, (i & 7) * 4, hraw[v], HPL);
    }
  };
  // weight slice of one (chunk, tap): raw buffer loads -- per-thread byte offsets are loop invariants, the (chunk, tap) part
  // travels in the scalar offset, rows past Cout fall beyond num_records and read 0: no address arithmetic and no predicated
  // load (whose join would make hipcc drain vmcnt(0) in front of the MFMA block) inside the tap loop
  int wofs_h[WV];
#pragma unroll
  for (int v = 0; v < WV; ++v) {
    const int i = tid + v * NTH;
    wofs_h[v] = (i < BN * 8) ? ((n_blk + (i >> 3)) * a.Kp + (i & 7) * 4) * 4 : (int)0x80000000;
  }
  const bool w_buf_ok = (size_t)a.Cout * a.Kp * 4 < (1ull << 31);
  const float* wg = a.w + (conv_group_of(a, (b * a.Hin + ty0) * a.Win + tx0) ? a.wgs : 0L);       // the tile's image decides the group
  const __amdgpu_buffer_rsrc_t wrs_h = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wg), 0, w_buf_ok ? a.Cout * a.Kp * 4 : 0, 0x00020000);
  auto issue_w = [&](int chunk, int stage) {       // the TPS taps stage * TPS ... of the chunk
#pragma unroll
    for (int tp = 0; tp < TPS; ++tp) {
      const size_t k0 = (size_t)(stage * TPS + tp) * a.cin + chunk * BK;
      if (w_buf_ok) {
#pragma unroll
        for (int v = 0; v < WV; ++v)
          wraw[tp * WV + v] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(wrs_h, wofs_h[v], (int)k0 * 4, 0));
        continue;
      }
#pragma unroll
      for (int v = 0; v < WV; ++v) {
        const int i = tid + v * NTH;
        const int r = i >> 3, c4 = (i & 7) * 4;
        wraw[tp * WV + v] = (i < BN * 8 && n_blk + r < a.Cout) ? *reinterpret_cast<const float4*>(wg + (size_t)(n_blk + r) * a.Kp + k0 + c4)
                                                                : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  };
  auto commit_w = [&](int buf) {
#pragma unroll
    for (int tp = 0; tp < TPS; ++tp)
#pragma unroll
      for (int v = 0; v < WV; ++v) {
        const int i = tid + v * NTH;
        if (i < BN * 8) put4(Wt + (size_t)(buf * TPS + tp) * NPL * BN * LDH, i >> 3, (i & 7) * 4, wraw[tp * WV + v], WPLH);
      }
  };

  const int lr = lane & 15, kq = lane >> 4;
  f32x4 acc[NT][MR];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < MR; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if (PREFETCH) { issue_halo(0); commit_halo(0); }
  issue_w(0, 0);
  commit_w(0);
  __syncthreads();
  int wb = 0;
  for (int chunk = 0; chunk < nchunks; ++chunk) {
    const int hb = PREFETCH ? (chunk & 1) : 0;
    const bool more = chunk + 1 < nchunks;
    if (PREFETCH) {
      if (more) issue_halo(chunk + 1);
    } else {
      stage_halo_direct(chunk);     // all reads of the previous chunk finished at the last barrier
      __syncthreads();
    }
    // 3x3: the nine taps fully unrolled -- (ky, kx), the halo offsets of the B-operand reads and the weight-buffer parity become
    // immediates instead of per-tap scalar / vector arithmetic (81 VALU + 81 SALU per 64-MFMA tap before; the vector ALU shares
    // its issue with the fp32 matrix pipe).  T = 9 is odd, so the buffer parity of tap t is (chunk + t) & 1.
    const float* hp0 = halo + (size_t)hb * NPX * LDH + ((MR * wave) * HW_ + lr) * LDH + kq * 4;      // (bf16: 8 halves = 4 float units)
    const float* wp0 = Wt + lr * LDH + kq * 4;
    constexpr int NS = T / TPS;
    constexpr int TAP_UNROLL = KS == 3 ? NS : 1;
#pragma unroll TAP_UNROLL
    for (int stage = 0; stage < NS; ++stage) {
      const bool lastt = stage == NS - 1;
      if (!lastt) issue_w(chunk, stage + 1);
      else if (more) issue_w(chunk + 1, 0);
#pragma unroll
      for (int tp = 0; tp < TPS; ++tp) {
      const int tap = stage * TPS + tp;
      const int ky = tap / KS, kx = tap % KS;
      const float* hp = hp0 + (ky * HW_ + kx) * LDH;
      const float* wp = wp0 + (size_t)(wb * TPS + tp) * NPL * BN * LDH;
      if constexpr (X3) {
        bf16x8 x0[MR], x1[MR], x2[MR], w0[NT], w1[NT], w2[NT];
#pragma unroll
        for (int j = 0; j < MR; ++j) {
          x0[j] = *reinterpret_cast<const bf16x8*>(hp + j * HW_ * LDH);
          x1[j] = *reinterpret_cast<const bf16x8*>(hp + HPL + j * HW_ * LDH);
          x2[j] = *reinterpret_cast<const bf16x8*>(hp + 2 * HPL + j * HW_ * LDH);
        }
#pragma unroll
        for (int i = 0; i < NT; ++i) {
          w0[i] = *reinterpret_cast<const bf16x8*>(wp + i * 16 * LDH);
          w1[i] = *reinterpret_cast<const bf16x8*>(wp + WPLH + i * 16 * LDH);
          w2[i] = *reinterpret_cast<const bf16x8*>(wp + 2 * WPLH + i * 16 * LDH);
        }
#define HL_X3_TERM(XF, WF)                                                  \
        _Pragma("unroll") for (int i = 0; i < NT; ++i)                      \
          _Pragma("unroll") for (int j = 0; j < MR; ++j) acc[i][j] = mfma16_bf16(WF[i], XF[j], acc[i][j]);
        HL_X3_TERM(x2, w0) HL_X3_TERM(x1, w1) HL_X3_TERM(x0, w2) HL_X3_TERM(x1, w0) HL_X3_TERM(x0, w1) HL_X3_TERM(x0, w0)
#undef HL_X3_TERM
      } else if constexpr (BF) {
        bf16x8 xf[MR], wf[NT];
#pragma unroll
        for (int j = 0; j < MR; ++j) xf[j] = *reinterpret_cast<const bf16x8*>(hp + j * HW_ * LDH);
#pragma unroll
        for (int i = 0; i < NT; ++i) wf[i] = *reinterpret_cast<const bf16x8*>(wp + i * 16 * LDH);
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
          for (int j = 0; j < MR; ++j) acc[i][j] = mfma16_bf16(wf[i], xf[j], acc[i][j]);
      } else
#pragma unroll
      for (int kc = 0; kc < BK; kc += 16) {
        f32x4 xf[MR];
#pragma unroll
        for (int j = 0; j < MR; ++j) xf[j] = *reinterpret_cast<const f32x4*>(hp + j * HW_ * LDH + kc);
        f32x4 wf[NT];
#pragma unroll
        for (int i = 0; i < NT; ++i) wf[i] = *reinterpret_cast<const f32x4*>(wp + i * 16 * LDH + kc);
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
          for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int j = 0; j < MR; ++j) acc[i][j] = mfma16(wf[i][s4], xf[j][s4], acc[i][j]);
      }
      }
      if (!lastt || more) commit_w(wb ^ 1);
      if (PREFETCH && lastt && more) commit_halo(hb ^ 1);
      __syncthreads();
      wb ^= 1;
    }
  }

  float ssum[NT][4], ssq[NT][4];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) { ssum[i][r] = 0.f; ssq[i][r] = 0.f; }
#pragma unroll
  for (int j = 0; j < MR; ++j) {
    const int oy = ty0 + MR * wave + j, ox = tx0 + lr;
    const int m = (b * a.Hin + oy) * a.Win + ox;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const int n = n_blk + i * 16 + kq * 4;
      if (n >= a.Cout) continue;
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      conv_store(a, m, n, v, ssum[i], ssq[i], a.ooy, a.oox);
    }
  }
  if (a.stats) {
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const int n = n_blk + i * 16 + kq * 4;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float s_ = ssum[i][r], q = ssq[i][r];
        s_ += xshfl<1>(s_); s_ += xshfl<2>(s_); s_ += xshfl<4>(s_); s_ += xshfl<8>(s_);
        q += xshfl<1>(q); q += xshfl<2>(q); q += xshfl<4>(q); q += xshfl<8>(q);
        if (lr == 0 && n + r < a.Cout) {
          double* st = reinterpret_cast<double*>(a.stats) + (size_t)(blockIdx.x % STAT_SLOTS) * 2 * a.Cout;
          atomicAdd(st + n + r, (double)(s_)); atomicAdd(st + a.Cout + n + r, (double)(q));
        }
      }
    }
  }
}


template <int KS, int BN, int TH, bool BF = false>
__global__ __launch_bounds__(256) void k_conv_halo(ConvArgs a) {
  conv_halo_body<KS, BN, TH, BF, false>(a);
}
template <int KS, int BN, int TH>
__global__ __launch_bounds__(256, 2) void k_conv_halo_x3(ConvArgs a) {
  conv_halo_body<KS, BN, TH, false, true>(a);
}
template <int KS, int BN>
__global__ __launch_bounds__(512, 1) void k_conv_halo_x3w(ConvArgs a) {      // 16 x 16 pixels, eight waves, 96-byte rows
  conv_halo_body<KS, BN, 16, false, true, 8, true>(a);
}

static inline double conv_flops(const ConvArgs& a) {
  return 2.0 * a.B * a.Hp * a.Wp * (a.nphase > 1 ? a.nphase : 1) * (double)a.Cout * a.KH * a.KW * a.cin;
}
static inline double conv_bytes(const ConvArgs& a) {
  const double nph = a.nphase > 1 ? a.nphase : 1;
  const double ng = a.groups > 1 ? a.groups : 1;
  return 4.0 * ((double)a.B * a.Hin * a.Win * a.cin + ng * nph * (double)a.Cout * a.KH * a.KW * a.cin + nph * a.B * a.Hp * a.Wp * (double)a.Cout);
}

}  // namespace
