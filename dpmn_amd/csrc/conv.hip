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
#include <cstdlib>
#include <type_traits>
#include "common.h"

namespace {

constexpr int PAD = 4, BK = 32, LDK = BK + PAD;
constexpr int STAT_SLOTS = 32;   // BatchNorm statistics are accumulated into (STAT_SLOTS, 2, Cout) DOUBLES and summed by bn_finalize.
// fp64 atomics: a block's fp32 partial sum is exact in fp64 and the fp64 additions of <= a few thousand partials lose nothing a
// final rounding to fp32 can see, so the statistics -- hence the whole training forward -- no longer depend on the order in which
// the blocks arrive (fp32 atomics made two runs of the same step differ by 1e-7 ... 5e-4 downstream)

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
template <int BM, int BN, int WM, int WN, bool UNI = false, int BKT = 32, bool SIMPLE = false, bool AFF = false, bool BF = false, bool M32 = false,
          bool XRED = false>
__device__ __forceinline__ void conv_igemm_body(const ConvArgs& a) {
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
  constexpr int LDKB = BKT + 8;                               // bf16 row: 32 + 8 elements = 80 bytes
  typedef typename std::conditional<BF, unsigned short, float>::type lds_t;
  __shared__ __attribute__((aligned(16))) lds_t Xs[2][BM * (BF ? LDKB : LDK)];
  __shared__ __attribute__((aligned(16))) lds_t Ws[2][BN * (BF ? LDKB : LDK)];

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
    if constexpr (BF) {
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
    if constexpr (!BF) {
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
  for (int kt = kt0; kt < nk; ++kt) {
    const int buf = (kt - kt0) & 1;
    if (SIMPLE) gload_simple(min(kt + 1, nk - 1));
    else if (UNI) gload_uni(min(kt + 1, nk - 1) * BK);      // unconditional: the refill past the end re-reads the last chunk
    else if (kt + 1 < nk) gload((kt + 1) * BK);
    // hipcc otherwise sinks the buffer loads deep into the MFMA block (the last ones ~100 MFMAs down): they must be in
    // flight for the WHOLE block to cover HBM / L2 latency before sstore waits for them
    if (UNI && SCHED_FENCE) __builtin_amdgcn_sched_barrier(0);
    if constexpr (BF) {
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
    if (SIMPLE) sstore_simple(buf ^ 1);
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

// ---------------------------------------------------------------------------------- halo-tile direct conv
// Stride-1 "same" KxK convs (3x3 of the CMM / PSN trunks, 9x9 output conv) with the INPUT tile resident in LDS:
// a block owns 8x16 output pixels of one image x BN output channels.  Per 32-channel chunk the (8+K-1)x(16+K-1)
// halo tile is fetched once (prologue affine/activation applied on the way in) and every tap reads its shifted
// window straight from LDS, so activations cross L2->LDS once instead of K*K times (the im2col redundancy that
// made k_conv_igemm L2-bound); only the (BN x 32) weight slice of each (chunk, tap) is streamed, double-buffered.
// One output row of the tile = one 16-pixel MFMA column block; waves 4(m: 2 rows each) x 1(n).
// BF: bf16 operands (halo tile and weight slices rounded on the way into LDS, 80-byte rows), one v_mfma_f32_16x16x32_bf16 per tap
// and tile pair instead of eight fp32 MFMAs; fp32 accumulation and epilogue.
template <int KS, int BN, int TH, bool BF = false>    // TH x 16 output pixels per block (TH = 8: 2 rows per wave, TH = 4: 1 row per wave)
__global__ __launch_bounds__(256) void k_conv_halo(ConvArgs a) {
  constexpr int LDH = BF ? (BK + 8) / 2 : LDK;           // LDS row stride in FLOAT units (bf16 rows: 40 halves = 20 floats = 80 bytes)
  constexpr int TW = 16, HH = TH + KS - 1, HW_ = TW + KS - 1, NPX = HH * HW_;
  constexpr int NT = BN / 16, T = KS * KS, MR = TH / 4;
  constexpr bool PREFETCH = false;
  static_assert(!BF || !PREFETCH, "bf16 variant: direct staging only");   // halo staged directly into ONE LDS buffer: 3 blocks per CU hide the staging latency
                                     // (measured: tatt 3x3 60.6 -> 55.3 us, en2b 118 -> 84 us vs the register-prefetch variant;
                                     //  weights straight from L1/L2 to registers instead of LDS measured 76 / 146 us: rejected)
  constexpr int HBUF = PREFETCH ? 2 : 1;
  constexpr int HV = PREFETCH ? (NPX * 8 + 255) / 256 : 1;   // halo float4 per thread held in registers
  constexpr int WV = (BN * 8 + 255) / 256;               // weight float4 per thread and tap
#ifndef DPMN_HALO_TPS
#define DPMN_HALO_TPS 1                                   // 3: the three taps of a kernel row share one weight stage and ONE barrier
#endif
  constexpr int TPS = (KS == 3 && !BF && TH == 4) ? DPMN_HALO_TPS : 1;      // taps per weight stage
  static_assert(T % TPS == 0, "whole stages");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* halo = smem;                                    // [HBUF][NPX][LDK]
  float* Wt = smem + HBUF * NPX * LDH;                   // [2][TPS][BN][LDK]

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
  auto put4 = [&](float* base, int row, int c4, const float4& v) {
    if constexpr (BF) *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(base + row * LDH) + c4) = pack_bf16x4(v.x, v.y, v.z, v.w);
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
  constexpr int HVD = (NPX * 8 + 255) / 256;
  constexpr bool HALO_BUF = TH == 4;      // (8-row tiles, 6-12 loads per thread: measured slower than the per-element loop)
  auto stage_halo_direct = [&](int chunk) {
    if constexpr (!HALO_BUF) {
      for (int i = tid; i < NPX * 8; i += 256)
        put4(halo, i >> 3, (i & 7) * 4, halo_elem(chunk, i));
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
      const int px = (tid >> 3) + v * 32;
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
      const int px = (tid >> 3) + v * 32;
      float4 val = raw[v];
      if (sc) { val.x = val.x * s4.x + h4.x; val.y = val.y * s4.y + h4.y; val.z = val.z * s4.z + h4.z; val.w = val.w * s4.w + h4.w; }
      if (a.pro_act != ACT_NONE) {
        val.x = apply_act(val.x, a.pro_act, 0.f); val.y = apply_act(val.y, a.pro_act, 0.f);
        val.z = apply_act(val.z, a.pro_act, 0.f); val.w = apply_act(val.w, a.pro_act, 0.f);
      }
      if (!((inb >> v) & 1u)) val = make_float4(0.f, 0.f, 0.f, 0.f);      // padding stays exactly 0 after the transform
      if (HVD * 32 == NPX || px < NPX) put4(halo, px, (tid & 7) * 4, val);
    }
  };
  auto issue_halo = [&](int chunk) {
#pragma unroll
    for (int v = 0; v < HV; ++v) {
      const int i = tid + v * 256;
      hraw[v] = (i < NPX * 8) ? halo_elem(chunk, i) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto commit_halo = [&](int buf) {
#pragma unroll
    for (int v = 0; v < HV; ++v) {
      const int i = tid + v * 256;
      if (i < NPX * 8) put4(halo + (size_t)buf * NPX * LDH, i >> 3, (i & 7) * 4, hraw[v]);
    }
  };
  // weight slice of one (chunk, tap): raw buffer loads -- per-thread byte offsets are loop invariants, the (chunk, tap) part
  // travels in the scalar offset, rows past Cout fall beyond num_records and read 0: no address arithmetic and no predicated
  // load (whose join would make hipcc drain vmcnt(0) in front of the MFMA block) inside the tap loop
  int wofs_h[WV];
#pragma unroll
  for (int v = 0; v < WV; ++v) {
    const int i = tid + v * 256;
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
        const int i = tid + v * 256;
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
        const int i = tid + v * 256;
        if (i < BN * 8) put4(Wt + (size_t)(buf * TPS + tp) * BN * LDH, i >> 3, (i & 7) * 4, wraw[tp * WV + v]);
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
      const float* wp = wp0 + (size_t)(wb * TPS + tp) * BN * LDH;
      if constexpr (BF) {
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

static inline double conv_flops(const ConvArgs& a) {
  return 2.0 * a.B * a.Hp * a.Wp * (a.nphase > 1 ? a.nphase : 1) * (double)a.Cout * a.KH * a.KW * a.cin;
}
static inline double conv_bytes(const ConvArgs& a) {
  const double nph = a.nphase > 1 ? a.nphase : 1;
  const double ng = a.groups > 1 ? a.groups : 1;
  return 4.0 * ((double)a.B * a.Hin * a.Win * a.cin + ng * nph * (double)a.Cout * a.KH * a.KW * a.cin + nph * a.B * a.Hp * a.Wp * (double)a.Cout);
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
int launch_halo_th(const ConvArgs& a, hipStream_t st) {
  if constexpr (!BF && KS == 3 && BN == 64) {
    if (g_dpmn_bf16) return launch_halo_th<KS, BN, TH, true>(a, st);
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
  if (KS == 3 && small) return launch_halo_th<KS, BN, 4>(a, st);
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
