// Shared device helpers for the DPMN gfx950 kernels (fp32 path).
// MFMA: v_mfma_f32_16x16x4_f32 -- exact f32, 32-cycle issue per SIMD, 157 TF chip peak
// (MI355X_MICROARCH.md "Matrix cores").  Wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/dpmn_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define DPMN_CHECK_LAUNCH()                                                         \
  do {                                                                              \
    hipError_t e__ = hipGetLastError();                                             \
    if (e__ != hipSuccess) return dpmn_set_error(DPMN_ERR_LAUNCH, hipGetErrorString(e__)); \
  } while (0)

#define DPMN_REQUIRE(cond, msg)                                    \
  do {                                                             \
    if (!(cond)) return dpmn_set_error(DPMN_ERR_ARG, msg);         \
  } while (0)

int dpmn_set_error(int code, const char* msg);

// D(16x16) += A(16x4) * B(4x16).  Lane l supplies A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15];
// result lane l holds D[row = (l>>4)*4 + r][col = l&15], r = 0..3.
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// erf by Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, i.e. fp32 round-off class) -- branch-free, ~12 VALU ops
// instead of libm erff's ~50; exact GELU (nn.GELU default, not the tanh approximation) to 2e-7 * |x|.
__device__ __forceinline__ float erf_as(float x) {
  // (v_rcp_f32, 1 ulp, instead of the ~10-instruction IEEE division, and explicit fmas under -ffp-contract=off: both far inside
  // the 1.5e-7 of the approximation itself; the GELU epilogue of the K = 96 GEMMs is paid in MFMA issue time)
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
  const float r = fmaf(-poly, __expf(-ax * ax), 1.0f);
  return copysignf(r, x);
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float softplus_t(float x) { return x > 20.0f ? x : log1pf(expf(x)); }  // F.softplus threshold 20
// mish(x) = x*tanh(softplus(x)) = x * n/(n+2) with n = e^x (e^x + 2)  (exact identity; one exp, no log/tanh).
// For x > 20 softplus(x) = x (F.softplus threshold) and tanh(x) = 1 in fp32.
__device__ __forceinline__ float mish_f(float x) {
  if (x > 20.0f) return x;
  const float t = __expf(x);
  const float n = t * (t + 2.0f);
  return x * (n / (n + 2.0f));
}
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }
// hardware-transcendental forms for latency chains (the GRU recurrence): v_exp_f32 + v_rcp_f32, abs error ~1e-7
__device__ __forceinline__ float sigmoid_fast(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanh_fast(float x) { return fmaf(2.0f, __builtin_amdgcn_rcpf(1.0f + __expf(-2.0f * x)), -1.0f); }

// activation codes shared by GEMM / conv epilogues and conv prologues
enum { ACT_NONE = 0, ACT_GELU = 1, ACT_RELU = 2, ACT_LEAKY02 = 3, ACT_LEAKY001 = 4, ACT_MISH = 5, ACT_PRELU = 6,
       ACT_TANH = 7, ACT_SIGMOID = 8, ACT_RELU_POST_RES = 9 };

__device__ __forceinline__ float apply_act(float v, int act, float slope) {
  switch (act) {
    case ACT_GELU: return gelu_erf(v);
    case ACT_RELU: return v > 0.f ? v : 0.f;
    case ACT_LEAKY02: return v > 0.f ? v : 0.2f * v;
    case ACT_LEAKY001: return v > 0.f ? v : 0.01f * v;
    case ACT_MISH: return mish_f(v);
    case ACT_PRELU: return v > 0.f ? v : slope * v;
    case ACT_TANH: return tanhf(v);
    case ACT_SIGMOID: return sigmoid_f(v);
    default: return v;
  }
}

// 4 values at once with the (wave-uniform) activation switch hoisted out of the per-value loop
__device__ __forceinline__ void apply_act4(float (&v)[4], int act, float slope) {
  switch (act) {
    case ACT_NONE: break;
    case ACT_GELU:
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
      break;
    case ACT_RELU:
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : 0.f;
      break;
    case ACT_MISH:
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = mish_f(v[r]);
      break;
    default:
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = apply_act(v[r], act, slope);
  }
}

// v[lane ^ O] for a compile-time O without the LDS pipe (__shfl_xor = ds_bpermute_b32 plus its address arithmetic and an LDS round
// trip per exchange): DPP inside a row of 16 lanes -- quad_perm for 1 / 2, row_shl:4 | row_shr:4 under bank masks for 4, row_ror:8
// for 8 -- and the gfx950 v_permlane16_swap / v_permlane32_swap for 16 / 32.  Exactly the partner's value, so every butterfly built
// on it gives bitwise the sums and maxima of the __shfl_xor version (checked on the device by dpmn_selftest_xshfl).
__device__ __forceinline__ unsigned wave_lane() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
template <int CTRL, int BANK>
__device__ __forceinline__ unsigned dpp_mov_u(unsigned old, unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, 0xF, BANK, false);
}
template <int O>
__device__ __forceinline__ unsigned xshfl_u(unsigned v) {
  typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
  if constexpr (O == 1) return dpp_mov_u<0xB1, 0xF>(v, v);               // quad_perm [1, 0, 3, 2]
  else if constexpr (O == 2) return dpp_mov_u<0x4E, 0xF>(v, v);          // quad_perm [2, 3, 0, 1]
  else if constexpr (O == 4) {
    const unsigned t = dpp_mov_u<0x104, 0x5>(v, v);                      // row_shl:4 into lanes 0-3, 8-11 of every row
    return dpp_mov_u<0x114, 0xA>(t, v);                                  // row_shr:4 into lanes 4-7, 12-15
  } else if constexpr (O == 8) return dpp_mov_u<0x128, 0xF>(v, v);       // row_ror:8
  else if constexpr (O == 16) {
    const u32x2_ r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    return (wave_lane() & 16) ? r[0] : r[1];                             // odd rows: the partner landed in the first result
  } else {
    static_assert(O == 32, "xshfl: O must be 1, 2, 4, 8, 16 or 32");
    const u32x2_ r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return (wave_lane() & 32) ? r[0] : r[1];
  }
}
template <int O> __device__ __forceinline__ float xshfl(float v) { return __uint_as_float(xshfl_u<O>(__float_as_uint(v))); }
template <int O> __device__ __forceinline__ int xshfl(int v) { return (int)xshfl_u<O>((unsigned)v); }
template <int O> __device__ __forceinline__ unsigned xshfl(unsigned v) { return xshfl_u<O>(v); }
template <int O> __device__ __forceinline__ unsigned long long xshfl(unsigned long long u) {
  const unsigned lo = xshfl_u<O>((unsigned)u), hi = xshfl_u<O>((unsigned)(u >> 32));
  return ((unsigned long long)hi << 32) | lo;
}
template <int O> __device__ __forceinline__ long long xshfl(long long v) { return (long long)xshfl<O>((unsigned long long)v); }
template <int O> __device__ __forceinline__ double xshfl(double v) {
  return __longlong_as_double((long long)xshfl<O>((unsigned long long)__double_as_longlong(v)));
}
// the same with the offset as an argument (a constant after unrolling; anything else falls back to ds_bpermute)
template <typename T>
__device__ __forceinline__ T xshfl_v(T v, int o) {
  switch (o) {
    case 1: return xshfl<1>(v);
    case 2: return xshfl<2>(v);
    case 4: return xshfl<4>(v);
    case 8: return xshfl<8>(v);
    case 16: return xshfl<16>(v);
    case 32: return xshfl<32>(v);
    default: return __shfl_xor(v, o, 64);
  }
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += xshfl_v(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, xshfl_v(v, o));
  return v;
}

// Train-mode Dropout / DropPath masks (nn.Dropout pgrm.py:24,180,494; timm DropPath pgrm.py:310).  The reference draws them
// from torch's Philox stream, which no other implementation can replay; here a mask element is a pure function of
// (seed, element index) -- splitmix64 finaliser, top 24 bits as a uniform in [0,1) -- so forward, backward and the CPU
// oracle (oracle/pgrm.py drop_mask) regenerate identical masks without storing them.  Returns 0 or 1/(1-p).
constexpr unsigned long long DROP_PHI = 0x9E3779B97F4A7C15ull;
// the hash in two halves: z0 = idx * PHI + seed is linear in idx, so a kernel that walks idx = base + c with compile-time c pays
// ONE 64-bit multiply per base (drop_z0) and a 64-bit constant add per element (z0 + c * DROP_PHI) instead of a multiply each
__device__ __forceinline__ unsigned long long drop_z0(unsigned long long seed, unsigned long long idx) { return idx * DROP_PHI + seed; }
__device__ __forceinline__ float drop_scale_z(unsigned long long z, float p, float inv_keep) {
  z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
  z ^= z >> 27; z *= 0x94D049BB133111EBull;
  z ^= z >> 31;
  const float u = (float)(unsigned)(z >> 40) * 5.9604644775390625e-8f;   // 2^-24
  return u >= p ? inv_keep : 0.0f;
}
__device__ __forceinline__ float drop_scale(unsigned long long seed, unsigned long long idx, float p, float inv_keep) {
  return drop_scale_z(drop_z0(seed, idx), p, inv_keep);
}

static inline hipStream_t as_stream(dpmn_stream_t s) { return (hipStream_t)s; }

// In-pipeline kernel timing (include/dpmn_hip.h dpmn_profile_*; runtime.hip).  A ProfScope placed around ONE kernel launch
// brackets it with a pair of HIP events on the launch stream while profiling is armed for its tag, and carries the launch's
// algorithmic FLOPs and bytes (from the launch arguments), so bench.py can report achieved TFLOP/s / GB/s per kernel family
// measured where the kernel runs -- inside the step, at the clocks and cache state of the timed region.  Disarmed cost: one
// load and one branch.
enum ProfTag { PT_CONV_IGEMM_128 = 0, PT_CONV_IGEMM_64, PT_CONV_IGEMM_NARROW, PT_CONV_SPLITK_REDUCE, PT_CONV_HALO, PT_CONV_HALO_C4,
               PT_GEMM_PW, PT_GEMM_WSTAT, PT_GEMM_KLOOP, PT_DWCONV_GELU, PT_WATTN8, PT_WATTN_SCALAR, PT_ATTN_FUSED, PT_BIGRU,
               PT_MHA32, PT_PATCH_EMBED, PT_SK_GATE, PT_TAIL, PT_DWPW_FUSED, PT_GEMM_WSTAT_LN, PT_CONV_IGEMM_SK, PT_ATTN_FUSED_BWD, PT_WATTN_BWD,
               PT_CONV_WGRAD, PT_GEMM_TN, PT_TN_REDUCE, PT_DWCONV_BWD, PT_WGRAD_UNPACK, PT_CONV_PACK, PT_AFFINE_ACT_BWD, PT_LN_BWD, PT_WATTN_MFMA32, PT_COUNT };
extern unsigned long long g_dpmn_prof_mask;
extern double g_dpmn_prof_hint_bytes;      // bytes of the next multi-descriptor launch (its descriptors live on the device)
extern int g_dpmn_bf16;
extern int g_dpmn_x3;       // dpmn_set_compute_dtype(2): fp32 products as six bf16 MFMAs of a three-term operand split (kernels that have the variant)
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// four fp32 -> four bf16 (round to nearest even, v_cvt_pk_bf16_f32) as two dwords
__device__ __forceinline__ uint2 pack_bf16x4(float a, float b, float c, float d) {
  typedef float f32x2_ __attribute__((ext_vector_type(2)));
  const bf16x2 lo = __builtin_convertvector((f32x2_){a, b}, bf16x2), hi = __builtin_convertvector((f32x2_){c, d}, bf16x2);
  return make_uint2(__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi));
}
// D(16x16) += A(16x32) * B(32x16) on bf16 operands, fp32 accumulation.  Lane l supplies A[i = l&15][k = 8*(l>>4) .. +7] and
// B[k = 8*(l>>4) .. +7][j = l&15]; the result layout is that of mfma16.
__device__ __forceinline__ f32x4 mfma16_bf16(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
// ---- "f32 via bf16x3" (dpmn_set_compute_dtype(2)): an fp32 value splits EXACTLY into three bf16 terms by truncation,
//   x = h + m + l,  h = x with the low 16 bits cleared, m = (x - h) with the low 16 bits cleared, l = x - h - m   (8 + 8 + <= 8 bits),
// and a product keeps the six terms of weight >= 2^-16:  x y ~= h h' + (h m' + m h') + (h l' + m m' + l h'); the three dropped ones
// are < 2^-21 |x y| in the worst case, ~2^-25 |x y| on average (uniform mantissas) -- the rounding class of one fp32 multiply.  The
// products run as six v_mfma_f32_16x16x32_bf16 (bf16 x bf16 is exact in fp32, accumulation fp32): 2500 / 6 = 417 TFLOP/s of
// fp32-equivalent work against the 157 TFLOP/s of v_mfma_f32_16x16x4_f32.  Truncation instead of round-to-nearest: the same 11
// vector instructions per pair of values, no overflow to inf for finite |x| up to FLT_MAX (rounding |x| > 3.39e38 to bf16 gives inf,
// and inf - inf a NaN in the lower planes).  Non-finite inputs give non-finite outputs (NaN, where the fp32 pipe would keep an inf).
// (a, b) -> three dwords of bf16 pairs (low half = a's term)
__device__ __forceinline__ void x3_split2t(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
  const unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
  h = __builtin_amdgcn_perm(ub, ua, 0x07060302u);                    // (hi16(b) << 16) | hi16(a)
  const float ra = a - __uint_as_float(ua & 0xffff0000u), rb = b - __uint_as_float(ub & 0xffff0000u);      // exact
  const unsigned va = __float_as_uint(ra), vb = __float_as_uint(rb);
  m = __builtin_amdgcn_perm(vb, va, 0x07060302u);
  const float sa = ra - __uint_as_float(va & 0xffff0000u), sb = rb - __uint_as_float(vb & 0xffff0000u);    // exact, <= 8 significant bits
  l = __builtin_amdgcn_perm(__float_as_uint(sb), __float_as_uint(sa), 0x07060302u);
}
__device__ __forceinline__ void x3_split4t(const float4& v, uint2& h, uint2& m, uint2& l) {
  x3_split2t(v.x, v.y, h.x, m.x, l.x);
  x3_split2t(v.z, v.w, h.y, m.y, l.y);
}
// debugging / A-B switch: DPMN_X3_OFF bit mask of kernel families that keep the fp32 kernel in mode 2
// (1 implicit-GEMM conv, 2 halo conv, 4 pointwise GEMM, 8 k-loop GEMM 64 x 96, 16 conv weight gradient, 32 k-loop GEMM 128 x 128,
// 64 Linear weight gradient dY^T X, 128 whole-K token GEMMs with the rows in registers: k_gemm_rowreg, k_sk_mlp_in)
int dpmn_x3_off_mask();
static inline bool x3_on(int family_bit) { return g_dpmn_x3 && !(dpmn_x3_off_mask() & family_bit); }
struct ProfScope {
  int slot;
  hipStream_t st;
  ProfScope(int tag, hipStream_t s, double flops, double bytes);
  ~ProfScope();
  void close();      // end the bracket before the scope does (further launches of the function are not part of the family)
};
int dpmn_prof_open(int tag, hipStream_t st, double flops, double bytes);
void dpmn_prof_close(int slot, hipStream_t st);
inline ProfScope::ProfScope(int tag, hipStream_t s, double flops, double bytes) : slot(-1), st(s) {
  if ((g_dpmn_prof_mask >> tag) & 1ull) slot = dpmn_prof_open(tag, s, flops, bytes);
}
inline void ProfScope::close() {
  if (slot >= 0) dpmn_prof_close(slot, st);
  slot = -1;
}
inline ProfScope::~ProfScope() { close(); }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
