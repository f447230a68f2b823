// Shared between gemm.hip and gemm_x3.hip: epilogue / prologue argument blocks and the generic tile epilogue of the token GEMMs.
#pragma once
#include <cstdlib>
#include "common.h"

namespace dpmn_gemm {
struct EpiArgs {
  const float* bias;   // (N) or null
  const float* res1;   // (M,N) or null, added after activation
  const float* res2;   // (M,N) or null
  float* colsum;       // (gridDim.x, N) per-block column sums of GELU(y) (SKConv GAP partials) or null
  int act;             // ACT_*
  float slope;         // PReLU slope
  int atomic;          // 1: y += acc with fp32 atomics (split-K weight gradients); bias/act/res ignored
  long zstride;        // k-loop split launches: != 0: split z STORES its partial result at y + z * zstride (no atomics; the caller adds
                       // the splits in order)
  // train-mode Dropout / DropPath on the Linear's output before the residual (Mlp.drop + DropPath, pgrm.py:40,330): k_gemm_kloop's
  // bias + one-residual epilogue only; y = res1 + (acc + bias) * m_elem(flat index) * m_row(flat index / row_len)
  float p_elem = 0.f, p_row = 0.f;
  unsigned long long seed_elem = 0ull, seed_row = 0ull;
  long row_len = 0;
};

struct ProArgs {
  const float* ln_w;   // LayerNorm affine (K) -- PRO_LN
  const float* ln_b;
  float eps;
  const float* sel;    // PRO_SKSEL: attention vectors A (B, G, K) ; x is (M, G*K)
  int rows_per_image;  // PRO_SKSEL: L
  int groups;          // PRO_SKSEL: G
  const float* addv;   // PRO_ADD: second (M,K) operand added to x before the GEMM (pos-embed add)
  const float* x2;     // PRO_CAT2: x = [x (M,k1) | x2 (M,K-k1)] channel concat read in place
  int k1;
};

// gemm_x3.hip ("f32 via bf16x3" instantiations, dpmn_set_compute_dtype(2)): same arguments as the fp32 kernels they stand in for
int x3_launch_kloop(const float* x, int ldx, const float* w, int ldw, float* y, int ldy, int M, int N, int K, const EpiArgs& e, int kb_len,
                    long x_bstride, long w_bstride, dim3 grid, hipStream_t st);
int x3_launch_kloop128(const float* x, const float* w, float* y, int M, int N, int L, int nchunks, int splits, long bstride, long zstride,
                       hipStream_t st);
int x3_launch_pw(const float* g, const float* w, const float* bias, float* z, int B, int Ch, int L, hipStream_t st);
// gemm_rowreg_x3.hip: k_gemm_rowreg<K, pro, epi> (K = 96 / 192, PRO_NONE / PRO_LN, RR_EPI 1 .. 5) and k_sk_mlp_in in mode 2, on the fp32
// launcher's grid (gx blocks of four waves per 96-column group); -1: no such instantiation
int x3_launch_rowreg(int K, int pro, int epi, const float* x, int ldx, const float* w, float* y, int ldy, int M, int N, const ProArgs& p,
                     const EpiArgs& e, int gx, hipStream_t st);
int x3_launch_sk_mlp_in(const float* cat, const float* sel, int rows_per_image, const float* w_head, const float* b_head, const float* feats,
                        const float* shortcut, float* x1, const float* ln_w, const float* ln_b, float eps, const float* w_fc1, const float* b_fc1,
                        float* y, int M, int N, float* v_out, float* n2_out, float p_row, unsigned long long seed_row, int gx, hipStream_t st);
}  // namespace dpmn_gemm

namespace {
using dpmn_gemm::EpiArgs;
using dpmn_gemm::ProArgs;
constexpr int PAD = 4;
enum { PRO_NONE = 0, PRO_LN = 1, PRO_SKSEL = 2, PRO_ADD = 3, PRO_CAT2 = 4 };

// ---------------------------------------------------------------------------------- epilogue
// tile (nt, mt): lane holds y[m = m_base + (l&15)][n = n_base + (l>>4)*4 + r]
// FULL: the caller guarantees that the whole tile lies inside (M, N) -- no edge predicates, i.e. no per-row / per-column
// branches around the stores (with branches hipcc cannot count its memory operations and drains vmcnt(0), which on gfx9
// also waits for the stores of the previous tile).
template <int NT, int MT, bool FULL = false>
__device__ __forceinline__ void epilogue(f32x4 (&acc)[NT][MT], int m0, int n0, int M, int N, int ldy, float* y,
                                         const EpiArgs& e, float* red /*LDS >= 4*BN floats or null*/, int bn_cols,
                                         int n_block0) {
  const int lane = threadIdx.x & 63;
  const int lm = lane & 15, lq = lane >> 4;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int n = n0 + nt * 16 + lq * 4;
    float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool nfull = FULL || (n + 3 < N);
    if (e.bias) {
      if (nfull) b4 = *reinterpret_cast<const float4*>(e.bias + n);
      else {
        float t[4] = {0, 0, 0, 0};
        for (int r = 0; r < 4; ++r) if (n + r < N) t[r] = e.bias[n + r];
        b4 = make_float4(t[0], t[1], t[2], t[3]);
      }
    }
    float cs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int m = m0 + mt * 16 + lm;
      float v[4] = {acc[nt][mt][0] + b4.x, acc[nt][mt][1] + b4.y, acc[nt][mt][2] + b4.z, acc[nt][mt][3] + b4.w};
      const bool min_ = FULL || m < M;
      if (min_ && e.atomic) {
        for (int r = 0; r < 4; ++r)
          if (n + r < N) atomicAdd(y + (size_t)m * ldy + n + r, acc[nt][mt][r]);
      } else if (min_) {
        if (e.colsum) {
#pragma unroll
          for (int r = 0; r < 4; ++r) cs[r] += gelu_erf(v[r]);
        }
        apply_act4(v, e.act, e.slope);
        const size_t off = (size_t)m * ldy + n;
        if (nfull) {
          if (e.res1) { float4 q = *reinterpret_cast<const float4*>(e.res1 + off); v[0] += q.x; v[1] += q.y; v[2] += q.z; v[3] += q.w; }
          if (e.res2) { float4 q = *reinterpret_cast<const float4*>(e.res2 + off); v[0] += q.x; v[1] += q.y; v[2] += q.z; v[3] += q.w; }
          *reinterpret_cast<float4*>(y + off) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
          for (int r = 0; r < 4; ++r) if (n + r < N) {
            float o = v[r];
            if (e.res1) o += e.res1[off + r];
            if (e.res2) o += e.res2[off + r];
            y[off + r] = o;
          }
        }
      }
    }
    if (e.colsum) {
      // reduce over the 16 token lanes of each quad (xor 1,2,4,8 stays inside l&15), then over waves via LDS
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float s = cs[r];
        s += xshfl<1>(s); s += xshfl<2>(s); s += xshfl<4>(s); s += xshfl<8>(s);
        cs[r] = s;
      }
      if (lm == 0) {
        const int wave = threadIdx.x >> 6;
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave * bn_cols + (n - n_block0) + r] = cs[r];
      }
    }
  }
}

}  // namespace
