// Kernels for the PSN backbones TSRN / TATT beyond plain convs and GEMMs:
//   * BiGRU recurrence of GruBlock (tsrn.py:139-150, tatt.py:1070-1083) -- persistent-register kernel,
//     one wave per sequence (lane = direction x hidden unit), W_hh rows held in VGPRs, h broadcast via LDS;
//     the input projection (conv1x1 folded with W_ih) is a separate implicit-GEMM launch;
//   * TPInterpreter pieces (tatt.py:193-223, transformer_v2.py:198-244, 455-469, 806-833): tiny linear,
//     fused 26-token encoder layer (one workgroup per image), 1024x26 cross attention, add+LayerNorm,
//     and the gate step of the query-embedding GRU (batch_first quirk Q5).
// Token tensors here are (N, L, E) row-major (N = image, L = 26 text slots or 1024 pixels, E = 64):
// the reference's (L, N, E) differs only by a transpose no kernel needs.
#include <cstdlib>
#include "common.h"

namespace {

// ---------------------------------------------------------------------------------- BiGRU recurrence
// gi: (pixels, 2*3*HID) = [dir0: r z n | dir1: r z n] precomputed input projection (bias folded)
// seq s -> base pixel = (s / inner) * outer_stride + (s % inner) * inner_stride ; step t adds t*step_stride (pixels)
typedef float v2f __attribute__((ext_vector_type(2)));

// PK: the recurrence's products as v_pk_fma_f32 (default) or, PK = false, as the same sums on scalar v_fma_f32.  The packed form is
// NOT safe next to bf16 MFMA work of another stream: with a v_mfma_f32_16x16x32_bf16 kernel (mode 2 "f32 via bf16x3", mode 1 bf16)
// resident on the same SIMDs, 7-25 % of the launches returned a few sequences off by 3e-3 -- never alone, never next to fp32-MFMA
// kernels, never with scalar fmas (tools/dbg_victim.py: the recurrence on one stream, a bf16x3 conv on another; 0 / 240 vs 28-60 / 240
// mismatching launches).  The launcher therefore takes the scalar form whenever the library is in a bf16-MFMA mode.
template <int HID, bool PK = true>
__global__ __launch_bounds__(256) void k_bigru(const float* gi, const float* w_hh /*(2,3H,H)*/,
                                                const float* b_hh /*(2,3H)*/, const float* res,
                                                float* out, int nseq, int T, int inner, long outer_stride,
                                                long inner_stride, long step_stride) {
  static_assert(HID == 32, "lane mapping assumes 32 hidden units per direction");
  __shared__ __attribute__((aligned(16))) float hs[4][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int dir = lane >> 5, j = lane & 31;
  const long s = (long)blockIdx.x * 4 + wave;
  if (s >= nseq) return;      // whole wave (no block-level barrier below: the hidden state goes through a wave-private LDS row)
  // W_hh rows of this lane's hidden unit as PACKED pairs: (r, z) share the h_k operand (v_pk_fma_f32 with a broadcast), the n row is
  // paired over k against (h_k, h_k+1): 48 packed fmas per step instead of 96 scalar ones on the step's critical path
  v2f wrz[HID], wnn[HID / 2];
  const float* wb = w_hh + (size_t)dir * 3 * HID * HID;
#pragma unroll
  for (int k = 0; k < HID; ++k) wrz[k] = v2f{wb[(size_t)j * HID + k], wb[(size_t)(HID + j) * HID + k]};
#pragma unroll
  for (int k = 0; k < HID / 2; ++k) wnn[k] = v2f{wb[(size_t)(2 * HID + j) * HID + 2 * k], wb[(size_t)(2 * HID + j) * HID + 2 * k + 1]};
  const float br = b_hh[dir * 3 * HID + j], bz = b_hh[dir * 3 * HID + HID + j], bn = b_hh[dir * 3 * HID + 2 * HID + j];
  // the weights are IN their registers before the first gate prefetch is issued (hipcc otherwise sinks these loads to their first
  // use inside the loop, behind the prefetches, and the static s_waitcnt it then needs there drains the ring on every iteration;
  // w_hh / b_hh are not __restrict__ so that the loads may not cross the clobber)
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0)
  asm volatile("" ::: "memory");
  const long base = (s / inner) * outer_stride + (s % inner) * inner_stride;
  float h = 0.f;
  // the recurrence is a pure latency chain (one wave per sequence, at most one wave per SIMD at these sizes): the hidden
  // state goes through a wave-private LDS row (in-order within a wave: no block barrier), and the gate pre-activations of the
  // next THREE steps are in flight while a step computes (a ring of four register sets, the loop unrolled by four: a step takes
  // ~0.3 us, an L2 / HBM load ~1 us -- one step of look-ahead left every step waiting for its gates; a "cur = next" copy at the end
  // of a step would make the compiler wait for the prefetch it just issued)
  float g0[4], g1[4], g2[4], g3[4];     // r, z, n pre-activations + residual
  // branch-free (a conditional load makes the compiler drain every outstanding load at the join): steps past the end
  // re-read a valid element, the residual pointer falls back to gi when there is no residual
  const float* rp = res ? res : gi;
  const float rmul = res ? 1.f : 0.f;
  // running pointers (one 64-bit add per step instead of the index arithmetic: fewer temporaries for the register allocator to
  // park in a gate register whose load it then has to wait for)
  const long p_first = base + (dir ? (long)(T - 1) * step_stride : 0);
  const long dpix = dir ? -step_stride : step_stride;
  const float* gp = gi + p_first * (6 * HID) + dir * 3 * HID + j;
  const float* rq = rp + p_first * (2 * HID) + dir * HID + j;
  float* op = out + p_first * (2 * HID) + dir * HID + j;
  const long gd = dpix * (6 * HID), rd = dpix * (2 * HID);
  int tf = 0;      // index of the step the next fetch is for (wave-uniform)
  auto fetch = [&](float (&g4)[4]) {
    g4[0] = gp[0]; g4[1] = gp[HID]; g4[2] = gp[2 * HID];
    asm volatile("" ::: "memory");                // (pins the four loads HERE: hipcc sinks a load to its first use, three steps later)
    g4[3] = rq[0];   // the residual too: a load feeding this step's store would stall it (x rmul at the use: an arithmetic op on
                     // the loaded value HERE makes the compiler wait for the load at the loop's back-edge)
    asm volatile("" ::: "memory");
    ++tf;
    const bool more = tf < T;      // fetches past the last step re-read it
    gp += more ? gd : 0;
    rq += more ? rd : 0;
  };
  auto pkfma = [](v2f a, v2f b, v2f c) -> v2f {
    if (PK) return __builtin_elementwise_fma(a, b, c);
    return v2f{fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y)};
  };
  auto step = [&](const float (&g4)[4], int t) {
    hs[wave][lane] = h;
    __builtin_amdgcn_wave_barrier();
    v2f rz0 = {br, bz}, rz1 = {0.f, 0.f}, n0 = {bn, 0.f}, n1 = {0.f, 0.f};
    const float* hp = &hs[wave][dir * HID];
#pragma unroll
    for (int k = 0; k < HID; k += 8) {
      const float4 h0 = *reinterpret_cast<const float4*>(hp + k);
      const float4 h1 = *reinterpret_cast<const float4*>(hp + k + 4);
      rz0 = pkfma(wrz[k], v2f{h0.x, h0.x}, rz0);
      rz1 = pkfma(wrz[k + 1], v2f{h0.y, h0.y}, rz1);
      n0 = pkfma(wnn[k / 2], v2f{h0.x, h0.y}, n0);
      rz0 = pkfma(wrz[k + 2], v2f{h0.z, h0.z}, rz0);
      rz1 = pkfma(wrz[k + 3], v2f{h0.w, h0.w}, rz1);
      n1 = pkfma(wnn[k / 2 + 1], v2f{h0.z, h0.w}, n1);
      rz0 = pkfma(wrz[k + 4], v2f{h1.x, h1.x}, rz0);
      rz1 = pkfma(wrz[k + 5], v2f{h1.y, h1.y}, rz1);
      n0 = pkfma(wnn[k / 2 + 2], v2f{h1.x, h1.y}, n0);
      rz0 = pkfma(wrz[k + 6], v2f{h1.z, h1.z}, rz0);
      rz1 = pkfma(wrz[k + 7], v2f{h1.w, h1.w}, rz1);
      n1 = pkfma(wnn[k / 2 + 3], v2f{h1.z, h1.w}, n1);
    }
    __builtin_amdgcn_wave_barrier();
    // v_exp / v_rcp gates: libm expf / tanhf and the IEEE divisions were ~100 dependent instructions per step of a pure latency chain
    const float r = sigmoid_fast(g4[0] + (rz0.x + rz1.x));
    const float z = sigmoid_fast(g4[1] + (rz0.y + rz1.y));
    const float n = tanh_fast(g4[2] + r * ((n0.x + n1.x) + (n0.y + n1.y)));
    h = (1.f - z) * n + z * h;
    op[0] = fmaf(g4[3], rmul, h);      // (one fma that needs h: as `h + g4[3] * rmul` the multiply is hoisted three steps up, and the wait for its load with it)
    op += rd;
  };
  fetch(g0);
  fetch(g1);
  fetch(g2);
  // whole groups of four steps without an exit inside the body: a `break` between the steps routes through the loop latch in the
  // structurised CFG, and the static path "fetch(g0) -> break -> header -> step(g0)" made hipcc wait for all but 5 loads there
  int t = 0;
  for (; t + 3 < T; t += 4) {
    fetch(g3);
    step(g0, t);
    fetch(g0);
    step(g1, t + 1);
    fetch(g1);
    step(g2, t + 2);
    fetch(g2);
    step(g3, t + 3);
  }
  if (t < T) step(g0, t);              // T % 4 leftover steps: their gates are already in the ring
  if (t + 1 < T) step(g1, t + 1);
  if (t + 2 < T) step(g2, t + 2);
}

// ---------------------------------------------------------------------------------- tiny linear
// y[m][n] = act((x[m][:] (+ add[m % add_rows][:])) . w[n][:] + b[n]) -- one thread per output; for problems of a few MFLOP
__global__ void k_small_linear(const float* __restrict__ x, const float* __restrict__ add, int add_rows,
                               const float* __restrict__ w, const float* __restrict__ b, float* __restrict__ y, int M, int N,
                               int K, int act, float slope) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)M * N) return;
  const int m = idx / N, n = idx % N;
  const float* xr = x + (size_t)m * K;
  const float* ar = add ? add + (size_t)(m % add_rows) * K : nullptr;
  const float* wr = w + (size_t)n * K;
  float a = b ? b[n] : 0.f;
  for (int k = 0; k < K; ++k) a += (xr[k] + (ar ? ar[k] : 0.f)) * wr[k];
  y[idx] = apply_act(a, act, slope);
}

// The same product staged through LDS: a block owns 32 rows x 64 columns, x (+ add) and w arrive with coalesced loads in 64-deep
// K chunks, a thread accumulates 8 rows of one column.  Every output is the same chain of multiply-then-add over k = 0 .. K-1 that
// k_small_linear runs (bitwise-equal results); that kernel's per-thread walk over a weight ROW made every wave load 64 different
// cache lines per k: 8 - 105 us per call on the 1248 x 64 x 64 products of the TATT interpreter (5 calls per batch).
__global__ __launch_bounds__(256) void k_small_linear_tiled(const float* __restrict__ x, const float* __restrict__ add, int add_rows,
                                                             const float* __restrict__ w, const float* __restrict__ b,
                                                             float* __restrict__ y, int M, int N, int K, int act, float slope) {
  __shared__ float xs[32][64];
  __shared__ float ws[64][65];
  const int m0 = blockIdx.x * 32, n0 = blockIdx.y * 64;
  const int c = threadIdx.x & 63, r = threadIdx.x >> 6;
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = (b && n0 + c < N) ? b[n0 + c] : 0.f;
  for (int k0 = 0; k0 < K; k0 += 64) {
    const int kc = K - k0 < 64 ? K - k0 : 64;
    for (int e = threadIdx.x; e < 32 * 64; e += 256) {
      const int row = e >> 6, k = e & 63, m = m0 + row;
      float v = 0.f;
      if (m < M && k < kc) {
        v = x[(size_t)m * K + k0 + k];
        if (add) v = v + add[(size_t)(m % add_rows) * K + k0 + k];
      }
      xs[row][k] = v;
    }
    for (int e = threadIdx.x; e < 64 * 64; e += 256) {
      const int col = e >> 6, k = e & 63;
      ws[col][k] = (n0 + col < N && k < kc) ? w[(size_t)(n0 + col) * K + k0 + k] : 0.f;
    }
    __syncthreads();
    for (int k = 0; k < kc; ++k) {
      const float wv = ws[c][k];
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] += xs[r + 4 * i][k] * wv;
    }
    __syncthreads();
  }
  if (n0 + c < N) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int m = m0 + r + 4 * i;
      if (m < M) y[(size_t)m * N + n0 + c] = apply_act(acc[i], act, slope);
    }
  }
}

// ---------------------------------------------------------------------------------- encoder layer (L <= 32 tokens)
// One workgroup per image n.  src (N, L, E); pos (L, E).  Implements TransformerEncoder with ONE layer:
// layer input = src + src (transformer_v2.py:272-277), forward_post (455-469).  E = 64, 4 heads.
struct EncW {
  const float *in_w, *in_b, *out_w, *out_b, *l1_w, *l1_b, *l2_w, *l2_b, *n1_w, *n1_b, *n2_w, *n2_b;
};
template <int E, int NH>
__global__ __launch_bounds__(256) void k_encoder_layer(const float* __restrict__ src, const float* __restrict__ pos, EncW w,
                                                        float* __restrict__ mem, int L) {
  constexpr int LMAX = 32, D = E / NH;
  __shared__ float s2[LMAX][E], Q[LMAX][E], Kk[LMAX][E], V[LMAX][E], O[LMAX][E], P[NH][LMAX][LMAX];
  const int n = blockIdx.x, tid = threadIdx.x;
  for (int i = tid; i < L * E; i += 256) {
    const int l = i / E, e = i % E;
    const float v = 2.0f * src[((size_t)n * L + l) * E + e];
    s2[l][e] = v;
  }
  __syncthreads();
  // Every E x E weight matrix goes through LDS transposed (Wt[k][e], in the score buffer P, which is idle whenever a projection
  // runs): lanes own consecutive e, so the global rows w[e][:] they used to read were 256 bytes apart -- 64 cache lines per
  // wave and k step.  (88 -> see DESIGN; the weights stay in the nn.Module layouts at the ABI.)
  float* Wt = &P[0][0][0];
  static_assert(NH * LMAX * LMAX >= E * E, "the score buffer doubles as the weight stage");
  auto stage = [&](const float* W) {
    __syncthreads();
    for (int i = tid; i < E * E; i += 256) Wt[(i % E) * E + i / E] = W[i];
    __syncthreads();
  };
  // q = k = src + pos, v = src (transformer_v2.py:462-465): O holds src + pos for the q / k projections
  for (int i = tid; i < L * E; i += 256) O[i / E][i % E] = s2[i / E][i % E] + pos[i];
  for (int which = 0; which < 3; ++which) {
    stage(w.in_w + (size_t)which * E * E);
    for (int i = tid; i < L * E; i += 256) {
      const int l = i / E, e = i % E;
      float a = w.in_b[which * E + e];
      const float* in = which == 2 ? &s2[l][0] : &O[l][0];
      for (int k = 0; k < E; ++k) a += in[k] * Wt[k * E + e];
      (which == 0 ? Q : (which == 1 ? Kk : V))[l][e] = a;
    }
  }
  __syncthreads();
  const float scale = 1.0f / sqrtf((float)D);
  for (int i = tid; i < NH * L * L; i += 256) {
    const int hh = i / (L * L), a = (i / L) % L, bb = i % L;
    float sc = 0.f;
    for (int d = 0; d < D; ++d) sc += Q[a][hh * D + d] * Kk[bb][hh * D + d];
    P[hh][a][bb] = sc * scale;
  }
  __syncthreads();
  for (int i = tid; i < NH * L; i += 256) {
    const int hh = i / L, a = i % L;
    float mx = -INFINITY;
    for (int bb = 0; bb < L; ++bb) mx = fmaxf(mx, P[hh][a][bb]);
    float den = 0.f;
    for (int bb = 0; bb < L; ++bb) { const float p = expf(P[hh][a][bb] - mx); P[hh][a][bb] = p; den += p; }
    const float inv = 1.0f / den;
    for (int bb = 0; bb < L; ++bb) P[hh][a][bb] *= inv;
  }
  __syncthreads();
  for (int i = tid; i < L * E; i += 256) {
    const int l = i / E, e = i % E, hh = e / D;
    float a = 0.f;
    for (int bb = 0; bb < L; ++bb) a += P[hh][l][bb] * V[bb][e];
    O[l][e] = a;
  }
  __syncthreads();
  // out_proj + residual -> Q (reuse) ; then LayerNorm1 -> s2
  stage(w.out_w);
  for (int i = tid; i < L * E; i += 256) {
    const int l = i / E, e = i % E;
    float a = w.out_b[e];
    for (int k = 0; k < E; ++k) a += O[l][k] * Wt[k * E + e];
    Q[l][e] = s2[l][e] + a;
  }
  __syncthreads();
  auto layer_norm_rows = [&](float (*in)[E], float (*outp)[E], const float* g, const float* be) {
    const int wave = tid >> 6, lane = tid & 63;
    for (int l = wave; l < L; l += 4) {
      const float v = in[l][lane];   // E == 64 == wave width
      const float mean = wave_sum(v) * (1.0f / E);
      const float dlt = v - mean;
      const float var = wave_sum(dlt * dlt) * (1.0f / E);
      outp[l][lane] = dlt * (1.0f / sqrtf(var + 1e-5f)) * g[lane] + be[lane];
    }
  };
  layer_norm_rows(Q, s2, w.n1_w, w.n1_b);
  __syncthreads();
  stage(w.l1_w);
  for (int i = tid; i < L * E; i += 256) {
    const int l = i / E, e = i % E;
    float a = w.l1_b[e];
    for (int k = 0; k < E; ++k) a += s2[l][k] * Wt[k * E + e];
    O[l][e] = a > 0.f ? a : 0.f;
  }
  stage(w.l2_w);
  for (int i = tid; i < L * E; i += 256) {
    const int l = i / E, e = i % E;
    float a = w.l2_b[e];
    for (int k = 0; k < E; ++k) a += O[l][k] * Wt[k * E + e];
    Q[l][e] = s2[l][e] + a;
  }
  __syncthreads();
  layer_norm_rows(Q, Kk, w.n2_w, w.n2_b);
  __syncthreads();
  for (int i = tid; i < L * E; i += 256) mem[(size_t)n * L * E + i] = Kk[i / E][i % E];
}

// ---------------------------------------------------------------------------------- cross attention (S <= 32 keys)
// q (N, L, E) projected queries; k, v (N, S, E) projected keys/values.  One wave = 64 queries, lane = query;
// all NH heads per lane.  Optional head-averaged weights pw (N, L, S) (MultiheadAttention need_weights).
template <int E, int NH>
__global__ __launch_bounds__(256) void k_cross_attn(const float* __restrict__ q, const float* __restrict__ k,
                                                     const float* __restrict__ v, float* __restrict__ o, float* __restrict__ pw,
                                                     int L, int S) {
  constexpr int SMAX = 32, D = E / NH;
  __shared__ __attribute__((aligned(16))) float Ks[SMAX][E], Vs[SMAX][E];
  const int n = blockIdx.y, tid = threadIdx.x;
  for (int i = tid; i < S * E; i += 256) {
    Ks[i / E][i % E] = k[(size_t)n * S * E + i];
    Vs[i / E][i % E] = v[(size_t)n * S * E + i];
  }
  __syncthreads();
  const int l = blockIdx.x * 256 + tid;
  if (l >= L) return;
  const float* qr = q + ((size_t)n * L + l) * E;
  float* orow = o + ((size_t)n * L + l) * E;
  const float scale = 1.0f / sqrtf((float)D);
  float pavg[SMAX];
#pragma unroll
  for (int s = 0; s < SMAX; ++s) pavg[s] = 0.f;
#pragma unroll
  for (int hh = 0; hh < NH; ++hh) {
    float qv[D];
#pragma unroll
    for (int d = 0; d < D; d += 4) {
      const float4 t4 = *reinterpret_cast<const float4*>(qr + hh * D + d);
      qv[d] = t4.x * scale; qv[d + 1] = t4.y * scale; qv[d + 2] = t4.z * scale; qv[d + 3] = t4.w * scale;
    }
    float sc[SMAX];
    float mx = -INFINITY;
#pragma unroll
    for (int s = 0; s < SMAX; ++s) {
      float a = -INFINITY;
      if (s < S) {
        a = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) a += qv[d] * Ks[s][hh * D + d];
      }
      sc[s] = a;
      mx = fmaxf(mx, a);
    }
    float den = 0.f;
#pragma unroll
    for (int s = 0; s < SMAX; ++s) { sc[s] = (s < S) ? expf(sc[s] - mx) : 0.f; den += sc[s]; }
    const float inv = 1.0f / den;
    float acc[D];
#pragma unroll
    for (int d = 0; d < D; ++d) acc[d] = 0.f;
#pragma unroll
    for (int s = 0; s < SMAX; ++s) {
      const float p = sc[s] * inv;
      pavg[s] += p;
      if (s < S) {
#pragma unroll
        for (int d = 0; d < D; ++d) acc[d] += p * Vs[s][hh * D + d];
      }
    }
#pragma unroll
    for (int d = 0; d < D; d += 4)
      *reinterpret_cast<float4*>(orow + hh * D + d) = make_float4(acc[d], acc[d + 1], acc[d + 2], acc[d + 3]);
  }
  if (pw) {
    float* pr = pw + ((size_t)n * L + l) * S;
#pragma unroll
    for (int s = 0; s < SMAX; ++s) if (s < S) pr[s] = pavg[s] * (1.0f / NH);
  }
}

// ---------------------------------------------------------------------------------- (x + res) -> LayerNorm (E = 64)
// y = LN(x + res) * g + b ; optionally acc_out += alpha * LN2(y) (decoder.norm on the intermediate, then mean over layers)
__global__ __launch_bounds__(256) void k_add_layernorm64(const float* __restrict__ x, const float* __restrict__ res,
                                                          const float* __restrict__ g, const float* __restrict__ b,
                                                          float* __restrict__ y, const float* __restrict__ g2,
                                                          const float* __restrict__ b2, float* __restrict__ acc_out, float alpha,
                                                          int accumulate, long M) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= M) return;
  float v = x[row * 64 + lane] + (res ? res[row * 64 + lane] : 0.f);
  float mean = wave_sum(v) * (1.0f / 64);
  float d = v - mean;
  float var = wave_sum(d * d) * (1.0f / 64);
  const float o = d * (1.0f / sqrtf(var + 1e-5f)) * g[lane] + b[lane];
  y[row * 64 + lane] = o;
  if (acc_out) {
    mean = wave_sum(o) * (1.0f / 64);
    d = o - mean;
    var = wave_sum(d * d) * (1.0f / 64);
    const float o2 = (d * (1.0f / sqrtf(var + 1e-5f)) * g2[lane] + b2[lane]) * alpha;
    acc_out[row * 64 + lane] = accumulate ? acc_out[row * 64 + lane] + o2 : o2;
  }
}

// ---------------------------------------------------------------------------------- GRU gate step (query-embed GRU)
// gi (R, 3H) constant input projection (bias folded), gh (R, 3H) = h_prev . W_hh^T + b_hh ; h (R, H) updated in place;
// hist (R, H) receives a copy (the GRU output at this step).
__global__ void k_gru_gate(const float* __restrict__ gi, const float* __restrict__ gh, float* __restrict__ h,
                           float* __restrict__ hist, long hist_row_stride, int R, int H) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)R * H) return;
  const int r_ = idx / H, j = idx % H;
  const float* a = gi + (size_t)r_ * 3 * H;
  const float* c = gh + (size_t)r_ * 3 * H;
  const float r = sigmoid_f(a[j] + c[j]);
  const float z = sigmoid_f(a[H + j] + c[H + j]);
  const float n = tanhf(a[2 * H + j] + r * c[2 * H + j]);
  const float hn = (1.f - z) * n + z * h[idx];
  h[idx] = hn;
  hist[(size_t)r_ * hist_row_stride + j] = hn;
}

// TPGSR (tsrn.py:226-227): the InfoGen map (N, 1, Win, C) stretched to the LR feature map (N, H, W, C) by F.interpolate(..., mode
// 'bilinear', align_corners=True): the single source row is repeated over H, columns x -> x (Win - 1) / (W - 1)
__global__ void k_tl_interp(const float* __restrict__ in, float* __restrict__ out, int N, int Win, int C, int H, int W) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int c4 = C / 4;
  if (idx >= (long)N * H * W * c4) return;
  const int c = (idx % c4) * 4;
  const int x = (idx / c4) % W, n = idx / ((long)c4 * W * H);
  const float sx = W > 1 ? (float)x * ((float)(Win - 1) / (float)(W - 1)) : 0.f;
  int x0 = (int)sx;
  if (x0 > Win - 1) x0 = Win - 1;
  const int x1 = x0 + 1 < Win ? x0 + 1 : Win - 1;
  const float l1 = sx - (float)x0, l0 = 1.f - l1;
  const float4 a = *reinterpret_cast<const float4*>(in + ((size_t)n * Win + x0) * C + c);
  const float4 b = *reinterpret_cast<const float4*>(in + ((size_t)n * Win + x1) * C + c);
  *reinterpret_cast<float4*>(out + idx * 4) = make_float4(l0 * a.x + l1 * b.x, l0 * a.y + l1 * b.y, l0 * a.z + l1 * b.z, l0 * a.w + l1 * b.w);
}

}  // namespace

extern "C" {

int dpmn_bigru_f32(const float* gi, const float* w_hh, const float* b_hh, const float* res, float* out, int nseq, int T,
                   int inner, long outer_stride, long inner_stride, long step_stride, int hidden, dpmn_stream_t stream) {
  DPMN_REQUIRE(gi && w_hh && b_hh && out && nseq > 0 && T > 0 && inner > 0, "bigru: bad arguments");
  DPMN_REQUIRE(hidden == 32, "bigru: built for hidden_units=32 per direction (hd_u default, main.py:47)");
  ProfScope prof(PT_BIGRU, as_stream(stream), 2.0 * 2 * 3 * 32 * 32 * (double)nseq * T, 4.0 * (2 * 96 + 64 + 64) * (double)nseq * T);
  static const int pk_env = getenv("DPMN_BIGRU_PK") ? atoi(getenv("DPMN_BIGRU_PK")) : -1;      // A/B switch: 1 / 0 force the packed / scalar form
  const bool pk = pk_env >= 0 ? pk_env != 0 : !(g_dpmn_x3 || g_dpmn_bf16);
  if (pk)
    hipLaunchKernelGGL((k_bigru<32, true>), dim3((unsigned)((nseq + 3) / 4)), dim3(256), 0, as_stream(stream), gi, w_hh, b_hh, res, out,
                       nseq, T, inner, outer_stride, inner_stride, step_stride);
  else
    hipLaunchKernelGGL((k_bigru<32, false>), dim3((unsigned)((nseq + 3) / 4)), dim3(256), 0, as_stream(stream), gi, w_hh, b_hh, res, out,
                       nseq, T, inner, outer_stride, inner_stride, step_stride);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

int dpmn_small_linear_f32(const float* x, const float* add, int add_rows, const float* w, const float* b, float* y, int M,
                          int N, int K, int act, float slope, dpmn_stream_t stream) {
  DPMN_REQUIRE(x && w && y && M > 0 && N > 0 && K > 0, "small_linear: bad arguments");
  static const int tiled = getenv("DPMN_SMALL_LINEAR_TILED") ? atoi(getenv("DPMN_SMALL_LINEAR_TILED")) : 1;
  if (tiled)
    hipLaunchKernelGGL(k_small_linear_tiled, dim3((unsigned)((M + 31) / 32), (unsigned)((N + 63) / 64)), dim3(256), 0, as_stream(stream), x,
                       add, add_rows > 0 ? add_rows : 1, w, b, y, M, N, K, act, slope);
  else
    hipLaunchKernelGGL(k_small_linear, dim3((unsigned)(((long)M * N + 255) / 256)), dim3(256), 0, as_stream(stream), x, add,
                       add_rows > 0 ? add_rows : 1, w, b, y, M, N, K, act, slope);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

int dpmn_tatt_encoder_layer_f32(const float* src, const float* pos, const float* const* w12, float* mem, int N, int L,
                                int E, int nhead, dpmn_stream_t stream) {
  DPMN_REQUIRE(src && pos && w12 && mem && N > 0, "encoder_layer: bad arguments");
  DPMN_REQUIRE(E == 64 && nhead == 4 && L <= 32, "encoder_layer: built for d_model=64, 4 heads, <=32 tokens (tatt.py:171-178)");
  EncW w{w12[0], w12[1], w12[2], w12[3], w12[4], w12[5], w12[6], w12[7], w12[8], w12[9], w12[10], w12[11]};
  hipLaunchKernelGGL((k_encoder_layer<64, 4>), dim3(N), dim3(256), 0, as_stream(stream), src, pos, w, mem, L);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

int dpmn_cross_attn_f32(const float* q, const float* k, const float* v, float* o, float* pw, int N, int L, int S, int E,
                        int nhead, dpmn_stream_t stream) {
  DPMN_REQUIRE(q && k && v && o && N > 0, "cross_attn: bad arguments");
  DPMN_REQUIRE(E == 64 && nhead == 4 && S <= 32, "cross_attn: built for d_model=64, 4 heads, <=32 keys");
  hipLaunchKernelGGL((k_cross_attn<64, 4>), dim3(cdiv(L, 256), N), dim3(256), 0, as_stream(stream), q, k, v, o, pw, L, S);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

int dpmn_add_layernorm64_f32(const float* x, const float* res, const float* g, const float* b, float* y, const float* g2,
                             const float* b2, float* acc_out, float alpha, int accumulate, long M, dpmn_stream_t stream) {
  DPMN_REQUIRE(x && g && b && y && M > 0, "add_layernorm64: bad arguments");
  DPMN_REQUIRE(!acc_out || (g2 && b2), "add_layernorm64: second norm parameters required with acc_out");
  hipLaunchKernelGGL(k_add_layernorm64, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, as_stream(stream), x, res, g, b, y, g2, b2,
                     acc_out, alpha, accumulate, M);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

int dpmn_gru_gate_f32(const float* gi, const float* gh, float* h, float* hist, long hist_row_stride, int R, int H,
                      dpmn_stream_t stream) {
  DPMN_REQUIRE(gi && gh && h && hist && R > 0 && H > 0, "gru_gate: bad arguments");
  const long total = (long)R * H;
  hipLaunchKernelGGL(k_gru_gate, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), gi, gh, h, hist,
                     hist_row_stride, R, H);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

int dpmn_tl_interp_f32(const float* in, float* out, int N, int Win, int C, int H, int W, dpmn_stream_t stream) {
  DPMN_REQUIRE(in && out && N > 0 && Win > 0 && H > 0 && W > 0 && C % 4 == 0, "tl_interp: bad arguments");
  const long total = (long)N * H * W * (C / 4);
  hipLaunchKernelGGL(k_tl_interp, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), in, out, N, Win, C, H, W);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

}  // extern "C"
