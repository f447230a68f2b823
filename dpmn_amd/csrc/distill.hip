// Native DistillModule (distill_module.py:4-31): two 3x3 convs over 6 / 3 image channels, train-mode BatchNorm2d(3), ReLU and the
// L1 between the two feature maps; returns (loss, feature_cat).  Forward = 4 launches, backward = 5 launches, all on NCHW
// (B, 3, H, W) tensors as the trainer hands them over (super_resolution.py:245-263) -- the module is 3-6 channels wide and purely
// memory / latency bound, so it gets its own direct kernels instead of the channel-padded NHWC implicit-GEMM path (which cost ~50
// launches and ~30 host-side tensor ops per module and step).
// Every reduction is a per-block partial row added in block order by a one-block finish kernel (BatchNorm statistics and the loss
// in fp64): bitwise reproducible, no atomics.
#include "common.h"

namespace {

constexpr int TPB = 256;

struct DistillW {
  float wc[3 * 6 * 9];      // conv_cat_feature.weight (3, 6, 3, 3)
  float wf[3 * 3 * 9];      // conv_feature.weight (3, 3, 3, 3)
  float bc[3], bf[3];
};
struct DistillWPtr { const float *wc, *bc, *wf, *bf; };
// the 249 weights of both convs, staged once per block (ends with a barrier)
__device__ __forceinline__ void stage_w(DistillW& w, const DistillWPtr& p) {
  for (int i = threadIdx.x; i < 162; i += TPB) w.wc[i] = p.wc[i];
  for (int i = threadIdx.x; i < 81; i += TPB) w.wf[i] = p.wf[i];
  if (threadIdx.x < 3) { w.bc[threadIdx.x] = p.bc ? p.bc[threadIdx.x] : 0.f; w.bf[threadIdx.x] = p.bf ? p.bf[threadIdx.x] : 0.f; }
  __syncthreads();
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += xshfl_v(v, o);
  return v;
}

// block-wide sums of NV per-thread values in fixed order (wave shuffle tree, then wave 0..3): row[v] = sum
template <int NV, typename T>
__device__ __forceinline__ void block_rows(const T (&val)[NV], T* __restrict__ row, T (*sm)[NV]) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    T s = val[v];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += xshfl_v(s, o);
    if (lane == 0) sm[wave][v] = s;
  }
  __syncthreads();
  for (int v = threadIdx.x; v < NV; v += TPB) row[v] = ((sm[0][v] + sm[1][v]) + sm[2][v]) + sm[3][v];
}


// sum of `nrows` rows of NV values in a FIXED order with 16 lanes per value: lane l adds the rows l, l + 16, ... (two independent
// chains), the 16 lane sums are then added in lane order by the value's first lane.  One block of NV * 16 threads (<= 1024).
template <int NV, typename T>
__device__ __forceinline__ T rows_sum16(const T* __restrict__ rows, int nrows, T (*sm)[16]) {
  const int v = threadIdx.x >> 4, l = threadIdx.x & 15;
  T s0 = 0, s1 = 0;
  if (v < NV) {
    int z = l;
    for (; z + 16 < nrows; z += 32) { s0 += rows[(size_t)z * NV + v]; s1 += rows[(size_t)(z + 16) * NV + v]; }
    if (z < nrows) s0 += rows[(size_t)z * NV + v];
    sm[v][l] = s0 + s1;
  }
  __syncthreads();
  T tot = 0;
  if (v < NV && l == 0)
    for (int i = 0; i < 16; ++i) tot += sm[v][i];
  return tot;      // valid in lane 0 of each value's 16-lane group
}

// ---- forward 1: both convs (raw outputs r: (B, 6, H, W) = [conv_cat (3) | conv_feature (3)]) + per-block (sum, sum of squares)
__global__ __launch_bounds__(TPB) void k_distill_conv_fwd(const float* __restrict__ xd, const float* __restrict__ xs, DistillWPtr wp,
                                                          float* __restrict__ r, double* __restrict__ stat_rows, int B, int H, int W) {
  __shared__ double sm[4][12];
  __shared__ DistillW w;
  stage_w(w, wp);
  const long HW = (long)H * W, total = (long)B * HW;
  const long idx = (long)blockIdx.x * TPB + threadIdx.x;
  double val[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) val[i] = 0.0;
  if (idx < total) {
    const int b = idx / HW, p = idx % HW, y = p / W, x = p % W;
    float a1[3] = {w.bc[0], w.bc[1], w.bc[2]}, a2[3] = {w.bf[0], w.bf[1], w.bf[2]};
#pragma unroll
    for (int ci = 0; ci < 6; ++ci) {
      const float* src = (ci < 3 ? xd : xs) + ((size_t)b * 3 + (ci % 3)) * HW;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int yy = y + ky - 1;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int xx = x + kx - 1;
          const float v = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? src[(size_t)yy * W + xx] : 0.f;
#pragma unroll
          for (int co = 0; co < 3; ++co) a1[co] += w.wc[((co * 6 + ci) * 3 + ky) * 3 + kx] * v;
          if (ci >= 3) {
#pragma unroll
            for (int co = 0; co < 3; ++co) a2[co] += w.wf[((co * 3 + ci - 3) * 3 + ky) * 3 + kx] * v;
          }
        }
      }
    }
#pragma unroll
    for (int co = 0; co < 3; ++co) {
      r[((size_t)b * 6 + co) * HW + p] = a1[co];
      r[((size_t)b * 6 + 3 + co) * HW + p] = a2[co];
      val[co] = a1[co]; val[3 + co] = a2[co];
      val[6 + co] = (double)a1[co] * a1[co]; val[9 + co] = (double)a2[co] * a2[co];
    }
  }
  block_rows<12, double>(val, stat_rows + (size_t)blockIdx.x * 12, sm);
}

// ---- forward 2 (one block): statistics rows added in block order -> scale / shift / mean / rstd, running statistics
// state: [scale (6) | shift (6) | mean (6) | rstd (6)]
__global__ void k_distill_bn_finalize(const double* __restrict__ stat_rows, int nrows, double count, const float* __restrict__ g1,
                                      const float* __restrict__ b1, const float* __restrict__ g2, const float* __restrict__ b2,
                                      float* __restrict__ rm1, float* __restrict__ rv1, long long* __restrict__ nbt1,
                                      float* __restrict__ rm2, float* __restrict__ rv2, long long* __restrict__ nbt2, float eps,
                                      float momentum, float* __restrict__ state) {
  __shared__ double tot[12];
  __shared__ double sm16[12][16];
  {
    const double s = rows_sum16<12, double>(stat_rows, nrows, sm16);
    if ((threadIdx.x & 15) == 0 && threadIdx.x < 192) tot[threadIdx.x >> 4] = s;
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    const int c = threadIdx.x, cc = c % 3;
    const double mean_d = tot[c] / count;
    const float mean = (float)mean_d;
    float var = (float)(tot[6 + c] / count - mean_d * mean_d);
    var = var > 0.f ? var : 0.f;
    const float rstd = 1.0f / sqrtf(var + eps);
    const float gamma = c < 3 ? g1[cc] : g2[cc], beta = c < 3 ? b1[cc] : b2[cc];
    const float s = gamma * rstd;
    state[c] = s; state[6 + c] = beta - mean * s; state[12 + c] = mean; state[18 + c] = rstd;
    float* rm = c < 3 ? rm1 : rm2;
    float* rv = c < 3 ? rv1 : rv2;
    if (rm) {
      rm[cc] = (1.f - momentum) * rm[cc] + momentum * mean;
      rv[cc] = (1.f - momentum) * rv[cc] + momentum * var * (float)(count / (count - 1.0));
    }
  }
  if (threadIdx.x == 0 && nbt1) { *nbt1 += 1; *nbt2 += 1; }
}

// eval mode: the running statistics as a fixed affine
__global__ void k_distill_bn_eval(const float* __restrict__ g1, const float* __restrict__ b1, const float* __restrict__ g2,
                                  const float* __restrict__ b2, const float* __restrict__ rm1, const float* __restrict__ rv1,
                                  const float* __restrict__ rm2, const float* __restrict__ rv2, float eps, float* __restrict__ state) {
  if (threadIdx.x < 6) {
    const int c = threadIdx.x, cc = c % 3;
    const float mean = c < 3 ? rm1[cc] : rm2[cc], var = c < 3 ? rv1[cc] : rv2[cc];
    const float rstd = 1.0f / sqrtf(var + eps), gamma = c < 3 ? g1[cc] : g2[cc], beta = c < 3 ? b1[cc] : b2[cc];
    state[c] = gamma * rstd; state[6 + c] = beta - mean * gamma * rstd; state[12 + c] = mean; state[18 + c] = rstd;
  }
}

// ---- forward 3: feature_cat = relu(bn_1(r1)) (written), feature_shallow = relu(bn_2(r2)); per-block sum |f1 - f2|
__global__ __launch_bounds__(TPB) void k_distill_act_loss(const float* __restrict__ r, const float* __restrict__ state,
                                                          float* __restrict__ feat, double* __restrict__ loss_rows, int B, int H, int W) {
  __shared__ double sm[4][1];
  const long HW = (long)H * W, total = (long)B * HW;
  const long idx = (long)blockIdx.x * TPB + threadIdx.x;
  double val[1] = {0.0};
  if (idx < total) {
    const int b = idx / HW, p = idx % HW;
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float f1 = fmaxf(r[((size_t)b * 6 + c) * HW + p] * state[c] + state[6 + c], 0.f);
      const float f2 = fmaxf(r[((size_t)b * 6 + 3 + c) * HW + p] * state[3 + c] + state[9 + c], 0.f);
      feat[((size_t)b * 3 + c) * HW + p] = f1;
      acc += fabsf(f1 - f2);
    }
    val[0] = acc;
  }
  block_rows<1, double>(val, loss_rows + blockIdx.x, sm);
}

__global__ void k_distill_loss_finalize(const double* __restrict__ loss_rows, int nrows, double inv_n, float* __restrict__ loss) {
  __shared__ double sm16[1][16];
  const double s = rows_sum16<1, double>(loss_rows, nrows, sm16);
  if (threadIdx.x == 0) *loss = (float)(s * inv_n);
}

// gradient wrt the two BatchNorm outputs before the ReLU: G1 = (gl * sign(f1 - f2) + dfeat) [f1 > 0], G2 = -gl * sign(f1 - f2) [f2 > 0]
__device__ __forceinline__ void distill_G(const float* __restrict__ r, const float* __restrict__ state, const float* __restrict__ dfeat,
                                          float gl, int b, long p, long HW, float (&G)[6], float (&xh)[6]) {
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float r1 = r[((size_t)b * 6 + c) * HW + p], r2 = r[((size_t)b * 6 + 3 + c) * HW + p];
    const float p1 = r1 * state[c] + state[6 + c], p2 = r2 * state[3 + c] + state[9 + c];
    const float f1 = fmaxf(p1, 0.f), f2 = fmaxf(p2, 0.f);
    const float sg = f1 > f2 ? 1.f : (f1 < f2 ? -1.f : 0.f);      // torch: sign(0) = 0
    const float d1 = gl * sg + (dfeat ? dfeat[((size_t)b * 3 + c) * HW + p] : 0.f);
    G[c] = p1 > 0.f ? d1 : 0.f;
    G[3 + c] = p2 > 0.f ? -gl * sg : 0.f;
    xh[c] = (r1 - state[12 + c]) * state[18 + c];
    xh[3 + c] = (r2 - state[15 + c]) * state[21 + c];
  }
}

// ---- backward 1: per-block (sum G, sum G xhat) for the 6 BatchNorm channels
__global__ __launch_bounds__(TPB) void k_distill_bn_bwd_stats(const float* __restrict__ r, const float* __restrict__ state,
                                                              const float* __restrict__ dfeat, const float* __restrict__ gloss, float inv_n,
                                                              double* __restrict__ rows, int B, int H, int W) {
  __shared__ double sm[4][12];
  const long HW = (long)H * W, total = (long)B * HW;
  const long idx = (long)blockIdx.x * TPB + threadIdx.x;
  double val[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) val[i] = 0.0;
  if (idx < total) {
    float G[6], xh[6];
    distill_G(r, state, dfeat, gloss[0] * inv_n, idx / HW, idx % HW, HW, G, xh);
#pragma unroll
    for (int c = 0; c < 6; ++c) { val[c] = G[c]; val[6 + c] = (double)G[c] * xh[c]; }
  }
  block_rows<12, double>(val, rows + (size_t)blockIdx.x * 12, sm);
}

// ---- backward 2 (one block): rows -> dbeta, dgamma (accumulated into the parameter gradients) and the per-channel means
// coef: [mean G (6) | mean G xhat (6)]
__global__ void k_distill_bn_bwd_finalize(const double* __restrict__ rows, int nrows, double count, float* __restrict__ dg1,
                                          float* __restrict__ db1, float* __restrict__ dg2, float* __restrict__ db2,
                                          float* __restrict__ coef) {
  __shared__ double sm16[12][16];
  const double s = rows_sum16<12, double>(rows, nrows, sm16);
  if ((threadIdx.x & 15) == 0 && threadIdx.x < 192) {
    const int v = threadIdx.x >> 4, c = v % 6, cc = c % 3;
    if (v < 6) { float* db = c < 3 ? db1 : db2; db[cc] += (float)s; }
    else { float* dg = c < 3 ? dg1 : dg2; dg[cc] += (float)s; }
    coef[v] = (float)(s / count);
  }
}

// ---- backward 3: dr = gamma rstd (G - mean G - xhat mean(G xhat)) for the 6 raw conv outputs (B, 6, H, W)
__global__ __launch_bounds__(TPB) void k_distill_dr(const float* __restrict__ r, const float* __restrict__ state, const float* __restrict__ dfeat,
                                                    const float* __restrict__ gloss, float inv_n, const float* __restrict__ coef,
                                                    float* __restrict__ dr, int B, int H, int W) {
  const long HW = (long)H * W, total = (long)B * HW;
  const long idx = (long)blockIdx.x * TPB + threadIdx.x;
  if (idx >= total) return;
  const int b = idx / HW;
  const long p = idx % HW;
  float G[6], xh[6];
  distill_G(r, state, dfeat, gloss[0] * inv_n, b, p, HW, G, xh);
#pragma unroll
  for (int c = 0; c < 6; ++c) dr[((size_t)b * 6 + c) * HW + p] = state[c] * (G[c] - coef[c] - xh[c] * coef[6 + c]);      // state[c] = gamma rstd
}

// ---- backward 4: data gradients (transposed convs) + per-block weight / bias gradient rows
// row: [dW_cat (162) | dW_feat (81) | db_cat (3) | db_feat (3)]
constexpr int NWG = 162 + 81 + 6;
__global__ __launch_bounds__(TPB) void k_distill_conv_bwd(const float* __restrict__ xd, const float* __restrict__ xs,
                                                          const float* __restrict__ dr, DistillWPtr wp, float* __restrict__ dxd,
                                                          float* __restrict__ dxs, float* __restrict__ wrows, int B, int H, int W) {
  __shared__ float sm[4][NWG];
  __shared__ DistillW w;
  stage_w(w, wp);
  const long HW = (long)H * W, total = (long)B * HW;
  const long idx = (long)blockIdx.x * TPB + threadIdx.x;
  const bool ok = idx < total;
  const int b = ok ? idx / HW : 0;
  const int p = ok ? idx % HW : 0, y = p / W, x = p % W;
  // data gradients: dx[ci][y][x] = sum_co sum_tap W[co][ci][ky][kx] dr[co][y - ky + 1][x - kx + 1]
  float gd[3] = {0.f, 0.f, 0.f}, gs[3] = {0.f, 0.f, 0.f};
  float drc[6];                 // dr at this pixel (for the weight gradients)
#pragma unroll
  for (int co = 0; co < 6; ++co) drc[co] = ok ? dr[((size_t)b * 6 + co) * HW + p] : 0.f;
  if (ok) {
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int yy = y - ky + 1;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int xx = x - kx + 1;
        if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
        const size_t q = (size_t)yy * W + xx;
#pragma unroll
        for (int co = 0; co < 3; ++co) {
          const float d1 = dr[((size_t)b * 6 + co) * HW + q], d2 = dr[((size_t)b * 6 + 3 + co) * HW + q];
#pragma unroll
          for (int ci = 0; ci < 3; ++ci) {
            gd[ci] += w.wc[((co * 6 + ci) * 3 + ky) * 3 + kx] * d1;
            gs[ci] += w.wc[((co * 6 + 3 + ci) * 3 + ky) * 3 + kx] * d1 + w.wf[((co * 3 + ci) * 3 + ky) * 3 + kx] * d2;
          }
        }
      }
    }
    if (dxd) {
#pragma unroll
      for (int ci = 0; ci < 3; ++ci) dxd[((size_t)b * 3 + ci) * HW + p] = gd[ci];
    }
    if (dxs) {
#pragma unroll
      for (int ci = 0; ci < 3; ++ci) dxs[((size_t)b * 3 + ci) * HW + p] = gs[ci];
    }
  }
  // weight gradients: dW[co][ci][ky][kx] = sum_pixels dr[co][pix] * x[ci][pix + (ky - 1, kx - 1)], one value at a time through the
  // wave shuffle tree (fixed order), the four waves added in wave order
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll 1
  for (int ci = 0; ci < 6; ++ci) {
    const float* src = (ci < 3 ? xd : xs) + ((size_t)b * 3 + (ci % 3)) * HW;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int yy = y + ky - 1;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int xx = x + kx - 1;
        const float v = (ok && yy >= 0 && yy < H && xx >= 0 && xx < W) ? src[(size_t)yy * W + xx] : 0.f;
#pragma unroll
        for (int co = 0; co < 3; ++co) {
          const float s = wave_sum(drc[co] * v);
          if (lane == 0) sm[wave][((co * 6 + ci) * 3 + ky) * 3 + kx] = s;
        }
        if (ci >= 3) {
#pragma unroll
          for (int co = 0; co < 3; ++co) {
            const float s = wave_sum(drc[3 + co] * v);
            if (lane == 0) sm[wave][162 + ((co * 3 + ci - 3) * 3 + ky) * 3 + kx] = s;
          }
        }
      }
    }
  }
#pragma unroll
  for (int co = 0; co < 6; ++co) {
    const float s = wave_sum(drc[co]);
    if (lane == 0) sm[wave][243 + co] = s;
  }
  __syncthreads();
  for (int v = threadIdx.x; v < NWG; v += TPB) wrows[(size_t)blockIdx.x * NWG + v] = ((sm[0][v] + sm[1][v]) + sm[2][v]) + sm[3][v];
}

// ---- backward 5: weight-gradient rows added in block order into the parameter gradients
__global__ __launch_bounds__(TPB) void k_distill_wgrad_finalize(const float* __restrict__ wrows, int nrows, float* __restrict__ dwc,
                                                                float* __restrict__ dbc, float* __restrict__ dwf, float* __restrict__ dbf) {
  // 16 values per block, 16 lanes per value (fixed order: lane l adds the rows l, l + 16, ..., the lanes are added in lane order)
  __shared__ float sm[16][16];
  const int v = blockIdx.x * 16 + (threadIdx.x >> 4), l = threadIdx.x & 15;
  float s0 = 0.f, s1 = 0.f;
  if (v < NWG) {
    int z = l;
    for (; z + 16 < nrows; z += 32) { s0 += wrows[(size_t)z * NWG + v]; s1 += wrows[(size_t)(z + 16) * NWG + v]; }
    if (z < nrows) s0 += wrows[(size_t)z * NWG + v];
  }
  sm[threadIdx.x >> 4][l] = s0 + s1;
  __syncthreads();
  if (v >= NWG || l != 0) return;
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += sm[threadIdx.x >> 4][i];
  if (v < 162) dwc[v] += s;
  else if (v < 243) dwf[v - 162] += s;
  else if (v < 246) dbc[v - 243] += s;
  else dbf[v - 246] += s;
}

struct DWs {
  double* rows;       // (nb, 12) statistics rows (forward: sum, sum of squares; backward: sum G, sum G xhat)
  double* lrows;      // (nb) loss rows
  float* coef;        // 12
  float* dr;          // (B, 6, H, W)
  float* wrows;       // (nb, NWG)
  size_t total;
};
DWs carve(int B, int H, int W, char* base) {
  const size_t nb = ((size_t)B * H * W + TPB - 1) / TPB;
  size_t off = 0;
  auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += (bytes + 255) / 256 * 256; return p; };
  DWs s;
  s.rows = reinterpret_cast<double*>(take(nb * 12 * sizeof(double)));
  s.lrows = reinterpret_cast<double*>(take(nb * sizeof(double)));
  s.coef = reinterpret_cast<float*>(take(12 * sizeof(float)));
  s.dr = reinterpret_cast<float*>(take((size_t)B * 6 * H * W * sizeof(float)));
  s.wrows = reinterpret_cast<float*>(take(nb * NWG * sizeof(float)));
  s.total = off;
  return s;
}

}  // namespace

extern "C" {

size_t dpmn_distill_workspace_bytes(int B, int H, int W) {
  return B > 0 && H > 0 && W > 0 ? carve(B, H, W, nullptr).total : 0;
}

int dpmn_distill_forward_f32(const dpmn_distill_params* p, const float* x_deep, const float* x_shallow, int training, float* r,
                             float* state, float* feat, float* loss, void* workspace, size_t workspace_bytes, int B, int H, int W,
                             dpmn_stream_t stream) {
  DPMN_REQUIRE(p && x_deep && x_shallow && r && state && feat && loss && workspace, "distill_forward: null pointer");
  DPMN_REQUIRE(p->conv_cat_w && p->conv_feat_w && p->bn1_w && p->bn1_b && p->bn2_w && p->bn2_b && p->bn1_rm && p->bn1_rv && p->bn2_rm &&
               p->bn2_rv, "distill_forward: incomplete parameter set");
  DPMN_REQUIRE(B > 0 && H > 0 && W > 0 && (long)B * H * W > 1, "distill_forward: empty batch");
  DWs s = carve(B, H, W, static_cast<char*>(workspace));
  if (s.total > workspace_bytes) return dpmn_set_error(DPMN_ERR_WORKSPACE, "distill_forward: workspace too small");
  hipStream_t st = as_stream(stream);
  const long total = (long)B * H * W;
  const int nb = (int)((total + TPB - 1) / TPB);
  const DistillWPtr wp{p->conv_cat_w, p->conv_cat_b, p->conv_feat_w, p->conv_feat_b};
  hipLaunchKernelGGL(k_distill_conv_fwd, dim3(nb), dim3(TPB), 0, st, x_deep, x_shallow, wp, r, s.rows, B, H, W);
  DPMN_CHECK_LAUNCH();
  if (training)
    hipLaunchKernelGGL(k_distill_bn_finalize, dim3(1), dim3(192), 0, st, s.rows, nb, (double)total, p->bn1_w, p->bn1_b, p->bn2_w, p->bn2_b,
                       p->bn1_rm, p->bn1_rv, p->bn1_nbt, p->bn2_rm, p->bn2_rv, p->bn2_nbt, 1e-5f, 0.1f, state);
  else
    hipLaunchKernelGGL(k_distill_bn_eval, dim3(1), dim3(64), 0, st, p->bn1_w, p->bn1_b, p->bn2_w, p->bn2_b, p->bn1_rm, p->bn1_rv, p->bn2_rm,
                       p->bn2_rv, 1e-5f, state);
  DPMN_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_distill_act_loss, dim3(nb), dim3(TPB), 0, st, r, state, feat, s.lrows, B, H, W);
  DPMN_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_distill_loss_finalize, dim3(1), dim3(64), 0, st, s.lrows, nb, 1.0 / (3.0 * (double)total), loss);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

int dpmn_distill_backward_f32(const dpmn_distill_params* p, const dpmn_distill_grads* g, const float* x_deep, const float* x_shallow,
                              const float* r, const float* state, const float* dfeat, const float* gloss, float* dx_deep,
                              float* dx_shallow, void* workspace, size_t workspace_bytes, int B, int H, int W, dpmn_stream_t stream) {
  DPMN_REQUIRE(p && g && x_deep && x_shallow && r && state && gloss && workspace, "distill_backward: null pointer");
  DPMN_REQUIRE(g->dconv_cat_w && g->dconv_cat_b && g->dbn1_w && g->dbn1_b && g->dconv_feat_w && g->dconv_feat_b && g->dbn2_w && g->dbn2_b,
               "distill_backward: incomplete gradient set");
  DWs s = carve(B, H, W, static_cast<char*>(workspace));
  if (s.total > workspace_bytes) return dpmn_set_error(DPMN_ERR_WORKSPACE, "distill_backward: workspace too small");
  hipStream_t st = as_stream(stream);
  const long total = (long)B * H * W;
  const int nb = (int)((total + TPB - 1) / TPB);
  const float inv_n = (float)(1.0 / (3.0 * (double)total));       // nn.L1Loss: mean over B * 3 * H * W elements
  const DistillWPtr wp{p->conv_cat_w, p->conv_cat_b, p->conv_feat_w, p->conv_feat_b};
  hipLaunchKernelGGL(k_distill_bn_bwd_stats, dim3(nb), dim3(TPB), 0, st, r, state, dfeat, gloss, inv_n, s.rows, B, H, W);
  DPMN_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_distill_bn_bwd_finalize, dim3(1), dim3(192), 0, st, s.rows, nb, (double)total, g->dbn1_w, g->dbn1_b, g->dbn2_w,
                     g->dbn2_b, s.coef);
  DPMN_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_distill_dr, dim3(nb), dim3(TPB), 0, st, r, state, dfeat, gloss, inv_n, s.coef, s.dr, B, H, W);
  DPMN_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_distill_conv_bwd, dim3(nb), dim3(TPB), 0, st, x_deep, x_shallow, s.dr, wp, dx_deep, dx_shallow, s.wrows, B, H, W);
  DPMN_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_distill_wgrad_finalize, dim3((NWG + 15) / 16), dim3(TPB), 0, st, s.wrows, nb, g->dconv_cat_w, g->dconv_cat_b, g->dconv_feat_w,
                     g->dconv_feat_b);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}

}  // extern "C"
