// CMM channel gate (cmm.py:135-147): concat -> global average pool -> fc_1 -> ReLU -> fc_2 -> sigmoid
// -> x * w + x.  x: NHWC (B, P, C) with P = 1*4 bottleneck positions.  Two launches, one wave per output
// neuron (coalesced weight-row reads, wave-wide dot product), so the 2 x 1 MB of fc weights are streamed by
// thousands of waves instead of one workgroup per image.
#include "common.h"

namespace {
// hid[b][j] = relu(fc1_b[j] + sum_k fc1_w[j][k] * mean_p x[b][p][k])
// Block = (image b, 16 hidden units): the pooled vector mean_p x[b][p][:] is built ONCE per block in LDS (it was rebuilt by every
// one of the Cm waves of an image: 245 MB of L2 reads for a 1 MB weight matrix), then each wave takes 4 hidden units.
__global__ __launch_bounds__(256) void k_se_fc1(const float* __restrict__ x, const float* __restrict__ fc1_w,
                                                 const float* __restrict__ fc1_b, float* __restrict__ hid, int B, int P, int C,
                                                 int Cm) {
  extern __shared__ float pooled[];       // [C]
  const int jb = (Cm + 15) / 16;
  const int b = blockIdx.x / jb, j0 = (blockIdx.x % jb) * 16;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float* xb = x + (size_t)b * P * C;
  for (int k = threadIdx.x; k < C; k += 256) {
    float s = 0.f;
    for (int p = 0; p < P; ++p) s += xb[p * C + k];
    pooled[k] = s / (float)P;
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int j = j0 + wave * 4 + u;
    if (j >= Cm) break;
    const float* w = fc1_w + (size_t)j * C;
    float a = 0.f;
    for (int k = lane; k < C; k += 64) a += w[k] * pooled[k];
    a = wave_sum(a);
    if (lane == 0) { a += fc1_b[j]; hid[(size_t)b * Cm + j] = a > 0.f ? a : 0.f; }
  }
}
// out[b][p][c] = x * sigmoid(fc2_b[c] + sum_j fc2_w[c][j] * hid[b][j]) + x
__global__ __launch_bounds__(256) void k_se_fc2_apply(const float* __restrict__ x, const float* __restrict__ hid,
                                                       const float* __restrict__ fc2_w, const float* __restrict__ fc2_b,
                                                       float* __restrict__ out, int B, int P, int C, int Cm) {
  const long o = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (o >= (long)B * C) return;
  const int b = o / C, c = o % C;
  const float* h = hid + (size_t)b * Cm;
  const float* w = fc2_w + (size_t)c * Cm;
  float a = 0.f;
  for (int k = lane; k < Cm; k += 64) a += w[k] * h[k];
  a = wave_sum(a);
  const float g = sigmoid_f(a + fc2_b[c]);
  if (lane < P) {
    const size_t i = ((size_t)b * P + lane) * C + c;
    const float v = x[i];
    out[i] = v * g + v;
  }
}
}  // namespace

extern "C" int dpmn_se_gate_f32(const float* x, const float* fc1_w, const float* fc1_b, const float* fc2_w,
                                const float* fc2_b, float* out, float* hidden_ws, int B, int P, int C, int Cmid,
                                dpmn_stream_t stream) {
  DPMN_REQUIRE(x && fc1_w && fc1_b && fc2_w && fc2_b && out && hidden_ws && B > 0 && P <= 64, "se_gate: bad arguments");
  DPMN_REQUIRE(C <= 12288, "se_gate: the pooled vector lives in LDS (C <= 12288)");
  hipLaunchKernelGGL(k_se_fc1, dim3((unsigned)(B * ((Cmid + 15) / 16))), dim3(256), (size_t)C * sizeof(float), as_stream(stream), x, fc1_w,
                     fc1_b, hidden_ws, B, P, C, Cmid);
  DPMN_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_se_fc2_apply, dim3((unsigned)(((long)B * C + 3) / 4)), dim3(256), 0, as_stream(stream), x, hidden_ws,
                     fc2_w, fc2_b, out, B, P, C, Cmid);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}
