// CMM channel gate (cmm.py:135-147): concat -> global average pool -> fc_1 -> ReLU -> fc_2 -> sigmoid
// -> x * w + x.  One workgroup per image; each output neuron is one wave-wide dot product with
// coalesced weight-row reads.  x: NHWC (B, P, C) with P = 1*4 bottleneck positions.
#include "common.h"

namespace {
__global__ __launch_bounds__(256) void k_se_gate(const float* __restrict__ x, const float* __restrict__ fc1_w,
                                                  const float* __restrict__ fc1_b, const float* __restrict__ fc2_w,
                                                  const float* __restrict__ fc2_b, float* __restrict__ out, int P, int C, int Cm) {
  extern __shared__ float sm[];
  float* S = sm;        // [C]
  float* Hd = sm + C;   // [Cm]
  float* Wg = Hd + Cm;  // [C]
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* xb = x + (size_t)b * P * C;
  for (int c = tid; c < C; c += 256) {
    float s = 0.f;
    for (int p = 0; p < P; ++p) s += xb[p * C + c];
    S[c] = s / (float)P;
  }
  __syncthreads();
  for (int j = wave; j < Cm; j += 4) {
    float a = 0.f;
    for (int k = lane; k < C; k += 64) a += fc1_w[(size_t)j * C + k] * S[k];
    a = wave_sum(a);
    if (lane == 0) { a += fc1_b[j]; Hd[j] = a > 0.f ? a : 0.f; }
  }
  __syncthreads();
  for (int c = wave; c < C; c += 4) {
    float a = 0.f;
    for (int k = lane; k < Cm; k += 64) a += fc2_w[(size_t)c * Cm + k] * Hd[k];
    a = wave_sum(a);
    if (lane == 0) Wg[c] = sigmoid_f(a + fc2_b[c]);
  }
  __syncthreads();
  float* ob = out + (size_t)b * P * C;
  for (int i = tid; i < P * C; i += 256) {
    const float v = xb[i];
    ob[i] = v * Wg[i % C] + v;
  }
}
}  // namespace

extern "C" int dpmn_se_gate_f32(const float* x, const float* fc1_w, const float* fc1_b, const float* fc2_w,
                                const float* fc2_b, float* out, int B, int P, int C, int Cmid, dpmn_stream_t stream) {
  DPMN_REQUIRE(x && fc1_w && fc1_b && fc2_w && fc2_b && out && B > 0, "se_gate: bad arguments");
  hipLaunchKernelGGL(k_se_gate, dim3(B), dim3(256), (size_t)(2 * C + Cmid) * 4, as_stream(stream), x, fc1_w, fc1_b, fc2_w,
                     fc2_b, out, P, C, Cmid);
  DPMN_CHECK_LAUNCH();
  return DPMN_OK;
}
