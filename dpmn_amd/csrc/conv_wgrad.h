// Argument block of the conv weight-gradient kernels, shared by conv_bwd.hip and conv_wgrad_x3.hip.
#pragma once
#include "common.h"

namespace dpmn_conv {
struct WgArgs {
  const float* in[3];
  const float* in_scale[3];
  const float* in_shift[3];
  int cseg[3];
  int cin;
  int B, Hin, Win, KH, KW, stride, dil_y, dil_x, pad_y, pad_x, Hp, Wp;
  int Hout, Wout, ostep, ooy, oox;
  int pro_act;
  const float* dy;      // NHWC (B, Hout, Wout, Cout)
  int Cout, K;          // K = KH*KW*cin
  float* dw;            // dw[base + co*s_co + ci*s_ci + ky*s_ky + kx*s_kx] += ..., for co < co_lim, ci < ci_lim
  long s_co, s_ci, s_ky, s_kx, base;
  int co_lim, ci_lim;
  int pix_per_block;
  int gx, gy, gz;       // logical grid (co tiles, k tiles, pixel splits)
  int nslots;           // > 1: split z accumulates into copy (z % nslots) of dw, copies slot_stride floats apart
  long slot_stride;
  int excl;             // packed destination with one slot PER split: plain stores, no atomics, slots need no zero-init
  float inv_hw, inv_w;
  int lgW, lgHW;        // P2 kernels: log2(Wp), log2(Hp * Wp)
};

// conv_wgrad_x3.hip ("f32 via bf16x3", dpmn_set_compute_dtype(2)): the 128 x 128, 64 x 256 and 64 x 128 (co x k) tiles of the power-of-two fast path
bool x3_wgrad_ok(const WgArgs& a, int bn, int bk);
int x3_launch_wgrad(const WgArgs& a, int bn, int bk, dim3 grid, hipStream_t st);
}  // namespace dpmn_conv
