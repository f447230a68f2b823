// Shared between backward.hip (fp32 kernels, the plan and the C entry points) and gemm_tn_x3.hip (mode 2 kernels): the descriptors of
// the grouped Linear weight-gradient launch.
#pragma once
#include "common.h"

namespace dpmn_gemm {
struct TnItem {
  const float* dy;      // (M, N)
  const float* x;       // (M, K)
  float* part;          // gz slabs of N K + N floats
  const float* db;      // non-null: the bias-gradient partials are wanted
  int M, N, K, rows;    // rows per split (multiple of 32)
  int gx, gy, gz;       // N tiles, K tiles, row splits
};
struct TnGroup {
  TnItem it[8];
  int first[9];         // first linear block id of item i; first[n] = the launch's block count
  int n;
};
// gemm_tn_x3.hip: the split partials of dW = dY^T X (+ db) on six bf16 MFMAs per tile -- the launches backward.hip k_gemm_tn_reg /
// k_gemm_tn_reg_multi would get
int x3_launch_tn(const float* dy, const float* x, int M, int N, int K, int rows, const float* db, float* part, dim3 grid, hipStream_t st);
int x3_launch_tn_multi(const TnGroup& g, hipStream_t st);
}  // namespace dpmn_gemm
