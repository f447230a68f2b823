// "f32 via bf16x3" instantiations of the conv kernels (dpmn_set_compute_dtype(2); common.h x3_split2t, conv_body.h X3 paths):
// fp32 tensors in HBM, operands split exactly into three bf16 planes on the way into LDS, six v_mfma_f32_16x16x32_bf16 per
// product, fp32 accumulation and epilogues.  Same call sites as the fp32 kernels (cmm.py:38-77 convs, tsrn.py / tatt.py trunks):
// conv.hip routes a launch here when the mode is set and the variant exists.
#include "conv_body.h"

namespace {
template <int BM, int BN, int WM, int WN, bool AFF, int ROWS = 2>
__global__ __launch_bounds__(256, (BM * BN > 128 * 64 ? 2 : 3)) void k_conv_igemm_x3(ConvArgs a) {      // two (128 x 128) / three resident blocks per CU
  conv_igemm_body<BM, BN, WM, WN, true, 32, true, AFF, false, false, false, ROWS>(a);
}
template <int KS, int BN, int TH>
int halo_x3(const ConvArgs& a, dim3 grid, hipStream_t st) {
  constexpr int NPX = (TH + KS - 1) * (16 + KS - 1);
  const size_t smem = (size_t)(3 * NPX + 2 * 3 * BN) * ((BK + 8) / 2) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_halo_x3<KS, BN, TH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_set = true;
  }
  hipLaunchKernelGGL((k_conv_halo_x3<KS, BN, TH>), grid, dim3(256), smem, st, a);
  return 0;
}
template <int KS, int BN>
int halo_x3w(const ConvArgs& a, dim3 grid, hipStream_t st) {
  constexpr int NPX = (16 + KS - 1) * (16 + KS - 1);
  const size_t smem = (size_t)(3 * NPX + 2 * 3 * BN) * ((BK + 16) / 2) * sizeof(float);      // 96-byte rows: 130 KB, one block per CU
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_halo_x3w<KS, BN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_set = true;
  }
  hipLaunchKernelGGL((k_conv_halo_x3w<KS, BN>), grid, dim3(512), smem, st, a);
  return 0;
}
}  // namespace

namespace dpmn_conv {
int x3_launch_igemm(int tile, bool aff, const ConvArgs& a, dim3 grid, hipStream_t st) {
  static const int rows96 = getenv("DPMN_X3_ROWS96") ? atoi(getenv("DPMN_X3_ROWS96")) : 1;      // 0: 80-byte LDS rows (A/B switch)
  if (tile == 128 && !rows96) {
    if (aff) hipLaunchKernelGGL((k_conv_igemm_x3<128, 128, 2, 2, true, 1>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((k_conv_igemm_x3<128, 128, 2, 2, false, 1>), grid, dim3(256), 0, st, a);
  } else if (tile == 128) {
    if (aff) hipLaunchKernelGGL((k_conv_igemm_x3<128, 128, 2, 2, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((k_conv_igemm_x3<128, 128, 2, 2, false>), grid, dim3(256), 0, st, a);
  } else if (tile == 12864) {      // 128 pixels x 64 channels: Cout <= 64 (a 64 x 64 tile splits as many weight values as it multiplies)
    if (aff) hipLaunchKernelGGL((k_conv_igemm_x3<128, 64, 2, 2, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((k_conv_igemm_x3<128, 64, 2, 2, false>), grid, dim3(256), 0, st, a);
  } else if (tile == 64) {
    if (aff) hipLaunchKernelGGL((k_conv_igemm_x3<64, 64, 2, 2, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((k_conv_igemm_x3<64, 64, 2, 2, false>), grid, dim3(256), 0, st, a);
  } else return -1;
  return 0;
}
int x3_launch_halo(int ks, int bn, int th, const ConvArgs& a, dim3 grid, hipStream_t st) {
  if (ks != 3 || bn != 64) return -1;
  if (th == 16) return halo_x3w<3, 64>(a, grid, st);
  if (th == 8) return halo_x3<3, 64, 8>(a, grid, st);
  if (th == 4) return halo_x3<3, 64, 4>(a, grid, st);
  return -1;
}
}  // namespace dpmn_conv
