// Native driver for one PGRM forward: launches the fused kernels in order on one stream.
// Replaces PGRM.forward / BasicLayer.forward / SwinTransformerBlock.forward / WindowAttention.forward
// / SKConv.forward / Mlp.forward (pgrm.py:546-565, 375-384, 315-331, 184-271, 79-96, 29-41).
#include <cstdlib>
#include "common.h"
#include <math.h>

namespace {
struct Ws {
  float *tq, *tkv, *q, *kv, *cat, *feats, *x1, *y, *g, *partial, *avec, *mid, *fold;
  size_t total;
};

Ws carve(const dpmn_pgrm_weights* w, int B, char* base) {
  const size_t L = (size_t)(w->img_h / w->patch) * (w->img_w / w->patch);
  const size_t C = w->dim, Ch = w->mlp_hidden;
  size_t off = 0;
  auto take = [&](size_t n) {
    float* p = base ? reinterpret_cast<float*>(base + off) : nullptr;
    off += ((n * sizeof(float) + 255) / 256) * 256;
    return p;
  };
  Ws s;
  s.tq = take(B * L * C);
  s.tkv = take(B * L * C);
  s.q = take(B * L * C);
  s.kv = take(B * L * 2 * C);
  s.cat = take(B * L * C);
  s.feats = take(B * L * C);
  s.x1 = take(B * L * C);
  s.y = take(B * L * Ch);
  s.g = take(B * L * Ch);
  s.partial = take((size_t)B * ((L + 31) / 32) * C);
  s.avec = take((size_t)B * C);
  s.mid = take(B * L * (size_t)w->hidden_size * w->patch * w->patch + (size_t)16 * (9 * C + 32));
  // folded attention weights of the two blocks (dim 96: attn_fused.hip; dim 192: attn_fused192.hip)
  s.fold = take(2 * (w->dim == 192 ? dpmn_ln_qkv_window_attn_d32_workspace_bytes() : dpmn_ln_qkv_window_attn_workspace_bytes()) / sizeof(float));
  s.total = off;
  return s;
}
}  // namespace

extern "C" {

size_t dpmn_pgrm_workspace_bytes(const dpmn_pgrm_weights* w, int B) {
  if (!w || B <= 0) return 0;
  return carve(w, B, nullptr).total;
}

int dpmn_pgrm_forward_f32(const dpmn_pgrm_weights* w, const float* x_q, int x_q_channels, const float* x_kv,
                          const float* const* residuals, int n_residuals, float* out, void* workspace,
                          size_t workspace_bytes, int B, dpmn_stream_t stream) {
  DPMN_REQUIRE(w && x_q && x_kv && out && workspace, "pgrm_forward: null pointer");
  DPMN_REQUIRE(B >= 1, "pgrm_forward: empty batch");
  DPMN_REQUIRE(w->n_groups >= 1 && w->n_groups <= 4, "pgrm_forward: 1..4 window groups");
  DPMN_REQUIRE(n_residuals <= w->n_weight_list && w->n_weight_list <= 16, "pgrm_forward: more residuals than weight_list entries (iter), or iter > 15");
  DPMN_REQUIRE((x_q_channels == 2) == (w->prior_fusion_w != nullptr) || x_q_channels == 3,
               "pgrm_forward: a 2-channel text prior needs prior_fusion weights (mode=False)");
  const int H = w->img_h / w->patch, Wd = w->img_w / w->patch, L = H * Wd, C = w->dim, Ch = w->mlp_hidden;
  const int r = (int)lrintf(sqrtf((float)L));
  DPMN_REQUIRE(r * r == L, "pgrm_forward: token count must be a perfect square (Mlp view, quirk Q2)");
  Ws s = carve(w, B, static_cast<char*>(workspace));
  if (s.total > workspace_bytes) return dpmn_set_error(DPMN_ERR_WORKSPACE, "pgrm_forward: workspace too small");
  const int M = B * L;
  int rc;
#define RUN(call) do { rc = (call); if (rc != DPMN_OK) return rc; } while (0)
  const bool fuse = (x_q_channels == 2);
  RUN(dpmn_patch_embed_ln_f32(x_q, x_q_channels, fuse ? w->prior_fusion_w : nullptr, fuse ? w->prior_fusion_b : nullptr,
                              w->pe_w, w->pe_b, w->pe_norm_w, w->pe_norm_b, s.tq, B, w->img_h, w->img_w, w->patch, C, stream));
  RUN(dpmn_patch_embed_ln_f32(x_kv, 3, nullptr, nullptr, w->pe_w, w->pe_b, w->pe_norm_w, w->pe_norm_b, s.tkv, B, w->img_h,
                              w->img_w, w->patch, C, stream));
  for (int blk = 0; blk < 2; ++blk) {
    const dpmn_pgrm_block& p = w->blocks[blk];
    int win[4], shift[4];
    for (int g = 0; g < w->n_groups; ++g) {
      win[g] = w->window[g];
      shift[g] = blk == 0 ? 0 : w->window[g] / 2;          // pgrm.py:362
      if ((H < Wd ? H : Wd) <= win[g]) { win[g] = H < Wd ? H : Wd; shift[g] = 0; }  // pgrm.py:147-150
    }
    if (dpmn_ln_qkv_window_attn_supported(C, w->n_groups, w->heads_per_group, win, H, Wd)) {
      // LayerNorm + q / kv projection + window attention in one kernel: q and kv never reach HBM (attn_fused.hip).  The folded
      // projection weights of block blk live at the end of the workspace and survive between calls (reuse_folded)
      RUN(dpmn_ln_qkv_window_attn_f32(s.tq, s.tkv, p.norm1_q_w, p.norm1_q_b, p.norm1_kv_w, p.norm1_kv_b, 1e-5f, p.q_w, p.q_b, p.kv_w,
                                      p.kv_b, p.bias_table, win, shift, w->n_groups, w->heads_per_group, s.cat,
                                      s.fold + blk * (dpmn_ln_qkv_window_attn_workspace_bytes() / sizeof(float)), w->reuse_folded ? 0 : 1,
                                      B, H, Wd, C, stream));
    } else if (dpmn_ln_qkv_window_attn_d32_supported(C, w->n_groups, w->heads_per_group, win, H, Wd)) {
      RUN(dpmn_ln_qkv_window_attn_d32_f32(s.tq, s.tkv, p.norm1_q_w, p.norm1_q_b, p.norm1_kv_w, p.norm1_kv_b, 1e-5f, p.q_w, p.q_b, p.kv_w,
                                          p.kv_b, p.bias_table, win, shift, w->n_groups, w->heads_per_group, s.cat,
                                          s.fold + blk * (dpmn_ln_qkv_window_attn_d32_workspace_bytes() / sizeof(float)), w->reuse_folded ? 0 : 1,
                                          B, H, Wd, C, stream));
    } else {
      RUN(dpmn_ln_linear_f32(s.tq, p.norm1_q_w, p.norm1_q_b, 1e-5f, p.q_w, p.q_b, s.q, M, C, C, DPMN_ACT_NONE, stream));
      RUN(dpmn_ln_linear_f32(s.tkv, p.norm1_kv_w, p.norm1_kv_b, 1e-5f, p.kv_w, p.kv_b, s.kv, M, 2 * C, C, DPMN_ACT_NONE, stream));
      RUN(dpmn_window_attn_f32(s.q, s.kv, p.bias_table, win, shift, w->n_groups, w->heads_per_group, s.cat, B, H, Wd, C, stream));
    }
    RUN(dpmn_sk_proj_f32(s.cat, p.sk_proj_w, p.sk_proj_b, s.feats, s.partial, M, C, stream));
    const int cg = C / w->n_groups;
    RUN(dpmn_sk_gate_f32(s.partial, (L + 31) / 32, L, p.sk_fc1_w, p.sk_fc1_b, p.sk_fc2_w, p.sk_fc2_b, s.avec, B, C,
                         w->n_groups, cg / 2, stream));
    static const int skmlp = getenv("DPMN_SKMLP") ? atoi(getenv("DPMN_SKMLP")) : 1;
    const bool fused_in = skmlp && dpmn_sk_mlp_in_supported(M, L, C, w->n_groups, Ch);
    if (fused_in)      // select + proj_head + residuals -> x1 -> LayerNorm2 -> fc1 in one launch (gemm.hip k_sk_mlp_in)
      RUN(dpmn_sk_mlp_in_f32(s.cat, s.avec, p.sk_head_w, p.sk_head_b, s.feats, s.tkv, s.x1, p.norm2_w, p.norm2_b, 1e-5f, p.fc1_w, p.fc1_b,
                             s.y, nullptr, nullptr, M, L, C, w->n_groups, Ch, stream));
    else
    RUN(dpmn_sk_select_f32(s.cat, s.avec, p.sk_head_w, p.sk_head_b, s.feats, s.tkv, s.x1, M, L, C, w->n_groups, stream));
    // fc1's GELU (pgrm.py:33): on load in the depthwise conv (HBM-bound, the erf is free there: fc1 48.7 -> 44.2 us, dwconv 36.6 -> 37.8 us), or (DPMN_GELU_ON_LOAD=0) in the epilogue of the MFMA-bound fc1 GEMM
    static const int gelu_on_load = getenv("DPMN_GELU_ON_LOAD") ? atoi(getenv("DPMN_GELU_ON_LOAD")) : 1;
    if (fused_in) {
      RUN(dpmn_dwconv3x3_gelu_in_f32(s.y, p.dw_w, p.dw_b, s.g, B, Ch, r, stream));
    } else if (gelu_on_load) {
      RUN(dpmn_ln_linear_f32(s.x1, p.norm2_w, p.norm2_b, 1e-5f, p.fc1_w, p.fc1_b, s.y, M, Ch, C, DPMN_ACT_NONE, stream));
      RUN(dpmn_dwconv3x3_gelu_in_f32(s.y, p.dw_w, p.dw_b, s.g, B, Ch, r, stream));
    } else {
      RUN(dpmn_ln_linear_f32(s.x1, p.norm2_w, p.norm2_b, 1e-5f, p.fc1_w, p.fc1_b, s.y, M, Ch, C, DPMN_ACT_GELU, stream));
      RUN(dpmn_dwconv3x3_gelu_f32(s.y, p.dw_w, p.dw_b, s.g, B, Ch, r, stream));
    }
    RUN(dpmn_pointwise_f32(s.g, p.pw_w, p.pw_b, s.y, B, Ch, L, stream));
    RUN(dpmn_linear_f32(s.y, p.fc2_w, p.fc2_b, s.x1, nullptr, s.tkv, M, C, Ch, DPMN_ACT_NONE, 0.f, stream));
  }
  RUN(dpmn_pgrm_tail_reuse_f32(s.tkv, w->tail0_w, w->tail0_b, w->tail1_w, w->tail1_b, w->weight_list, residuals, n_residuals,
                               s.mid, out, B, H, Wd, C, w->hidden_size, w->patch, w->reuse_folded ? 1 : 0, stream));
#undef RUN
  return DPMN_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Training forward of one PGRM as ONE call: the launches of train/pgrm_train.py::forward (patch embedding x 2, pos_drop x 2, per
// block fused LayerNorm + q / kv + window attention (+ attn_drop), SK projection, SK gate, select + proj_head + DropPath + LayerNorm2
// + fc1, depthwise conv with both GELUs and the Mlp dropout, pointwise conv, fc2 + Dropout + DropPath + residual; the two tail
// convs and the pixel-shuffle / weight_list epilogue) issued from native code.  The host thread needed 0.39 ms per module to issue
// these ~22 launches through ctypes / torch allocations, 0.29 ms through this call (tools/host_profile_pgrm.py; what is left is the
// ~5 us per hipLaunchKernel plus the saved-tensor views).  The step time did not move: the forward phase is bound by the GPU (six
// modules x 0.81 ms back to back; two of them side by side take as long as one after the other), the gain is host head-room.
// Every tensor the backward reads is written into the caller's dpmn_pgrm_saved slots.
int dpmn_pgrm_forward_train_supported(const dpmn_pgrm_weights* w, int B) {
  if (!w || B < 1 || w->n_groups < 1 || w->n_groups > 4) return 0;
  const int H = w->img_h / w->patch, Wd = w->img_w / w->patch, L = H * Wd, C = w->dim, Ch = w->mlp_hidden;
  const int r = (int)lrintf(sqrtf((float)L));
  if (r * r != L || w->hidden_size != 3 || w->patch != 2) return 0;
  const long M = (long)B * L;
  for (int blk = 0; blk < 2; ++blk) {
    int win[4];
    for (int g = 0; g < w->n_groups; ++g) win[g] = (H < Wd ? H : Wd) <= w->window[g] ? (H < Wd ? H : Wd) : w->window[g];
    if (!dpmn_ln_qkv_window_attn_supported(C, w->n_groups, w->heads_per_group, win, H, Wd)) return 0;
  }
  if (!dpmn_sk_mlp_in_supported((int)M, L, C, w->n_groups, Ch)) return 0;
  return (M % 64 == 0 && C % 96 == 0 && Ch % 32 == 0 && Ch > 192) ? 1 : 0;     // dpmn_linear_drop_f32's tiling
}

int dpmn_pgrm_forward_train_f32(const dpmn_pgrm_weights* w, const float* x_q, int x_q_channels, const float* x_kv,
                                const float* const* residuals, int n_residuals, const float* tail0_packed, const float* tail1_packed,
                                const dpmn_pgrm_drop* drop, const dpmn_pgrm_saved* sv, const dpmn_cmm_scratch* scratch, float* out,
                                int B, dpmn_stream_t stream) {
  DPMN_REQUIRE(w && x_q && x_kv && out && sv && tail0_packed && tail1_packed, "pgrm_forward_train: null pointer");
  DPMN_REQUIRE(dpmn_pgrm_forward_train_supported(w, B), "pgrm_forward_train: geometry outside the fused training kernels (use the per-op path)");
  DPMN_REQUIRE(n_residuals <= w->n_weight_list && w->n_weight_list <= 16, "pgrm_forward_train: more residuals than weight_list entries");
  DPMN_REQUIRE((x_q_channels == 2 && w->prior_fusion_w) || x_q_channels == 3,
               "pgrm_forward_train: a 2-channel text prior needs prior_fusion weights (mode=False)");
  const int H = w->img_h / w->patch, Wd = w->img_w / w->patch, L = H * Wd, C = w->dim, Ch = w->mlp_hidden, G = w->n_groups;
  const int r = (int)lrintf(sqrtf((float)L));
  const int M = B * L;
  const float pd = drop ? drop->p : 0.f, pa = drop ? drop->pa : 0.f;
  static const unsigned long long zero_seeds[12] = {0};
  const unsigned long long* sd = drop ? drop->seeds : zero_seeds;
  DPMN_REQUIRE(sv->tq && sv->tkv0 && sv->c0 && sv->c1, "pgrm_forward_train: saved-tensor slots missing");
  int rc;
#define RUN(call) do { rc = (call); if (rc != DPMN_OK) return rc; } while (0)
  const bool fuse = (x_q_channels == 2);
  // pos_drop (pgrm.py:550-551) rides in the patch embedding's epilogue
  RUN(dpmn_patch_embed_ln_drop_f32(x_q, x_q_channels, fuse ? w->prior_fusion_w : nullptr, fuse ? w->prior_fusion_b : nullptr,
                                   w->pe_w, w->pe_b, w->pe_norm_w, w->pe_norm_b, sv->tq, B, w->img_h, w->img_w, w->patch, C, pd, sd[0],
                                   stream));
  RUN(dpmn_patch_embed_ln_drop_f32(x_kv, 3, nullptr, nullptr, w->pe_w, w->pe_b, w->pe_norm_w, w->pe_norm_b, sv->tkv0, B, w->img_h,
                                   w->img_w, w->patch, C, pd, sd[1], stream));
  const float* tkv = sv->tkv0;
  for (int blk = 0; blk < 2; ++blk) {
    const dpmn_pgrm_block& p = w->blocks[blk];
    const dpmn_pgrm_saved_block& s = sv->blk[blk];
    DPMN_REQUIRE(s.cat && s.fold && s.feats && s.partial && s.avec && s.x1 && s.ypre && s.V && s.n2 && s.gpre && s.g && s.z && s.tkv_out,
                 "pgrm_forward_train: saved-tensor slots missing");
    const unsigned long long* sb = sd + 2 + 5 * blk;
    const float dpb = drop ? drop->dp[blk] : 0.f;
    int win[4], shift[4];
    for (int g = 0; g < G; ++g) {
      win[g] = w->window[g];
      shift[g] = blk == 0 ? 0 : w->window[g] / 2;
      if ((H < Wd ? H : Wd) <= win[g]) { win[g] = H < Wd ? H : Wd; shift[g] = 0; }
    }
    RUN(dpmn_ln_qkv_window_attn_train_f32(sv->tq, tkv, p.norm1_q_w, p.norm1_q_b, p.norm1_kv_w, p.norm1_kv_b, 1e-5f, p.q_w, p.q_b, p.kv_w,
                                          p.kv_b, p.bias_table, win, shift, G, w->heads_per_group, s.cat, nullptr, nullptr, pa, sb[0],
                                          s.fold, B, H, Wd, C, stream));
    RUN(dpmn_sk_proj_f32(s.cat, p.sk_proj_w, p.sk_proj_b, s.feats, s.partial, M, C, stream));
    RUN(dpmn_sk_gate_f32(s.partial, (L + 31) / 32, L, p.sk_fc1_w, p.sk_fc1_b, p.sk_fc2_w, p.sk_fc2_b, s.avec, B, C, G, C / G / 2, stream));
    RUN(dpmn_sk_mlp_in_drop_f32(s.cat, s.avec, p.sk_head_w, p.sk_head_b, s.feats, tkv, s.x1, p.norm2_w, p.norm2_b, 1e-5f, p.fc1_w, p.fc1_b,
                                s.ypre, s.V, s.n2, M, L, C, G, Ch, dpb, sb[1], stream));
    RUN(dpmn_dwconv3x3_train_f32(s.ypre, p.dw_w, p.dw_b, s.gpre, s.g, 1, pd, sb[2], B, Ch, r, stream));
    RUN(dpmn_pointwise_f32(s.g, p.pw_w, p.pw_b, s.z, B, Ch, L, stream));
    if (pd > 0.f || dpb > 0.f)
      RUN(dpmn_linear_drop_f32(s.z, p.fc2_w, p.fc2_b, s.x1, s.tkv_out, M, C, Ch, pd, sb[3], dpb, sb[4], L * C, stream));
    else
      RUN(dpmn_linear_f32(s.z, p.fc2_w, p.fc2_b, s.x1, nullptr, s.tkv_out, M, C, Ch, DPMN_ACT_NONE, 0.f, stream));
    tkv = s.tkv_out;
  }
  const int Cm = w->hidden_size * w->patch * w->patch;
  dpmn_conv_desc d{};
  d.in[0] = tkv; d.cseg[0] = C; d.B = B; d.Hin = H; d.Win = Wd;
  d.KH = 3; d.KW = 3; d.stride = 1; d.dil_y = 1; d.dil_x = 1; d.pad_y = 1; d.pad_x = 1;
  d.Hp = H; d.Wp = Wd; d.Hout = H; d.Wout = Wd; d.ostep = 1;
  d.w = tail0_packed; d.bias = w->tail0_b; d.Cout = Cm; d.out = sv->c0;
  if (scratch) { d.splitk_ws = scratch->splitk_ws; d.splitk_ws_bytes = scratch->splitk_ws_bytes; d.arrive_cnt = scratch->arrive_cnt; d.arrive_cnt_len = scratch->arrive_cnt_len; }
  RUN(dpmn_conv2d_nhwc_f32(&d, stream));
  d.in[0] = sv->c0; d.cseg[0] = Cm; d.w = tail1_packed; d.bias = w->tail1_b; d.out = sv->c1;
  RUN(dpmn_conv2d_nhwc_f32(&d, stream));
  RUN(dpmn_pgrm_tail_elem_f32(sv->c1, w->weight_list, residuals, n_residuals, out, B, H, Wd, stream));
#undef RUN
  return DPMN_OK;
}

}  // extern "C"
