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
  RUN(dpmn_pgrm_tail_f32(s.tkv, w->tail0_w, w->tail0_b, w->tail1_w, w->tail1_b, w->weight_list, residuals, n_residuals,
                         s.mid, out, B, H, Wd, C, w->hidden_size, w->patch, stream));
#undef RUN
  return DPMN_OK;
}

}  // extern "C"
