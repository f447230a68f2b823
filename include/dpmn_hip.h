/* dpmn_hip.h -- C ABI of libdpmn_hip.so, the MI355X (gfx950) drop-in for the DPMN SR hot path.
 *
 * The reference (jdfxzzy/DPMN) is pure PyTorch and has no FFI of its own, so
 * every entry point below cites the reference Python call site it replaces.  Conventions:
 *   - plain device pointers (fp32 unless said otherwise), explicit sizes, no torch types;
 *   - caller owns all memory, including scratch ("workspace" pointers, sizes from *_workspace_bytes);
 *   - every call enqueues on the given hipStream_t (passed as void*) and returns immediately;
 *   - returns DPMN_OK (0) or a negative DPMN_ERR_* code; dpmn_last_error() gives the message of the
 *     calling thread's last failure.  No exceptions cross the boundary.  Re-entrant per stream.
 *   - tensors use the reference's own layouts: images NCHW, PGRM tokens (B, L, C) row-major,
 *     weights exactly as stored in the reference state_dict unless a
 *     *_pack function is named.
 */
#ifndef DPMN_HIP_H
#define DPMN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* dpmn_stream_t; /* hipStream_t */

enum { DPMN_OK = 0, DPMN_ERR_ARG = -1, DPMN_ERR_LAUNCH = -2, DPMN_ERR_WORKSPACE = -3, DPMN_ERR_RUNTIME = -4 };

/* activation codes (epilogues / conv prologues) */
enum { DPMN_ACT_NONE = 0, DPMN_ACT_GELU = 1, DPMN_ACT_RELU = 2, DPMN_ACT_LEAKY02 = 3, DPMN_ACT_LEAKY001 = 4,
       DPMN_ACT_MISH = 5, DPMN_ACT_PRELU = 6, DPMN_ACT_TANH = 7, DPMN_ACT_SIGMOID = 8,
       DPMN_ACT_RELU_POST_RES = 9 /* conv epilogue only: ReLU applied AFTER the residual add (ResNet BasicBlock, VisionLAN resnet.py:34-36) */ };

int dpmn_abi_version(void);
const char* dpmn_last_error(void);
/* Arithmetic of the GEMM-shaped kernels (BASELINE.json configs[2..4] name bf16; the reference itself is fp32 and fp32 is the
 * default and the headline).  mode 0: v_mfma_f32_16x16x4_f32 on fp32 operands.  mode 1: the kernels that have the variant
 * (implicit-GEMM conv, pointwise GEMM) round their MFMA operands to bf16 on the way into LDS and run v_mfma_f32_16x16x32_bf16
 * with fp32 accumulation.  mode 2 ("f32 via bf16x3"): the kernels that have the variant split every fp32 operand EXACTLY into
 * three bf16 terms and keep the six product terms of weight >= 2^-16 -- fp32-class products (relative error <= 2^-23 each)
 * on the bf16 pipe, 6 MFMAs per 16x16x32 step; kernels without the variant stay on mode 0.  Since round 6 the variant exists for the
 * implicit-GEMM conv (128 x 128 tiles), the 3 x 3 halo conv (layers with >= 8 row tiles per image: a per-image rule, so the kernel choice
 * does not depend on the batch size), the conv weight gradient, the pointwise GEMM, the k-loop GEMMs and the Linear weight gradient
 * (dpmn_conv2d_*, dpmn_conv_wgrad_*, dpmn_pointwise_f32, dpmn_linear_f32 with K > 192, dpmn_gemm_tn_*); no kernel of mode 2 owns memory.
 * Tensors in HBM, LayerNorm / softmax / BatchNorm statistics and every epilogue are fp32 in all modes.  Process-wide switch, not
 * thread-safe.  The whole parity suite runs in mode 0 and in mode 2 with the same tolerances (tests/conftest.py). */
int dpmn_set_compute_dtype(int mode);
int dpmn_get_compute_dtype(void);

/* ------------------------------------------------------------------ GEMM family (gemm.hip) */
/* y = act(x . w^T + bias) + res1 + res2 ; x (M,K), w (N,K), y/res (M,N).  nn.Linear call sites:
 * pgrm.py:39 (Mlp.fc2 + residual 330), tatt.py:209, transformer_v2.py:453/785 FFNs. */
int dpmn_linear_f32(const float* x, const float* w, const float* bias, const float* res1, const float* res2,
                    float* y, int M, int N, int K, int act, float slope, dpmn_stream_t stream);
/* y = res + Dropout(x w^T + bias): element dropout (p_elem, seed_elem) and / or per-sample DropPath (p_row, seed_row, row_len elements per
 * sample) on the Linear's output before the residual, the masks of dpmn_dropout_f32 -- Mlp.fc2 -> Mlp.drop -> DropPath -> + shortcut
 * (pgrm.py:39-40, 330) in one launch.  M % 64 == 0, N % 96 == 0, K % 32 == 0, K > 192. */
int dpmn_linear_drop_f32(const float* x, const float* w, const float* bias, const float* res, float* y, int M, int N, int K, float p_elem,
                         unsigned long long seed_elem, float p_row, unsigned long long seed_row, long row_len, dpmn_stream_t stream);
/* y = act((x + addv) . w^T + bias): with_pos_embed + in-projection, transformer_v2.py:462,826-828 */
int dpmn_add_linear_f32(const float* x, const float* addv, const float* w, const float* bias, float* y, int M,
                        int N, int K, int act, dpmn_stream_t stream);
/* y = act([x1 | x2] . w^T + bias): 1x1 conv over a channel concat read in place (GruBlock.conv1 folded into the
 * GRU input projection on cat([residual, text_emb]), tatt.py:902-907, 1078) */
int dpmn_cat2_linear_f32(const float* x1, int k1, const float* x2, int k2, const float* w, const float* bias, float* y,
                         int M, int N, int act, dpmn_stream_t stream);
/* y = act(LayerNorm(x) . w^T + bias): pgrm.py:322-323 + 188/194 (q, kv), pgrm.py:330 + 30-31 (norm2+fc1+GELU) */
int dpmn_ln_linear_f32(const float* x, const float* ln_w, const float* ln_b, float eps, const float* w,
                       const float* bias, float* y, int M, int N, int K, int act, dpmn_stream_t stream);
/* SKConv.proj (pgrm.py:82) + per-32-row-tile column sums of GELU(feats) for the global average pool
 * (pgrm.py:84-86).  colsum_partials: (ceil(M/32), C); rows per image must be a multiple of 32. */
int dpmn_sk_proj_f32(const float* cat, const float* w, const float* bias, float* feats, float* colsum_partials,
                     int M, int C, dpmn_stream_t stream);
/* out = shortcut + feats + (sum_g A[b,g,:] * cat[:, g-th slice]) . w_head^T + b_head
 * (pgrm.py:92-95 + residual 329).  attn_vec: (B, groups, C/groups). */
int dpmn_sk_select_f32(const float* cat, const float* attn_vec, const float* w_head, const float* b_head,
                       const float* feats, const float* shortcut, float* out, int M, int rows_per_image, int C,
                       int groups, dpmn_stream_t stream);
/* SKConv select + proj_head + both residuals (= dpmn_sk_select_f32) -> LayerNorm2 -> Mlp.fc1 (= dpmn_ln_linear_f32) in ONE launch
 * (pgrm.py:91-96, 327-331, 31): x1 = proj_head(sum_g A[b][g] cat_g) + b_head + feats + shortcut is written once (fc2's residual)
 * and fed to fc1 from registers; y (M, N) = fc1(LayerNorm(x1)) WITHOUT activation (the depthwise conv applies fc1's GELU on load).
 * v_out (M, C / groups), n2_out (M, C): both NULL, or (training forward) the group sum fed to proj_head and LayerNorm2(x1), which the
 * backward's weight-gradient GEMMs read.  Bitwise equal to the two calls it replaces.  dim 96 / three window groups only: ask dpmn_sk_mlp_in_supported first. */
int dpmn_sk_mlp_in_f32(const float* cat, const float* attn_vec, const float* w_head, const float* b_head, const float* feats,
                       const float* shortcut, float* x1, const float* ln_w, const float* ln_b, float eps, const float* w_fc1,
                       const float* b_fc1, float* y, float* v_out, float* n2_out, int M, int rows_per_image, int C, int groups, int N,
                       dpmn_stream_t stream);
/* the same with timm DropPath on the attention branch (training, pgrm.py:329): x1 = shortcut + m_b (proj_head(sel) + b_head + feats),
 * m_b the per-sample mask of dpmn_dropout_f32(p_row, seed_row) */
int dpmn_sk_mlp_in_drop_f32(const float* cat, const float* attn_vec, const float* w_head, const float* b_head, const float* feats,
                       const float* shortcut, float* x1, const float* ln_w, const float* ln_b, float eps, const float* w_fc1,
                       const float* b_fc1, float* y, float* v_out, float* n2_out, int M, int rows_per_image, int C, int groups, int N,
                       float p_row, unsigned long long seed_row, dpmn_stream_t stream);
int dpmn_sk_mlp_in_supported(int M, int rows_per_image, int C, int groups, int N);
/* z[b] = w (Ch,Ch) . g[b] (Ch,L) + bias : Mlp.pointwise_conv on the raw (B,Ch,r,r) view (pgrm.py:34,37) */
int dpmn_pointwise_f32(const float* g, const float* w, const float* bias, float* z, int B, int Ch, int L,
                       dpmn_stream_t stream);
/* Fused norm1_q / norm1_kv + q / kv projections + multi-size window attention of one SwinTransformerBlock
 * (pgrm.py:322-323 LayerNorms, 188/194 Linear q / kv, 197-266 per-group roll + window partition + 2 heads x 16 + relative
 * position bias + shift mask + softmax + P.V, written window-major without un-roll, quirk Q1).  tq / tkv: (B, H*W, C) token
 * streams BEFORE the LayerNorms; out: (B, H*W, C) = the `cat` tensor fed to SKConv.  q and kv never reach HBM.
 * Built for C = 96 = 3 groups x 2 heads x 16 with windows in {2, 4, 8} (configs 1-3); _supported() says whether a shape
 * qualifies (0: use dpmn_ln_linear_f32 x2 + dpmn_window_attn_f32). */
int dpmn_ln_qkv_window_attn_supported(int C, int n_groups, int heads_per_group, const int* windows, int H, int W);
int dpmn_ln_qkv_window_attn_f32(const float* tq, const float* tkv, const float* lnq_w, const float* lnq_b, const float* lnkv_w,
                                const float* lnkv_b, float eps, const float* wq, const float* bq, const float* wkv,
                                const float* bkv, const float* const* bias_tables, const int* windows, const int* shifts,
                                int n_groups, int heads_per_group, float* out, void* workspace, int refold, int B, int H, int W,
                                int C, dpmn_stream_t stream);
/* workspace: dpmn_ln_qkv_window_attn_workspace_bytes() bytes of device memory, 16-byte aligned, private to the call's stream until
 * the call has run -- it receives the projection weights with the LayerNorm affine folded in (one tiny kernel, when refold != 0).
 * refold = 0: the caller vouches that the workspace still holds the fold of THESE weights and bias tables (frozen weights in
 * evaluation: the fold kernel then runs once per module, not once per call). */
size_t dpmn_ln_qkv_window_attn_workspace_bytes(void);
/* The same fused operator at embed_dim 192 = 3 groups x 2 heads x head dim 32 with windows in {4, 8, 16} (BASELINE.json configs[4];
 * csrc/attn_fused192.hip): same arguments and result as dpmn_ln_qkv_window_attn_f32, own workspace size (the folded weights of the
 * three groups, 3 x 160 KB).  Token grid sides powers of two >= 16, H W a multiple of 256 (one unit = 256 window-major tokens). */
int dpmn_ln_qkv_window_attn_d32_supported(int C, int n_groups, int heads_per_group, const int* windows, int H, int W);
size_t dpmn_ln_qkv_window_attn_d32_workspace_bytes(void);
int dpmn_ln_qkv_window_attn_d32_f32(const float* tq, const float* tkv, const float* lnq_w, const float* lnq_b, const float* lnkv_w,
                                    const float* lnkv_b, float eps, const float* wq, const float* bq, const float* wkv,
                                    const float* bkv, const float* const* bias_tables, const int* windows, const int* shifts,
                                    int n_groups, int heads_per_group, float* out, void* workspace, int refold, int B, int H, int W,
                                    int C, dpmn_stream_t stream);
/* Training forward of the same fused kernel (interfaces/super_resolution.py:140-278 runs the PGRMs in .train()): also writes
 * the projections q_out (B L, C) and kv_out (B L, 2 C) in raster token order -- the tensors a.q(norm1_q(x_q)) and
 * a.kv(norm1_kv(x_kv)) of pgrm.py:188,194, which the backward kernels read -- and applies attn_drop (pgrm.py:248) with the
 * counter-based masks of dpmn_window_attn_f32 (same element index, same seed => same masks as the unfused kernels). */
int dpmn_ln_qkv_window_attn_train_f32(const float* tq, const float* tkv, const float* lnq_w, const float* lnq_b, const float* lnkv_w,
                                      const float* lnkv_b, float eps, const float* wq, const float* bq, const float* wkv,
                                      const float* bkv, const float* const* bias_tables, const int* windows, const int* shifts,
                                      int n_groups, int heads_per_group, float* out, float* q_out, float* kv_out, float p_drop,
                                      unsigned long long seed, void* workspace, int B, int H, int W, int C, dpmn_stream_t stream);
/* q_out and kv_out may both be NULL: the recomputing backward below needs neither.
 *
 * Backward of the fused kernel, recomputing q / k / v from the forward's inputs (WindowAttention.forward, pgrm.py:184-271, from the
 * gradient of its concatenated head outputs back to the outputs of a.q / a.kv):
 *   dout (B, L, C): gradient of `out` above (same window-major layout);  dq (B L, C), dkv (B L, 2 C): gradients of the q / kv
 *   Linear outputs in raster token order (what the LayerNorm backward + weight-gradient GEMMs of norm1_* / a.q / a.kv consume);
 *   dtable_parts[g]: (dpmn_ln_qkv_window_attn_bwd_part_rows(B, H, W), (2 ws_g - 1)^2 * heads_per_group) -- every block stores its
 *   partial row of group g's relative-position-bias table gradient (rows of blocks without work for g are zero); the caller adds
 *   the rows in row order (dpmn_tn_reduce_*), so the result is bitwise reproducible.
 * p_drop / seed: the attn_drop masks of the forward call.  workspace / refold as in dpmn_ln_qkv_window_attn_f32 (refold = 0: the
 * workspace still holds the forward's fold of these weights). */
int dpmn_ln_qkv_window_attn_bwd_f32(const float* tq, const float* tkv, const float* lnq_w, const float* lnq_b, const float* lnkv_w,
                                    const float* lnkv_b, float eps, const float* wq, const float* bq, const float* wkv,
                                    const float* bkv, const float* const* bias_tables, const int* windows, const int* shifts,
                                    int n_groups, int heads_per_group, const float* dout, float* dq, float* dkv,
                                    float* const* dtable_parts, float p_drop, unsigned long long seed, void* workspace, int refold,
                                    int B, int H, int W, int C, dpmn_stream_t stream);
int dpmn_ln_qkv_window_attn_bwd_part_rows(int B, int H, int W);

/* Measurement hooks for bench.py's roofline objects (no reference counterpart: the reference has no profiler, SURVEY.md
 * section 5).  While armed, every launch of a kernel family whose tag bit is set in tag_mask -- issued directly or from
 * inside a module driver such as dpmn_pgrm_forward_f32 -- is bracketed by HIP events on the stream it is launched on, up to
 * max_launches.  dpmn_profile_end disarms and aggregates per tag: launches, summed duration, summed ALGORITHMIC FLOPs and
 * bytes of those launches (computed from the launch arguments; DESIGN.md (d) states the formulas).  Synchronise the
 * stream before calling it.  Returns the number of rows written (<= max_rows) or a negative error.  Not thread-safe. */
typedef struct {
  int tag;          /* index for dpmn_profile_tag_name */
  int launches;
  double total_ms;  /* sum of event-to-event durations */
  double flops;     /* sum over the timed launches */
  double bytes;     /* compulsory HBM bytes (inputs read once + outputs written once) */
} dpmn_profile_row;
int dpmn_profile_tag_count(void);
/* compulsory bytes of the NEXT multi-descriptor pack / unpack launches (their descriptor tables live on the device, the host
 * side that built them knows the sizes); only read while that family is armed */
int dpmn_profile_hint_bytes(double bytes);
const char* dpmn_profile_tag_name(int tag);
int dpmn_profile_begin(unsigned long long tag_mask, int max_launches);
int dpmn_profile_end(dpmn_profile_row* rows, int max_rows);

/* ------------------------------------------------------------------ NHWC implicit-GEMM conv (conv.hip) */
/* One descriptor drives nn.Conv2d / nn.ConvTranspose2d call sites of cmm.py:44-71,86-118 and
 * tsrn.py/tatt.py conv stacks.  Inputs are up to 3 channel-concatenated NHWC segments (torch.cat of
 * cmm.py:150-158 is never materialised); weights are pre-packed (Cout, roundup(KH*KW*Cin,32)) with
 * K index = (ky*KW + kx)*Cin + ci (see dpmn_amd/model/packing.py). */
typedef struct {
  const float* in[3];        /* NHWC (B,Hin,Win,cseg[s]) ; unused = NULL */
  const float* in_scale[3];  /* optional per-channel affine applied on load (train-mode BatchNorm) */
  const float* in_shift[3];
  int cseg[3];
  int B, Hin, Win;
  int KH, KW, stride, dil_y, dil_x, pad_y, pad_x; /* iy = oy'*stride + ky*dil_y - pad_y (dil may be -1) */
  int Hp, Wp;                                      /* pixels computed per image (phase grid) */
  int Hout, Wout, ostep, ooy, oox;                 /* oy = oy'*ostep + ooy within the (Hout,Wout) output */
  int pro_act;                                     /* DPMN_ACT_* applied to inputs after the affine */
  const float* w;
  const float* bias;
  int Cout;
  int epi_act;
  float slope;
  const float* res;          /* residual in output layout, or NULL */
  float* out;
  int out_ld, out_coff;      /* NHWC channel stride (0 = Cout) and channel offset */
  int out_nchw;              /* store NCHW instead */
  int pixel_shuffle;         /* PixelShuffle(2) store: NHWC (B,2Hout,2Wout,Cout/4) (tsrn.py:110-111) */
  float* stats;              /* (32,2,Cout) DOUBLES (8-byte aligned, 64*Cout floats of storage x 2), slotted += sum / sum of squares of
                              * the pre-activation outputs by fp64 atomics -- order-independent after the final rounding -- (summed by
                              * dpmn_bn_finalize_f32), or NULL */
  float* splitk_ws;          /* optional scratch enabling split-K for small-M / large-K convs (deep CMM levels) */
  size_t splitk_ws_bytes;
  int nphase;                /* 0/1: one conv.  4: all phases of nn.ConvTranspose2d(4,2,1) (cmm.py:100-118) in one launch:
                              * phase p = 2*py+px reads w + p*w_phase_stride, pad = -(py,px), writes output pixels
                              * (2y+py, 2x+px); pad_y/pad_x/ooy/oox of the descriptor are ignored */
  long w_phase_stride;       /* floats between consecutive packed phase weights */
  int groups;                /* 0/1: one weight set.  2: images [B/2, B) use w + w_group_stride and bias + Cout -- the twin
                              * encoder branches of the CMM (cmm.py:86-99: same shapes, own weights) in one launch; needs an
                              * even B, no stats, and (implicit-GEMM path) B/2*Hp*Wp % 128 == 0 */
  long w_group_stride;
  unsigned* arrive_cnt;      /* optional: arrive_cnt_len tile-arrival counters, ZERO before the first launch that uses them (the
                              * library leaves them zero; one array per stream, like splitk_ws).  With them the split-K layers run
                              * as ONE persistent "stream-K" launch: every workgroup walks an equal share of all (tile, k chunk)
                              * steps, partial tiles go to splitk_ws and the LAST workgroup to arrive at a tile sums them in a
                              * fixed order and runs the epilogue -- no reduce kernel, run-to-run bitwise reproducible.
                              * The fixed-split layers on 128 x 128 tiles use the same counters for the in-launch reduction
                              * through ONE XCD's L2 (round 4, csrc/conv.hip XRED: all k splits of a tile run on one XCD, the
                              * last-arriving workgroup adds the partial tiles in split order and runs the epilogue).
                              * NULL: the two-launch split-K path */
  int arrive_cnt_len;
} dpmn_conv_desc;
int dpmn_conv2d_nhwc_f32(const dpmn_conv_desc* d, dpmn_stream_t stream);
/* Diagnostics of the in-L2 split-K reduction: number of tiles whose contributing workgroups were NOT all placed on one XCD and
 * that were therefore recomputed by the last-arriving workgroup (bitwise-equal result, slower).  0 on the observed round-robin
 * placement.  Synchronises the device.  reset != 0 zeroes the counter. */
int dpmn_xred_fallbacks(unsigned* count_out, int reset);
/* Device self-test of the DPP / permlane-swap lane exchanges the reductions are built on (csrc/common.h xshfl): *mismatches_out = the
 * number of lanes whose exchange differs from __shfl_xor (0 on a correct build).  Test hook; no reference counterpart. */
int dpmn_selftest_xshfl(unsigned* mismatches_out);
/* Test hook: `blocks` workgroups that fill their CU's whole LDS allocation (160 KB) with `pattern` and exit -- launched next to a kernel
 * under test it changes what that kernel would find in LDS words it reads without having written them (a kernel must not depend on
 * LDS contents it did not produce: under stream concurrency they are another kernel's leftovers). */
int dpmn_selftest_lds_poison(unsigned pattern, int blocks, dpmn_stream_t stream);
/* 1 / 0: use / do not use the in-L2 split-K reduction for the layers that qualify; -1: the DPMN_CONV_XRED environment variable
 * (default 0: on MI355X the reduce launch measured faster, DESIGN.md "Measured and rejected", round 4). */
int dpmn_xred_enable(int on);
/* Test hook: on != 0 makes every split-K tile of the in-L2 reduction take the recompute path (as if misplaced). */
int dpmn_xred_test_force_recompute(int on);
/* layout plumbing at the module boundary: NCHW images <-> NHWC (channels zero-padded to Cpad) */
int dpmn_nchw_to_nhwc_f32(const float* in, float* out, int B, int C, int H, int W, int Cpad, dpmn_stream_t stream);
int dpmn_nhwc_to_nchw_f32(const float* in, float* out, int B, int C, int H, int W, dpmn_stream_t stream);

/* ---- DistillModule (distill_module.py:4-31) as a native module: conv_cat_feature (6 -> 3, 3x3) and conv_feature (3 -> 3, 3x3) on NCHW
 * (B, 3, H, W) images, train-mode BatchNorm2d(3) (batch statistics, running statistics updated: momentum 0.1, eps 1e-5) or the
 * running statistics in eval mode, ReLU, and loss = mean |feature_cat - feature_shallow|; returns the loss (device scalar) and
 * feature_cat.  Forward = 4 launches, backward = 5, every reduction a per-block partial row added in block order (statistics and
 * loss in fp64): bitwise reproducible.  Pointers are the reference module's own parameter tensors (state_dict layout). */
typedef struct {
  const float *conv_cat_w, *conv_cat_b;   /* conv_cat_feature.weight (3,6,3,3), .bias (3) */
  const float *bn1_w, *bn1_b;             /* bn_1.weight / .bias (3) */
  float *bn1_rm, *bn1_rv;                 /* bn_1.running_mean / running_var (3), updated in training mode */
  long long* bn1_nbt;                     /* bn_1.num_batches_tracked or NULL */
  const float *conv_feat_w, *conv_feat_b; /* conv_feature.weight (3,3,3,3), .bias (3) */
  const float *bn2_w, *bn2_b;
  float *bn2_rm, *bn2_rv;
  long long* bn2_nbt;
} dpmn_distill_params;
typedef struct {                          /* gradients are ACCUMULATED (+=) into these */
  float *dconv_cat_w, *dconv_cat_b, *dbn1_w, *dbn1_b, *dconv_feat_w, *dconv_feat_b, *dbn2_w, *dbn2_b;
} dpmn_distill_grads;
size_t dpmn_distill_workspace_bytes(int B, int H, int W);
/* r (B,6,H,W): the raw conv outputs [conv_cat | conv_feature], state (24): [scale | shift | mean | rstd] x 6 channels -- both kept
 * by the caller for the backward; feat (B,3,H,W) = feature_cat; loss: one float on the device */
int dpmn_distill_forward_f32(const dpmn_distill_params* p, const float* x_deep, const float* x_shallow, int training, float* r,
                             float* state, float* feat, float* loss, void* workspace, size_t workspace_bytes, int B, int H, int W,
                             dpmn_stream_t stream);
/* gloss: d(objective) / d(loss), one float on the device; dfeat (B,3,H,W) = gradient wrt feature_cat or NULL; dx_deep / dx_shallow
 * (B,3,H,W) are WRITTEN (NULL: not needed) */
int dpmn_distill_backward_f32(const dpmn_distill_params* p, const dpmn_distill_grads* g, const float* x_deep, const float* x_shallow,
                              const float* r, const float* state, const float* dfeat, const float* gloss, float* dx_deep,
                              float* dx_shallow, void* workspace, size_t workspace_bytes, int B, int H, int W, dpmn_stream_t stream);

/* CMM channel gate (cmm.py:135-147) on the NHWC bottleneck x (B,P,C): out = x * sigmoid(fc2(relu(fc1(mean_p x)))) + x.
 * hidden_ws: B*Cmid floats of scratch. */
int dpmn_se_gate_f32(const float* x, const float* fc1_w, const float* fc1_b, const float* fc2_w, const float* fc2_b,
                     float* out, float* hidden_ws, int B, int P, int C, int Cmid, dpmn_stream_t stream);

/* ------------------------------------------------------------------ TSRN / TATT kernels (tatt.hip) */
/* BiGRU recurrence of GruBlock (tsrn.py:139-150; tatt.py:1070-1083).  gi: (pixels, 6*hidden) input
 * projection for both directions [fwd r z n | bwd r z n] with b_ih (and the folded conv1x1) already added;
 * w_hh (2, 3*hidden, hidden), b_hh (2, 3*hidden).  Sequence s starts at pixel
 * (s / inner)*outer_stride + (s % inner)*inner_stride and advances step_stride pixels per time step, so
 * the same kernel runs along W (rows as batch) or along H (the transposed gru1 call, tsrn.py:99).
 * out (pixels, 2*hidden) = [h_fwd | h_bwd] + res. */
int dpmn_bigru_f32(const float* gi, const float* w_hh, const float* b_hh, const float* res, float* out, int nseq,
                   int T, int inner, long outer_stride, long inner_stride, long step_stride, int hidden,
                   dpmn_stream_t stream);
/* y = act((x + add[m % add_rows]) . w^T + b) for tiny problems (tatt.py:209 fc_in, K/V projections of 26 slots) */
int dpmn_small_linear_f32(const float* x, const float* add, int add_rows, const float* w, const float* b, float* y,
                          int M, int N, int K, int act, float slope, dpmn_stream_t stream);
/* TransformerEncoder with one TransformerEncoderLayer.forward_post on (N, L<=32, 64) tokens
 * (transformer_v2.py:256-281, 455-469).  w12: HOST array of 12 device pointers
 * {in_proj_w, in_proj_b, out_proj_w, out_proj_b, linear1_w, linear1_b, linear2_w, linear2_b, norm1_w, norm1_b, norm2_w, norm2_b}. */
int dpmn_tatt_encoder_layer_f32(const float* src, const float* pos, const float* const* w12, float* mem, int N,
                                int L, int E, int nhead, dpmn_stream_t stream);
/* softmax(q k^T / sqrt(d)) v over S<=32 keys, 4 heads (transformer_v2.py:821-824); pw (N,L,S) head-averaged or NULL */
int dpmn_cross_attn_f32(const float* q, const float* k, const float* v, float* o, float* pw, int N, int L, int S, int E,
                        int nhead, dpmn_stream_t stream);
/* TPGSR's TSRN_TL (tsrn.py:226-227): F.interpolate(InfoGen map (N, 1, Win, C) NHWC, (H, W), mode 'bilinear', align_corners=True)
 * -> (N, H, W, C) NHWC, the text-prior map concatenated into every RecurrentResidualBlockTL */
int dpmn_tl_interp_f32(const float* in, float* out, int N, int Win, int C, int H, int W, dpmn_stream_t stream);
/* y = LayerNorm64(x + res); optional acc_out (+)= alpha * LayerNorm64'(y) (decoder.norm on intermediates,
 * transformer_v2.py:377-378, then mean over layers tatt.py:218) */
int dpmn_add_layernorm64_f32(const float* x, const float* res, const float* g, const float* b, float* y,
                             const float* g2, const float* b2, float* acc_out, float alpha, int accumulate, long M,
                             dpmn_stream_t stream);
/* one gate step of the query-embedding GRU (transformer_v2.py:177,215-218; quirk Q5) */
int dpmn_gru_gate_f32(const float* gi, const float* gh, float* h, float* hist, long hist_row_stride, int R, int H,
                      dpmn_stream_t stream);

/* ------------------------------------------------------------------ PGRM kernels (pgrm.hip) */
/* prior_fusion (optional, pf_w != NULL; pgrm.py:548) + PatchEmbed conv k=s=patch + LayerNorm (pgrm.py:419-426).
 * img NCHW (B,cin,Hi,Wi) -> tokens (B, Hi/patch*Wi/patch, C). */
int dpmn_patch_embed_ln_f32(const float* img, int cin, const float* pf_w, const float* pf_b, const float* pe_w,
                            const float* pe_b, const float* ln_w, const float* ln_b, float* tokens, int B, int Hi,
                            int Wi, int patch, int C, dpmn_stream_t stream);
/* + pos_drop (pgrm.py:550-551) in the epilogue: tokens = dropout(PatchEmbed(img)) with the mask of
 * dpmn_dropout_f32(tokens, n = B L C, p_drop, seed) -- bitwise that launch sequence */
int dpmn_patch_embed_ln_drop_f32(const float* img, int cin, const float* pf_w, const float* pf_b, const float* pe_w,
                                 const float* pe_b, const float* ln_w, const float* ln_b, float* tokens, int B, int Hi, int Wi,
                                 int patch, int C, float p_drop, unsigned long long seed, dpmn_stream_t stream);
/* multi-window cross attention core (pgrm.py:197-266): q (B,L,C), kv (B,L,2C) -> out (B,L,C) in
 * window-major order per group (quirk Q1).  bias_tables / windows / shifts are HOST arrays of length
 * n_groups (the table pointers themselves are device pointers). */
int dpmn_window_attn_f32(const float* q, const float* kv, const float* const* bias_tables, const int* windows,
                         const int* shifts, int n_groups, int heads_per_group, float* out, int B, int H, int W,
                         int C, dpmn_stream_t stream);
/* SKConv gate (pgrm.py:86-91): GAP partials -> fc1 -> GELU -> fc2 -> softmax over groups -> (B,G,C/G) */
int dpmn_sk_gate_f32(const float* colsum_partials, int parts_per_image, int L, const float* fc1_w,
                     const float* fc1_b, const float* fc2_w, const float* fc2_b, float* attn_vec, int B, int C,
                     int groups, int dmid, dpmn_stream_t stream);
/* Mlp.depthwise_conv + act_2 on the raw (B,Ch,r,r) view (pgrm.py:34-36) */
int dpmn_dwconv3x3_gelu_f32(const float* y, const float* w, const float* bias, float* g, int B, int Ch, int r,
                            dpmn_stream_t stream);
/* conv_before_upsample (2 convs + LeakyReLU) + PixelShuffle + weight_list scaling + residuals
 * (pgrm.py:559-565).  weight_list / residuals: HOST arrays of device pointers; residuals[0] is
 * ignored like the reference does (quirk Q11).  mid_ws: B*H*W*hidden*patch^2 + 16*(9*C+32) floats. */
int dpmn_pgrm_tail_f32(const float* tokens, const float* w0, const float* b0, const float* w1, const float* b1,
                       const float* const* weight_list, const float* const* residuals, int n_residuals,
                       float* mid_ws, float* out, int B, int H, int W, int C, int hidden, int patch,
                       dpmn_stream_t stream);
/* reuse_pack != 0: mid_ws still holds conv_before_upsample[0]'s packed weights of a previous call with the same (unchanged) weights */
int dpmn_pgrm_tail_reuse_f32(const float* tokens, const float* w0, const float* b0, const float* w1, const float* b1,
                             const float* const* weight_list, const float* const* residuals, int n_residuals, float* mid_ws,
                             float* out, int B, int H, int W, int C, int hidden, int patch, int reuse_pack, dpmn_stream_t stream);

/* ------------------------------------------------------------------ image-space helpers (misc.hip) */
/* toMask (utils/util.py:27-35) for a batch: img NCHW, first 3 channels used, img_stride = floats between images;
 * out (B,3,H,W) in {0,1}. */
int dpmn_to_mask_f32(const float* img, long img_stride, float* out, int B, int H, int W, dpmn_stream_t stream);
/* TBSRN FeatureEnhancer (model/tbsrn.py:76-92, config 3's PSN), tbsrn.hip.
 * mha32: MultiHeadedAttention core (tbsrn.py:110-150) -- qkv (B*L, 3*heads*32) rows [q|k|v] (the three input linears
 * fused into one GEMM by the caller), out (B*L, heads*32) = softmax(q k^T * scale) v per head over the L positions of an
 * image; L % 64 == 0.  layernorm_std: tbsrn.py:23-36, a2 * (x - mean) / (unbiased std + eps) + b2, C in {64,128,256}. */
int dpmn_mha32_f32(const float* qkv, float* out, int B, int L, int heads, float scale, dpmn_stream_t stream);
/* the same core with d_k = 64: VisionLAN's MultiHeadAttention (model/VisionLAN/modules/modules.py:43-81, 8 heads x 64, L = 256) */
int dpmn_mha64_f32(const float* qkv, float* out, int B, int L, int heads, float scale, dpmn_stream_t stream);

/* ------------------------------------------------------------------ in-loop text prior (visionlan.hip), SURVEY.md section 8(f)-1:
 * replaces the per-image host loop of interfaces/super_resolution.py:174-199.
 * vl_resize:   parse_visionlan_data (interfaces/base.py:473-478): img (B, >=3, H, W) planes with batch stride img_stride ->
 *              uint8 quantisation, bilinear (half-pixel centres, edge clamp) to Ho x Wo, /255, stored NHWC with 4 channels (ch 3 = 0).
 * vl_tokens:   MLM_VRM.forward (VisionLAN.py:70-75) + PositionalEncoding: NHWC features (B,Hf,Wf,C) -> (B, Wf*Hf, C), token w*Hf+h, + table.
 * vl_pp_pool:  PP_layer.forward (modules.py:168-171) from the position scores on + Prediction.w_vrm (199-202):
 *              scores (B*256, ld) [column n = step n], enc (B,256,512) -> logits (B, n_steps, n_class).
 * vl_decode:   MLM_VRM.forward 107-126: cls (B, max_len) = first arg-max per step, length = first EOS (class 0) step + 1, else max_len.
 * text_prior_compose: replaces utils/render_standard_text.py (pygame + cv2): the decoded string is laid out from a glyph atlas
 *              (2 cases, n_glyph classes, GH x GW cells, per-glyph advance) and stretched bilinearly to (B, 2, Ho, Wo), values
 *              rounded to integers 0..255 (quirk Q6).  Specification: oracle/visionlan.py compose_text_prior. */
int dpmn_vl_resize_f32(const float* img, long img_stride, float* out_nhwc4, int B, int H, int W, int Ho, int Wo, dpmn_stream_t stream);
int dpmn_vl_tokens_f32(const float* feat_nhwc, const float* pos_table, float* tokens, int B, int Hf, int Wf, int C, dpmn_stream_t stream);
int dpmn_vl_pp_pool_f32(const float* scores, int ld_scores, const float* enc, const float* w_vrm, const float* b_vrm, float* logits,
                        int B, int L, int C, int n_steps, int n_class, dpmn_stream_t stream);
int dpmn_vl_decode_i32(const float* logits, int* cls, int* length, int B, int n_steps, int n_class, int max_len, dpmn_stream_t stream);
int dpmn_text_prior_compose_f32(const int* cls, const int* length, const float* atlas, const int* advance, float* out, int B, int max_len,
                                int n_glyph, int GH, int GW, int Ho, int Wo, dpmn_stream_t stream);
int dpmn_layernorm_std_f32(const float* x, const float* a2, const float* b2, float eps, float* y, long M, int C,
                           dpmn_stream_t stream);
/* GPU half of the TextZoom collate (dataset/dataset.py:1266-1319 resizeNormalize, 2007-2013 alignCollate_realWTLAMask.__call__):
 * img (B, H, W, 3) uint8 = the PIL-resized RGB pixels, out (B, 3 + with_mask, H, W) = ToTensor (/255, CHW) and, with_mask, the mask
 * channel (PIL RGB -> L, threshold at the image's mean L: L > mean ? 0 : 1). */
int dpmn_collate_u8_f32(const unsigned char* img, float* out, int B, int H, int W, int with_mask, dpmn_stream_t stream);

/* rotation augmentation of the trainer (utils/util.py:37-58 torch_rotate_img; super_resolution.py:144-151, 358-365):
 * per-image affine with aspect-ratio jitter -> affine_grid (align_corners=False) -> bilinear grid_sample, zeros padding.
 * img / out: contiguous NCHW (N,C,H,W); arc, rand_offs: (N) */
int dpmn_rotate_img_f32(const float* img, const float* arc, const float* rand_offs, float off_range, float* out, int N, int C,
                        int H, int W, dpmn_stream_t stream);
/* out = alpha*a + (1-alpha)*b over chw floats per image (interfaces/super_resolution.py:449) */
int dpmn_blend_f32(const float* a, long a_stride, const float* b, long b_stride, float* out, float alpha, int B,
                   int chw, dpmn_stream_t stream);
/* out2[0] = PSNR (utils/ssim_psnr.py:9-13), out2[1] = SSIM (28-48, 62-79) over the first C channels */
size_t dpmn_psnr_ssim_workspace_bytes(int B, int C, int H, int W);
int dpmn_psnr_ssim_f32(const float* x, long x_stride, const float* y, long y_stride, float* out2, void* workspace,
                       int B, int C, int H, int W, dpmn_stream_t stream);

/* ------------------------------------------------------------------ training: backward building blocks
 * (backward.hip, backward_pgrm.hip).  Autograd of the reference (loss.backward(), super_resolution.py:270) is
 * replaced by explicit kernels; weight-gradient entries ACCUMULATE into their output (caller zeroes, like
 * optimizer.zero_grad, super_resolution.py:141). */
/* dw (N,K) += dy (M,N)^T . x (M,K): nn.Linear weight gradient */
int dpmn_gemm_tn_f32(const float* dy, const float* x, float* dw, float* db /* (N) += column sums of dy, or NULL */, int M,
                     int N, int K, float* ws /* split partials; NULL or too small: fp32 atomics instead */, size_t ws_bytes,
                     dpmn_stream_t stream);
/* the same in two steps, for callers that issue many of them: dpmn_gemm_tn_partial_f32 launches only the split partial sums (into
 * ws, >= dpmn_gemm_tn_partial_bytes; every pending call needs its OWN region) and fills *pending; dpmn_tn_reduce_multi_f32 then adds
 * any number of pending results into their dw / db in one launch per 16 (same arithmetic and order as the single-call form). */
typedef struct {
  const float* part;
  float* dw;
  float* db;
  int NK, N, splits;
} dpmn_tn_pending;
size_t dpmn_gemm_tn_partial_bytes(int M, int N, int K);
/* n weight gradients whose operands are live at the same time (the Linears of one Swin block, pgrm.py:315-331): the same bits as n
 * dpmn_gemm_tn_f32 calls in array order; the products with N, K multiples of 48 and a workspace share ONE partial-sum launch. */
typedef struct {
  const float* dy;
  const float* x;
  float* dw;
  float* db;
  int M, N, K;
  float* ws;
  size_t ws_bytes;
} dpmn_tn_item;
int dpmn_gemm_tn_group_f32(const dpmn_tn_item* items /* HOST array */, int n, dpmn_stream_t stream);
int dpmn_gemm_tn_partial_f32(const float* dy, const float* x, float* dw, float* db, int M, int N, int K, float* ws, size_t ws_bytes,
                             dpmn_tn_pending* pending, dpmn_stream_t stream);
int dpmn_tn_reduce_multi_f32(const dpmn_tn_pending* pending /* HOST array */, int n, dpmn_stream_t stream);
/* Deferred ordered reductions: between dpmn_reduce_defer_begin() and dpmn_reduce_defer_flush(end = 1, stream) every "partial rows added
 * in order" finish of the backward entry points (dpmn_gemm_tn_f32, dpmn_colsum_det_f32, dpmn_layernorm_bwd_det_f32,
 * dpmn_rows_reduce_f32 and the functions built on it) is queued instead of launched, and the flush runs the queue as multi-descriptor
 * launches (sums into the same tensor in separate launches, in queue order).  The CALLER keeps every workspace it passed to those
 * functions untouched until the flush.  Per host thread.  dpmn_reduce_defer_enable(0 / 1) pauses / resumes queueing (for a reduction
 * whose result is read immediately); dpmn_reduce_defer_push queues a descriptor of the caller's own. */
int dpmn_reduce_defer_begin(void);
int dpmn_reduce_defer_enable(int on);
int dpmn_reduce_defer_push(const dpmn_tn_pending* p);
int dpmn_reduce_defer_pending(void);
int dpmn_reduce_defer_flush(int end, dpmn_stream_t stream);
/* db (N) += column sums of dy (M,N) */
int dpmn_colsum_f32(const float* dy, float* db, long M, int N, dpmn_stream_t stream);
/* the same without atomics (per-block partial sums in ws, >= ceil(M / 256) * N floats, added in block order): bitwise reproducible */
int dpmn_colsum_det_f32(const float* dy, float* db, long M, int N, float* ws, size_t ws_bytes, dpmn_stream_t stream);
/* LayerNorm backward from the saved pre-norm input x; dx written or accumulated; dgamma/dbeta accumulated */
int dpmn_layernorm_bwd_f32(const float* x, const float* dy, const float* gamma, float eps, float* dx, int accumulate_dx,
                           float* dgamma, float* dbeta, long M, int C, dpmn_stream_t stream);
/* the same without atomics: per-block [dgamma | dbeta] partials in ws (>= 512 * 2 C floats), added in block order */
int dpmn_layernorm_bwd_det_f32(const float* x, const float* dy, const float* gamma, float eps, float* dx, int accumulate_dx,
                               float* dgamma, float* dbeta, long M, int C, float* ws, size_t ws_bytes, dpmn_stream_t stream);
/* + a second output masked_out = dx_final * Dropout mask * DropPath mask (the masks of dpmn_dropout_f32(dx, n = M C, row_len, p_elem,
 * seed_elem, p_row, seed_row)): the masked copy the next Linear's backward consumes (pgrm.py:329-330 reversed), C = 96 / 192 */
int dpmn_layernorm_bwd_det_drop_f32(const float* x, const float* dy, const float* gamma, float eps, float* dx, int accumulate_dx,
                                    float* dgamma, float* dbeta, long M, int C, float* ws, size_t ws_bytes, float* masked_out,
                                    float p_elem, unsigned long long seed_elem, float p_row, unsigned long long seed_row, long row_len,
                                    dpmn_stream_t stream);
int dpmn_layernorm_f32(const float* x, const float* gamma, const float* beta, float eps, float* y, long M, int C,
                       dpmn_stream_t stream);
/* dpre = dy * act'(pre) ; y = act(x) */
int dpmn_act_bwd_f32(const float* dy, const float* pre, float* dpre, int act, float slope, long n, dpmn_stream_t stream);
int dpmn_act_fwd_f32(const float* x, float* y, int act, float slope, long n, dpmn_stream_t stream);
/* y (+)= a*x + b*z (z may be NULL) */
int dpmn_axpby_f32(const float* x, const float* z, float* y, float a, float b, int accumulate, long n, dpmn_stream_t stream);
/* out[row % mod] += sum_c x[row][c] */
int dpmn_rowsum_mod_f32(const float* x, float* out, long rows, int cols, int mod, dpmn_stream_t stream);
/* atomics-free forms (bitwise reproducible): per-image partial rows in ws, added in image order by dpmn_rows_reduce_f32
 * (dw[e] += sum_z part[z][e], db[n] += sum_z part[z][NK + n] over `rows` rows of NK + N floats) */
int dpmn_rows_reduce_f32(const float* part, float* dw, float* db, int NK, int N, int rows, dpmn_stream_t stream);
int dpmn_rowsum_mod_det_f32(const float* x, float* out, long rows, int cols, int mod, float* ws /* rows floats */, size_t ws_bytes,
                            dpmn_stream_t stream);
size_t dpmn_dwconv3x3_bwd_det_bytes(int B, int Ch, int r);      /* workspace bytes of dpmn_dwconv3x3_bwd_fused_det_f32 */
int dpmn_dwconv3x3_bwd_fused_det_f32(const float* P, const float* dg, const float* gpre, const float* w, float* dP, float* dw, float* db,
                                     int in_gelu, int out_gelu_bwd, float p_drop, unsigned long long seed, int B, int Ch, int r,
                                     float* ws /* B * Ch * 10 floats */, size_t ws_bytes, dpmn_stream_t stream);
/* ImageLoss (loss/image_loss.py:15-43): loss = w_mse*MSE + w_grad*L1(gradient maps of the first 3 channels).
 * U, V: (B,3,H,W) scratch written by the forward and consumed by the backward (may be NULL when gradient == 0).
 * grad_out (B,C,H,W) (+)= grad_scale[0] * dloss/dout. */
size_t dpmn_image_loss_workspace_bytes(int B, int C, int H, int W);
int dpmn_image_loss_fwd_f32(const float* out, long out_stride, const float* tgt, long tgt_stride, float w_mse,
                            float w_grad, int gradient, float* loss, float* U, float* V, void* workspace, int B, int C,
                            int H, int W, dpmn_stream_t stream);
int dpmn_image_loss_bwd_f32(const float* out, long out_stride, const float* tgt, long tgt_stride, const float* U,
                            const float* V, const float* grad_scale, float w_mse, float w_grad, int gradient,
                            float* grad_out, int accumulate, int B, int C, int H, int W, dpmn_stream_t stream);
/* window attention backward (pgrm.py:197-266): dq (B,L,C), dkv (B,L,2C) written; dtables[g] accumulated */
int dpmn_window_attn_bwd_f32(const float* q, const float* kv, const float* const* bias_tables, const int* windows,
                             const int* shifts, int n_groups, int heads_per_group, const float* dout, float* dq,
                             float* dkv, float* const* dtables, int B, int H, int W, int C, dpmn_stream_t stream);
/* Train-mode stochastic regularisers of PGRM (nn.Dropout pgrm.py:24,32,40,494,554-555; attn_drop pgrm.py:180,248; timm DropPath
 * pgrm.py:310,329-330).  torch's Philox stream cannot be replayed outside torch, so a mask element is a pure function of
 * (seed, element index): splitmix64 finaliser of idx*0x9E3779B97F4A7C15 + seed, top 24 bits / 2^24 = u, keep iff u >= p, kept
 * values scaled by 1/(1-p).  Forward and backward regenerate the mask from the seed; nothing is stored.
 *   y = res + x * m_elem(i) * m_row(i / row_len)      res may be NULL; p_elem / p_row = 0 disables that factor; y may alias x.
 * window attention: mask index ((((b*n_groups + g)*2 + head)*L + window-major query token)*N + key row in window). */
int dpmn_dropout_f32(const float* x, const float* res, float* y, long n, long row_len, float p_elem,
                     unsigned long long seed_elem, float p_row, unsigned long long seed_row, dpmn_stream_t stream);
int dpmn_window_attn_drop_f32(const float* q, const float* kv, const float* const* bias_tables, const int* windows,
                              const int* shifts, int n_groups, int heads_per_group, float* out, int B, int H, int W, int C,
                              float p_drop, unsigned long long seed, dpmn_stream_t stream);
int dpmn_window_attn_drop_bwd_f32(const float* q, const float* kv, const float* const* bias_tables, const int* windows,
                                  const int* shifts, int n_groups, int heads_per_group, const float* dout, float* dq,
                                  float* dkv, float* const* dtables, int B, int H, int W, int C, float p_drop,
                                  unsigned long long seed, dpmn_stream_t stream);
/* the same with the bias-table gradients as per-block partial rows instead of atomics (bitwise reproducible): dtable_parts[g] is a
 * (dpmn_window_attn_bwd_part_rows(B, H, W), (2 ws_g - 1)^2 * heads_per_group) buffer; rows_out[g] = the rows group g wrote, which
 * the caller adds in order into the table gradient (dpmn_rows_reduce_f32) */
int dpmn_window_attn_drop_bwd_det_f32(const float* q, const float* kv, const float* const* bias_tables, const int* windows,
                                      const int* shifts, int n_groups, int heads_per_group, const float* dout, float* dq, float* dkv,
                                      float* const* dtable_parts, int* rows_out, int B, int H, int W, int C, float p_drop,
                                      unsigned long long seed, dpmn_stream_t stream);
int dpmn_window_attn_bwd_part_rows(int B, int H, int W);
/* Row a15 -- the PSNs' spatial-transformer front end (reached only in PSN train mode: tatt.py / tbsrn.py `if self.stn and
 * self.training`).
 *   maxpool : nn.MaxPool2d(k, stride k) over NHWC (stn_head.py:36-46); scale/shift != NULL applies the producing
 *             conv3x3_block's BatchNorm affine + ReLU on load (train mode: batch statistics from dpmn_bn_finalize_f32).
 *   stn_fc  : STNHead.forward after the conv stack (stn_head.py:94-100): x = last block's raw NHWC output (B,1,W2,512/W2)
 *             (+ affine/ReLU on load), NCHW flatten, stn_fc1 = Linear(512,512) [w1t = weight transposed, (in,out)] +
 *             BatchNorm1d (training != 0: batch statistics and running-stat update, needs B > 1) + ReLU -> img_feat (B,512);
 *             ctrl (B,n_out) = stn_fc2(0.1 * img_feat).
 *   tps_sample : TPSSpatialTransformer.forward (tps_spatial_transformer.py:97-112) incl. F.grid_sample(bilinear, zeros,
 *             align_corners=False): img (B,C,Hin,Win) NCHW, ctrl (B,N,2), inverse_kernel (N+3,N+3), coord_repr (Hout*Wout,N+3)
 *             -> out (B,C,Hout,Wout), src_coord (B,Hout*Wout,2) (the unclamped source coordinates the reference also returns). */
int dpmn_maxpool_f32(const float* x, const float* scale, const float* shift, float* y, int B, int H, int W, int C, int kh,
                     int kw, dpmn_stream_t stream);
int dpmn_stn_fc_f32(const float* x, const float* in_scale, const float* in_shift, int W2, const float* w1t, const float* b1,
                    const float* bn_gamma, const float* bn_beta, float* running_mean, float* running_var, int training,
                    float momentum, float eps, const float* w2, const float* b2, float* img_feat, float* ctrl, int B,
                    int n_out, dpmn_stream_t stream);
int dpmn_tps_sample_f32(const float* img, const float* ctrl, const float* inverse_kernel, const float* coord_repr, float* out,
                        float* src_coord, int B, int C, int Hin, int Win, int Hout, int Wout, int N, dpmn_stream_t stream);
/* SKConv backward pieces (pgrm.py:79-96) */
int dpmn_sk_select_only_f32(const float* cat, const float* attn_vec, float* V, long M, int L, int C, int G, dpmn_stream_t stream);
int dpmn_sk_select_bwd_f32(const float* cat, const float* attn_vec, const float* dV, float* dcat, float* dA, int B, int L,
                           int C, int G, dpmn_stream_t stream);
int dpmn_sk_gate_bwd_f32(const float* colsum_partials, int parts_per_image, int L, const float* fc1_w, const float* fc1_b,
                         const float* fc2_w, const float* attn_vec, const float* dA, float* dS, float* dfc1_w,
                         float* dfc1_b, float* dfc2_w, float* dfc2_b, int B, int C, int G, int dmid, dpmn_stream_t stream);
/* atomics-free forms (bitwise reproducible): dA_part is (ceil(L / 32), B, C) partial rows that the gate backward adds in order; the
 * gate's weight gradients come back as per-image rows wpart2 (B, C dmid + C) = [dfc2_w | dfc2_b] and wpart1 (B, dmid C + dmid) =
 * [dfc1_w | dfc1_b] for dpmn_rows_reduce_f32 (pgrm.py:86-93 backward) */
int dpmn_sk_select_bwd_det_f32(const float* cat, const float* attn_vec, const float* dV, float* dcat, float* dA_part, int B, int L,
                               int C, int G, dpmn_stream_t stream);
/* dcat = A dV written, not accumulated (fresh buffer: no zero fill, no read-modify-write) */
int dpmn_sk_select_bwd_det_set_f32(const float* cat, const float* attn_vec, const float* dV, float* dcat, float* dA_part, int B, int L,
                                   int C, int G, dpmn_stream_t stream);
int dpmn_sk_gate_bwd_det_f32(const float* colsum_partials, int parts_per_image, int L, const float* fc1_w, const float* fc1_b,
                             const float* fc2_w, const float* attn_vec, const float* dA_part, int nparts, float* dS, float* wpart2,
                             float* wpart1, int B, int C, int G, int dmid, dpmn_stream_t stream);
int dpmn_sk_feats_grad_f32(const float* dout, const float* feats, const float* dS, float* dfeats, long M, int L, int C,
                           dpmn_stream_t stream);
/* Mlp depthwise conv without the activation (training keeps the pre-activation), its backward, and the
 * pointwise-conv weight gradient dw (Ch,Ch) += sum_b dz_b . g_b^T over the raw (B,Ch,L) views */
/* y = fc1's pre-activation: GELU applied on load (pgrm.py:33-35 in one pass), g = GELU(dwconv(GELU(y))) */
int dpmn_dwconv3x3_gelu_in_f32(const float* y, const float* w, const float* bias, float* g, int B, int Ch, int r, dpmn_stream_t stream);
int dpmn_dwconv3x3_f32(const float* y, const float* w, const float* bias, float* g, int B, int Ch, int r, dpmn_stream_t stream);
/* the same two kernels with the GELUs of the Mlp chain (pgrm.py:31-37: fc1 -> GELU -> dwconv -> GELU -> pointwise) fused in:
 *   _train : gpre = dwconv(in_gelu ? GELU(y) : y) and g = GELU(gpre) in one pass (the backward needs both)
 *   _bwd_fused : dg is multiplied by GELU'(gpre) on load (gpre may be NULL), P is GELU'd on load when in_gelu (then it is fc1's
 *                pre-activation), and dP is multiplied by GELU'(P) before the store when out_gelu_bwd
 *   p_drop > 0 : nn.Dropout(p_drop) between fc1's GELU and the conv (pgrm.py:34), mask = f(seed, element index) as in
 *                dpmn_dropout_f32, applied on load (forward input) and on dP (before GELU') */
int dpmn_dwconv3x3_train_f32(const float* y, const float* w, const float* bias, float* gpre, float* g, int in_gelu, float p_drop,
                             unsigned long long seed, int B, int Ch, int r, dpmn_stream_t stream);
int dpmn_dwconv3x3_bwd_fused_f32(const float* P, const float* dg, const float* gpre, const float* w, float* dP, float* dw, float* db,
                                 int in_gelu, int out_gelu_bwd, float p_drop, unsigned long long seed, int B, int Ch, int r,
                                 dpmn_stream_t stream);
int dpmn_dwconv3x3_bwd_f32(const float* P, const float* dg, const float* w, float* dP, float* dw, float* db, int B, int Ch,
                           int r, dpmn_stream_t stream);
int dpmn_pointwise_wgrad_f32(const float* dz, const float* g, float* dw, int B, int Ch, int L, dpmn_stream_t stream);
/* atomics-free form: ws >= 32 * Ch * Ch floats (the k splits' partial results, added in split order) */
int dpmn_pointwise_wgrad_det_f32(const float* dz, const float* g, float* dw, int B, int Ch, int L, float* ws, size_t ws_bytes,
                                 dpmn_stream_t stream);
size_t dpmn_pointwise_wgrad_det_bytes(int Ch, int L);      /* workspace of the call above (its split count depends on Ch) */

/* PGRM tail in training form: out = lrelu(c1) pixel-shuffled * weight_list_0 + sum_i residual_i * weight_list_i
 * (pgrm.py:560-565; residual 0 skipped, Q11) and its backward (dresiduals[i] may be NULL; dweight_list accumulated) */
int dpmn_pgrm_tail_elem_f32(const float* c1, const float* const* weight_list, const float* const* residuals,
                            int n_residuals, float* out, int B, int H, int W, dpmn_stream_t stream);
int dpmn_pgrm_tail_elem_bwd_f32(const float* dout, const float* c1, const float* const* weight_list,
                                const float* const* residuals, float* const* dresiduals, float* const* dweight_list,
                                int n_residuals, float* dc1, int B, int H, int W, dpmn_stream_t stream);
/* PatchEmbed (+prior_fusion) backward (pgrm.py:419-426, 548): see backward_pgrm.hip */
int dpmn_patch_embed_bwd_f32(const float* img, int cin, const float* pf_w, const float* pf_b, const float* pe_w,
                             const float* pe_b, const float* ln_w, const float* dtok, float* dconv, float* patches,
                             float* dgamma, float* dbeta, int B, int Hi, int Wi, int C, dpmn_stream_t stream);
int dpmn_patch_scatter_f32(const float* din, float* dimg, int cimg, int B, int Hi, int Wi, dpmn_stream_t stream);
int dpmn_prior_fusion_wgrad_f32(const float* din, const float* prior, float* dpf_w, float* dpf_b, int B, int Hi, int Wi,
                                dpmn_stream_t stream);
/* atomics-free forms of the two calls above (bitwise reproducible): per-block partial rows for dpmn_rows_reduce_f32 --
 * ln_part (ceil(tokens / 64), 2 C) = [dgamma | dbeta] of the patch-embed LayerNorm, part (ceil(B Hi Wi / 256), 57) = [dw (54) | db (3)]
 * of prior_fusion */
int dpmn_patch_embed_bwd_det_f32(const float* img, int cin, const float* pf_w, const float* pf_b, const float* pe_w,
                                 const float* pe_b, const float* ln_w, const float* dtok, float* dconv, float* patches,
                                 float* ln_part, int B, int Hi, int Wi, int C, dpmn_stream_t stream);
/* dtok is the gradient behind pos_drop: the forward's mask (p_drop, seed) is applied on load */
int dpmn_patch_embed_bwd_det_drop_f32(const float* img, int cin, const float* pf_w, const float* pf_b, const float* pe_w,
                                      const float* pe_b, const float* ln_w, const float* dtok, float* dconv, float* patches,
                                      float* ln_part, int B, int Hi, int Wi, int C, float p_drop, unsigned long long seed,
                                      dpmn_stream_t stream);
/* the same (embed_dim 96) with the PatchEmbed conv's weight / bias gradient as per-block partial rows instead of the `patches` output:
 * w_part is (ceil(tokens / 64), 12 C + C) rows of [dW (C, 3, 2, 2) flattened | db (C)], one row per block, to be added in row order
 * (dpmn_rows_reduce_f32 with NK = 12 C, N = C) -- replaces dW = dconv^T . patches as a separate skinny GEMM.
 * dimg (optional; 3-channel img, no prior_fusion): the gradient of img, (B, 3, Hi, Wi), written directly (replaces the
 * dconv . W Linear + dpmn_patch_scatter_f32 into a zero-filled image) */
int dpmn_patch_embed_bwd_det_wgrad_f32(const float* img, int cin, const float* pf_w, const float* pf_b, const float* pe_w,
                                       const float* pe_b, const float* ln_w, const float* dtok, float* dconv, float* ln_part,
                                       float* w_part, float* dimg, int B, int Hi, int Wi, int C, float p_drop, unsigned long long seed,
                                       dpmn_stream_t stream);
int dpmn_prior_fusion_wgrad_det_f32(const float* din, const float* prior, float* part, int B, int Hi, int Wi, dpmn_stream_t stream);
/* conv weight gradient in the packed (Cout, Kp) layout, train-mode BatchNorm plumbing, CMM gate backward (conv_bwd.hip) */
int dpmn_conv2d_wgrad_f32(const dpmn_conv_desc* d, const float* dy, float* dwp,
                          int nslots /* > 1: dwp holds nslots copies (Cout*Kp apart); pixel splits spread over them so that
                                      * small tiles (DistillModule: 4 x 72) do not pile thousands of atomics on one address */,
                          dpmn_stream_t stream);
/* same, accumulated straight into the parameter's own layout (nn.Conv2d (Cout,Cin,KH,KW), nn.ConvTranspose2d
 * (Cin,Cout,KH,KW) flipped, or one phase of ConvTranspose2d(4,2,1)):
 *   dw[base + co*s_co + ci*s_ci + ky*s_ky + kx*s_kx] += ...   for co < co_lim, ci < ci_lim (padding channels dropped) */
/* parameter layout -> packed (Cout,Kp) weights for dpmn_conv2d_nhwc_f32 (zero filled beyond co_lim / ci_lim / K):
 *   wp[co][(ky*KW+kx)*cin + ci] = w[base + co*s_co + ci*s_ci + ky*s_ky + kx*s_kx]
 * covers nn.Conv2d, flipped nn.ConvTranspose2d, the ConvTranspose2d(4,2,1) phases and every data-gradient re-pack */
int dpmn_conv_pack_f32(const float* w, float* wp, int Cout, int cin, int KH, int KW, int co_lim, int ci_lim, long s_co, long s_ci,
                       long s_ky, long s_kx, long base, dpmn_stream_t stream);
/* every pack of a training step in one launch: descs = device array of
 *   { const float* w; float* wp; long s_co, s_ci, s_ky, s_kx, base, n_elems; int Cout, Kp, K, cin, KW, co_lim, ci_lim,
 *     co_t, ci_t, order, nci, pad; }                                                       (112 bytes)
 * (the arguments of dpmn_conv_pack_f32, n_elems = Cout * Kp; (co_t, ci_t, order) from dpmn_conv_pack_tile_shape,
 * nci = ceil(cin / ci_t)); a descriptor owns ceil(Cout / co_t) * nci blocks, block_prefix[d] = its first block,
 * n_blocks = total.  Built and cached by dpmn_amd/model/packing.py (PackCache). */
int dpmn_conv_pack_multi_f32(const void* descs, const int* block_prefix, int n_desc, int n_blocks, dpmn_stream_t stream);
/* LDS tile of the pack / unpack kernels for one weight: shape3 = {co_t, ci_t, order} (order 1: co is the faster axis of the
 * parameter layout, i.e. |s_co| < |s_ci|) */
int dpmn_conv_pack_tile_shape(int Cout, int cin, int taps, long s_co, long s_ci, int* shape3);
/* the same gradient without atomics: the pixel range is cut into `slots` splits (dpmn_conv2d_wgrad_excl_slots reports the
 * count for a descriptor; 0 = tiny gradient over very many pixels, use dpmn_conv2d_wgrad_f32 with 32 slotted copies) and
 * every (tile, split) block STORES its partial tile into copy `split` of dwp (slots, Cout, Kp):
 * deterministic, the copies need no zero-init; sum them with dpmn_conv2d_wgrad_unpack_f32(clear = 0, nslots = slots) */
int dpmn_conv2d_wgrad_excl_slots(const dpmn_conv_desc* d, int* slots);
int dpmn_conv2d_wgrad_excl_f32(const dpmn_conv_desc* d, const float* dy, float* dwp, int slots, dpmn_stream_t stream);
/* packed (Cout,Kp) gradient -> += into the parameter layout (same stride convention as below); clear != 0 zeroes the
 * packed buffer afterwards so that a persistent workspace needs no memset before its next dpmn_conv2d_wgrad_f32 */
int dpmn_conv2d_wgrad_unpack_f32(float* dwp, float* dw, int Cout, int cin, int KH, int KW, int co_lim, int ci_lim, long s_co,
                                 long s_ci, long s_ky, long s_kx, long base, int clear, int nslots /* copies to sum */,
                                 dpmn_stream_t stream);
/* Every weight-gradient unpack of a backward pass in ONE launch: descs = device array of the 112-byte pack descriptors of
 * dpmn_conv_pack_multi_f32 (w = the gradient tensor, wp = the exclusive-slot workspace, last int = slot count), block_prefix[i] = first
 * block of descriptor i.  The descriptors' destination elements must be disjoint. */
int dpmn_conv2d_wgrad_unpack_multi_f32(const void* descs, const int* block_prefix, int n_desc, int n_blocks, dpmn_stream_t stream);
int dpmn_conv2d_wgrad_strided_f32(const dpmn_conv_desc* d, const float* dy, float* dw, int co_lim, int ci_lim, long s_co,
                                  long s_ci, long s_ky, long s_kx, long base, dpmn_stream_t stream);
/* num_batches_tracked (int64, may be NULL) is incremented by one; clear_stats != 0 zeroes the (32,2,C) fp64 slots after they are
 * read, so that a persistent statistics buffer serves the next convolution without a memset */
int dpmn_bn_finalize_f32(float* stats, const float* gamma, const float* beta, float count, float eps, float momentum,
                         float* scale, float* shift, float* mean, float* rstd, float* running_mean, float* running_var,
                         int C, long long* num_batches_tracked, int clear_stats, dpmn_stream_t stream);
int dpmn_affine_act_bwd_f32(const float* dA, const float* r, const float* scale, const float* shift, int act, float* G,
                            int accumulate, long pixels, int C, dpmn_stream_t stream);
/* dpmn_affine_act_bwd_f32 by the LAST consumer of a BatchNorm output, with the producer's BatchNorm-backward reduction folded in:
 * sums (2, C) DOUBLES, zero on entry, += (sum G, sum G * xhat) of the completed G (xhat = (r - mean) * rstd of the producer's
 * batch statistics).  dpmn_bn_bwd_apply_f32 is dpmn_bn_bwd_f32 without its own reduction pass over G and r (autograd of
 * nn.BatchNorm2d in batch-statistics mode, cmm.py:12); sums_ws: 2 C floats of scratch. */
int dpmn_affine_act_bwd_stats_f32(const float* dA, const float* r, const float* scale, const float* shift, int act, float* G,
                                  int accumulate, long pixels, int C, const float* mean, const float* rstd, double* sums,
                                  dpmn_stream_t stream);
int dpmn_bn_bwd_apply_f32(const float* G, const float* r, const float* gamma, const float* mean, const float* rstd, const double* sums,
                          float* sums_ws, float* dr, float* dgamma, float* dbeta, long pixels, int C, dpmn_stream_t stream);
int dpmn_bn_bwd_f32(const float* G, const float* r, const float* gamma, const float* mean, const float* rstd, float* sums_ws,
                    float* dr, float* dgamma, float* dbeta, long pixels, int C, dpmn_stream_t stream);
int dpmn_se_gate_bwd_f32(const float* x, const float* dg, const float* fc1_w, const float* fc1_b, const float* fc2_w,
                         const float* fc2_b, float* dx, float* dfc1_w, float* dfc1_b, float* dfc2_w, float* dfc2_b,
                         float* ws /* B*(5C+2Cmid) + 2*C*Cmid floats */, int B, int P, int C, int Cmid, dpmn_stream_t stream);

/* DistillModule pieces (distill_module.py:18-31): y = act(scale*r+shift); L1 loss forward / backward */
int dpmn_affine_act_fwd_f32(const float* r, const float* scale, const float* shift, int act, float* y, long pixels, int C,
                            dpmn_stream_t stream);
int dpmn_l1_loss_fwd_f32(const float* a, const float* b, float inv_count, float* loss, float* part_ws, long n, dpmn_stream_t stream);
int dpmn_l1_loss_bwd_f32(const float* a, const float* b, const float* grad_scale, float inv_count, const float* extra_a,
                         float* da, float* db, long n, dpmn_stream_t stream);
/* optimizer (super_resolution.py:272-278, base.py:221): ||g||^2 of a flat gradient bucket, then
 * clip_grad_norm_(max_norm) fused with torch.optim.Adam's update on flat (param, grad, exp_avg, exp_avg_sq) buffers */
int dpmn_sumsq_f32(const float* x, float* out, float* part_ws, long n, dpmn_stream_t stream);
int dpmn_adam_clip_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, const float* grad_normsq,
                       float max_norm, float lr, float beta1, float beta2, float eps, int step,
                       const float* step_dev /* optional device scalar overriding `step` (hipGraph replay) */, long n,
                       dpmn_stream_t stream);

/* ------------------------------------------------------------------ PGRM module (pgrm_forward.hip) */
typedef struct {
  const float *norm1_q_w, *norm1_q_b, *norm1_kv_w, *norm1_kv_b;
  const float *q_w, *q_b, *kv_w, *kv_b;
  const float* bias_table[4];
  const float *sk_proj_w, *sk_proj_b, *sk_fc1_w, *sk_fc1_b, *sk_fc2_w, *sk_fc2_b, *sk_head_w, *sk_head_b;
  const float *norm2_w, *norm2_b;
  const float *fc1_w, *fc1_b, *dw_w, *dw_b, *pw_w, *pw_b, *fc2_w, *fc2_b;
} dpmn_pgrm_block;

typedef struct {
  int img_h, img_w, patch, dim, n_groups, heads_per_group, mlp_hidden, hidden_size, n_weight_list;
  int window[4];
  const float *prior_fusion_w, *prior_fusion_b; /* NULL when mode=True (mask prior, 3 channels) */
  const float *pe_w, *pe_b, *pe_norm_w, *pe_norm_b;
  dpmn_pgrm_block blocks[2];
  const float *tail0_w, *tail0_b, *tail1_w, *tail1_b;
  const float* weight_list[16];   /* weight_list_0 .. weight_list_iter (iter <= 11 in the 6+6 stress stack) */
  int reuse_folded;               /* != 0: the workspace (same pointer, same B) still holds the folded attention weights of a previous
                                   * dpmn_pgrm_forward_f32 call with these weights -- skip the two fold kernels (frozen weights) */
} dpmn_pgrm_weights;

size_t dpmn_pgrm_workspace_bytes(const dpmn_pgrm_weights* w, int B);
/* PGRM.forward(x_q, x_kv, residual_list) (pgrm.py:546-565), eval semantics (dropout / DropPath identity).
 * x_q (B, 2|3, H, W), x_kv (B,3,H,W), residuals: HOST array of n_residuals device pointers (B,hid,H,W),
 * out (B, hidden_size, H, W). */
int dpmn_pgrm_forward_f32(const dpmn_pgrm_weights* w, const float* x_q, int x_q_channels, const float* x_kv,
                          const float* const* residuals, int n_residuals, float* out, void* workspace,
                          size_t workspace_bytes, int B, dpmn_stream_t stream);

/* PGRM.forward in TRAINING mode as one call (pgrm.py:546-565 with pos_drop / attn_drop / Mlp.drop / DropPath active, and every
 * activation the hand-written backward of dpmn_amd/train/pgrm_train.py reads written out).  Replaces the ~22 per-op calls of
 * train/pgrm_train.py::forward (host issue time 0.39 -> 0.29 ms per module at B = 48; the GPU time is unchanged).  The caller owns
 * every buffer:
 *   saved.tq, saved.tkv0                       (B L, C)   patch-embedded (+pos_drop) token streams
 *   per block: cat, feats, x1, n2, tkv_out     (B L, C);  ypre, gpre, g, z (B L, Ch);  V (B L, C / groups);  avec (B, groups, C / groups);
 *              partial (B ceil(L / 32), C);  fold: dpmn_ln_qkv_window_attn_workspace_bytes() bytes (folded LayerNorm + q / kv weights,
 *              reused by dpmn_ln_qkv_window_attn_bwd_f32)
 *   saved.c0, saved.c1                         (B, H, W, hidden_size patch^2) NHWC outputs of conv_before_upsample[0] / [1]
 * tail0_packed / tail1_packed: conv_before_upsample weights in the packed layout of dpmn_conv2d_nhwc_f32 (dpmn_conv_pack_f32).
 * drop == NULL: all rates zero.  seeds: [0] pos_drop(x_q) [1] pos_drop(x_kv); block b at 2 + 5 b: attn_drop, DropPath(attention),
 * Mlp.drop after act_1, Mlp.drop after fc2, DropPath(mlp) -- the masks the backward regenerates.
 * dpmn_pgrm_forward_train_supported: 1 when the geometry runs on the fused training kernels (dim 96 / head dim 16 / windows
 * <= 8 / hidden_size 3 / patch 2); otherwise callers keep the per-op sequence. */
typedef struct {
  float *cat, *fold, *feats, *partial, *avec, *x1, *ypre, *V, *n2, *gpre, *g, *z, *tkv_out;
} dpmn_pgrm_saved_block;
typedef struct {
  float *tq, *tkv0;
  dpmn_pgrm_saved_block blk[2];
  float *c0, *c1;
} dpmn_pgrm_saved;
typedef struct {
  float p, pa, dp[2];               /* drop_rate, attn_drop_rate, DropPath rate of block 0 / 1 */
  unsigned long long seeds[12];
} dpmn_pgrm_drop;
int dpmn_pgrm_forward_train_supported(const dpmn_pgrm_weights* w, int B);

/* ------------------------------------------------------------------ native CMM forward (cmm_forward.hip) */
/* ComplementationModulationModule.forward(x1, x2) in eval mode (cmm.py:120-161) as ONE call: 2 layout kernels, 10 grouped
 * encoder convs (the twin branches of cmm.py:86-99 share a launch), the channel gate, 5 phase-fused transposed convs and
 * 5 three-segment decoder convs.  All weights are the PACKED forms of dpmn_amd/model/packing.py with the eval BatchNorm folded:
 * encoder entries are (2, Cout, Kp) / (2, Cout) stacks of the two branches in the order en_1, en_2.a, en_2.b, ..., en_5.b, en_6;
 * transposed 4x4 convs are (4, Cout, Kp) phase-major packs; dea / deb entries are de_5 .. de_2. */
typedef struct {
  int c_img, cnum, img_h, img_w;
  const float *en_w[10], *en_b[10];
  const float *fc1_w, *fc1_b, *fc2_w, *fc2_b;     /* nn.Linear layouts of fc_1 / fc_2 (cmm.py:97-98) */
  const float *de6_w, *de6_b;
  const float *dea_w[4], *dea_b[4];
  const float *deb_w[4], *deb_b[4];
  const float *de1_w, *de1_b;
} dpmn_cmm_weights;
typedef struct {            /* the per-stream conv scratch of dpmn_conv_desc (split-K partial sums, stream-K arrival counters) */
  float* splitk_ws;
  size_t splitk_ws_bytes;
  unsigned* arrive_cnt;
  int arrive_cnt_len;
} dpmn_cmm_scratch;
/* The Swin-block part of one PGRM backward as ONE call (csrc/pgrm_backward.hip): the two-block loop of
 * dpmn_amd/train/pgrm_train.py::backward -- SwinTransformerBlock.forward reversed (pgrm.py:315-331) -- on the activations
 * dpmn_pgrm_forward_train_f32 saved.  grads[2]: the gradient sinks, laid out as dpmn_pgrm_block (every pointer is WRITTEN: += of the
 * parameter gradient); wt[2]: the transposed (in, out) copies of the seven Linear / pointwise weights the data gradients multiply by;
 * table_numel[g]: elements of relative_position_bias_table_g; dtkv: in dL/d(tokens behind block 1), out dL/d(tokens in front of
 * block 0); dtq: zero-filled, out dL/d(query tokens); dcat_zero[2]: two (B L, C) scratch buffers (written, need no fill); zero_bias:
 * mlp_hidden zeros.
 * scratch >= dpmn_pgrm_blocks_backward_scratch_bytes and must stay untouched until the caller's dpmn_reduce_defer_flush (it holds
 * partial rows of queued ordered reductions); arena / arena_used: the slice allocator of those reductions (*arena_used advances; a
 * request that does not fit flushes the queue and starts the arena over).  Requires dpmn_pgrm_forward_train_supported(w, B). */
typedef struct {
  const float *fc2_t, *pw_t, *fc1_t, *head_t, *proj_t, *q_t, *kv_t;
} dpmn_pgrm_block_t;
size_t dpmn_pgrm_blocks_backward_scratch_bytes(const dpmn_pgrm_weights* w, int B, const int* table_numel);
int dpmn_pgrm_blocks_backward_f32(const dpmn_pgrm_weights* w, const dpmn_pgrm_block* grads, const dpmn_pgrm_block_t* wt,
                                  const dpmn_pgrm_saved* sv, const dpmn_pgrm_drop* drop, const int* table_numel, float* dtkv, float* dtq,
                                  float* const* dcat_zero, const float* zero_bias, void* scratch, size_t scratch_bytes, void* arena,
                                  size_t arena_bytes, size_t* arena_used, int B, dpmn_stream_t stream);
/* The same with the LEAVES of the backward graph -- the Linear / pointwise-conv weight gradients and the ordered row reductions of the
 * gate and bias-table gradients (autograd of pgrm.py:315-331: nothing in the call consumes them) -- issued on `leaf_stream` behind events
 * recorded on `stream`; `stream` waits for `leaf_stream` before a buffer a leaf reads is overwritten and at the end of the call, so the
 * caller's view is unchanged: everything is complete in `stream` order on return.  leaf_stream == NULL or == stream: one stream.
 * Same kernels, same arguments, same reduction order: bitwise the gradients of dpmn_pgrm_blocks_backward_f32. */
int dpmn_pgrm_blocks_backward_leaf_f32(const dpmn_pgrm_weights* w, const dpmn_pgrm_block* grads, const dpmn_pgrm_block_t* wt,
                                       const dpmn_pgrm_saved* sv, const dpmn_pgrm_drop* drop, const int* table_numel, float* dtkv, float* dtq,
                                       float* const* dcat_zero, const float* zero_bias, void* scratch, size_t scratch_bytes, void* arena,
                                       size_t arena_bytes, size_t* arena_used, int B, dpmn_stream_t stream, dpmn_stream_t leaf_stream);
/* see dpmn_pgrm_saved above; scratch: the per-stream conv scratch (may be NULL: no split-K) */
int dpmn_pgrm_forward_train_f32(const dpmn_pgrm_weights* w, const float* x_q, int x_q_channels, const float* x_kv,
                                const float* const* residuals, int n_residuals, const float* tail0_packed, const float* tail1_packed,
                                const dpmn_pgrm_drop* drop, const dpmn_pgrm_saved* saved, const dpmn_cmm_scratch* scratch, float* out,
                                int B, dpmn_stream_t stream);
size_t dpmn_cmm_workspace_bytes(const dpmn_cmm_weights* w, int B);
/* x1, x2 (B, c_img, H, W) NCHW; out (B, c_img, H, W) NCHW; workspace >= dpmn_cmm_workspace_bytes (activations; contents are
 * scratch).  H, W multiples of 32. */
int dpmn_cmm_forward_f32(const dpmn_cmm_weights* w, const float* x1, const float* x2, float* out, void* workspace,
                         size_t workspace_bytes, const dpmn_cmm_scratch* scratch, int B, dpmn_stream_t stream);

/* ------------------------------------------------------------------ native PSN trunk (psn_forward.hip) */
/* The SRBs and the tail of TSRN.forward (tsrn.py:58-74) / TSRN_TL_TRANS.forward (tatt.py:645-691) in eval mode as ONE call:
 * per SRB conv+bn+mish, conv+bn, [cat with tp] -> GRU input projection (1x1 conv folded in) -> BiGRU along H (+ x) -> projection ->
 * BiGRU along W; then conv+bn + block1, conv + PixelShuffle + mish, 9x9 conv + tanh.  Weights are the packed forms of
 * dpmn_amd/model/packing.py / tsrn.py::_pack_gru_block (eval BatchNorm folded). */
typedef struct {
  const float *c1_w, *c1_b, *c2_w, *c2_b;              /* packed 3x3 convs */
  const float *g1_w, *g1_b, *g1_whh, *g1_bhh;          /* gru1: (6 hidden, Cin [+ tp channels]) input projection, (2, 3 hidden, hidden) recurrent */
  const float *g2_w, *g2_b, *g2_whh, *g2_bhh;
} dpmn_psn_srb;
typedef struct {
  int in_planes, ch, hidden, srb_nums;
  dpmn_psn_srb srb[8];
  const float *b7_w, *b7_b, *up_w, *up_b, *last_w, *last_b;
} dpmn_psn_weights;
size_t dpmn_psn_trunk_workspace_bytes(const dpmn_psn_weights* w, int B, int H, int W);
/* b1 (B,H,W,ch) NHWC = block1's output (9x9 conv + PReLU, run by the caller); tp (B,H,W,tp_channels) NHWC = TATT's text-prior map
 * or NULL (TSRN); out (B, in_planes, 2H, 2W) NCHW. */
int dpmn_psn_trunk_f32(const dpmn_psn_weights* w, const float* b1, const float* tp, int tp_channels, float* out, void* workspace,
                       size_t workspace_bytes, const dpmn_cmm_scratch* scratch, int B, int H, int W, dpmn_stream_t stream);

/* TPInterpreter.forward of TATT (tatt.py:196-237; InfoTransformer, transformer_v2.py) in eval mode as ONE call: fc_in + PReLU, one
 * encoder layer over the S text slots, n_dec decoder layers (cross attention of the feature map + query embedding against the
 * encoded slots, FFN), mean of the finally-normalised layer outputs.  nn.Linear / nn.LayerNorm / in_proj slices in their own layouts. */
typedef struct {
  const float *wq, *bq, *wk, *bk, *wv, *bv;       /* multihead_attn.in_proj rows [0,E) / [E,2E) / [2E,3E) */
  const float *out_w, *out_b, *norm2_w, *norm2_b, *lin1_w, *lin1_b, *lin2_w, *lin2_b, *norm3_w, *norm3_b;
} dpmn_tatt_dec_layer;
typedef struct {
  int n_dec, nhead;
  const float *fc_in_w, *fc_in_b;
  float fc_in_slope;                               /* nn.PReLU() single slope */
  const float* enc[12];                            /* encoder layer: in_proj w/b, out_proj w/b, linear1 w/b, linear2 w/b, norm1 w/b, norm2 w/b */
  dpmn_tatt_dec_layer dec[4];
  const float *dec_norm_w, *dec_norm_b;            /* decoder.norm */
} dpmn_tatt_interp_weights;
size_t dpmn_tatt_interpreter_workspace_bytes(int B, int L, int S);
/* x (B*S, t_emb) text-prior rows, b1 (B*L, 64) block1's NHWC output, qe (B*L, 64) query embedding, pos (S, 64);
 * tp (B*L, 64) out, pw (B, L, S) attention weights of the last layer or NULL. */
int dpmn_tatt_interpreter_f32(const dpmn_tatt_interp_weights* w, const float* x, int t_emb, const float* b1, const float* qe,
                              const float* pos, float* tp, float* pw, void* workspace, size_t workspace_bytes, int B, int L, int S,
                              dpmn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DPMN_HIP_H */
