// Native driver for the Swin-block part of one PGRM backward: the ~64 launches of the two-block loop of
// train/pgrm_train.py::backward (SwinTransformerBlock.forward reversed, pgrm.py:315-331: Mlp.fc2 / pointwise conv / depthwise conv /
// fc1 / LayerNorm2, SKConv select / gate / projection, the fused LayerNorm + q / kv + window-attention backward, the q / kv Linear
// weight gradients and the two LayerNorm1 backwards) issued from one call.  Same kernels, same order, same arguments as the per-op
// sequence: every gradient is bitwise equal (tests/test_gpu_train.py).  The tail convs in front of the loop and the patch embedding
// behind it stay with the caller (their weight gradients go through the conv weight-gradient plumbing of train/pgrm_train.py).
//
// Workspaces, all owned by the caller:
//   scratch   activation gradients of one block (reused by the next block in stream order) + the per-block partial-row buffers that
//             must survive until the caller's ordered-reduction flush (dpmn_reduce_defer_flush): dpmn_pgrm_blocks_backward_scratch_bytes
//   arena     the slice allocator of the deferred reductions (train/pgrm_train.py _ws): *arena_used is advanced; when a request does
//             not fit, the queued reductions are flushed (dpmn_reduce_defer_flush(0)) and the arena starts over, as the host code does
#include <cstdlib>
#include "common.h"
#include <math.h>

namespace {
struct Carve {
  char* base;
  size_t off;
  float* take(size_t n) {
    float* p = base ? reinterpret_cast<float*>(base + off) : nullptr;
    off += ((n * sizeof(float) + 255) / 256) * 256;
    return p;
  }
};

struct Scratch {
  float *dbr, *dz, *dg, *dypre, *dn2, *dat, *dV, *dS, *dA, *dfeats, *dcat2, *dq, *dkv, *nrm, *nrm2, *dnrm;
  float *wp2[2], *wp1[2], *tparts[2][4];
  size_t total;
};

Scratch carve_scratch(const dpmn_pgrm_weights* w, int B, const int* table_numel, char* base) {
  const size_t L = (size_t)(w->img_h / w->patch) * (w->img_w / w->patch), C = w->dim, Ch = w->mlp_hidden, G = w->n_groups;
  const size_t M = (size_t)B * L, parts = (L + 31) / 32, dmid = C / G / 2;
  const int H = w->img_h / w->patch, Wd = w->img_w / w->patch;
  const size_t rows = (size_t)dpmn_ln_qkv_window_attn_bwd_part_rows(B, H, Wd);
  Carve c{base, 0};
  Scratch s;
  s.dbr = c.take(M * C); s.dz = c.take(M * Ch); s.dg = c.take(M * Ch); s.dypre = c.take(M * Ch); s.dn2 = c.take(M * C);
  s.dat = c.take(M * C); s.dV = c.take(M * (C / G)); s.dS = c.take((size_t)B * C); s.dA = c.take(parts * B * C);
  s.dfeats = c.take(M * C); s.dcat2 = c.take(M * C); s.dq = c.take(M * C); s.dkv = c.take(M * 2 * C);
  s.nrm = c.take(M * C); s.nrm2 = c.take(M * C); s.dnrm = c.take(M * C);      // (nrm2: the kv LayerNorm output while the q weight gradient, a leaf, may still read nrm)
  for (int b = 0; b < 2; ++b) {
    s.wp2[b] = c.take((size_t)B * (C * dmid + C));
    s.wp1[b] = c.take((size_t)B * (dmid * C + dmid));
    for (size_t g = 0; g < 4; ++g) s.tparts[b][g] = g < G ? c.take(rows * (size_t)table_numel[g]) : nullptr;
  }
  s.total = c.off;
  return s;
}
// Leaf stream (dpmn_pgrm_blocks_backward_leaf_f32): the weight gradients of a Swin block -- six dY^T X products, the pointwise-conv weight
// gradient with its bias row sums, the ordered row reductions of the gate and bias-table gradients -- are LEAVES of the backward graph:
// nothing in the call consumes them.  On the main stream they sit between the data-gradient kernels the next step waits for (3.3 ms of a
// 27 ms step, measured by removal in round 5).  With a leaf stream every such launch goes there behind an event recorded on the main
// stream (its inputs are complete), and the main stream waits for the leaf stream only where a buffer a leaf reads is about to be
// overwritten (the in-place token gradient before the two LayerNorm backwards that accumulate into it, the scratch buffers at the
// block boundary) and at the end of the call (the caller's ordered-reduction flush runs on the main stream).
// Events come from a process-wide ring created on first use (disable-timing; a wait captures the record that precedes it, so reuse
// after a full turn of the ring is harmless).
struct EventRing {
  static constexpr int N = 64;
  hipEvent_t ev[N];
  int next = 0;
  bool ok = false;
  hipEvent_t get() {
    if (!ok) {
      for (int i = 0; i < N; ++i)
        if (hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) != hipSuccess) return nullptr;
      ok = true;
    }
    hipEvent_t e = ev[next];
    next = (next + 1) % N;
    return e;
  }
};
thread_local EventRing g_ring;
}  // namespace

extern "C" {

size_t dpmn_pgrm_blocks_backward_scratch_bytes(const dpmn_pgrm_weights* w, int B, const int* table_numel) {
  if (!w || B <= 0 || !table_numel) return 0;
  return carve_scratch(w, B, table_numel, nullptr).total;
}

int dpmn_pgrm_blocks_backward_f32(const dpmn_pgrm_weights* w, const dpmn_pgrm_block* grads, const dpmn_pgrm_block_t* wt,
                                  const dpmn_pgrm_saved* sv, const dpmn_pgrm_drop* drop, const int* table_numel, float* dtkv, float* dtq,
                                  float* const* dcat_zero, const float* zero_bias, void* scratch, size_t scratch_bytes, void* arena,
                                  size_t arena_bytes, size_t* arena_used, int B, dpmn_stream_t stream) {
  return dpmn_pgrm_blocks_backward_leaf_f32(w, grads, wt, sv, drop, table_numel, dtkv, dtq, dcat_zero, zero_bias, scratch, scratch_bytes, arena,
                                            arena_bytes, arena_used, B, stream, nullptr);
}

int dpmn_pgrm_blocks_backward_leaf_f32(const dpmn_pgrm_weights* w, const dpmn_pgrm_block* grads, const dpmn_pgrm_block_t* wt,
                                       const dpmn_pgrm_saved* sv, const dpmn_pgrm_drop* drop, const int* table_numel, float* dtkv, float* dtq,
                                       float* const* dcat_zero, const float* zero_bias, void* scratch, size_t scratch_bytes, void* arena,
                                       size_t arena_bytes, size_t* arena_used, int B, dpmn_stream_t stream, dpmn_stream_t leaf_stream) {
  DPMN_REQUIRE(w && grads && wt && sv && table_numel && dtkv && dtq && dcat_zero && dcat_zero[0] && dcat_zero[1] && zero_bias && scratch &&
               arena && arena_used, "pgrm_blocks_backward: null pointer");
  DPMN_REQUIRE(dpmn_pgrm_forward_train_supported(w, B), "pgrm_blocks_backward: geometry outside the fused training kernels (use the per-op path)");
  const int H = w->img_h / w->patch, Wd = w->img_w / w->patch, L = H * Wd, C = w->dim, Ch = w->mlp_hidden, G = w->n_groups;
  const int r = (int)lrintf(sqrtf((float)L));
  const int M = B * L, parts = (L + 31) / 32, dmid = C / G / 2, cg = C / G;
  Scratch s = carve_scratch(w, B, table_numel, static_cast<char*>(scratch));
  if (s.total > scratch_bytes) return dpmn_set_error(DPMN_ERR_WORKSPACE, "pgrm_blocks_backward: scratch too small");
  const float pd = drop ? drop->p : 0.f, pa = drop ? drop->pa : 0.f;
  static const unsigned long long zero_seeds[12] = {0};
  const unsigned long long* sd = drop ? drop->seeds : zero_seeds;
  const int rows = dpmn_ln_qkv_window_attn_bwd_part_rows(B, H, Wd);
  int rc;
#define RUN(call) do { rc = (call); if (rc != DPMN_OK) return rc; } while (0)
  // leaf launches: `leaf` = the leaf stream behind the main stream's present position (fork), or the main stream itself
  const bool forked = leaf_stream != nullptr && leaf_stream != stream;
  bool leaf_dirty = false;           // the leaf stream holds work the main stream has not waited for
  bool fork_failed = false;          // (the main stream may be the null stream: a null handle is not an error)
  auto fork = [&]() -> dpmn_stream_t {
    if (!forked) return stream;
    hipEvent_t e = g_ring.get();
    if (!e || hipEventRecord(e, as_stream(stream)) != hipSuccess || hipStreamWaitEvent(as_stream(leaf_stream), e, 0) != hipSuccess) {
      fork_failed = true;
      return stream;
    }
    leaf_dirty = true;
    return leaf_stream;
  };
  auto join = [&]() -> int {
    if (!forked || !leaf_dirty) return DPMN_OK;
    hipEvent_t e = g_ring.get();
    if (!e || hipEventRecord(e, as_stream(leaf_stream)) != hipSuccess || hipStreamWaitEvent(as_stream(stream), e, 0) != hipSuccess)
      return dpmn_set_error(DPMN_ERR_LAUNCH, "pgrm_blocks_backward: leaf-stream join failed");
    leaf_dirty = false;
    return DPMN_OK;
  };
#define LEAF(var) dpmn_stream_t var = fork(); if (fork_failed) return dpmn_set_error(DPMN_ERR_LAUNCH, "pgrm_blocks_backward: leaf-stream fork failed")
  // a slice of the deferred-reduction arena (train/pgrm_train.py _ws): flush and start over when the request does not fit
  float* ws_ptr = nullptr;
  size_t ws_n = 0;
  // Linear weight gradients collected while their operands stay live, launched as ONE grouped product (dpmn_gemm_tn_group_f32): a
  // product alone puts one short block on each CU; their arena slices are taken at collection time, so the group goes out before the
  // arena wraps
  static const int tn_group = getenv("DPMN_BWD_TN_GROUP") ? atoi(getenv("DPMN_BWD_TN_GROUP")) : 1;
  dpmn_tn_item tn_items[8];
  int tn_n = 0;
  auto tn_flush = [&]() -> int {
    if (tn_n == 0) return DPMN_OK;
    dpmn_stream_t ls = fork();
    if (fork_failed) return dpmn_set_error(DPMN_ERR_LAUNCH, "pgrm_blocks_backward: leaf-stream fork failed");
    const int e = dpmn_gemm_tn_group_f32(tn_items, tn_n, ls);
    tn_n = 0;
    return e;
  };
  auto take = [&](size_t nbytes) -> int {
    nbytes = (nbytes + 255) / 256 * 256;
    if (nbytes > arena_bytes) return dpmn_set_error(DPMN_ERR_WORKSPACE, "pgrm_blocks_backward: reduction arena smaller than one request");
    if (*arena_used + nbytes > arena_bytes) {
      const int f = tn_flush();    // collected products own slices of the arena that is about to be handed out again
      if (f != DPMN_OK) return f;
      const int j = join();        // the queued partial rows were written on the leaf stream
      if (j != DPMN_OK) return j;
      const int e = dpmn_reduce_defer_flush(0, stream);
      if (e != DPMN_OK) return e;
      *arena_used = 0;
    }
    ws_ptr = reinterpret_cast<float*>(static_cast<char*>(arena) + *arena_used);
    ws_n = nbytes;
    *arena_used += nbytes;
    return DPMN_OK;
  };
  // dw (N, K) += dy^T x, db += colsum(dy) as split partials in the arena; then dx = dy . W through the transposed weight
#define LINEAR_BWD(dy, x, w_t, dw, db, N_, K_, dx)                                                                   \
  do {                                                                                                             \
    RUN(take(dpmn_gemm_tn_partial_bytes(M, (N_), (K_))));                                                            \
    if (tn_group) tn_items[tn_n++] = dpmn_tn_item{(dy), (x), (dw), (db), M, (N_), (K_), ws_ptr, ws_n};                \
    else { LEAF(ls_); RUN(dpmn_gemm_tn_f32((dy), (x), (dw), (db), M, (N_), (K_), ws_ptr, ws_n, ls_)); }              \
    RUN(dpmn_linear_f32((dy), (w_t), nullptr, nullptr, nullptr, (dx), M, (K_), (N_), DPMN_ACT_NONE, 0.f, stream));     \
  } while (0)
#define LN_BWD(x, dy, gamma, dx, dgamma, dbeta)                                                                      \
  do {                                                                                                             \
    RUN(take((size_t)512 * 2 * C * 4));                                                                             \
    RUN(dpmn_layernorm_bwd_det_f32((x), (dy), (gamma), 1e-5f, (dx), 1, (dgamma), (dbeta), M, C, ws_ptr, ws_n, stream)); \
  } while (0)
  // the same with the masked copy of the finished gradient as a second output (instead of a dropout launch re-reading dx)
#define LN_BWD_DROP(x, dy, gamma, dx, dgamma, dbeta, out2, pe, se, pr, sr)                                            \
  do {                                                                                                             \
    RUN(take((size_t)512 * 2 * C * 4));                                                                             \
    RUN(dpmn_layernorm_bwd_det_drop_f32((x), (dy), (gamma), 1e-5f, (dx), 1, (dgamma), (dbeta), M, C, ws_ptr, ws_n, (out2), (pe), (se), (pr), \
                                        (sr), (long)L * C, stream));                                                 \
  } while (0)
  static const int fuse_masks = getenv("DPMN_BWD_FUSE_MASKS") ? atoi(getenv("DPMN_BWD_FUSE_MASKS")) : 1;
  bool dbr_ready = false;      // block 0's masked fc2 gradient was written by block 1's last LayerNorm backward
  float* dx2 = dtkv;
  for (int bi = 1; bi >= 0; --bi) {
    const dpmn_pgrm_block& p = w->blocks[bi];
    const dpmn_pgrm_block& g = grads[bi];              // gradient sinks, same fields (written through)
    const dpmn_pgrm_block_t& t = wt[bi];
    const dpmn_pgrm_saved_block& b = sv->blk[bi];
    const float* tkv_in = bi == 0 ? sv->tkv0 : sv->blk[0].tkv_out;
    const float dpb = drop ? drop->dp[bi] : 0.f;
    const unsigned long long* sb = sd + 2 + 5 * bi;
    auto sink = [](const float* q) { return const_cast<float*>(q); };
    // fc2 (+ residual x1); with dropout the branch gradient is dx2 under the forward's masks
    const float* dbr = dx2;
    if (pd > 0.f || dpb > 0.f) {
      if (!dbr_ready) RUN(dpmn_dropout_f32(dx2, nullptr, s.dbr, (long)M * C, (long)L * C, pd, sb[3], dpb, sb[4], stream));
      dbr = s.dbr;
    }
    LINEAR_BWD(dbr, b.z, t.fc2_t, sink(g.fc2_w), sink(g.fc2_b), C, Ch, s.dz);
    // pointwise conv on the raw (B, Ch, L) views
    RUN(dpmn_pointwise_f32(s.dz, t.pw_t, zero_bias, s.dg, B, Ch, L, stream));
    {
      LEAF(ls_);
      RUN(take(dpmn_pointwise_wgrad_det_bytes(Ch, L)));
      RUN(dpmn_pointwise_wgrad_det_f32(s.dz, b.g, sink(g.pw_w), B, Ch, L, ws_ptr, ws_n, ls_));
      RUN(take((size_t)B * Ch * 4));
      RUN(dpmn_rowsum_mod_det_f32(s.dz, sink(g.pw_b), (long)B * Ch, L, Ch, ws_ptr, ws_n, ls_));
    }
    // depthwise conv: GELU'(gpre) on the way in, GELU (+ the dropout mask) on the forward input, mask and GELU'(ypre) on the way out
    RUN(take(dpmn_dwconv3x3_bwd_det_bytes(B, Ch, r)));
    RUN(dpmn_dwconv3x3_bwd_fused_det_f32(b.ypre, s.dg, b.gpre, p.dw_w, s.dypre, sink(g.dw_w), sink(g.dw_b), 1, 1, pd, sb[2], B, Ch, r, ws_ptr,
                                         ws_n, stream));
    LINEAR_BWD(s.dypre, b.n2, t.fc1_t, sink(g.fc1_w), sink(g.fc1_b), Ch, C, s.dn2);
    float* dx1 = dx2;       // in place: every reader of dx2 is already queued on this stream ...
    RUN(tn_flush());        // (group A = fc2 + fc1: the fc2 weight gradient reads dx2 when no mask was applied)
    RUN(join());            // ... or on the leaf stream
    // x1 = tkv_in + DropPath(feats + V Wh^T + bh): the DropPath-masked gradient rides out of the LayerNorm2 backward
    const float* dat = dx1;
    if (dpb > 0.f && fuse_masks) {
      LN_BWD_DROP(b.x1, s.dn2, p.norm2_w, dx1, sink(g.norm2_w), sink(g.norm2_b), s.dat, 0.f, 0ull, dpb, sb[1]);
      dat = s.dat;
    } else {
      LN_BWD(b.x1, s.dn2, p.norm2_w, dx1, sink(g.norm2_w), sink(g.norm2_b));
      if (dpb > 0.f) {
        RUN(dpmn_dropout_f32(dx1, nullptr, s.dat, (long)M * C, (long)L * C, 0.f, 0ull, dpb, sb[1], stream));
        dat = s.dat;
      }
    }
    LINEAR_BWD(dat, b.V, t.head_t, sink(g.sk_head_w), sink(g.sk_head_b), C, cg, s.dV);
    float* dcat = dcat_zero[bi];
    RUN(dpmn_sk_select_bwd_det_set_f32(b.cat, b.avec, s.dV, dcat, s.dA, B, L, C, G, stream));      // dcat written: the buffer needs no fill
    RUN(dpmn_sk_gate_bwd_det_f32(b.partial, parts, L, p.sk_fc1_w, p.sk_fc1_b, p.sk_fc2_w, b.avec, s.dA, parts, s.dS, s.wp2[bi], s.wp1[bi], B, C, G,
                                 dmid, stream));
    {
      LEAF(ls_);
      RUN(dpmn_rows_reduce_f32(s.wp2[bi], sink(g.sk_fc2_w), sink(g.sk_fc2_b), C * dmid, C, B, ls_));
      RUN(dpmn_rows_reduce_f32(s.wp1[bi], sink(g.sk_fc1_w), sink(g.sk_fc1_b), dmid * C, dmid, B, ls_));
    }
    RUN(dpmn_sk_feats_grad_f32(dat, b.feats, s.dS, s.dfeats, M, L, C, stream));
    RUN(take(dpmn_gemm_tn_partial_bytes(M, C, C)));
    if (tn_group) tn_items[tn_n++] = dpmn_tn_item{s.dfeats, b.cat, sink(g.sk_proj_w), sink(g.sk_proj_b), M, C, C, ws_ptr, ws_n};
    else { LEAF(ls_); RUN(dpmn_gemm_tn_f32(s.dfeats, b.cat, sink(g.sk_proj_w), sink(g.sk_proj_b), M, C, C, ws_ptr, ws_n, ls_)); }
    RUN(dpmn_linear_f32(s.dfeats, t.proj_t, nullptr, dcat, nullptr, s.dcat2, M, C, C, DPMN_ACT_NONE, 0.f, stream));
    // window attention: q / k / v recomputed, all window sizes on MFMA, bias-table gradients as per-block partial rows
    int win[4], shift[4];
    for (int k = 0; k < G; ++k) {
      win[k] = w->window[k];
      shift[k] = bi == 0 ? 0 : w->window[k] / 2;
      if ((H < Wd ? H : Wd) <= win[k]) { win[k] = H < Wd ? H : Wd; shift[k] = 0; }
    }
    RUN(dpmn_ln_qkv_window_attn_bwd_f32(sv->tq, tkv_in, p.norm1_q_w, p.norm1_q_b, p.norm1_kv_w, p.norm1_kv_b, 1e-5f, p.q_w, p.q_b, p.kv_w, p.kv_b,
                                        p.bias_table, win, shift, G, w->heads_per_group, s.dcat2, s.dq, s.dkv, s.tparts[bi], pa, sb[0], b.fold, 0,
                                        B, H, Wd, C, stream));
    {
      LEAF(ls_);
      for (int k = 0; k < G; ++k)
        RUN(dpmn_rows_reduce_f32(s.tparts[bi][k], sink(g.bias_table[k]), nullptr, table_numel[k], 0, rows, ls_));
    }
    RUN(dpmn_layernorm_f32(sv->tq, p.norm1_q_w, p.norm1_q_b, 1e-5f, s.nrm, M, C, stream));
    LINEAR_BWD(s.dq, s.nrm, t.q_t, sink(g.q_w), sink(g.q_b), C, C, s.dnrm);
    LN_BWD(sv->tq, s.dnrm, p.norm1_q_w, dtq, sink(g.norm1_q_w), sink(g.norm1_q_b));
    RUN(dpmn_layernorm_f32(tkv_in, p.norm1_kv_w, p.norm1_kv_b, 1e-5f, s.nrm2, M, C, stream));
    LINEAR_BWD(s.dkv, s.nrm2, t.kv_t, sink(g.kv_w), sink(g.kv_b), 2 * C, C, s.dnrm);
    RUN(tn_flush());        // group B = SKConv head + proj, q, kv
    RUN(join());            // block boundary: the last LayerNorm backward accumulates into the token gradient the SKConv head's weight
                            // gradient may still read, and the next block overwrites the scratch buffers this block's leaves read
    // block 1's last step finishes dL/d(tokens behind block 0) = block 0's dx2: its Dropout / DropPath-masked copy (block 0's masks)
    // comes out of the same kernel
    const float dpb0 = drop ? drop->dp[0] : 0.f;
    if (bi == 1 && fuse_masks && (pd > 0.f || dpb0 > 0.f)) {
      LN_BWD_DROP(tkv_in, s.dnrm, p.norm1_kv_w, dx1, sink(g.norm1_kv_w), sink(g.norm1_kv_b), s.dbr, pd, sd[2 + 3], dpb0, sd[2 + 4]);
      dbr_ready = true;
    } else {
      LN_BWD(tkv_in, s.dnrm, p.norm1_kv_w, dx1, sink(g.norm1_kv_w), sink(g.norm1_kv_b));
    }
    dx2 = dx1;
  }
#undef LN_BWD_DROP
#undef LN_BWD
#undef LINEAR_BWD
  RUN(join());
#undef LEAF
#undef RUN
  return DPMN_OK;
}

}  // extern "C"
