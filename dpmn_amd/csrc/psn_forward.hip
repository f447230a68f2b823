// Native driver for the conv / BiGRU trunk of the frozen PSN backbones in eval mode: the five SRBs and the tail of
// TSRN.forward (tsrn.py:58-74, RecurrentResidualBlock 83-110, UpsampleBLock 113-125) and of TSRN_TL_TRANS.forward (tatt.py:645-691,
// RecurrentResidualBlockTL 885-910) from ONE C call -- 33 launches that the host mirror (dpmn_amd/model/tsrn.py, tatt.py) used
// to issue one ctypes call at a time.  Same kernels, order and arguments as the composed path: bitwise-equal results
// (tests/test_gpu_psn.py).  The head conv and TATT's text-prior interpreter (whose output tp feeds every SRB) stay on the host side.
#include "common.h"

namespace {
struct PsnWs {
  float *r1, *r2, *gi, *s, *f[2], *t, *u;
  size_t total;
};
PsnWs carve(const dpmn_psn_weights* w, int B, int H, int W, char* base) {
  size_t off = 0;
  auto take = [&](size_t n) {
    float* p = base ? reinterpret_cast<float*>(base + off) : nullptr;
    off += ((n * sizeof(float) + 255) / 256) * 256;
    return p;
  };
  const size_t px = (size_t)B * H * W, ch = w->ch;
  PsnWs s;
  s.r1 = take(px * ch); s.r2 = take(px * ch); s.gi = take(px * 6 * w->hidden); s.s = take(px * ch);
  s.f[0] = take(px * ch); s.f[1] = take(px * ch); s.t = take(px * ch); s.u = take(px * 4 * ch);
  s.total = off;
  return s;
}
dpmn_conv_desc conv_same(const float* in, int cin, int B, int H, int W, int k, const float* wp, const float* bias, int cout, float* out,
                         const dpmn_cmm_scratch* sc) {
  dpmn_conv_desc d{};
  d.in[0] = in; d.cseg[0] = cin;
  d.B = B; d.Hin = H; d.Win = W; d.KH = k; d.KW = k;
  d.stride = 1; d.dil_y = 1; d.dil_x = 1; d.pad_y = (k - 1) / 2; d.pad_x = (k - 1) / 2;
  d.Hp = H; d.Wp = W; d.Hout = H; d.Wout = W; d.ostep = 1;
  d.w = wp; d.bias = bias; d.Cout = cout; d.out = out;
  d.splitk_ws = sc->splitk_ws; d.splitk_ws_bytes = sc->splitk_ws_bytes; d.arrive_cnt = sc->arrive_cnt; d.arrive_cnt_len = sc->arrive_cnt_len;
  return d;
}
}  // namespace

extern "C" {

size_t dpmn_psn_trunk_workspace_bytes(const dpmn_psn_weights* w, int B, int H, int W) {
  if (!w || B <= 0 || H <= 0 || W <= 0) return 0;
  return carve(w, B, H, W, nullptr).total;
}

int dpmn_psn_trunk_f32(const dpmn_psn_weights* w, const float* b1, const float* tp, int tp_channels, float* out, void* workspace,
                       size_t workspace_bytes, const dpmn_cmm_scratch* sc, int B, int H, int W, dpmn_stream_t stream) {
  DPMN_REQUIRE(w && b1 && out && workspace && sc, "psn_trunk: null pointer");
  DPMN_REQUIRE(B >= 1 && w->srb_nums >= 1 && w->srb_nums <= 8 && w->ch == 2 * w->hidden, "psn_trunk: bad shape (1..8 SRBs, ch = 2 * hidden)");
  DPMN_REQUIRE((tp == nullptr) == (tp_channels == 0), "psn_trunk: tp and tp_channels go together");
  PsnWs s = carve(w, B, H, W, static_cast<char*>(workspace));
  if (s.total > workspace_bytes) return dpmn_set_error(DPMN_ERR_WORKSPACE, "psn_trunk: workspace too small");
  const int ch = w->ch, M = B * H * W, hid = w->hidden;
  int rc;
#define RUN(call) do { rc = (call); if (rc != DPMN_OK) return rc; } while (0)
  const float* f = b1;
  for (int i = 0; i < w->srb_nums; ++i) {
    const dpmn_psn_srb& p = w->srb[i];
    // conv1 + bn1 (folded) + mish, conv2 + bn2 (folded)  (tsrn.py:97-98 / tatt.py:899-900)
    dpmn_conv_desc d1 = conv_same(f, ch, B, H, W, 3, p.c1_w, p.c1_b, ch, s.r1, sc);
    d1.epi_act = DPMN_ACT_MISH;
    RUN(dpmn_conv2d_nhwc_f32(&d1, stream));
    dpmn_conv_desc d2 = conv_same(s.r1, ch, B, H, W, 3, p.c2_w, p.c2_b, ch, s.r2, sc);
    RUN(dpmn_conv2d_nhwc_f32(&d2, stream));
    // gru1 on the transposed map, + x (tsrn.py:99 / tatt.py:902-907): GruBlock.conv1 (1x1, on cat([residual, tp]) in TATT) is folded
    // into the GRU input projection -- one (pixels, Cin) x (Cin, 6 hidden) GEMM -- then the recurrence along H
    if (tp) RUN(dpmn_cat2_linear_f32(s.r2, ch, tp, tp_channels, p.g1_w, p.g1_b, s.gi, M, 6 * hid, DPMN_ACT_NONE, stream));
    else RUN(dpmn_linear_f32(s.r2, p.g1_w, p.g1_b, nullptr, nullptr, s.gi, M, 6 * hid, ch, DPMN_ACT_NONE, 0.f, stream));
    RUN(dpmn_bigru_f32(s.gi, p.g1_whh, p.g1_bhh, f, s.s, B * W, H, W, (long)H * W, 1, W, hid, stream));
    // gru2 along W (tsrn.py:100 / tatt.py:909)
    RUN(dpmn_linear_f32(s.s, p.g2_w, p.g2_b, nullptr, nullptr, s.gi, M, 6 * hid, ch, DPMN_ACT_NONE, 0.f, stream));
    float* fo = s.f[i & 1];
    RUN(dpmn_bigru_f32(s.gi, p.g2_whh, p.g2_bhh, nullptr, fo, B * H, W, 1, W, 0, 1, hid, stream));
    f = fo;
  }
  // block(srb+2): conv + bn, + block1 (tsrn.py:68-69); UpsampleBLock: conv -> PixelShuffle(2) -> mish; last 9x9 conv; tanh (tsrn.py:70-72)
  dpmn_conv_desc d7 = conv_same(f, ch, B, H, W, 3, w->b7_w, w->b7_b, ch, s.t, sc);
  d7.res = b1;
  RUN(dpmn_conv2d_nhwc_f32(&d7, stream));
  dpmn_conv_desc du = conv_same(s.t, ch, B, H, W, 3, w->up_w, w->up_b, 4 * ch, s.u, sc);
  du.epi_act = DPMN_ACT_MISH; du.pixel_shuffle = 1;
  RUN(dpmn_conv2d_nhwc_f32(&du, stream));
  dpmn_conv_desc dl = conv_same(s.u, ch, B, 2 * H, 2 * W, 9, w->last_w, w->last_b, w->in_planes, out, sc);
  dl.epi_act = DPMN_ACT_TANH; dl.out_nchw = 1;
  RUN(dpmn_conv2d_nhwc_f32(&dl, stream));
#undef RUN
  return DPMN_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------- TATT text-prior interpreter
// TPInterpreter.forward (tatt.py:196-237) + InfoTransformer (transformer_v2.py: one encoder layer over the 26 text slots, three
// decoder layers whose queries are the image feature map + the cached query embedding, mean of the normalised layer outputs):
// 2 + 3 x 9 launches from one C call.  x: the (B * S, t_emb) text-prior rows, qe: (B * L, 64) query embedding (quirk Q5: computed
// once per batch size by the host mirror), pos: (S, 64) sinusoid rows; out tp (B * L, 64), pw (B, L, S) or NULL.
namespace {
struct TpiWs {
  float *src, *mem, *q, *k, *v, *o, *t1, *out[2], *f;
  size_t total;
};
TpiWs carve_tpi(int B, int L, int S, int E, char* base) {
  size_t off = 0;
  auto take = [&](size_t n) {
    float* p = base ? reinterpret_cast<float*>(base + off) : nullptr;
    off += ((n * sizeof(float) + 255) / 256) * 256;
    return p;
  };
  TpiWs s;
  const size_t BL = (size_t)B * L, BS = (size_t)B * S;
  s.src = take(BS * E); s.mem = take(BS * E); s.q = take(BL * E); s.k = take(BS * E); s.v = take(BS * E);
  s.o = take(BL * E); s.t1 = take(BL * E); s.out[0] = take(BL * E); s.out[1] = take(BL * E); s.f = take(BL * E);
  s.total = off;
  return s;
}
}  // namespace

extern "C" {

size_t dpmn_tatt_interpreter_workspace_bytes(int B, int L, int S) {
  if (B <= 0 || L <= 0 || S <= 0) return 0;
  return carve_tpi(B, L, S, 64, nullptr).total;
}

int dpmn_tatt_interpreter_f32(const dpmn_tatt_interp_weights* w, const float* x, int t_emb, const float* b1, const float* qe,
                              const float* pos, float* tp, float* pw, void* workspace, size_t workspace_bytes, int B, int L, int S,
                              dpmn_stream_t stream) {
  DPMN_REQUIRE(w && x && b1 && qe && pos && tp && workspace, "tatt_interpreter: null pointer");
  DPMN_REQUIRE(w->n_dec >= 1 && w->n_dec <= 4 && w->nhead > 0, "tatt_interpreter: 1..4 decoder layers");
  const int E = 64;
  TpiWs s = carve_tpi(B, L, S, E, static_cast<char*>(workspace));
  if (s.total > workspace_bytes) return dpmn_set_error(DPMN_ERR_WORKSPACE, "tatt_interpreter: workspace too small");
  const int BL = B * L, BS = B * S;
  int rc;
#define RUN(call) do { rc = (call); if (rc != DPMN_OK) return rc; } while (0)
  // fc_in + PReLU (tatt.py:209-210), encoder layer with the sinusoid added to q / k (transformer_v2.py:453-470)
  RUN(dpmn_small_linear_f32(x, nullptr, 0, w->fc_in_w, w->fc_in_b, s.src, BS, E, t_emb, DPMN_ACT_PRELU, w->fc_in_slope, stream));
  RUN(dpmn_tatt_encoder_layer_f32(s.src, pos, w->enc, s.mem, B, S, E, w->nhead, stream));
  const float* out = b1;
  for (int li = 0; li < w->n_dec; ++li) {
    const dpmn_tatt_dec_layer& d = w->dec[li];
    const bool last = li == w->n_dec - 1;
    // cross attention: q = (tgt + query_embed) Wq, k = (memory + pos) Wk, v = memory Wv (transformer_v2.py:826-838)
    RUN(dpmn_add_linear_f32(out, qe, d.wq, d.bq, s.q, BL, E, E, DPMN_ACT_NONE, stream));
    RUN(dpmn_small_linear_f32(s.mem, pos, S, d.wk, d.bk, s.k, BS, E, E, DPMN_ACT_NONE, 0.f, stream));
    RUN(dpmn_small_linear_f32(s.mem, nullptr, 0, d.wv, d.bv, s.v, BS, E, E, DPMN_ACT_NONE, 0.f, stream));
    RUN(dpmn_cross_attn_f32(s.q, s.k, s.v, s.o, last ? pw : nullptr, B, L, S, E, w->nhead, stream));
    RUN(dpmn_linear_f32(s.o, d.out_w, d.out_b, out, nullptr, s.t1, BL, E, E, DPMN_ACT_NONE, 0.f, stream));
    float* o1 = s.out[0];
    RUN(dpmn_add_layernorm64_f32(s.t1, nullptr, d.norm2_w, d.norm2_b, o1, nullptr, nullptr, nullptr, 1.0f, 0, BL, stream));
    // FFN (transformer_v2.py:785) + norm3; the decoder's final norm of every layer output is averaged into tp (return_intermediate)
    RUN(dpmn_linear_f32(o1, d.lin1_w, d.lin1_b, nullptr, nullptr, s.f, BL, E, E, DPMN_ACT_RELU, 0.f, stream));
    RUN(dpmn_linear_f32(s.f, d.lin2_w, d.lin2_b, o1, nullptr, s.t1, BL, E, E, DPMN_ACT_NONE, 0.f, stream));
    float* o2 = s.out[1];
    RUN(dpmn_add_layernorm64_f32(s.t1, nullptr, d.norm3_w, d.norm3_b, o2, w->dec_norm_w, w->dec_norm_b, tp, 1.0f / w->n_dec, li > 0 ? 1 : 0,
                                 BL, stream));
    out = o2;       // (the next layer last reads it as out_proj's residual, long before its own norm3 rewrites the buffer)
  }
#undef RUN
  return DPMN_OK;
}

}  // extern "C"
