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
