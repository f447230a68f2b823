// Native driver for one eval-mode CMM forward: every launch of ComplementationModulationModule.forward (cmm.py:120-161) issued
// from ONE C call on one stream -- the host mirror (dpmn_amd/model/cmm.py) only packs the weights (BatchNorm folded, quirk Q12)
// and supplies device memory.  Same kernels, same order and same arguments as the composed per-op path, so the results are
// bitwise equal to it (tests/test_gpu_cmm.py).  The twin encoder branches (cmm.py:86-99) run as grouped launches over the
// batch-concatenated inputs; decoder skip concats (cmm.py:150-158) are extra input segments, never materialised.
#include "common.h"

namespace {
struct Lvl { int H, W, C; };

struct CmmWs {
  float* xin;        // (2B, H, W, 4) NHWC inputs of the two branches, channel-padded
  float* enc[6];     // (2B, Hl, Wl, Cl) outputs of en_1 .. en_6 (first half = branch 1)
  float* tmp;        // largest intermediate of an Encode / DecodeBlock
  float* dec;        // running decoder tensor
  float* bott;       // (B, 1*4, 16c) channel-concatenated bottleneck, gated in place into bott + half
  float* hid;        // (B, 4c) gate hidden units
  size_t total;
};

void levels(const dpmn_cmm_weights* w, Lvl* l) {
  const int c = w->cnum;
  const int ch[6] = {c, 2 * c, 4 * c, 8 * c, 8 * c, 8 * c};
  for (int i = 0; i < 6; ++i) l[i] = Lvl{w->img_h >> i, w->img_w >> i, ch[i]};
}

CmmWs carve(const dpmn_cmm_weights* w, int B, char* base) {
  Lvl l[6];
  levels(w, l);
  size_t off = 0;
  auto take = [&](size_t n) {
    float* p = base ? reinterpret_cast<float*>(base + off) : nullptr;
    off += ((n * sizeof(float) + 255) / 256) * 256;
    return p;
  };
  CmmWs s;
  s.xin = take((size_t)2 * B * w->img_h * w->img_w * 4);
  size_t big = 0;
  for (int i = 0; i < 6; ++i) {
    const size_t n = (size_t)2 * B * l[i].H * l[i].W * l[i].C;
    s.enc[i] = take(n);
    if (n > big) big = n;
  }
  s.tmp = take(big);
  s.dec = take(big);
  s.bott = take((size_t)2 * B * l[5].H * l[5].W * 2 * l[5].C);
  s.hid = take((size_t)B * 4 * w->cnum);
  s.total = off;
  return s;
}

dpmn_conv_desc conv_desc(const float* in0, int c0, const float* in1, int c1, const float* in2, int c2, int B, int H, int W, int k,
                         int stride, int pad, int dil, int pro_act, const float* wp, const float* bias, int cout, float* out,
                         const dpmn_cmm_scratch* sc) {
  dpmn_conv_desc d{};
  d.in[0] = in0; d.cseg[0] = c0; d.in[1] = in1; d.cseg[1] = c1; d.in[2] = in2; d.cseg[2] = c2;
  d.B = B; d.Hin = H; d.Win = W; d.KH = k; d.KW = k;
  d.stride = stride; d.dil_y = dil; d.dil_x = dil; d.pad_y = pad; d.pad_x = pad;
  const int Ho = (H + 2 * pad - dil * (k - 1) - 1) / stride + 1, Wo = (W + 2 * pad - dil * (k - 1) - 1) / stride + 1;
  d.Hp = Ho; d.Wp = Wo; d.Hout = Ho; d.Wout = Wo; d.ostep = 1;
  d.pro_act = pro_act; d.w = wp; d.bias = bias; d.Cout = cout; d.out = out;
  d.splitk_ws = sc->splitk_ws; d.splitk_ws_bytes = sc->splitk_ws_bytes; d.arrive_cnt = sc->arrive_cnt; d.arrive_cnt_len = sc->arrive_cnt_len;
  return d;
}
}  // namespace

extern "C" {

size_t dpmn_cmm_workspace_bytes(const dpmn_cmm_weights* w, int B) {
  if (!w || B <= 0) return 0;
  return carve(w, B, nullptr).total;
}

int dpmn_cmm_forward_f32(const dpmn_cmm_weights* w, const float* x1, const float* x2, float* out, void* workspace,
                         size_t workspace_bytes, const dpmn_cmm_scratch* sc, int B, dpmn_stream_t stream) {
  DPMN_REQUIRE(w && x1 && x2 && out && workspace && sc, "cmm_forward: null pointer");
  DPMN_REQUIRE(B >= 1 && w->c_img >= 1 && w->c_img <= 4 && w->cnum >= 8 && w->cnum % 8 == 0, "cmm_forward: bad shape");
  DPMN_REQUIRE(w->img_h % 32 == 0 && w->img_w % 32 == 0, "cmm_forward: five stride-2 levels need H, W multiples of 32");
  CmmWs s = carve(w, B, static_cast<char*>(workspace));
  if (s.total > workspace_bytes) return dpmn_set_error(DPMN_ERR_WORKSPACE, "cmm_forward: workspace too small");
  Lvl l[6];
  levels(w, l);
  const int c = w->cnum, B2 = 2 * B;
  hipStream_t st = static_cast<hipStream_t>(stream);
  int rc;
#define RUN(call) do { rc = (call); if (rc != DPMN_OK) return rc; } while (0)
  // module boundary: NCHW images -> NHWC, channels zero-padded to 4, the two branches back to back (cmm.py:121-133 inputs)
  RUN(dpmn_nchw_to_nhwc_f32(x1, s.xin, B, w->c_img, w->img_h, w->img_w, 4, stream));
  RUN(dpmn_nchw_to_nhwc_f32(x2, s.xin + (size_t)B * w->img_h * w->img_w * 4, B, w->c_img, w->img_h, w->img_w, 4, stream));
  auto grouped = [&](dpmn_conv_desc d, long wstride) {
    d.groups = 2; d.w_group_stride = wstride;
    return dpmn_conv2d_nhwc_f32(&d, stream);
  };
  auto kp = [](int k, int cin) { return ((k * k * cin + 31) / 32) * 32; };
  // en_1 (cmm.py:86, 93)
  RUN(grouped(conv_desc(s.xin, 4, nullptr, 0, nullptr, 0, B2, l[0].H, l[0].W, 3, 1, 1, 1, DPMN_ACT_NONE, w->en_w[0], w->en_b[0], c, s.enc[0], sc),
              (long)c * kp(3, 4)));
  // en_2 .. en_5: EncodeBlock = LeakyReLU, Conv2d(cin, cin, 4, 2, dilation 2, padding 3), BN, LeakyReLU, Conv2d(cin, cout, 3, 1, 1), BN (cmm.py:38-55)
  for (int i = 1; i <= 4; ++i) {
    const int ci = l[i - 1].C, co = l[i].C;
    // the block's inner LeakyReLU runs in the first conv's epilogue (once per element) instead of on load in the second
    dpmn_conv_desc da = conv_desc(s.enc[i - 1], ci, nullptr, 0, nullptr, 0, B2, l[i - 1].H, l[i - 1].W, 4, 2, 3, 2, DPMN_ACT_LEAKY02,
                                  w->en_w[2 * i - 1], w->en_b[2 * i - 1], ci, s.tmp, sc);
    da.epi_act = DPMN_ACT_LEAKY02;
    RUN(grouped(da, (long)ci * kp(4, ci)));
    RUN(grouped(conv_desc(s.tmp, ci, nullptr, 0, nullptr, 0, B2, l[i].H, l[i].W, 3, 1, 1, 1, DPMN_ACT_NONE, w->en_w[2 * i], w->en_b[2 * i], co,
                          s.enc[i], sc), (long)co * kp(3, ci)));
  }
  // en_6: LeakyReLU, Conv2d(8c, 8c, 4, 2, 1) (cmm.py:91-92)
  RUN(grouped(conv_desc(s.enc[4], 8 * c, nullptr, 0, nullptr, 0, B2, l[4].H, l[4].W, 4, 2, 1, 1, DPMN_ACT_LEAKY02, w->en_w[9], w->en_b[9], 8 * c,
                        s.enc[5], sc), (long)8 * c * kp(4, 8 * c)));
  // torch.cat([out_6_1, out_6_2], dim=1) (cmm.py:135): two strided copies into the (B, P, 16c) gate input
  const int P = l[5].H * l[5].W;
  const size_t half = (size_t)B * P * 8 * c;
  for (int g = 0; g < 2; ++g) {
    const hipError_t e = hipMemcpy2DAsync(s.bott + g * 8 * c, (size_t)16 * c * 4, s.enc[5] + g * half, (size_t)8 * c * 4, (size_t)8 * c * 4,
                                          (size_t)B * P, hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) return dpmn_set_error(DPMN_ERR_LAUNCH, hipGetErrorString(e));
  }
  float* gated = s.bott + (size_t)B * P * 16 * c;
  RUN(dpmn_se_gate_f32(s.bott, w->fc1_w, w->fc1_b, w->fc2_w, w->fc2_b, gated, s.hid, B, P, 16 * c, 4 * c, stream));   // cmm.py:136-147
  // de_6: ReLU, ConvTranspose2d(16c, 8c, 4, 2, 1), BN (cmm.py:100-102) -- the 4 output phases in one launch
  auto convT = [&](const float* in, int cin, int H, int W, const float* wp, const float* bias, int cout, float* o, int pro_act) {
    dpmn_conv_desc d = conv_desc(in, cin, nullptr, 0, nullptr, 0, B, H, W, 2, 1, 0, -1, pro_act, wp, bias, cout, o, sc);
    d.Hp = H; d.Wp = W; d.Hout = 2 * H; d.Wout = 2 * W; d.ostep = 2;
    d.nphase = 4; d.w_phase_stride = (long)cout * kp(2, cin);
    return dpmn_conv2d_nhwc_f32(&d, stream);
  };
  RUN(convT(gated, 16 * c, l[5].H, l[5].W, w->de6_w, w->de6_b, 8 * c, s.dec, DPMN_ACT_RELU));
  // de_5 .. de_2: DecodeBlock on cat([d, skip_1, skip_2]) (cmm.py:150-158): ReLU, ConvTranspose2d(cin, cout, 3, 1, 1), BN, ReLU,
  // ConvTranspose2d(cout, cout, 4, 2, 1), BN
  const int outc[4] = {8 * c, 4 * c, 2 * c, c};
  int dch = 8 * c;
  for (int j = 0; j < 4; ++j) {
    const int lv = 4 - j;          // skip level index: enc[4] .. enc[1]
    const float* a = s.enc[lv];
    const float* b = s.enc[lv] + (size_t)B * l[lv].H * l[lv].W * l[lv].C;
    dpmn_conv_desc d = conv_desc(s.dec, dch, a, l[lv].C, b, l[lv].C, B, l[lv].H, l[lv].W, 3, 1, 1, 1, DPMN_ACT_RELU, w->dea_w[j], w->dea_b[j], outc[j],
                                 s.tmp, sc);
    d.epi_act = DPMN_ACT_RELU;      // the block's inner ReLU: epilogue of the first conv, none on load in the second
    RUN(dpmn_conv2d_nhwc_f32(&d, stream));
    RUN(convT(s.tmp, outc[j], l[lv].H, l[lv].W, w->deb_w[j], w->deb_b[j], outc[j], s.dec, DPMN_ACT_NONE));
    dch = outc[j];
  }
  // de_1: ReLU, ConvTranspose2d(3c, c_img, 3, 1, 1) (cmm.py:117-118), NCHW store at the module boundary
  {
    const float* a = s.enc[0];
    const float* b = s.enc[0] + (size_t)B * l[0].H * l[0].W * l[0].C;
    dpmn_conv_desc d = conv_desc(s.dec, c, a, c, b, c, B, l[0].H, l[0].W, 3, 1, 1, 1, DPMN_ACT_RELU, w->de1_w, w->de1_b, w->c_img, out, sc);
    d.out_nchw = 1;
    RUN(dpmn_conv2d_nhwc_f32(&d, stream));
  }
#undef RUN
  return DPMN_OK;
}

}  // extern "C"
