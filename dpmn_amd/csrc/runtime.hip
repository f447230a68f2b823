// Error plumbing shared by every entry point (thread-local last-error string).
#include "common.h"
#include <string.h>
#include <stdlib.h>

static thread_local char g_err[512] = "";

int dpmn_set_error(int code, const char* msg) {
  strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
  g_err[sizeof(g_err) - 1] = 0;
  return code;
}

// ------------------------------------------------------------------ in-pipeline kernel profiler (common.h ProfScope)
unsigned long long g_dpmn_prof_mask = 0ull;
double g_dpmn_prof_hint_bytes = 0.0;
int g_dpmn_x3 = 0;
int g_dpmn_bf16 = 0;      // dpmn_set_compute_dtype: 1 = bf16 MFMA operands (fp32 accumulation) in the kernels that have the variant
namespace {
const char* const kTagNames[PT_COUNT] = {
    "k_conv_igemm<128,128>", "k_conv_igemm<64,64>", "k_conv_igemm<128,16|32>", "k_conv_splitk_reduce", "k_conv_halo", "k_conv_halo_c4",
    "k_gemm_pw", "k_gemm_wstat|rowreg", "k_gemm_kloop", "k_dwconv_gelu", "k_window_attn8_mfma", "k_window_attn<2|4|16>", "k_ln_qkv_window_attn",
    "k_bigru", "k_mha32", "k_patch_embed_ln", "k_sk_gate", "k_tail_conv2", "k_mlp_dw_pw", "k_gemm_wstat|rowreg<LN prologue>", "k_conv_igemm_sk", "k_ln_qkv_window_attn_bwd", "k_window_attn_bwd_mfma",
    "k_conv_wgrad", "k_gemm_tn_reg", "k_tn_reduce_multi", "k_dwconv_bwd", "k_wgrad_unpack_multi", "k_conv_pack_multi", "k_affine_act_bwd", "k_ln_bwd", "k_window_attn_mfma<4|8|16,32>"};
struct Rec { int tag; double flops, bytes; };
struct Prof {
  int cap = 0, count = 0, limit = 0;      // cap: allocated event pairs (only grows); limit: max_launches of the current dpmn_profile_begin
  hipEvent_t* ev = nullptr;
  Rec* rec = nullptr;
} g_prof;
}  // namespace

int dpmn_x3_off_mask() {
  static const int m = getenv("DPMN_X3_OFF") ? atoi(getenv("DPMN_X3_OFF")) : 0;
  return m;
}

int dpmn_prof_open(int tag, hipStream_t st, double flops, double bytes) {
  if (g_prof.count >= g_prof.limit) return -1;
  const int i = g_prof.count++;
  g_prof.rec[i] = Rec{tag, flops, bytes};
  (void)hipEventRecord(g_prof.ev[2 * i], st);
  return i;
}
void dpmn_prof_close(int slot, hipStream_t st) { (void)hipEventRecord(g_prof.ev[2 * slot + 1], st); }

extern "C" {
int dpmn_abi_version(void) { return 5; }
int dpmn_set_compute_dtype(int mode) {
  DPMN_REQUIRE(mode >= 0 && mode <= 2, "set_compute_dtype: 0 = f32, 1 = bf16 operands, 2 = f32 via three bf16 terms");
  g_dpmn_bf16 = mode == 1;
  g_dpmn_x3 = mode == 2;
  return DPMN_OK;
}
int dpmn_get_compute_dtype(void) { return g_dpmn_x3 ? 2 : g_dpmn_bf16; }
const char* dpmn_last_error(void) { return g_err; }

int dpmn_profile_tag_count(void) { return PT_COUNT; }
int dpmn_profile_hint_bytes(double bytes) { g_dpmn_prof_hint_bytes = bytes; return DPMN_OK; }
const char* dpmn_profile_tag_name(int tag) { return tag >= 0 && tag < PT_COUNT ? kTagNames[tag] : ""; }

int dpmn_profile_begin(unsigned long long tag_mask, int max_launches) {
  DPMN_REQUIRE(max_launches > 0 && max_launches <= (1 << 18), "profile_begin: 1..262144 launches");
  if (g_prof.cap < max_launches) {
    for (int i = 0; i < 2 * g_prof.cap; ++i) (void)hipEventDestroy(g_prof.ev[i]);
    delete[] g_prof.ev;
    delete[] g_prof.rec;
    g_prof.ev = new hipEvent_t[2 * (size_t)max_launches];
    g_prof.rec = new Rec[(size_t)max_launches];
    for (int i = 0; i < 2 * max_launches; ++i)
      if (hipEventCreate(&g_prof.ev[i]) != hipSuccess) return dpmn_set_error(DPMN_ERR_LAUNCH, "profile_begin: hipEventCreate failed");
    g_prof.cap = max_launches;
  }
  g_prof.count = 0;
  g_prof.limit = max_launches;
  g_dpmn_prof_mask = tag_mask;
  return DPMN_OK;
}

int dpmn_profile_end(dpmn_profile_row* rows, int max_rows) {
  g_dpmn_prof_mask = 0ull;
  dpmn_profile_row acc[PT_COUNT];
  for (int t = 0; t < PT_COUNT; ++t) acc[t] = dpmn_profile_row{t, 0, 0.0, 0.0, 0.0};
  for (int i = 0; i < g_prof.count; ++i) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]) != hipSuccess)
      return dpmn_set_error(DPMN_ERR_LAUNCH, "profile_end: events not complete (synchronise the stream first)");
    dpmn_profile_row& r = acc[g_prof.rec[i].tag];
    r.launches += 1; r.total_ms += ms; r.flops += g_prof.rec[i].flops; r.bytes += g_prof.rec[i].bytes;
  }
  int n = 0;
  for (int t = 0; t < PT_COUNT && n < max_rows; ++t)
    if (acc[t].launches > 0) rows[n++] = acc[t];
  g_prof.count = 0;
  return n;
}
}
