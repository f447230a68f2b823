// Error plumbing shared by every entry point (thread-local last-error string).
#include "common.h"
#include <string.h>

static thread_local char g_err[512] = "";

int dpmn_set_error(int code, const char* msg) {
  strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
  g_err[sizeof(g_err) - 1] = 0;
  return code;
}

extern "C" {
int dpmn_abi_version(void) { return 1; }
const char* dpmn_last_error(void) { return g_err; }
}
