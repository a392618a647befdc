// error plumbing + version for libnudf
#include "nudf_common.h"
#include "../../include/nudf.h"
#include <stdio.h>
#include <string.h>

static thread_local char g_err[512] = "";

extern "C" void nudf_set_error(const char* where, hipError_t e) {
  snprintf(g_err, sizeof(g_err), "%s: %s", where, hipGetErrorString(e));
}
extern "C" const char* nudf_last_error(void) { return g_err; }
extern "C" int nudf_version(void) { return 100; }
