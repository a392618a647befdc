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
extern "C" int nudf_version(void) { return 105; }   // 105: NudfChain.tile_scale / tile_amax_in / tile_amax_out; 104: NudfChain.absmax_out, NudfGemmTNGroup.amax_a / amax_b + prec 4 (struct sizes); 103: NudfChainStep.X3 / ldx3, prec 4 chains, NudfPackFrag.dtype 4

// The non-finite status word (include/nudf.h): ONE int32 in device memory the caller owns.  The launchers of the three
// kernels that can see a non-finite value first hand the pointer to their kernels as a plain kernel argument (baked into a
// captured HIP graph like every other pointer of the step); the kernels OR a bit into it, nobody ever reads it on the
// device, and the host reads it when it chooses to (no per-step sync).
static int32_t* g_status = nullptr;
extern "C" int nudf_set_status_flag(int32_t* device_word) {
  g_status = device_word;
  return 0;
}
extern "C" int32_t* nudf_status_flag(void) { return g_status; }
