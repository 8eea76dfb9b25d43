// sitk core: error reporting and ABI bookkeeping.
#include <stdarg.h>
#include <stdio.h>

#include "common.h"

namespace sitk {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// Launch-time errors only (bad configuration, missing code object): never synchronises.
int check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return SITK_ERR_LAUNCH;
  }
  return SITK_OK;
}

}  // namespace sitk

extern "C" int sitk_abi_version(void) { return SITK_ABI_VERSION; }
extern "C" const char* sitk_last_error(void) { return sitk::g_err; }
extern "C" int sitk_dtype_size(int dtype) { return dtype == SITK_BF16 ? 2 : (dtype == SITK_F32 ? 4 : 0); }
