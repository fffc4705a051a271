// Error plumbing and version of the C ABI (include/unflow_hip.h).
#include "common.h"

namespace unflow {

char *error_buffer() {
  static thread_local char buf[512] = {0};
  return buf;
}

int fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(error_buffer(), 512, fmt, ap);
  va_end(ap);
  return code;
}

}  // namespace unflow

extern "C" int unflow_abi_version(void) { return UNFLOW_ABI_VERSION; }
extern "C" const char *unflow_last_error(void) { return unflow::error_buffer(); }
