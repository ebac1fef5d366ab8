// Error text + version of libmsi_hip.so.
#include <cstring>

#include "msi_common.h"

namespace msi {

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

}  // namespace msi

extern "C" {

const char *msi_version(void) { return "msi_hip 0.1 (gfx950)"; }

const char *msi_last_error_string(void) { return msi::error_buffer(); }

}  // extern "C"
