// Shared host-side helpers of libmsi_hip.so (error text, launch checks).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>

#include "msi_hip.h"

namespace msi {

char *error_buffer();          // thread-local, 512 bytes
int fail(int code, const char *fmt, ...);

inline hipStream_t as_stream(msi_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// Checks the launch that has just been enqueued (no synchronisation).
inline int check_launch(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(MSI_E_LAUNCH, "%s: %s", what, hipGetErrorString(e));
  return MSI_OK;
}

}  // namespace msi

#define MSI_REQUIRE(cond, ...)                          \
  do {                                                  \
    if (!(cond)) return msi::fail(MSI_E_BADARG, __VA_ARGS__); \
  } while (0)
