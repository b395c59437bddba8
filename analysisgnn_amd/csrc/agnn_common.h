// Internal helpers shared by the kernel translation units (not part of the C-ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "agnn.h"

namespace agnn {

char* last_error_buf();
constexpr int kErrBuf = 512;

inline int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(last_error_buf(), kErrBuf, fmt, ap);
  va_end(ap);
  return code;
}

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(AGNN_ERUNTIME, "%s: %s", what, hipGetErrorString(e));
  return AGNN_OK;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

constexpr int kWave = 64;  // gfx950 wavefront

}  // namespace agnn
