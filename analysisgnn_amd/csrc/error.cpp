#include "agnn_common.h"

namespace agnn {
char* last_error_buf() {
  static thread_local char buf[kErrBuf] = {0};
  return buf;
}
}  // namespace agnn

extern "C" const char* agnn_last_error(void) { return agnn::last_error_buf(); }
extern "C" int agnn_version(void) { return (0 << 16) | 1; }
