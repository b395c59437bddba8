#include "agnn_common.h"

namespace agnn {
char* last_error_buf() {
  static thread_local char buf[kErrBuf] = {0};
  return buf;
}
}  // namespace agnn

extern "C" const char* agnn_last_error(void) { return agnn::last_error_buf(); }
extern "C" int agnn_version(void) { return (0 << 16) | 2; }

// The device-side status word of a caller (agnn_csr_build's `status`): waits for the stream, reads the word, and turns a
// non-zero count into AGNN_ERUNTIME.  The one synchronising entry point of the library — call it outside hot loops.
extern "C" int agnn_check_status(const int32_t* status, agnn_stream_t stream_) {
  using namespace agnn;
  if (!status) return fail(AGNN_EINVAL, "check_status: null argument");
  int32_t host = 0;
  hipError_t e = hipMemcpyAsync(&host, status, sizeof(host), hipMemcpyDeviceToHost, static_cast<hipStream_t>(stream_));
  if (e == hipSuccess) e = hipStreamSynchronize(static_cast<hipStream_t>(stream_));
  if (e != hipSuccess) return fail(AGNN_ERUNTIME, "check_status: %s", hipGetErrorString(e));
  if (host != 0)
    return fail(AGNN_ERUNTIME, "device status word = %d: kernels flagged %d events since it was zeroed (agnn_csr_build: an edge position outside "
                "its row, i.e. counters not clean when the build started; agnn_sample_hops: a subgraph proposed more distinct "
                "out-of-window sources than the kernel's table holds and the batch is incomplete)", host, host);
  return AGNN_OK;
}
