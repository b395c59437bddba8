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

#if defined(__HIPCC__)
// ReLU that lets a NaN through, as torch.relu does (fmaxf(NaN, 0) is 0: a poisoned row would come out of the forward pass as
// zeros, with a finite loss, and only show as non-finite GRADIENTS — seen in round 3 when an un-normalised stack overflowed)
__device__ __forceinline__ float relu_nan(float v) { return v < 0.f ? 0.f : v; }


// Wave-wide reductions on the DPP cross-lane paths (no LDS round trip as with __shfl_xor / ds_bpermute):
// quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror reduce each 16-lane row; v_readlane joins the 4 rows.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float lane_value(float v, int lane) {   // lane must be wave-uniform
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_mov<0xB1>(v);
  v += dpp_mov<0x4E>(v);
  v += dpp_mov<0x141>(v);
  v += dpp_mov<0x140>(v);
  return v;
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, dpp_mov<0xB1>(v));
  v = fmaxf(v, dpp_mov<0x4E>(v));
  v = fmaxf(v, dpp_mov<0x141>(v));
  v = fmaxf(v, dpp_mov<0x140>(v));
  return v;
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v = row16_sum(v);
  return (lane_value(v, 0) + lane_value(v, 16)) + (lane_value(v, 32) + lane_value(v, 48));
}
__device__ __forceinline__ float wave_max_dpp(float v) {
  v = fmaxf(v, dpp_mov<0xB1>(v));
  v = fmaxf(v, dpp_mov<0x4E>(v));
  v = fmaxf(v, dpp_mov<0x141>(v));
  v = fmaxf(v, dpp_mov<0x140>(v));
  return fmaxf(fmaxf(lane_value(v, 0), lane_value(v, 16)), fmaxf(lane_value(v, 32), lane_value(v, 48)));
}
#endif

// wgrad.hip: fixed-order sum of S partial-result slabs (used by the weight-gradient kernels)
int launch_slab_reduce(const float* slab, const float* slab_b, int S, int out_f, int in_f, int out_pad, int in_pad, float* dw,
                       int64_t ld_dw, float* db, hipStream_t s);

}  // namespace agnn
