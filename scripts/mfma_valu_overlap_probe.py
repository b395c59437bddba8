"""Do a wave's VALU instructions issue while ANOTHER wave of the same SIMD streams MFMAs?  (gfx950 probe, one-off)
A 512-thread workgroup: waves 0..3 (one per SIMD) run a chain-free stream of v_mfma_f32_32x32x2_f32, waves 4..7 (the same
SIMDs) a dependent chain of v_fma_f32; each kind is timed alone and together with s_memtime.  Also: the same VALU chain
interleaved into the MFMA wave's own instruction stream."""
import ctypes, os, subprocess, tempfile
import numpy as np
src = r'''
#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
extern "C" __global__ __launch_bounds__(512) void k(int mode, int n, float* out, long long* t) {
  const int wave = threadIdx.x >> 6;
  const bool mf = wave < 4;
  f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
  float x = threadIdx.x * 1e-3f, y = 1.0001f;
  __syncthreads();
  const long long t0 = clock64();
  if (mf && (mode & 1)) {
    for (int i = 0; i < n; ++i) {
      a0 = MFMA(x, y, a0); a1 = MFMA(x, y, a1); a2 = MFMA(x, y, a2); a3 = MFMA(x, y, a3);
      if (mode & 4) {                      // the VALU chain inside the MFMA wave's own stream: 16 dependent FMAs per 4 MFMAs
#pragma unroll
        for (int j = 0; j < 16; ++j) x = __builtin_fmaf(x, y, 0.5f);
      }
    }
  }
  if (!mf && (mode & 2)) {
    for (int i = 0; i < n; ++i) {
#pragma unroll
      for (int j = 0; j < 16; ++j) x = __builtin_fmaf(x, y, 0.5f);
    }
  }
  const long long t1 = clock64();
  if ((threadIdx.x & 63) == 0) t[blockIdx.x * 8 + wave] = t1 - t0;
  out[blockIdx.x * 512 + threadIdx.x] = x + a0[0] + a1[1] + a2[2] + a3[3];
}
extern "C" int run(int mode, int n, long long* host) {
  float* o; long long* t; hipMalloc(&o, 512 * 4); hipMalloc(&t, 64);
  hipLaunchKernelGGL(k, dim3(1), dim3(512), 0, 0, mode, n, o, t);
  hipMemcpy(host, t, 64, hipMemcpyDeviceToHost); hipFree(o); hipFree(t); return 0;
}
'''
d = tempfile.mkdtemp()
open(os.path.join(d, "p.hip"), "w").write(src)
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "--offload-arch=gfx950", "-shared", "-fPIC", os.path.join(d, "p.hip"), "-o", os.path.join(d, "p.so")])
lib = ctypes.CDLL(os.path.join(d, "p.so"))
n = 2000
for mode, name in ((1, "MFMA waves alone (4 per step)"), (2, "VALU waves alone (16 dependent FMAs per step)"), (3, "both kinds, different waves of the same SIMDs"),
                   (5, "MFMA waves with the 16 FMAs inside their own stream")):
    t = np.zeros(8, dtype=np.int64)
    lib.run(mode, n, t.ctypes.data_as(ctypes.c_void_p))
    lib.run(mode, n, t.ctypes.data_as(ctypes.c_void_p))
    print(f"{name:62s} cycles per step: MFMA waves {t[:4].mean() / n:7.1f}   VALU waves {t[4:].mean() / n:7.1f}")
