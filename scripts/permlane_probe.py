"""gfx950 v_permlane16_swap / v_permlane32_swap semantics with both operands = the same register (one-off probe;
the GRU kernels' lane-group broadcasts rely on it).  Compiles a tiny kernel with hipcc at run time."""
import ctypes, os, subprocess, sys, tempfile
import numpy as np
src = r'''
#include <hip/hip_runtime.h>
extern "C" __global__ void k(unsigned* o) {
  unsigned x = threadIdx.x;
  auto a = __builtin_amdgcn_permlane16_swap(x, x, false, false);
  auto b = __builtin_amdgcn_permlane32_swap(x, x, false, false);
  o[threadIdx.x] = a[0]; o[64 + threadIdx.x] = a[1]; o[128 + threadIdx.x] = b[0]; o[192 + threadIdx.x] = b[1];
}
extern "C" int run(unsigned* host) {
  unsigned* d; hipMalloc(&d, 256 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(host, d, 256 * 4, hipMemcpyDeviceToHost); hipFree(d); return 0;
}
'''
d = tempfile.mkdtemp()
open(os.path.join(d, "p.hip"), "w").write(src)
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "--offload-arch=gfx950", "-shared", "-fPIC", os.path.join(d, "p.hip"), "-o", os.path.join(d, "p.so")])
lib = ctypes.CDLL(os.path.join(d, "p.so"))
out = np.zeros(256, dtype=np.uint32)
lib.run(out.ctypes.data_as(ctypes.c_void_p))
for name, row in zip(("permlane16_swap[0]", "permlane16_swap[1]", "permlane32_swap[0]", "permlane32_swap[1]"), out.reshape(4, 64)):
    print(name, "rows of 16 come from source rows:", [int(row[16 * r] // 16) for r in range(4)], "lane order kept:", bool(all(row[i] % 16 == i % 16 for i in range(64))))
