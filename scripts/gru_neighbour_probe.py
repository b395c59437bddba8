"""Why do the backward recurrence launches take 5 - 20 % longer inside the training step than alone?  (gfx950 probe, one-off)
k_gru_bwd (64 workgroups, each alone on its CU) is timed alone, beside a kernel that only streams MFMAs on the other CUs (no
memory traffic at all), and beside one that only streams memory (no MFMA): if pure-MFMA neighbours slow it, the cause is the chip's
clock under load, not the memory system."""
import ctypes, os, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from analysisgnn_amd import _lib
src = r'''
#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
extern "C" __global__ __launch_bounds__(256) void k_mfma(int n, float* out) {
  f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
  float x = threadIdx.x * 1e-3f, y = 1.0001f;
  for (int i = 0; i < n; ++i) { a0 = MFMA(x, y, a0); a1 = MFMA(x, y, a1); a2 = MFMA(x, y, a2); a3 = MFMA(x, y, a3); }
  if (x == 12345.f) out[threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
}
extern "C" __global__ __launch_bounds__(256) void k_stream(const float4* src, float4* dst, long n, int reps) {
  for (int r = 0; r < reps; ++r)
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) dst[i] = src[i];
}
extern "C" void launch_mfma(int blocks, int n, float* out, hipStream_t s) { hipLaunchKernelGGL(k_mfma, dim3(blocks), dim3(256), 0, s, n, out); }
extern "C" void launch_stream(int blocks, const void* a, void* b, long n, int reps, hipStream_t s) {
  hipLaunchKernelGGL(k_stream, dim3(blocks), dim3(256), 0, s, (const float4*)a, (float4*)b, n, reps); }
'''
d = tempfile.mkdtemp()
open(os.path.join(d, "p.hip"), "w").write(src)
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "--offload-arch=gfx950", "-shared", "-fPIC", os.path.join(d, "p.hip"), "-o", os.path.join(d, "p.so")])
P = ctypes.c_void_p
probe = ctypes.CDLL(os.path.join(d, "p.so"))
probe.launch_mfma.argtypes = [ctypes.c_int, ctypes.c_int, P, P]
probe.launch_stream.argtypes = [ctypes.c_int, P, P, ctypes.c_long, ctypes.c_int, P]
lib = _lib.load()
dev = torch.device("cuda", 0)
B, T, Hh = 32, 500, 128
torch.manual_seed(0)
gi = torch.randn(B, T, 2, 3 * Hh, device=dev); w = torch.randn(2, 3 * Hh, Hh, device=dev) * 0.05; bh = torch.randn(2, 3 * Hh, device=dev) * 0.05
y = torch.empty(B, T, 2 * Hh, device=dev); saved = torch.empty(B, T, 2, 4, Hh, device=dev)
dgi = torch.empty_like(gi); dgh = torch.empty_like(gi); dy = torch.randn_like(y); hp = torch.empty(B, T, 2, Hh, device=dev)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
_lib.check(lib.agnn_gru_fwd_f32(gi.data_ptr(), w.data_ptr(), bh.data_ptr(), B, T, Hh, y.data_ptr(), saved.data_ptr(), None, None, sa.cuda_stream), "fwd")
out = torch.empty(256, device=dev)
big_a = torch.empty(64 << 20, dtype=torch.float32, device=dev); big_b = torch.empty_like(big_a)      # 256 MB each
torch.cuda.synchronize()
def gru():
    _lib.check(lib.agnn_gru_bwd_f32(dy.data_ptr(), y.data_ptr(), saved.data_ptr(), w.data_ptr(), B, T, Hh, dgi.data_ptr(), dgh.data_ptr(), None, hp.data_ptr(), sa.cuda_stream), "bwd")
def run(neigh):
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        gru_first = True
        e0.record(sa); gru(); e1.record(sa)                     # the recurrence first: its 64 workgroups take their CUs
        if neigh == "mfma":
            probe.launch_mfma(192 * 2, 6000, out.data_ptr(), sb.cuda_stream)        # ~0.6 ms of MFMAs on the other 192 CUs
        elif neigh == "stream":
            probe.launch_stream(192 * 4, big_a.data_ptr(), big_b.data_ptr(), big_a.numel() // 4, 6, sb.cuda_stream)
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]
for neigh, what in ((None, "alone"), ("mfma", "beside 384 workgroups that only stream MFMAs (no memory traffic)"), ("stream", "beside a 256 MB copy stream (no MFMA)"), (None, "alone again")):
    print(f"k_gru_bwd {what:70s} {run(neigh):7.1f} us", flush=True)
