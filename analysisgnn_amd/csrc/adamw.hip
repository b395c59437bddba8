// Gradient-norm clipping + AdamW over the flat parameter / gradient buffers in two launches.
//
// The reference steps `torch.optim.AdamW` after Lightning's gradient clipping (analysisgnn/models/analysis.py:1380-1381,
// train/train_analysisgnn.py trainer flags).  Over one flat 5 M-element buffer that is still ~20 element-wise torch
// launches (norm, scale, clamp, mul, addcmul, sqrt, addcdiv, ...), each a full pass over 20 MB: 0.25 ms per step.
// Here: k_gnorm writes per-block partial sums of g^2 (fixed order), k_adamw re-adds the partials in the same fixed
// order in every block (bitwise identical coefficient everywhere, no atomics, no host round trip), and applies
//     g' = g * min(max_norm / (|g| + 1e-6), 1);  p *= 1 - lr*wd;  m = b1 m + (1-b1) g';  v = b2 v + (1-b2) g'^2
//     p -= lr * (m / (1 - b1^t)) / (sqrt(v) / sqrt(1 - b2^t) + eps)                      (torch.optim.AdamW's update)
// The step counter t lives on the device so a captured hipGraph advances it on every replay.
#include "agnn_common.h"

namespace {

constexpr int kPartials = 1024;

__global__ __launch_bounds__(256) void k_gnorm(const float* __restrict__ g, int64_t n, float* __restrict__ partial, float* __restrict__ step) {
  __shared__ float sm[256];
  const int64_t per = ((n + kPartials - 1) / kPartials + 3) & ~int64_t{3};
  const int64_t b0 = static_cast<int64_t>(blockIdx.x) * per;
  int64_t b1 = b0 + per;
  if (b1 > n) b1 = n;
  float a = 0.f;
  for (int64_t i = b0 + threadIdx.x; i < b1; i += 256) a = fmaf(g[i], g[i], a);
  sm[threadIdx.x] = a;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (static_cast<int>(threadIdx.x) < o) sm[threadIdx.x] += sm[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    partial[blockIdx.x] = sm[0];
    if (blockIdx.x == 0) step[0] += 1.f;
  }
}

struct AdamArgs {
  float* p; float* g; float* m; float* v;
  int64_t n;
  float lr, b1, b2, eps, wd, max_norm;
  const float* partial; const float* step; float* norm_out;
  int write_g;
};

__global__ __launch_bounds__(256) void k_adamw(AdamArgs a) {
  __shared__ float sm[256];
  float s = 0.f;
  for (int i = threadIdx.x; i < kPartials; i += 256) s += a.partial[i];       // same order in every block
  sm[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (static_cast<int>(threadIdx.x) < o) sm[threadIdx.x] += sm[threadIdx.x + o];
    __syncthreads();
  }
  const float norm = sqrtf(sm[0]);
  if (blockIdx.x == 0 && threadIdx.x == 0 && a.norm_out != nullptr) a.norm_out[0] = norm;
  float coef = 1.f;
  if (a.max_norm > 0.f) {
    coef = a.max_norm / (norm + 1e-6f);
    if (coef > 1.f) coef = 1.f;
  }
  const float t = a.step[0];
  const float bc1 = 1.f - powf(a.b1, t), bc2 = 1.f - powf(a.b2, t);
  const float inv_bc1 = 1.f / bc1, inv_sqrt_bc2 = 1.f / sqrtf(bc2);
  const float decay = 1.f - a.lr * a.wd;
  const int64_t n4 = a.n >> 2;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < n4; i += static_cast<int64_t>(gridDim.x) * 256) {
    float4 p = reinterpret_cast<float4*>(a.p)[i], g = reinterpret_cast<const float4*>(a.g)[i];
    float4 m = reinterpret_cast<float4*>(a.m)[i], v = reinterpret_cast<float4*>(a.v)[i];
#define AGNN_ADAM1(c)                                                          \
    {                                                                          \
      const float gc = g.c * coef;                                             \
      g.c = gc;                                                                \
      m.c = a.b1 * m.c + (1.f - a.b1) * gc;                                    \
      v.c = a.b2 * v.c + (1.f - a.b2) * gc * gc;                               \
      const float denom = sqrtf(v.c) * inv_sqrt_bc2 + a.eps;                   \
      p.c = p.c * decay - a.lr * (m.c * inv_bc1) / denom;                      \
    }
    AGNN_ADAM1(x) AGNN_ADAM1(y) AGNN_ADAM1(z) AGNN_ADAM1(w)
    reinterpret_cast<float4*>(a.p)[i] = p;
    reinterpret_cast<float4*>(a.m)[i] = m;
    reinterpret_cast<float4*>(a.v)[i] = v;
    if (a.write_g) reinterpret_cast<float4*>(a.g)[i] = g;
  }
  if (blockIdx.x == 0 && threadIdx.x < (a.n & 3)) {                            // tail (n not a multiple of 4)
    const int64_t i = (n4 << 2) + threadIdx.x;
    const float gc = a.g[i] * coef;
    const float m = a.b1 * a.m[i] + (1.f - a.b1) * gc;
    const float v = a.b2 * a.v[i] + (1.f - a.b2) * gc * gc;
    a.m[i] = m;
    a.v[i] = v;
    a.p[i] = a.p[i] * decay - a.lr * (m * inv_bc1) / (sqrtf(v) * inv_sqrt_bc2 + a.eps);
    if (a.write_g) a.g[i] = gc;
  }
}
#undef AGNN_ADAM1

}  // namespace

extern "C" size_t agnn_adamw_workspace_bytes(void) { return kPartials * sizeof(float); }

extern "C" int agnn_adamw_f32(float* p, float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                              float weight_decay, float max_norm, float* step, float* norm_out, int32_t write_clipped_grad,
                              void* workspace, size_t workspace_bytes, agnn_stream_t stream_) {
  using namespace agnn;
  if (n < 0) return fail(AGNN_EINVAL, "adamw: n=%lld", (long long)n);
  if (n == 0) return AGNN_OK;
  if (!p || !g || !m || !v || !step || !workspace) return fail(AGNN_EINVAL, "adamw: null argument");
  if (!aligned16(p) || !aligned16(g) || !aligned16(m) || !aligned16(v)) return fail(AGNN_EALIGN, "adamw: buffers must be 16-byte aligned");
  if (workspace_bytes < agnn_adamw_workspace_bytes()) return fail(AGNN_ENOMEM, "adamw: workspace too small");
  if (!(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f)) return fail(AGNN_EINVAL, "adamw: betas (%f, %f)", beta1, beta2);
  hipStream_t s = static_cast<hipStream_t>(stream_);
  float* partial = static_cast<float*>(workspace);
  hipLaunchKernelGGL(k_gnorm, dim3(kPartials), dim3(256), 0, s, g, n, partial, step);
  if (int rc = check_launch("adamw_gnorm")) return rc;
  AdamArgs a{p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, max_norm, partial, step, norm_out, write_clipped_grad ? 1 : 0};
  int64_t blocks = ((n >> 2) + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(k_adamw, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s, a);
  return check_launch("adamw");
}
