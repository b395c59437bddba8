// Fused multi-task cross entropy with label smoothing over column segments of one logits matrix.
//
// The reference computes 21 independent `nn.CrossEntropyLoss(ignore_index=-1, label_smoothing=0.1)` terms
// (analysisgnn/models/analysis.py:881-888, summed by models/chord.py:39-49), each a chain of
// log_softmax / nll / smoothing kernels forward and backward.  Here the task logits live side by side in
// one [N, ld] matrix (segment t = columns [off[t], off[t+1])) and one wavefront per row produces, for all
// tasks in one pass, the per-row loss terms and the FINAL gradient w.r.t. the logits (already divided by
// the number of non-ignored rows of the task), so backward is a scale by the incoming scalar.
//   p = softmax(z);  loss_row = (1-eps) * (-log p_y) + eps * (-(1/C) sum_c log p_c)
//   dz_c = inv_cnt[t] * (p_c - (1-eps) [c == y] - eps / C)        (rows with y == ignore: 0)
// Memory-bound: reads and writes the logits matrix once.  No atomics: per-row losses go to [N, T] and are
// summed by the caller in a fixed order.
#include <cmath>

#include "agnn_common.h"

namespace {

__device__ __forceinline__ float wave_max(float v) {
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__global__ __launch_bounds__(256) void k_mtce(const float* __restrict__ z, int64_t ld, const int32_t* __restrict__ off, int T,
                                              const int64_t* __restrict__ labels, int64_t n_rows, float eps, int64_t ignore,
                                              const float* __restrict__ inv_cnt, float* __restrict__ row_loss,
                                              float* __restrict__ dz) {
  const int lane = threadIdx.x & 63;
  const int64_t row = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (row >= n_rows) return;
  const float* zr = z + row * ld;
  float* dr = dz + row * ld;
  for (int t = 0; t < T; ++t) {
    const int a = off[t], b = off[t + 1];
    const int C = b - a;
    const int64_t y = labels[static_cast<int64_t>(t) * n_rows + row];
    const bool valid = (y != ignore);
    float mx = -INFINITY;
    for (int c = a + lane; c < b; c += 64) mx = fmaxf(mx, zr[c]);
    mx = wave_max(mx);
    float se = 0.f, sz = 0.f;
    for (int c = a + lane; c < b; c += 64) {
      const float v = zr[c];
      se += expf(v - mx);
      sz += v;
    }
    se = wave_sum(se);
    sz = wave_sum(sz);
    const float lse = mx + logf(se);
    float loss = 0.f;
    if (valid) {
      const float zy = zr[a + static_cast<int>(y)];
      loss = (1.f - eps) * (lse - zy) + eps * (lse - sz / static_cast<float>(C));
    }
    if (lane == 0) row_loss[row * T + t] = loss;
    const float sc = valid ? inv_cnt[t] : 0.f;
    const float sm = eps / static_cast<float>(C);
    for (int c = a + lane; c < b; c += 64) {
      const float p = expf(zr[c] - lse);
      const float tgt = ((c - a) == y ? (1.f - eps) : 0.f) + sm;
      dr[c] = sc * (p - tgt);
    }
  }
}

}  // namespace

extern "C" int agnn_multitask_ce_f32(const float* logits, int64_t ld, const int32_t* seg_off, int32_t n_tasks,
                                     const int64_t* labels, int64_t n_rows, float label_smoothing, int64_t ignore_index,
                                     const float* inv_count, float* row_loss, float* dlogits, agnn_stream_t stream_) {
  using namespace agnn;
  if (n_rows < 0 || n_tasks < 0 || ld < 0) return fail(AGNN_EINVAL, "multitask_ce: negative size");
  if (n_rows == 0 || n_tasks == 0) return AGNN_OK;
  if (!logits || !seg_off || !labels || !inv_count || !row_loss || !dlogits) return fail(AGNN_EINVAL, "multitask_ce: null argument");
  if (label_smoothing < 0.f || label_smoothing >= 1.f) return fail(AGNN_EINVAL, "multitask_ce: label_smoothing=%f", label_smoothing);
  const unsigned blocks = static_cast<unsigned>((n_rows + 3) / 4);
  hipLaunchKernelGGL(k_mtce, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream_), logits, ld, seg_off, n_tasks,
                     labels, n_rows, label_smoothing, ignore_index, inv_count, row_loss, dlogits);
  return check_launch("multitask_ce");
}
