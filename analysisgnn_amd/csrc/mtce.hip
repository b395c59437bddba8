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

using agnn::lane_value;
using agnn::wave_max_dpp;
using agnn::wave_sum_dpp;

// One wavefront per row.  A task's logits sit in up to three registers per lane (C <= 192 covers every head of the
// reference; wider segments take the strided loops below); the next task's values and label are fetched while the
// current task is reduced (DPP reductions, v_exp / v_log), so the 21 tasks of a row do not pay 21 load latencies.
__global__ __launch_bounds__(256) void k_mtce(const float* __restrict__ z, int64_t ld, const int32_t* __restrict__ off, int T,
                                              const int64_t* __restrict__ labels, int64_t n_rows, float eps, int64_t ignore,
                                              const float* __restrict__ inv_cnt, float* __restrict__ row_loss,
                                              float* __restrict__ dz) {
  const int lane = threadIdx.x & 63;
  const int64_t row = static_cast<int64_t>(blockIdx.x) * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (row >= n_rows) return;
  const float* zr = z + row * ld;
  float* dr = dz + row * ld;
  float n0, n1, n2;
  int64_t ny;
  auto fetch = [&](int t) {
    const int a = off[t], b = off[t + 1];
    n0 = (a + lane < b) ? zr[a + lane] : -INFINITY;
    n1 = (a + 64 + lane < b) ? zr[a + 64 + lane] : -INFINITY;
    n2 = (a + 128 + lane < b) ? zr[a + 128 + lane] : -INFINITY;
    ny = labels[static_cast<int64_t>(t) * n_rows + row];
  };
  fetch(0);
  for (int t = 0; t < T; ++t) {
    const int a = off[t], b = off[t + 1];
    const int C = b - a;
    const float v0 = n0, v1 = n1, v2 = n2;
    const int64_t y = ny;
    if (t + 1 < T) fetch(t + 1);
    const bool valid = (y != ignore);
    const float sc = valid ? inv_cnt[t] : 0.f;
    const float sm = eps / static_cast<float>(C);
    if (C <= 192) {
      const float mx = wave_max_dpp(fmaxf(fmaxf(v0, v1), v2));
      const float e0 = __expf(v0 - mx), e1 = __expf(v1 - mx), e2 = __expf(v2 - mx);        // exp(-inf) = 0 for absent classes
      const float se = wave_sum_dpp((e0 + e1) + e2);
      const float sz = wave_sum_dpp(((a + lane < b ? v0 : 0.f) + (a + 64 + lane < b ? v1 : 0.f)) + (a + 128 + lane < b ? v2 : 0.f));
      const float lse = mx + __logf(se);
      float loss = 0.f;
      if (valid) {
        const int yi = static_cast<int>(y);
        const float zy = lane_value(yi < 64 ? v0 : (yi < 128 ? v1 : v2), yi & 63);
        loss = (1.f - eps) * (lse - zy) + eps * (lse - sz / static_cast<float>(C));
      }
      if (lane == 0) row_loss[row * T + t] = loss;
      const float inv = 1.f / se;
      const int yl = static_cast<int>(y) - lane;                      // class c = lane + 64 k is the label when yl == 64 k
      if (a + lane < b) dr[a + lane] = sc * (e0 * inv - ((yl == 0 ? 1.f - eps : 0.f) + sm));
      if (a + 64 + lane < b) dr[a + 64 + lane] = sc * (e1 * inv - ((yl == 64 ? 1.f - eps : 0.f) + sm));
      if (a + 128 + lane < b) dr[a + 128 + lane] = sc * (e2 * inv - ((yl == 128 ? 1.f - eps : 0.f) + sm));
    } else {
      float mx = -INFINITY;
      for (int c = a + lane; c < b; c += 64) mx = fmaxf(mx, zr[c]);
      mx = wave_max_dpp(mx);
      float se = 0.f, sz = 0.f;
      for (int c = a + lane; c < b; c += 64) {
        const float v = zr[c];
        se += __expf(v - mx);
        sz += v;
      }
      se = wave_sum_dpp(se);
      sz = wave_sum_dpp(sz);
      const float lse = mx + __logf(se);
      float loss = 0.f;
      if (valid) {
        const float zy = zr[a + static_cast<int>(y)];
        loss = (1.f - eps) * (lse - zy) + eps * (lse - sz / static_cast<float>(C));
      }
      if (lane == 0) row_loss[row * T + t] = loss;
      for (int c = a + lane; c < b; c += 64) {
        const float p = __expf(zr[c] - lse);
        const float tgt = ((c - a) == y ? (1.f - eps) : 0.f) + sm;
        dr[c] = sc * (p - tgt);
      }
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
