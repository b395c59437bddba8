// Fused multi-task cross entropy with label smoothing over column segments of one logits matrix.
//
// The reference computes 21 independent `nn.CrossEntropyLoss(ignore_index=-1, label_smoothing=0.1)` terms
// (analysisgnn/models/analysis.py:881-888, summed by models/chord.py:39-49), each a chain of
// log_softmax / nll / smoothing kernels forward and backward.  Here the task logits live side by side in
// one [N, ld] matrix (segment t = columns [off[t], off[t+1])) and 16 lanes per row produce, for all
// tasks in one pass, the per-row loss terms and the FINAL gradient w.r.t. the logits (already divided by
// the number of non-ignored rows of the task), so backward is a scale by the incoming scalar.
//   p = softmax(z);  loss_row = (1-eps) * (-log p_y) + eps * (-(1/C) sum_c log p_c)
//   dz_c = p_c - (1-eps) [c == y] - eps / C        (rows with y == ignore: 0); scaled per task by k_mtce_scale
// Memory-bound: reads and writes the logits matrix once.  No atomics: per-row losses go to [N, T] and are
// summed by the caller in a fixed order.
#include <cmath>
#include <cstdlib>

#include "agnn_common.h"

namespace {

using agnn::row16_max;
using agnn::row16_sum;

// Most heads have few classes (2 ... 50; one has 185), so a wavefront per row leaves most lanes idle and pays three
// full-wave reductions per task: the kernel was instruction-issue bound at ~1 TB/s.  Layout here: 16 lanes per row,
// FOUR rows per wavefront; reductions are four DPP steps inside a 16-lane row (no cross-row traffic at all).
// Segments up to 64 classes keep their logits in registers between the max / sum-exp / gradient passes (NK = 1..4
// values per lane); wider ones re-read them (L1-resident).
template <int NK, class PZ, class PD>
__device__ __forceinline__ void task_regs(PZ zr, PD dr, int a, int b, int sub, int64_t y, bool valid, float sc, float eps, float& loss) {
  float v[NK];
  bool in[NK];
  float mx = -INFINITY;
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    in[k] = a + sub + 16 * k < b;
    v[k] = in[k] ? zr[a + sub + 16 * k] : -INFINITY;
    mx = fmaxf(mx, v[k]);
  }
  mx = row16_max(mx);
  float e[NK], se = 0.f, sz = 0.f;
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    e[k] = __expf(v[k] - mx);                          // exp(-inf) = 0 for absent classes
    se += e[k];
    sz += in[k] ? v[k] : 0.f;
  }
  se = row16_sum(se);
  sz = row16_sum(sz);
  const float C = static_cast<float>(b - a);
  const float lse = mx + __logf(se);
  const float zy = valid ? zr[a + static_cast<int>(y)] : 0.f;
  loss = valid ? (1.f - eps) * (lse - zy) + eps * (lse - sz / C) : 0.f;
  const float inv = 1.f / se, sm = eps / C;
  const int yl = static_cast<int>(y) - sub;
#pragma unroll
  for (int k = 0; k < NK; ++k)
    if (in[k]) dr[a + sub + 16 * k] = sc * (e[k] * inv - ((yl == 16 * k ? 1.f - eps : 0.f) + sm));
}

// all tasks of one row (16 lanes): zr = the row's logits, dr = where its gradient goes (may be the same memory)
template <class PZ, class PD>
__device__ __forceinline__ void mtce_row(PZ zr, PD dr, const int32_t* __restrict__ off, int T, const int64_t* __restrict__ labels,
                                         int64_t n_rows, int64_t row, int sub, float eps, int64_t ignore, float* __restrict__ row_loss,
                                         const float* __restrict__ scale) {
  for (int t = 0; t < T; ++t) {
    const int a = off[t], b = off[t + 1];
    const int C = b - a;
    if (C <= 0) continue;
    const int64_t y_raw = labels[static_cast<int64_t>(t) * n_rows + row];
    const bool valid = (y_raw != ignore);
    // A label outside [0, C) that is not the ignore value: torch's CrossEntropyLoss raises a device assert.  Here the row's
    // loss and gradient become NaN (loud in the total and in every gradient downstream) and nothing is read out of range.
    const bool bad = valid && (y_raw < 0 || y_raw >= C);
    const int64_t y = bad ? 0 : y_raw;
    // scale == nullptr: per-task 1/count and the incoming gradient are applied later (k_mtce_scale / k_train_loss_bwd);
    // otherwise scale[t] is the task's final factor (k_task_scale) and the gradient leaves this kernel finished
    const float sc = valid ? (bad ? NAN : (scale != nullptr ? scale[t] : 1.f)) : 0.f;
    float loss;
    const int nk = (C + 15) >> 4;
    if (nk == 1) task_regs<1>(zr, dr, a, b, sub, y, valid, sc, eps, loss);
    else if (nk == 2) task_regs<2>(zr, dr, a, b, sub, y, valid, sc, eps, loss);
    else if (nk == 3) task_regs<3>(zr, dr, a, b, sub, y, valid, sc, eps, loss);
    else if (nk == 4) task_regs<4>(zr, dr, a, b, sub, y, valid, sc, eps, loss);
    else {
      float mx = -INFINITY;
      for (int c = a + sub; c < b; c += 16) mx = fmaxf(mx, zr[c]);
      mx = row16_max(mx);
      float se = 0.f, sz = 0.f;
      for (int c = a + sub; c < b; c += 16) {
        const float v = zr[c];
        se += __expf(v - mx);
        sz += v;
      }
      se = row16_sum(se);
      sz = row16_sum(sz);
      const float lse = mx + __logf(se);
      const float zy = valid ? zr[a + static_cast<int>(y)] : 0.f;
      loss = valid ? (1.f - eps) * (lse - zy) + eps * (lse - sz / static_cast<float>(C)) : 0.f;
      const float inv = 1.f / se, sm = eps / static_cast<float>(C);
      for (int c = a + sub; c < b; c += 16) {
        const float p = __expf(zr[c] - mx) * inv;
        dr[c] = sc * (p - (((c - a) == y ? 1.f - eps : 0.f) + sm));
      }
    }
    if (sub == 0) row_loss[static_cast<int64_t>(t) * n_rows + row] = bad ? NAN : loss;     // task-major: the reduction reads contiguously
  }
}

__global__ __launch_bounds__(256) void k_mtce(const float* __restrict__ z, int64_t ld, const int32_t* __restrict__ off, int T,
                                              const int64_t* __restrict__ labels, int64_t n_rows, float eps, int64_t ignore,
                                              float* __restrict__ row_loss, float* __restrict__ dz, const float* __restrict__ scale) {
  const int lane = threadIdx.x & 63, sub = lane & 15;
  const int64_t row = (static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6)) * 4 + (lane >> 4);
  if (row >= n_rows) return;                                // whole 16-lane rows drop out; DPP never crosses a row
  mtce_row(z + row * ld, dz + row * ld, off, T, labels, n_rows, row, sub, eps, ignore, row_loss, scale);
}

// The same through LDS: the 16 lanes of a row first fetch the WHOLE row (all loads in flight together, 64 contiguous bytes
// per row and instruction), the task loop then reads and overwrites the LDS image (logits -> gradient in place), and the
// gradient leaves as one pass of stores.  In k_mtce every task is its own global round trip — load, three 16-lane
// reductions, store, 21 times in a row — and the kernel ran at 0.75 TB/s of traffic it touches exactly once (55 us at C2).
// A row's 16 lanes only ever touch their own LDS row: no barrier, a wave-level fence between the phases.
constexpr int kMtceMaxCols = 1024;      // LDS image: 4 waves x 4 rows x W floats (W = 634 at C2: 40 KB per workgroup)

__global__ __launch_bounds__(256) void k_mtce_lds(const float* __restrict__ z, int64_t ld, const int32_t* __restrict__ off, int T,
                                                  const int64_t* __restrict__ labels, int64_t n_rows, float eps, int64_t ignore,
                                                  float* __restrict__ row_loss, float* __restrict__ dz,
                                                  const float* __restrict__ scale) {
  extern __shared__ float s_rows[];                         // [16 rows][ld]
  const int lane = threadIdx.x & 63, sub = lane & 15;
  const int slot = (threadIdx.x >> 6) * 4 + (lane >> 4);
  const int64_t row = static_cast<int64_t>(blockIdx.x) * 16 + slot;
  if (row >= n_rows) return;
  const int lo = off[0], hi = off[T];                       // the columns the segments cover; anything else is left alone
  float* sr = s_rows + static_cast<size_t>(slot) * ld;
  const float* zr = z + row * ld;
  float* dr = dz + row * ld;
  // 8-byte pieces when the geometry allows (even row stride, even segment range, 8-byte aligned matrices: the C2 logits are
  // [N, 658]): 128 contiguous bytes per row and instruction instead of 64, half the memory instructions, twice the bytes in flight
  const bool vec2 = ((ld | lo | hi) & 1) == 0 && ((reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(dz)) & 7u) == 0;
  if (vec2) {
    for (int c0 = lo + 2 * sub; c0 < hi; c0 += 32 * 8) {    // eight loads in flight per lane and trip
      float2 v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = c0 + 32 * k < hi ? *reinterpret_cast<const float2*>(zr + c0 + 32 * k) : make_float2(0.f, 0.f);
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (c0 + 32 * k < hi) *reinterpret_cast<float2*>(sr + c0 + 32 * k) = v[k];
    }
  } else {
    for (int c0 = lo + sub; c0 < hi; c0 += 16 * 8) {        // eight loads in flight per lane and trip
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = c0 + 16 * k < hi ? zr[c0 + 16 * k] : 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (c0 + 16 * k < hi) sr[c0 + 16 * k] = v[k];
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  mtce_row(sr, sr, off, T, labels, n_rows, row, sub, eps, ignore, row_loss, scale);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  if (vec2) {
    for (int c = lo + 2 * sub; c < hi; c += 32) *reinterpret_cast<float2*>(dr + c) = *reinterpret_cast<const float2*>(sr + c);
  } else {
    for (int c = lo + sub; c < hi; c += 16) dr[c] = sr[c];
  }
}

void launch_mtce(hipStream_t s, const float* logits, int64_t ld, const int32_t* seg_off, int n_tasks, const int64_t* labels, int64_t n_rows,
                 float eps, int64_t ignore, float* row_loss, float* dlogits, const float* scale = nullptr) {
  const unsigned blocks = static_cast<unsigned>((n_rows + 15) / 16);   // 4 waves x 4 rows
  if (ld > 0 && ld <= kMtceMaxCols)
    hipLaunchKernelGGL(k_mtce_lds, dim3(blocks), dim3(256), static_cast<size_t>(16) * ld * sizeof(float), s, logits, ld, seg_off, n_tasks,
                       labels, n_rows, eps, ignore, row_loss, dlogits, scale);
  else
    hipLaunchKernelGGL(k_mtce, dim3(blocks), dim3(256), 0, s, logits, ld, seg_off, n_tasks, labels, n_rows, eps, ignore, row_loss, dlogits, scale);
}

// Before the cross entropy: wscale[t] = ce_scale * w_t / max(count_t, 1), count_t = #{n : labels[t][n] != ignore}, w_t = 0.5 / p_t^2
// with task weights (1 without) — everything the logits' gradient is scaled by except the incoming scalar.  With it the
// cross-entropy kernel writes the finished gradient and the backward pass has no launch of its own (the scale pass read and
// wrote the [N, C] matrix once more, 23 us at C2, on the step's serial stretch between the heads and their backward).
__global__ __launch_bounds__(1024) void k_task_scale(const int64_t* __restrict__ labels, int64_t n_rows, int64_t ignore,
                                                    const float* __restrict__ task_param, float ce_scale, float* __restrict__ wscale) {
  __shared__ int sc[1024];
  const int t = blockIdx.x;
  const int64_t* lb = labels + static_cast<int64_t>(t) * n_rows;
  int c = 0;
  int64_t i = threadIdx.x;
  for (; i + 3 * 1024 < n_rows; i += 4 * 1024) {
    const int64_t l0 = lb[i], l1 = lb[i + 1024], l2 = lb[i + 2048], l3 = lb[i + 3072];
    c += (l0 != ignore) + (l1 != ignore) + (l2 != ignore) + (l3 != ignore);
  }
  for (; i < n_rows; i += 1024) c += lb[i] != ignore ? 1 : 0;
  sc[threadIdx.x] = c;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if (static_cast<int>(threadIdx.x) < o) sc[threadIdx.x] += sc[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    float w = 1.f;
    if (task_param != nullptr) w = 0.5f / (task_param[t] * task_param[t]);
    wscale[t] = ce_scale * w * (1.f / static_cast<float>(sc[0] > 0 ? sc[0] : 1));     // the same expression as k_train_loss_reduce's
  }
}

// loss[t] = sum_n row_loss[t][n] / max(count_t, 1),  inv_cnt[t] = 1 / max(count_t, 1),  count_t = #{n : labels[t][n] != ignore}.
// One block per task, fixed-order tree: bitwise reproducible.
__global__ __launch_bounds__(1024) void k_mtce_reduce(const float* __restrict__ row_loss, const int64_t* __restrict__ labels, int64_t n_rows,
                                                     int64_t ignore, float* __restrict__ loss, float* __restrict__ inv_cnt) {
  __shared__ float sl[1024];
  __shared__ int sc[1024];
  const int t = blockIdx.x;
  const float* rl = row_loss + static_cast<int64_t>(t) * n_rows;
  const int64_t* lb = labels + static_cast<int64_t>(t) * n_rows;
  float a = 0.f;
  int c = 0;
  for (int64_t i = threadIdx.x; i < n_rows; i += 1024) {
    a += rl[i];
    c += lb[i] != ignore ? 1 : 0;
  }
  sl[threadIdx.x] = a;
  sc[threadIdx.x] = c;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if (static_cast<int>(threadIdx.x) < o) { sl[threadIdx.x] += sl[threadIdx.x + o]; sc[threadIdx.x] += sc[threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float inv = 1.f / static_cast<float>(sc[0] > 0 ? sc[0] : 1);
    loss[t] = sl[0] * inv;
    inv_cnt[t] = inv;
  }
}

// out[n, c] = dz[n, c] * scale[task(c)] for the columns of the T segments; columns outside every segment are copied
__global__ __launch_bounds__(256) void k_mtce_scale(const float* __restrict__ dz, int64_t ld, const int32_t* __restrict__ off, int T,
                                                    int64_t n_rows, int total_cols, const float* __restrict__ scale,
                                                    float* __restrict__ out, int64_t ld_out) {
  const int lane = threadIdx.x & 63;
  const int64_t row = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (row >= n_rows) return;
  const float* zr = dz + row * ld;
  float* orow = out + row * ld_out;
  const int lo = off[0], hi = off[T];
  int t = 0;
  for (int c = lane; c < total_cols; c += 64) {
    while (t + 1 < T && c >= off[t + 1]) ++t;        // columns ascend per lane: the task index only moves forward
    orow[c] = (c >= lo && c < hi) ? zr[c] * scale[t] : zr[c];
  }
}

// Training loss in two launches:  total = ce_scale * sum_t (w_t * loss[t] + reg_t) + lambda * mean(feat^2)
// (ref: models/analysis.py:1034-1036 `total_loss = loss_dict.pop("total") / len(labels_dict)`, :984 `feature_loss =
// x.pow(2).mean()`, :1072 `total_loss += ... feature_loss * self.lambda_featl`).  With learned uncertainty weights p
// (MultiTaskLoss, models/chord.py:39-49, the CLI default --mt_strategy wloss): w_t = 0.5 / p_t^2, reg_t = log(1 + p_t^2);
// without: w_t = 1, reg_t = 0.
// Blocks 0 .. T-1 reduce one task each (as k_mtce_reduce), blocks T .. T+kFeatBlocks-1 reduce a fixed slice of feat^2;
// the last block to finish (integer ticket) adds the T + kFeatBlocks partial results in index order and resets the ticket:
// the result does not depend on which block that was.
constexpr int kFeatBlocks = 256;

__global__ __launch_bounds__(1024) void k_train_loss_reduce(const float* __restrict__ row_loss, const int64_t* __restrict__ labels,
                                                            int64_t n_rows, int64_t ignore, int T, float* __restrict__ loss,
                                                            float* __restrict__ inv_cnt, const float* __restrict__ feat, int64_t ld_feat,
                                                            int feat_cols, float lam_over_numel, float* __restrict__ fpart,
                                                            unsigned int* __restrict__ ticket, float* __restrict__ total,
                                                            const float* __restrict__ task_param, float ce_scale,
                                                            float* __restrict__ wscale, float* __restrict__ dparam,
                                                            float* __restrict__ dfeat, int64_t ld_dfeat) {
  __shared__ float sl[1024];
  __shared__ int sc[1024];
  __shared__ bool last;
  const float dcoef = 2.f * lam_over_numel;            // dfeat = d (lambda * mean(feat^2)) / d feat, for an incoming gradient of 1
  const int t = blockIdx.x;
  float a = 0.f;
  int c = 0;
  if (t < T) {
    const float* rl = row_loss + static_cast<int64_t>(t) * n_rows;
    const int64_t* lb = labels + static_cast<int64_t>(t) * n_rows;
    // four independent loads in flight per thread (a plain loop is one L2 round trip per iteration); fixed order
    float a4[4] = {0.f, 0.f, 0.f, 0.f};
    int64_t i = threadIdx.x;
    for (; i + 3 * 1024 < n_rows; i += 4 * 1024) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a4[u] += rl[i + u * 1024];
        c += lb[i + u * 1024] != ignore ? 1 : 0;
      }
    }
    for (; i < n_rows; i += 1024) {
      a4[0] += rl[i];
      c += lb[i] != ignore ? 1 : 0;
    }
    a = (a4[0] + a4[1]) + (a4[2] + a4[3]);
  } else if (feat != nullptr) {
    const int64_t numel = n_rows * feat_cols;
    const int64_t per = (numel + kFeatBlocks - 1) / kFeatBlocks;
    const int64_t e0 = (t - T) * per;
    int64_t e1 = e0 + per;
    if (e1 > numel) e1 = numel;
    if (ld_feat == feat_cols && (dfeat == nullptr || ld_dfeat == feat_cols)) {   // contiguous: no index arithmetic, four loads in flight
      float a4[4] = {0.f, 0.f, 0.f, 0.f};
      int64_t e = e0 + threadIdx.x;
      for (; e + 3 * 1024 < e1; e += 4 * 1024) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float v = feat[e + u * 1024];
          a4[u] = fmaf(v, v, a4[u]);
          if (dfeat != nullptr) dfeat[e + u * 1024] = dcoef * v;
        }
      }
      for (; e < e1; e += 1024) {
        const float v = feat[e];
        a4[0] = fmaf(v, v, a4[0]);
        if (dfeat != nullptr) dfeat[e] = dcoef * v;
      }
      a = (a4[0] + a4[1]) + (a4[2] + a4[3]);
    } else {
      for (int64_t e = e0 + threadIdx.x; e < e1; e += 1024) {
        const int64_t r = e / feat_cols;
        const float v = feat[r * ld_feat + (e - r * feat_cols)];
        a = fmaf(v, v, a);
        if (dfeat != nullptr) dfeat[r * ld_dfeat + (e - r * feat_cols)] = dcoef * v;
      }
    }
  }
  sl[threadIdx.x] = a;
  sc[threadIdx.x] = c;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if (static_cast<int>(threadIdx.x) < o) { sl[threadIdx.x] += sl[threadIdx.x + o]; sc[threadIdx.x] += sc[threadIdx.x + o]; }
    __syncthreads();
  }
  // The partial results travel through returning device-scope atomics (performed at the coherence point of the 8 XCDs'
  // L2s) instead of plain stores + __threadfence(): an agent-scope release writes back every dirty line of the L2,
  // and the cross-entropy kernel has just left 50 MB of them there (measured: 19.6 -> 11.8 us).
  unsigned int* pub = reinterpret_cast<unsigned int*>(fpart);
  if (threadIdx.x == 0) {
    float val;
    if (t < T) {
      const float inv = 1.f / static_cast<float>(sc[0] > 0 ? sc[0] : 1);
      val = sl[0] * inv;
      loss[t] = val;                                   // for later kernels (stream order)
      inv_cnt[t] = inv;
    } else {
      val = sl[0];
    }
    const unsigned int before = atomicExch(pub + t, __float_as_uint(val));
    asm volatile("" ::"v"(before));                    // the exchange has completed before the ticket is taken
    if (t < T) {                                       // the count travels the same way (the weighted gradient scale needs it)
      const unsigned int b2 = atomicExch(pub + 1024 + t, __float_as_uint(inv_cnt[t]));
      asm volatile("" ::"v"(b2));
    }
    last = atomicAdd(ticket, 1u) == gridDim.x - 1;
  }
  __syncthreads();
  if (last) {                                          // block-uniform
    const int tid = threadIdx.x;                       // T + kFeatBlocks <= 1024 (host-checked): one value per thread, then a serial sum in LDS
    if (tid < T + kFeatBlocks) sl[tid] = __uint_as_float(atomicOr(pub + tid, 0u));
    if (tid < T) {                                     // per-task weight, gradient scale and d total / d p_t
      float w = 1.f, dp = 0.f;
      if (task_param != nullptr) {
        const float p = task_param[tid], p2 = p * p;
        w = 0.5f / p2;
        dp = ce_scale * (2.f * p / (1.f + p2) - sl[tid] / (p2 * p));
      }
      if (wscale != nullptr) wscale[tid] = ce_scale * w * __uint_as_float(atomicOr(pub + 1024 + tid, 0u));
      if (dparam != nullptr) dparam[tid] = dp;
    }
    __syncthreads();
    if (tid == 0) {
      float tot = 0.f, f = 0.f;
      for (int i = 0; i < T; ++i) {
        if (task_param != nullptr) {
          const float p2 = task_param[i] * task_param[i];
          tot += 0.5f / p2 * sl[i] + logf(1.f + p2);
        } else {
          tot += sl[i];
        }
      }
      for (int i = 0; i < kFeatBlocks; ++i) f += sl[T + i];
      *total = fmaf(f, lam_over_numel, ce_scale * tot);
      atomicExch(ticket, 0u);
    }
  }
}

// Backward of the above in one launch: out[n, c] = dz[n, c] * g * inv_cnt[task(c)] (blocks over the logits rows), and
// dfeat[n, c] = g * 2 * lambda / numel * feat[n, c] (the remaining blocks).
constexpr int kBwdMaxCols = 4096;                     // per-column scale table in LDS (16 KB)

template <int VEC>      // floats per access on the logits rows: 4, 2 or 0 (scalar, any layout)
__global__ __launch_bounds__(256) void k_train_loss_bwd(const float* __restrict__ dz, int64_t ld, const int32_t* __restrict__ off, int T,
                                                        int64_t n_rows, int total_cols, const float* __restrict__ inv_cnt,
                                                        const float* __restrict__ g, float* __restrict__ out, int64_t ld_out,
                                                        const float* __restrict__ feat, int64_t ld_feat, int feat_cols, float coef,
                                                        float* __restrict__ dfeat, int64_t ld_dfeat, unsigned logit_blocks) {
  __shared__ __attribute__((aligned(16))) float s_scale[VEC ? kBwdMaxCols : 1];
  const int lane = threadIdx.x & 63;
  const float gg = *g;
  if (blockIdx.x < logit_blocks) {
    if (VEC) {                                        // column -> g / count of its task, once per block; rows as float4
      const int lo = off[0], hi = off[T];
      for (int c = threadIdx.x; c < total_cols; c += 256) {
        int t = 0;
        while (t + 1 < T && c >= off[t + 1]) ++t;
        s_scale[c] = (c >= lo && c < hi) ? gg * inv_cnt[t] : 0.f;
      }
      __syncthreads();
      for (int64_t row = static_cast<int64_t>(blockIdx.x) * 16 + (threadIdx.x >> 6); row < n_rows && row < static_cast<int64_t>(blockIdx.x + 1) * 16; row += 4) {
        if (VEC == 4) {
          const float4* zr = reinterpret_cast<const float4*>(dz + row * ld);
          float4* orow = reinterpret_cast<float4*>(out + row * ld_out);
          for (int c = lane; c < (total_cols >> 2); c += 64) {
            const float4 v = zr[c];
            const float4 sc = *reinterpret_cast<const float4*>(&s_scale[4 * c]);
            orow[c] = make_float4(v.x * sc.x, v.y * sc.y, v.z * sc.z, v.w * sc.w);
          }
        } else {
          const float2* zr = reinterpret_cast<const float2*>(dz + row * ld);
          float2* orow = reinterpret_cast<float2*>(out + row * ld_out);
          for (int c = lane; c < (total_cols >> 1); c += 64) {
            const float2 v = zr[c];
            const float2 sc = *reinterpret_cast<const float2*>(&s_scale[2 * c]);
            orow[c] = make_float2(v.x * sc.x, v.y * sc.y);
          }
        }
      }
    } else {
      for (int64_t row = static_cast<int64_t>(blockIdx.x) * 16 + (threadIdx.x >> 6); row < n_rows && row < static_cast<int64_t>(blockIdx.x + 1) * 16; row += 4) {
        const float* zr = dz + row * ld;
        float* orow = out + row * ld_out;
        const int lo = off[0], hi = off[T];
        int t = 0;
        for (int c = lane; c < total_cols; c += 64) {
          while (t + 1 < T && c >= off[t + 1]) ++t;    // columns ascend per lane: the task index only moves forward
          orow[c] = (c >= lo && c < hi) ? zr[c] * (gg * inv_cnt[t]) : 0.f;
        }
      }
    }
  } else {
    const float s = gg * coef;
    const int64_t b = blockIdx.x - logit_blocks;
    for (int64_t row = b * 16 + (threadIdx.x >> 6); row < n_rows && row < (b + 1) * 16; row += 4)
      for (int c = lane; c < feat_cols; c += 64) dfeat[row * ld_dfeat + c] = s * feat[row * ld_feat + c];
  }
}

}  // namespace

extern "C" int agnn_multitask_ce_f32(const float* logits, int64_t ld, const int32_t* seg_off, int32_t n_tasks,
                                     const int64_t* labels, int64_t n_rows, float label_smoothing, int64_t ignore_index,
                                     float* row_loss, float* dlogits, float* loss, float* inv_count, agnn_stream_t stream_) {
  using namespace agnn;
  if (n_rows < 0 || n_tasks < 0 || ld < 0) return fail(AGNN_EINVAL, "multitask_ce: negative size");
  if (n_rows == 0 || n_tasks == 0) return AGNN_OK;
  if (!logits || !seg_off || !labels || !row_loss || !dlogits || !loss || !inv_count) return fail(AGNN_EINVAL, "multitask_ce: null argument");
  if (label_smoothing < 0.f || label_smoothing >= 1.f) return fail(AGNN_EINVAL, "multitask_ce: label_smoothing=%f", label_smoothing);
  hipStream_t s = static_cast<hipStream_t>(stream_);
  launch_mtce(s, logits, ld, seg_off, n_tasks, labels, n_rows, label_smoothing, ignore_index, row_loss, dlogits);
  if (int rc = check_launch("multitask_ce")) return rc;
  hipLaunchKernelGGL(k_mtce_reduce, dim3(n_tasks), dim3(1024), 0, s, row_loss, labels, n_rows, ignore_index, loss, inv_count);
  return check_launch("multitask_ce_reduce");
}

extern "C" int agnn_multitask_ce_scale_f32(const float* dlogits, int64_t ld, const int32_t* seg_off, int32_t n_tasks, int64_t n_rows,
                                           int32_t n_cols, const float* scale, float* out, int64_t ld_out, agnn_stream_t stream_) {
  using namespace agnn;
  if (n_rows < 0 || n_tasks < 0 || n_cols < 0 || n_cols > ld || n_cols > ld_out) return fail(AGNN_EINVAL, "multitask_ce_scale: bad size");
  if (n_rows == 0 || n_tasks == 0) return AGNN_OK;
  if (!dlogits || !seg_off || !scale || !out) return fail(AGNN_EINVAL, "multitask_ce_scale: null argument");
  hipLaunchKernelGGL(k_mtce_scale, dim3(static_cast<unsigned>((n_rows + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream_),
                     dlogits, ld, seg_off, n_tasks, n_rows, n_cols, scale, out, ld_out);
  return check_launch("multitask_ce_scale");
}

extern "C" size_t agnn_train_loss_workspace_bytes(void) { return 256 + 2048 * sizeof(float); }

extern "C" int agnn_train_loss_f32(const float* logits, int64_t ld, const int32_t* seg_off, int32_t n_tasks, const int64_t* labels,
                                   int64_t n_rows, float label_smoothing, int64_t ignore_index, const float* feat, int64_t ld_feat,
                                   int32_t feat_cols, float lambda_feat, const float* task_param, float ce_scale, float* row_loss,
                                   float* dlogits, float* loss, float* inv_count, float* total, float* wscale, float* dparam,
                                   void* workspace, size_t workspace_bytes, agnn_stream_t stream_) {
  using namespace agnn;
  if (n_rows <= 0 || n_tasks <= 0 || n_tasks > 1024 - kFeatBlocks || ld < 0) return fail(AGNN_EINVAL, "train_loss: n_rows=%lld n_tasks=%d", (long long)n_rows, n_tasks);
  if (!logits || !seg_off || !labels || !row_loss || !dlogits || !loss || !inv_count || !total || !workspace) return fail(AGNN_EINVAL, "train_loss: null argument");
  if (label_smoothing < 0.f || label_smoothing >= 1.f) return fail(AGNN_EINVAL, "train_loss: label_smoothing=%f", label_smoothing);
  if (feat && (feat_cols <= 0 || ld_feat < feat_cols)) return fail(AGNN_EINVAL, "train_loss: feat_cols=%d ld_feat=%lld", feat_cols, (long long)ld_feat);
  if (workspace_bytes < agnn_train_loss_workspace_bytes() || (reinterpret_cast<uintptr_t>(workspace) & 255u)) return fail(AGNN_ENOMEM, "train_loss: workspace too small or not 256-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream_);
  launch_mtce(s, logits, ld, seg_off, n_tasks, labels, n_rows, label_smoothing, ignore_index, row_loss, dlogits);
  if (int rc = check_launch("train_loss/ce")) return rc;
  unsigned int* ticket = reinterpret_cast<unsigned int*>(workspace);
  float* fpart = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + 256);
  const float lam = feat ? lambda_feat / (static_cast<float>(n_rows) * static_cast<float>(feat_cols)) : 0.f;
  hipLaunchKernelGGL(k_train_loss_reduce, dim3(n_tasks + kFeatBlocks), dim3(1024), 0, s, row_loss, labels, n_rows, ignore_index, n_tasks,
                     loss, inv_count, feat, ld_feat, feat_cols, lam, fpart, ticket, total, task_param, ce_scale, wscale, dparam,
                     static_cast<float*>(nullptr), static_cast<int64_t>(0));
  return check_launch("train_loss/reduce");
}

extern "C" int agnn_train_loss_final_f32(const float* logits, int64_t ld, const int32_t* seg_off, int32_t n_tasks, const int64_t* labels,
                                         int64_t n_rows, float label_smoothing, int64_t ignore_index, const float* feat, int64_t ld_feat,
                                         int32_t feat_cols, float lambda_feat, const float* task_param, float ce_scale, float* row_loss,
                                         float* dlogits, float* loss, float* inv_count, float* total, float* wscale, float* dparam,
                                         float* dfeat, int64_t ld_dfeat, void* workspace, size_t workspace_bytes, agnn_stream_t stream_) {
  using namespace agnn;
  if (n_rows <= 0 || n_tasks <= 0 || n_tasks > 1024 - kFeatBlocks || ld < 0) return fail(AGNN_EINVAL, "train_loss_final: n_rows=%lld n_tasks=%d", (long long)n_rows, n_tasks);
  if (!logits || !seg_off || !labels || !row_loss || !dlogits || !loss || !inv_count || !total || !wscale || !workspace) return fail(AGNN_EINVAL, "train_loss_final: null argument");
  if (label_smoothing < 0.f || label_smoothing >= 1.f) return fail(AGNN_EINVAL, "train_loss_final: label_smoothing=%f", label_smoothing);
  if (feat && (feat_cols <= 0 || ld_feat < feat_cols)) return fail(AGNN_EINVAL, "train_loss_final: feat_cols=%d ld_feat=%lld", feat_cols, (long long)ld_feat);
  if (dfeat && (!feat || ld_dfeat < feat_cols)) return fail(AGNN_EINVAL, "train_loss_final: dfeat without feat, or ld_dfeat=%lld", (long long)ld_dfeat);
  if (workspace_bytes < agnn_train_loss_workspace_bytes() || (reinterpret_cast<uintptr_t>(workspace) & 255u)) return fail(AGNN_ENOMEM, "train_loss_final: workspace too small or not 256-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream_);
  hipLaunchKernelGGL(k_task_scale, dim3(n_tasks), dim3(1024), 0, s, labels, n_rows, ignore_index, task_param, ce_scale, wscale);
  if (int rc = check_launch("train_loss_final/scale")) return rc;
  launch_mtce(s, logits, ld, seg_off, n_tasks, labels, n_rows, label_smoothing, ignore_index, row_loss, dlogits, wscale);
  if (int rc = check_launch("train_loss_final/ce")) return rc;
  unsigned int* ticket = reinterpret_cast<unsigned int*>(workspace);
  float* fpart = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + 256);
  const float lam = feat ? lambda_feat / (static_cast<float>(n_rows) * static_cast<float>(feat_cols)) : 0.f;
  hipLaunchKernelGGL(k_train_loss_reduce, dim3(n_tasks + kFeatBlocks), dim3(1024), 0, s, row_loss, labels, n_rows, ignore_index, n_tasks,
                     loss, inv_count, feat, ld_feat, feat_cols, lam, fpart, ticket, total, task_param, ce_scale, wscale, dparam,
                     dfeat, ld_dfeat);
  return check_launch("train_loss_final/reduce");
}

extern "C" int agnn_train_loss_bwd_f32(const float* dlogits, int64_t ld, const int32_t* seg_off, int32_t n_tasks, int64_t n_rows,
                                       int32_t n_cols, const float* inv_count, const float* g, float* out, int64_t ld_out,
                                       const float* feat, int64_t ld_feat, int32_t feat_cols, float lambda_feat, float* dfeat,
                                       int64_t ld_dfeat, agnn_stream_t stream_) {
  using namespace agnn;
  if (n_rows <= 0 || n_tasks <= 0 || n_cols < 0 || n_cols > ld || n_cols > ld_out) return fail(AGNN_EINVAL, "train_loss_bwd: bad size");
  if (!dlogits || !seg_off || !inv_count || !g || !out) return fail(AGNN_EINVAL, "train_loss_bwd: null argument");
  if (dfeat && (!feat || feat_cols <= 0 || ld_feat < feat_cols || ld_dfeat < feat_cols)) return fail(AGNN_EINVAL, "train_loss_bwd: bad feature arguments");
  const unsigned lb = static_cast<unsigned>((n_rows + 15) / 16);      // 16 rows per workgroup
  const unsigned fb = dfeat ? lb : 0u;
  const float coef = dfeat ? 2.f * lambda_feat / (static_cast<float>(n_rows) * static_cast<float>(feat_cols)) : 0.f;
  const uintptr_t al = reinterpret_cast<uintptr_t>(dlogits) | reinterpret_cast<uintptr_t>(out);
  const int64_t geo = n_cols | ld | ld_out;
  const int vec = n_cols > kBwdMaxCols ? 0 : ((geo & 3) == 0 && (al & 15u) == 0 ? 4 : ((geo & 1) == 0 && (al & 7u) == 0 ? 2 : 0));
  hipStream_t s = static_cast<hipStream_t>(stream_);
#define AGNN_TLB(V)                                                                                                              \
  hipLaunchKernelGGL(k_train_loss_bwd<V>, dim3(lb + fb), dim3(256), 0, s, dlogits, ld, seg_off, n_tasks, n_rows, n_cols, inv_count, \
                     g, out, ld_out, feat, ld_feat, feat_cols, coef, dfeat, ld_dfeat, lb)
  if (vec == 4) AGNN_TLB(4); else if (vec == 2) AGNN_TLB(2); else AGNN_TLB(0);
#undef AGNN_TLB
  return check_launch("train_loss_bwd");
}
