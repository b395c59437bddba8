// Multi-relation segmented gather-reduce ("hetero SpMM") for gfx950.
//
// One 64-lane wavefront owns one output row; a lane owns 4 consecutive floats of every
// 256-float chunk of the row, so a 1 KiB row (H = 256) is ONE global_load_dwordx4 per neighbour
// and one global_store_dwordx4 per relation slot.  Column indices of a segment are read
// coalesced (lane p reads col[start + p]) and broadcast with v_readlane, which makes every
// neighbour-row base address wave-uniform (scalar base + lane offset addressing).  Neighbour
// loads are issued four at a time before the first use so several HBM/L2 requests are in flight
// per wave; with 8-16 waves per SIMD resident that hides the dependent rowptr -> col -> row chain.
// Sums run in CSR order: bitwise reproducible, no atomics.  HBM/L2-bound integer+fp32-add work:
// no MFMA here on purpose.
//
// Replaces: `h[edge_index[1]]` + torch_scatter.scatter(..., out=x.clone(), reduce='mean')
// (reference analysisgnn/models/core/gnn.py:70-74), the zero-initialised scatter_add calls
// (core/gnn.py:511,539; core/hgnn.py:406-407), PyG SAGEConv's mean aggregation
// (models/cadence.py:147-159) and the onset pooling (models/analysis.py:580-586).
#include "agnn_common.h"

namespace {

struct RelTable {
  agnn_rel_t r[AGNN_MAX_SEG];
  int n_rel;
};

struct SpmmArgs {
  int32_t n_rows;
  int32_t H;
  float* out;
  int64_t ld_out;
  int64_t rel_stride;
  const float* self;
  int64_t ld_self;
  float* inv_cnt;
  int32_t col_limit;
  uint32_t flags;
};

__device__ __forceinline__ float4 f4_zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ void f4_fma(float4& a, float w, const float4& v) {
  a.x = fmaf(w, v.x, a.x);
  a.y = fmaf(w, v.y, a.y);
  a.z = fmaf(w, v.z, a.z);
  a.w = fmaf(w, v.w, a.w);
}
__device__ __forceinline__ void f4_add(float4& a, const float4& v) {
  a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
}
__device__ __forceinline__ void f4_div(float4& a, float d) {  // true division, as torch_scatter's out.div_(count)
  a.x /= d; a.y /= d; a.z /= d; a.w /= d;
}

// CH = number of 256-float chunks that cover a row (H <= 256*CH)
template <int CH>
__global__ __launch_bounds__(256) void k_spmm(RelTable t, SpmmArgs a) {
  const int lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  const int wave = threadIdx.x >> 6;
  const bool mean = (a.flags & AGNN_SPMM_MEAN) != 0;
  const bool skip_self = (a.flags & AGNN_SPMM_SKIP_SELF) != 0;
  const bool shared_slot = a.rel_stride == 0;
  bool lane_on[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) lane_on[c] = (c * 256 + lane * 4) < a.H;

  for (int row = blockIdx.x * wpb + wave; row < a.n_rows; row += gridDim.x * wpb) {
    float4 selfv[CH];
    if (a.self != nullptr) {
      const float4* sp = reinterpret_cast<const float4*>(a.self + static_cast<int64_t>(row) * a.ld_self);
#pragma unroll
      for (int c = 0; c < CH; ++c) selfv[c] = lane_on[c] ? sp[c * 64 + lane] : f4_zero();
    }
    float4 tot[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) tot[c] = f4_zero();

    for (int r = 0; r < t.n_rel; ++r) {
      const agnn_rel_t& R = t.r[r];
      const int start = R.rowptr[row];
      const int end = (R.rowend != nullptr) ? R.rowend[row] : R.rowptr[row + 1];
      float4 acc[CH];
#pragma unroll
      for (int c = 0; c < CH; ++c) acc[c] = f4_zero();
      int cnt = 0;
      for (int base = start; base < end; base += 64) {
        const int p = base + lane;
        int col = -1;
        float w = 0.f;
        if (p < end) {
          const int c0 = R.col[p];
          const bool valid = !(skip_self && c0 == row) && (c0 < a.col_limit) && (c0 >= 0);
          if (valid) {
            col = c0;
            w = 1.f;
            if (R.ew != nullptr) w *= R.ew[p];
            if (R.colscale != nullptr) w *= R.colscale[c0];
          }
        }
        cnt += __popcll(__ballot(col >= 0));
        const int n = (end - base) < 64 ? (end - base) : 64;
        for (int k = 0; k < n; k += 4) {
          int ck[4];
          float wk[4];
          float4 v[4][CH];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int kk = (k + u < n) ? (k + u) : k;  // clamp: duplicates get weight 0 below
            ck[u] = __builtin_amdgcn_readlane(col, kk);
            wk[u] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w), kk));
            if (k + u >= n) ck[u] = -1;
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            if (ck[u] >= 0) {  // wave-uniform
              const float4* sp = reinterpret_cast<const float4*>(R.src + static_cast<int64_t>(ck[u]) * R.ld_src);
#pragma unroll
              for (int c = 0; c < CH; ++c) v[u][c] = lane_on[c] ? sp[c * 64 + lane] : f4_zero();
            } else {
#pragma unroll
              for (int c = 0; c < CH; ++c) v[u][c] = f4_zero();
              wk[u] = 0.f;
            }
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int c = 0; c < CH; ++c) f4_fma(acc[c], wk[u], v[u][c]);
          }
        }
      }
      const float denom = static_cast<float>(cnt > 1 ? cnt : 1);
      if (a.inv_cnt != nullptr && lane == 0)
        a.inv_cnt[static_cast<int64_t>(r) * a.n_rows + row] = 1.f / denom;
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        if (a.self != nullptr) f4_add(acc[c], selfv[c]);
        if (mean) f4_div(acc[c], denom);
      }
      if (shared_slot) {
#pragma unroll
        for (int c = 0; c < CH; ++c) f4_add(tot[c], acc[c]);
      } else {
        float4* op = reinterpret_cast<float4*>(a.out + static_cast<int64_t>(row) * a.ld_out + static_cast<int64_t>(r) * a.rel_stride);
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          if (!lane_on[c]) continue;
          float4 o = acc[c];
          if (a.flags & AGNN_SPMM_ACCUM) f4_add(o, op[c * 64 + lane]);
          op[c * 64 + lane] = o;
        }
      }
    }
    if (shared_slot) {
      float4* op = reinterpret_cast<float4*>(a.out + static_cast<int64_t>(row) * a.ld_out);
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        if (!lane_on[c]) continue;
        float4 o = tot[c];
        if (a.flags & AGNN_SPMM_ACCUM) f4_add(o, op[c * 64 + lane]);
        op[c * 64 + lane] = o;
      }
    }
  }
}

}  // namespace

extern "C" int agnn_spmm_f32(int n_rel, const agnn_rel_t* rels, int64_t n_rows, int32_t H, float* out,
                             int64_t ld_out, int64_t rel_stride, const float* self, int64_t ld_self,
                             float* inv_cnt, int32_t col_limit, uint32_t flags, agnn_stream_t stream_) {
  using namespace agnn;
  if (n_rel <= 0 || n_rel > AGNN_MAX_SEG) return fail(AGNN_EINVAL, "spmm: n_rel=%d not in [1,%d]", n_rel, AGNN_MAX_SEG);
  if (n_rows < 0 || n_rows >= (int64_t{1} << 31)) return fail(AGNN_EINVAL, "spmm: n_rows=%lld", (long long)n_rows);
  if (H <= 0 || (H & 3) != 0 || H > 1024) return fail(AGNN_EINVAL, "spmm: H=%d must be a multiple of 4 in [4,1024]", H);
  if (n_rows == 0) return AGNN_OK;
  if (!rels || !out) return fail(AGNN_EINVAL, "spmm: null argument");
  if (!aligned16(out) || (ld_out & 3) || (rel_stride & 3) || ld_out < H) return fail(AGNN_EALIGN, "spmm: out/ld_out/rel_stride must be 16-byte aligned and ld_out >= H");
  if (rel_stride != 0 && rel_stride < H) return fail(AGNN_EINVAL, "spmm: rel_stride=%lld < H", (long long)rel_stride);
  if (self && (!aligned16(self) || (ld_self & 3) || ld_self < H)) return fail(AGNN_EALIGN, "spmm: self misaligned");
  RelTable t{};
  t.n_rel = n_rel;
  for (int r = 0; r < n_rel; ++r) {
    if (!rels[r].rowptr) return fail(AGNN_EINVAL, "spmm: relation %d has null rowptr", r);
    if (rels[r].src && (!aligned16(rels[r].src) || (rels[r].ld_src & 3) || rels[r].ld_src < H)) return fail(AGNN_EALIGN, "spmm: relation %d src misaligned or ld_src < H", r);
    t.r[r] = rels[r];
  }
  SpmmArgs a{static_cast<int32_t>(n_rows), H, out, ld_out, rel_stride, self, ld_self, inv_cnt, col_limit, flags};
  const int threads = 256;  // 4 waves = 4 rows per block
  int64_t blocks = (n_rows + 3) / 4;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (H <= 256) hipLaunchKernelGGL(k_spmm<1>, dim3(blocks), dim3(threads), 0, stream, t, a);
  else if (H <= 512) hipLaunchKernelGGL(k_spmm<2>, dim3(blocks), dim3(threads), 0, stream, t, a);
  else hipLaunchKernelGGL(k_spmm<4>, dim3(blocks), dim3(threads), 0, stream, t, a);
  return check_launch("spmm");
}
