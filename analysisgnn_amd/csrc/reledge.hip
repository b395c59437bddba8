// Aggregation side of the reference's in-tree RelEdgeConv (analysisgnn/models/core/gnn.py:99-105):
//     m_ij = W_e [h_j || |h_i - h_j|] + b_e ,    s_i = (h_i + sum_{(i,j)} m_ij) / max(deg_i, 1)
// The reference gathers h twice into [E, F], concatenates to [E, 2F], runs the edge GEMM [E, 2F] x [2F, F] and scatters
// with atomics.  W_e is linear, so the sum over a row's edges moves inside it:
//     sum_j m_ij = W_e [ S_i || D_i ] + deg_i b_e ,    S_i = sum_j h_j ,   D_i = sum_j |h_i - h_j|
// — two per-row aggregates [N, F] from ONE pass over the row's neighbours (one wavefront per row, h_i in registers, each
// h_j row gathered once) and a node-level GEMM [N, 2F] x [2F, F]: E/N times fewer GEMM flops, nothing of size E in HBM.
// Backward of D needs both ends of an edge:
//     dh_i += gD_i * sum_j sign(h_i - h_j)                (CSR by the aggregating row)
//     dh_j += sum_i sign(h_j - h_i) * gD_i  (+ gS_i)      (transposed CSR; the S term rides along)
// Same kernel, the second launch accumulates onto the first's output.  CSR order, no atomics, bitwise reproducible.
// Gather-bound fp32 work (per edge 4H + 4 bytes), no MFMA.
#include "agnn_common.h"

namespace {

__device__ __forceinline__ float4 r4z() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float sgn(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }     // torch.abs' subgradient: sign(0) = 0

struct AbsDiffArgs {
  const int32_t* rowptr;
  const int32_t* col;
  const float* h;
  int64_t ld;
  int32_t n_rows, H;
};

// out = [ S | D ] side by side in one [n, 2 * ld_half] matrix when both are wanted (the GEMM operand), else one of them.
template <int CH>
__global__ __launch_bounds__(256) void k_absdiff_fwd(AbsDiffArgs g, float* __restrict__ out_s, float* __restrict__ out_d, int64_t ld_out) {
  const int lane = threadIdx.x & 63;
  const int row = ((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3)) * 4 + (threadIdx.x >> 6);   // XCD-contiguous row slabs
  if (row >= g.n_rows) return;
  bool on[CH];
  float4 hv[CH], as[CH], ad[CH];
  const float4* hp = reinterpret_cast<const float4*>(g.h + static_cast<int64_t>(row) * g.ld);
#pragma unroll
  for (int k = 0; k < CH; ++k) {
    on[k] = (k * 256 + lane * 4) < g.H;
    hv[k] = on[k] ? hp[k * 64 + lane] : r4z();
    as[k] = r4z();
    ad[k] = r4z();
  }
  const int start = g.rowptr[row], end = g.rowptr[row + 1];
  for (int p = start; p < end; ++p) {
    const int j = __builtin_amdgcn_readfirstlane(g.col[p]);
    const float4* np = reinterpret_cast<const float4*>(g.h + static_cast<int64_t>(j) * g.ld);
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      if (!on[k]) continue;
      const float4 v = np[k * 64 + lane];
      as[k].x += v.x; as[k].y += v.y; as[k].z += v.z; as[k].w += v.w;
      ad[k].x += fabsf(hv[k].x - v.x); ad[k].y += fabsf(hv[k].y - v.y);
      ad[k].z += fabsf(hv[k].z - v.z); ad[k].w += fabsf(hv[k].w - v.w);
    }
  }
#pragma unroll
  for (int k = 0; k < CH; ++k) {
    if (!on[k]) continue;
    if (out_s != nullptr) reinterpret_cast<float4*>(out_s + static_cast<int64_t>(row) * ld_out)[k * 64 + lane] = as[k];
    if (out_d != nullptr) reinterpret_cast<float4*>(out_d + static_cast<int64_t>(row) * ld_out)[k * 64 + lane] = ad[k];
  }
}

// out[r] (+)= sum_{p in row r} sign(h_r - h_col[p]) * (BY_COL ? gd[col[p]] : gd[r])  (+ BY_COL && gs ? gs[col[p]] : 0)
template <int CH, bool BY_COL>
__global__ __launch_bounds__(256) void k_absdiff_bwd(AbsDiffArgs g, const float* __restrict__ gd, const float* __restrict__ gs, int64_t ld_g,
                                                     int accumulate, float* __restrict__ out, int64_t ld_out) {
  const int lane = threadIdx.x & 63;
  const int row = ((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3)) * 4 + (threadIdx.x >> 6);
  if (row >= g.n_rows) return;
  bool on[CH];
  float4 hv[CH], gv[CH], acc[CH];
  const float4* hp = reinterpret_cast<const float4*>(g.h + static_cast<int64_t>(row) * g.ld);
  const float4* gp = gd != nullptr ? reinterpret_cast<const float4*>(gd + static_cast<int64_t>(row) * ld_g) : nullptr;
#pragma unroll
  for (int k = 0; k < CH; ++k) {
    on[k] = (k * 256 + lane * 4) < g.H;
    hv[k] = on[k] ? hp[k * 64 + lane] : r4z();
    gv[k] = (!BY_COL && on[k] && gp != nullptr) ? gp[k * 64 + lane] : r4z();
    acc[k] = r4z();
  }
  const int start = g.rowptr[row], end = g.rowptr[row + 1];
  for (int p = start; p < end; ++p) {
    const int j = __builtin_amdgcn_readfirstlane(g.col[p]);
    const float4* np = reinterpret_cast<const float4*>(g.h + static_cast<int64_t>(j) * g.ld);
    const float4* gdp = (BY_COL && gd != nullptr) ? reinterpret_cast<const float4*>(gd + static_cast<int64_t>(j) * ld_g) : nullptr;
    const float4* gsp = (BY_COL && gs != nullptr) ? reinterpret_cast<const float4*>(gs + static_cast<int64_t>(j) * ld_g) : nullptr;
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      if (!on[k]) continue;
      const float4 v = np[k * 64 + lane];
      float4 w = BY_COL ? (gdp ? gdp[k * 64 + lane] : r4z()) : gv[k];
      acc[k].x = fmaf(sgn(hv[k].x - v.x), w.x, acc[k].x); acc[k].y = fmaf(sgn(hv[k].y - v.y), w.y, acc[k].y);
      acc[k].z = fmaf(sgn(hv[k].z - v.z), w.z, acc[k].z); acc[k].w = fmaf(sgn(hv[k].w - v.w), w.w, acc[k].w);
      if (gsp) {
        const float4 s = gsp[k * 64 + lane];
        acc[k].x += s.x; acc[k].y += s.y; acc[k].z += s.z; acc[k].w += s.w;
      }
    }
  }
  float4* op = reinterpret_cast<float4*>(out + static_cast<int64_t>(row) * ld_out);
#pragma unroll
  for (int k = 0; k < CH; ++k) {
    if (!on[k]) continue;
    float4 r = acc[k];
    if (accumulate) {
      const float4 o = op[k * 64 + lane];
      r.x += o.x; r.y += o.y; r.z += o.z; r.w += o.w;
    }
    op[k * 64 + lane] = r;
  }
}

int ad_check(const char* who, const int32_t* rowptr, const int32_t* col, const float* h, int64_t ld, int64_t n_rows, int32_t H) {
  using namespace agnn;
  if (n_rows < 0 || n_rows >= (int64_t{1} << 31)) return fail(AGNN_EINVAL, "%s: n_rows=%lld", who, (long long)n_rows);
  if (H <= 0 || (H & 3) || H > 1024) return fail(AGNN_EINVAL, "%s: H=%d must be a multiple of 4 in [4,1024]", who, H);
  if (n_rows == 0) return 1;
  if (!rowptr || !col || !h) return fail(AGNN_EINVAL, "%s: null argument", who);
  if (!aligned16(h) || (ld & 3) || ld < H) return fail(AGNN_EALIGN, "%s: h misaligned or ld < H", who);
  return 0;
}

inline unsigned ad_grid(int64_t n_rows) { return static_cast<unsigned>((((n_rows + 3) / 4) + 7) & ~int64_t{7}); }

}  // namespace

extern "C" int agnn_absdiff_fwd_f32(const int32_t* rowptr, const int32_t* col, const float* h, int64_t ld, int64_t n_rows, int32_t H,
                                    float* out_s, float* out_d, int64_t ld_out, agnn_stream_t stream_) {
  using namespace agnn;
  if (int rc = ad_check("absdiff_fwd", rowptr, col, h, ld, n_rows, H)) return rc > 0 ? AGNN_OK : rc;
  if (!out_s && !out_d) return fail(AGNN_EINVAL, "absdiff_fwd: no output requested");
  if ((out_s && !aligned16(out_s)) || (out_d && !aligned16(out_d)) || (ld_out & 3) || ld_out < H) return fail(AGNN_EALIGN, "absdiff_fwd: outputs misaligned or ld_out < H");
  const AbsDiffArgs g{rowptr, col, h, ld, static_cast<int32_t>(n_rows), H};
  const dim3 grid(ad_grid(n_rows)), block(256);
  hipStream_t s = static_cast<hipStream_t>(stream_);
  if (H <= 256) hipLaunchKernelGGL(k_absdiff_fwd<1>, grid, block, 0, s, g, out_s, out_d, ld_out);
  else if (H <= 512) hipLaunchKernelGGL(k_absdiff_fwd<2>, grid, block, 0, s, g, out_s, out_d, ld_out);
  else hipLaunchKernelGGL(k_absdiff_fwd<4>, grid, block, 0, s, g, out_s, out_d, ld_out);
  return check_launch("absdiff_fwd");
}

extern "C" int agnn_absdiff_bwd_f32(const int32_t* rowptr, const int32_t* col, const float* h, int64_t ld, int64_t n_rows, int32_t H,
                                    const float* gd, const float* gs, int64_t ld_g, int32_t by_col, int32_t accumulate, float* out,
                                    int64_t ld_out, agnn_stream_t stream_) {
  using namespace agnn;
  if (int rc = ad_check("absdiff_bwd", rowptr, col, h, ld, n_rows, H)) return rc > 0 ? AGNN_OK : rc;
  if (!out || !aligned16(out) || (ld_out & 3) || ld_out < H) return fail(AGNN_EALIGN, "absdiff_bwd: out misaligned or ld_out < H");
  if (!gd && !gs) return fail(AGNN_EINVAL, "absdiff_bwd: no incoming gradient");
  if ((gd && !aligned16(gd)) || (gs && !aligned16(gs)) || (ld_g & 3) || ld_g < H) return fail(AGNN_EALIGN, "absdiff_bwd: gradients misaligned or ld_g < H");
  if (gs && !by_col) return fail(AGNN_EINVAL, "absdiff_bwd: the sum term's gradient flows to the gathered end only (by_col = 1)");
  const AbsDiffArgs g{rowptr, col, h, ld, static_cast<int32_t>(n_rows), H};
  const dim3 grid(ad_grid(n_rows)), block(256);
  hipStream_t s = static_cast<hipStream_t>(stream_);
#define AD_LAUNCH(CH)                                                                                              \
  do {                                                                                                             \
    if (by_col) hipLaunchKernelGGL((k_absdiff_bwd<CH, true>), grid, block, 0, s, g, gd, gs, ld_g, accumulate, out, ld_out);  \
    else hipLaunchKernelGGL((k_absdiff_bwd<CH, false>), grid, block, 0, s, g, gd, gs, ld_g, accumulate, out, ld_out);        \
  } while (0)
  if (H <= 256) AD_LAUNCH(1);
  else if (H <= 512) AD_LAUNCH(2);
  else AD_LAUNCH(4);
#undef AD_LAUNCH
  return check_launch("absdiff_bwd");
}
