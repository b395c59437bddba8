// Edge-gated aggregation of the reference's in-tree ResGatedGraphConv (analysisgnn/models/core/gnn.py:243-258):
//     S_i = sum_{e=(i,j)} sigmoid(a_i + b_j [+ c_e]) * h_j          (a = W3 x, b = W4 x, h = W2 x, c = W5 e_feat)
// The reference materialises three [E, H] gathers, the [E, H] gate and an atomic scatter.  Here one
// wavefront owns one destination row (a lane owns 4 floats of every 256-float chunk): a_i stays in
// registers, b_j / h_j rows are gathered once, the gate lives only in registers.  Backward recomputes
// the gate: pass by destination gives da_i (+ dc_e rows), pass by source over the transposed CSR gives
// db_j and dh_j.  CSR order, no atomics, bitwise reproducible.  Gather-bound fp32 work; no MFMA.
#include <cmath>

#include "agnn_common.h"

namespace {

__device__ __forceinline__ float4 g4z() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float sig(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float4 sig4(const float4& a, const float4& b, const float4& c) {
  return make_float4(sig(a.x + b.x + c.x), sig(a.y + b.y + c.y), sig(a.z + b.z + c.z), sig(a.w + b.w + c.w));
}

struct GatedArgs {
  const int32_t* rowptr;
  const int32_t* col;
  const int32_t* perm;
  const float* a;      // indexed by THIS kernel's row in fwd / bwd_dst, by col in bwd_src
  const float* b;      // indexed by col in fwd / bwd_dst, by row in bwd_src
  const float* h;
  const float* c;      // optional per-edge term, rows indexed by perm (COO edge id)
  int64_t ld, ld_c;
  int32_t n_rows, H;
};

template <int CH>
__global__ __launch_bounds__(256) void k_gated_fwd(GatedArgs g, float* __restrict__ out, int64_t ld_out) {
  const int lane = threadIdx.x & 63;
  const int row = ((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3)) * 4 + (threadIdx.x >> 6);
  if (row >= g.n_rows) return;
  bool on[CH];
  float4 av[CH], acc[CH];
  const float4* ap = reinterpret_cast<const float4*>(g.a + static_cast<int64_t>(row) * g.ld);
#pragma unroll
  for (int k = 0; k < CH; ++k) {
    on[k] = (k * 256 + lane * 4) < g.H;
    av[k] = on[k] ? ap[k * 64 + lane] : g4z();
    acc[k] = g4z();
  }
  const int start = g.rowptr[row], end = g.rowptr[row + 1];
  for (int p = start; p < end; ++p) {
    const int j = __builtin_amdgcn_readfirstlane(g.col[p]);
    const float4* bp = reinterpret_cast<const float4*>(g.b + static_cast<int64_t>(j) * g.ld);
    const float4* hp = reinterpret_cast<const float4*>(g.h + static_cast<int64_t>(j) * g.ld);
    const float4* cp = nullptr;
    if (g.c != nullptr) cp = reinterpret_cast<const float4*>(g.c + static_cast<int64_t>(__builtin_amdgcn_readfirstlane(g.perm[p])) * g.ld_c);
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      if (!on[k]) continue;
      const float4 bv = bp[k * 64 + lane], hv = hp[k * 64 + lane];
      const float4 cv = cp ? cp[k * 64 + lane] : g4z();
      const float4 z = sig4(av[k], bv, cv);
      acc[k].x = fmaf(z.x, hv.x, acc[k].x); acc[k].y = fmaf(z.y, hv.y, acc[k].y);
      acc[k].z = fmaf(z.z, hv.z, acc[k].z); acc[k].w = fmaf(z.w, hv.w, acc[k].w);
    }
  }
  float4* op = reinterpret_cast<float4*>(out + static_cast<int64_t>(row) * ld_out);
#pragma unroll
  for (int k = 0; k < CH; ++k)
    if (on[k]) op[k * 64 + lane] = acc[k];
}

// by destination: da_i = sum_e dS_i * h_j * z(1-z); optional dc[perm[p]] = that summand
template <int CH>
__global__ __launch_bounds__(256) void k_gated_bwd_dst(GatedArgs g, const float* __restrict__ ds, int64_t ld_ds,
                                                       float* __restrict__ da, float* __restrict__ dc) {
  const int lane = threadIdx.x & 63;
  const int row = ((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3)) * 4 + (threadIdx.x >> 6);
  if (row >= g.n_rows) return;
  bool on[CH];
  float4 av[CH], dv[CH], acc[CH];
  const float4* ap = reinterpret_cast<const float4*>(g.a + static_cast<int64_t>(row) * g.ld);
  const float4* dp = reinterpret_cast<const float4*>(ds + static_cast<int64_t>(row) * ld_ds);
#pragma unroll
  for (int k = 0; k < CH; ++k) {
    on[k] = (k * 256 + lane * 4) < g.H;
    av[k] = on[k] ? ap[k * 64 + lane] : g4z();
    dv[k] = on[k] ? dp[k * 64 + lane] : g4z();
    acc[k] = g4z();
  }
  const int start = g.rowptr[row], end = g.rowptr[row + 1];
  for (int p = start; p < end; ++p) {
    const int j = __builtin_amdgcn_readfirstlane(g.col[p]);
    const int64_t e = __builtin_amdgcn_readfirstlane(g.perm[p]);
    const float4* bp = reinterpret_cast<const float4*>(g.b + static_cast<int64_t>(j) * g.ld);
    const float4* hp = reinterpret_cast<const float4*>(g.h + static_cast<int64_t>(j) * g.ld);
    const float4* cp = g.c != nullptr ? reinterpret_cast<const float4*>(g.c + e * g.ld_c) : nullptr;
    float4* dcp = dc != nullptr ? reinterpret_cast<float4*>(dc + e * g.ld_c) : nullptr;
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      if (!on[k]) continue;
      const float4 bv = bp[k * 64 + lane], hv = hp[k * 64 + lane];
      const float4 cv = cp ? cp[k * 64 + lane] : g4z();
      const float4 z = sig4(av[k], bv, cv);
      const float4 t = make_float4(dv[k].x * hv.x * z.x * (1.f - z.x), dv[k].y * hv.y * z.y * (1.f - z.y),
                                   dv[k].z * hv.z * z.z * (1.f - z.z), dv[k].w * hv.w * z.w * (1.f - z.w));
      acc[k].x += t.x; acc[k].y += t.y; acc[k].z += t.z; acc[k].w += t.w;
      if (dcp) dcp[k * 64 + lane] = t;
    }
  }
  float4* op = reinterpret_cast<float4*>(da + static_cast<int64_t>(row) * g.ld);
#pragma unroll
  for (int k = 0; k < CH; ++k)
    if (on[k]) op[k * 64 + lane] = acc[k];
}

// by source (transposed CSR: row = source j, col = destination i):
//   dh_j = sum_e z * dS_i ;  db_j = sum_e dS_i * h_j * z(1-z)
template <int CH>
__global__ __launch_bounds__(256) void k_gated_bwd_src(GatedArgs g, const float* __restrict__ ds, int64_t ld_ds,
                                                       float* __restrict__ db, float* __restrict__ dh) {
  const int lane = threadIdx.x & 63;
  const int row = ((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3)) * 4 + (threadIdx.x >> 6);
  if (row >= g.n_rows) return;
  bool on[CH];
  float4 bv[CH], hv[CH], accb[CH], acch[CH];
  const float4* bp = reinterpret_cast<const float4*>(g.b + static_cast<int64_t>(row) * g.ld);
  const float4* hp = reinterpret_cast<const float4*>(g.h + static_cast<int64_t>(row) * g.ld);
#pragma unroll
  for (int k = 0; k < CH; ++k) {
    on[k] = (k * 256 + lane * 4) < g.H;
    bv[k] = on[k] ? bp[k * 64 + lane] : g4z();
    hv[k] = on[k] ? hp[k * 64 + lane] : g4z();
    accb[k] = g4z();
    acch[k] = g4z();
  }
  const int start = g.rowptr[row], end = g.rowptr[row + 1];
  for (int p = start; p < end; ++p) {
    const int i = __builtin_amdgcn_readfirstlane(g.col[p]);
    const int64_t e = __builtin_amdgcn_readfirstlane(g.perm[p]);
    const float4* ap = reinterpret_cast<const float4*>(g.a + static_cast<int64_t>(i) * g.ld);
    const float4* dp = reinterpret_cast<const float4*>(ds + static_cast<int64_t>(i) * ld_ds);
    const float4* cp = g.c != nullptr ? reinterpret_cast<const float4*>(g.c + e * g.ld_c) : nullptr;
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      if (!on[k]) continue;
      const float4 av = ap[k * 64 + lane], dv = dp[k * 64 + lane];
      const float4 cv = cp ? cp[k * 64 + lane] : g4z();
      const float4 z = sig4(av, bv[k], cv);
      acch[k].x = fmaf(z.x, dv.x, acch[k].x); acch[k].y = fmaf(z.y, dv.y, acch[k].y);
      acch[k].z = fmaf(z.z, dv.z, acch[k].z); acch[k].w = fmaf(z.w, dv.w, acch[k].w);
      accb[k].x += dv.x * hv[k].x * z.x * (1.f - z.x); accb[k].y += dv.y * hv[k].y * z.y * (1.f - z.y);
      accb[k].z += dv.z * hv[k].z * z.z * (1.f - z.z); accb[k].w += dv.w * hv[k].w * z.w * (1.f - z.w);
    }
  }
  float4* obp = reinterpret_cast<float4*>(db + static_cast<int64_t>(row) * g.ld);
  float4* ohp = reinterpret_cast<float4*>(dh + static_cast<int64_t>(row) * g.ld);
#pragma unroll
  for (int k = 0; k < CH; ++k) {
    if (!on[k]) continue;
    obp[k * 64 + lane] = accb[k];
    ohp[k * 64 + lane] = acch[k];
  }
}

int gated_check(const char* who, const agnn_gated_t* g) {
  using namespace agnn;
  if (!g) return fail(AGNN_EINVAL, "%s: null descriptor", who);
  if (g->n_rows < 0 || g->n_rows >= (int64_t{1} << 31)) return fail(AGNN_EINVAL, "%s: n_rows=%lld", who, (long long)g->n_rows);
  if (g->H <= 0 || (g->H & 3) || g->H > 1024) return fail(AGNN_EINVAL, "%s: H=%d must be a multiple of 4 in [4,1024]", who, g->H);
  if (g->n_rows == 0) return 1;
  if (!g->rowptr || !g->a || !g->b || !g->h) return fail(AGNN_EINVAL, "%s: null argument", who);
  if (!aligned16(g->a) || !aligned16(g->b) || !aligned16(g->h) || (g->ld & 3) || g->ld < g->H) return fail(AGNN_EALIGN, "%s: a/b/h misaligned or ld < H", who);
  if (g->c && (!aligned16(g->c) || (g->ld_c & 3) || g->ld_c < g->H || !g->perm)) return fail(AGNN_EALIGN, "%s: per-edge term misaligned or perm missing", who);
  return 0;
}

inline GatedArgs to_args(const agnn_gated_t* g) {
  return GatedArgs{g->rowptr, g->col, g->perm, g->a, g->b, g->h, g->c, g->ld, g->ld_c, static_cast<int32_t>(g->n_rows), g->H};
}
inline unsigned ggrid(int64_t n_rows) { return static_cast<unsigned>((((n_rows + 3) / 4) + 7) & ~int64_t{7}); }

#define AGNN_GATED_LAUNCH(KERN, ...)                                                                          \
  do {                                                                                                        \
    const dim3 grid(ggrid(g->n_rows)), block(256);                                                            \
    hipStream_t s = static_cast<hipStream_t>(stream_);                                                        \
    if (g->H <= 256) hipLaunchKernelGGL(KERN<1>, grid, block, 0, s, to_args(g), __VA_ARGS__);                 \
    else if (g->H <= 512) hipLaunchKernelGGL(KERN<2>, grid, block, 0, s, to_args(g), __VA_ARGS__);            \
    else hipLaunchKernelGGL(KERN<4>, grid, block, 0, s, to_args(g), __VA_ARGS__);                             \
  } while (0)

}  // namespace

extern "C" int agnn_gated_fwd_f32(const agnn_gated_t* g, float* out, int64_t ld_out, agnn_stream_t stream_) {
  using namespace agnn;
  if (int rc = gated_check("gated_fwd", g)) return rc > 0 ? AGNN_OK : rc;
  if (!out || !aligned16(out) || (ld_out & 3)) return fail(AGNN_EALIGN, "gated_fwd: out misaligned");
  AGNN_GATED_LAUNCH(k_gated_fwd, out, ld_out);
  return check_launch("gated_fwd");
}

extern "C" int agnn_gated_bwd_dst_f32(const agnn_gated_t* g, const float* ds, int64_t ld_ds, float* da, float* dc,
                                      agnn_stream_t stream_) {
  using namespace agnn;
  if (int rc = gated_check("gated_bwd_dst", g)) return rc > 0 ? AGNN_OK : rc;
  if (!ds || !da || !aligned16(ds) || !aligned16(da) || (ld_ds & 3) || !g->perm) return fail(AGNN_EALIGN, "gated_bwd_dst: ds/da misaligned or perm missing");
  AGNN_GATED_LAUNCH(k_gated_bwd_dst, ds, ld_ds, da, dc);
  return check_launch("gated_bwd_dst");
}

extern "C" int agnn_gated_bwd_src_f32(const agnn_gated_t* g, const float* ds, int64_t ld_ds, float* db, float* dh,
                                      agnn_stream_t stream_) {
  using namespace agnn;
  if (int rc = gated_check("gated_bwd_src", g)) return rc > 0 ? AGNN_OK : rc;
  if (!ds || !db || !dh || !aligned16(ds) || !aligned16(db) || !aligned16(dh) || (ld_ds & 3) || !g->perm) return fail(AGNN_EALIGN, "gated_bwd_src: misaligned or perm missing");
  AGNN_GATED_LAUNCH(k_gated_bwd_src, ds, ld_ds, db, dh);
  return check_launch("gated_bwd_src");
}
