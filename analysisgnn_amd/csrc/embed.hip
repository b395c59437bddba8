// Note-input assembly of `TorchAnalysisGNN.encode`:
//     z = cat([x_note, pitch_embedding(pitch_spelling), key_embedding(key_signature)], dim=-1)
// (reference analysisgnn/models/analysis.py:399-400 tables, :574 the cat) and the gradient of the small tables.
//
// Forward: one launch writes the [N, ld_out] operand of the input projection — feature columns copied, one table row
// per index appended per table, the columns up to ld_out zero-filled (rows are padded to 16 bytes for the GEMMs).
// It replaces two index_select launches, a fill and a cat.
//
// Backward: dTable[v, :] = sum over rows n with idx[n] == v of dOut[n, table's columns].  The tables have 35 and 15
// rows, so no sort and no atomics: a workgroup owns one table row v and one slice of the N rows; its waves read 64
// indices at a time (coalesced), ballot the matches and add the matching dOut rows in row order (lane = column).  The
// per-wave partial sums go to slabs that `launch_slab_reduce` (wgrad.hip) adds in a fixed order: bitwise
// reproducible.  It replaces onehot(idx)^T @ dOut (arange, compare, two copies, split-N GEMM, reduce — per table).
// Byte-bound work (N x D x 4 B read once), no MFMA on purpose.
#include <cstdlib>

#include "agnn_common.h"

namespace {

constexpr int kMaxTab = AGNN_EMBED_MAX_TABLES;
constexpr int kSlices = 128;        // row slices; x 4 waves = 512 slabs (a wave's rows: 32 at N = 16 000)

struct EmbFwd {
  const float* x;
  int64_t ld_x;
  int32_t in_x, n_tab, dim;
  int64_t n_rows;
  const int64_t* idx[kMaxTab];
  const float* tab[kMaxTab];
  int32_t vocab[kMaxTab];
  float* out;
  int64_t ld_out;
};

__global__ __launch_bounds__(256) void k_embed_cat_fwd(EmbFwd a) {
  const int64_t total = a.n_rows * a.ld_out;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; e < total; e += static_cast<int64_t>(gridDim.x) * 256) {
    const int64_t row = e / a.ld_out;
    const int col = static_cast<int>(e - row * a.ld_out);
    float v = 0.f;
    if (col < a.in_x) {
      v = a.x[row * a.ld_x + col];
    } else {
      const int c = col - a.in_x;
      const int t = c / a.dim;
      if (t < a.n_tab) {
        int64_t id = a.idx[t][row];
        id = id < 0 ? 0 : (id >= a.vocab[t] ? a.vocab[t] - 1 : id);     // out-of-range ids are clamped, never read outside the table
        v = a.tab[t][id * a.dim + (c - t * a.dim)];
      }
    }
    a.out[e] = v;
  }
}

struct EmbBwd {
  const float* dout;
  int64_t ld_dout;
  int32_t col0, n_tab, dim, dim_pad, v_total, v_pad;
  int64_t n_rows;
  const int64_t* idx[kMaxTab];
  int32_t vbase[kMaxTab + 1];       // first global table row of each table
  float* slab;                      // [kSlices * 4][v_pad][dim_pad]
};

__global__ __launch_bounds__(256) void k_embed_bwd(EmbBwd a) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int vg = blockIdx.x;                       // global table row
  int t = 0;
  while (t + 1 < a.n_tab && vg >= a.vbase[t + 1]) ++t;
  const int64_t v = vg - a.vbase[t];
  const int64_t* idx = a.idx[t];
  const int col_base = a.col0 + t * a.dim;
  const int part = blockIdx.y * 4 + wave;          // this wave's slab
  const int64_t per = ((a.n_rows + kSlices * 4 - 1) / (kSlices * 4) + 63) & ~int64_t{63};
  const int64_t r0 = part * per;
  int64_t r1 = r0 + per;
  if (r1 > a.n_rows) r1 = a.n_rows;
  float* dst = a.slab + (static_cast<int64_t>(part) * a.v_pad + vg) * a.dim_pad;
  for (int c0 = 0; c0 < a.dim; c0 += 64) {
    const bool on = c0 + lane < a.dim;
    float acc = 0.f;
    for (int64_t base = r0; base < r1; base += 64) {
      const int64_t r = base + lane;
      const bool hit = r < r1 && idx[r] == v;
      uint64_t m = __ballot(hit);
      while (m) {                                  // wave-uniform loop over the matching rows, in row order
        const int b = __builtin_ctzll(m);
        m &= m - 1;
        if (on) acc += a.dout[(base + b) * a.ld_dout + col_base + c0 + lane];
      }
    }
    if (on) dst[c0 + lane] = acc;
  }
}

// The same sums with every index read ONCE: a wave walks its slice of the rows (all tables' columns of a row side by side
// in the lanes, 256 contiguous bytes per load, eight rows' loads in flight) and adds each value into its own LDS image of
// all the tables (lane = column: no two lanes share an address; rows in order: bitwise reproducible), then writes the
// image to its slab.  The kernel above scans the whole index vector once per TABLE ROW and fetches the matching rows one
// dependent load at a time: 35 - 48 us per launch on the serial tail of the step (profiles/r02_step_stamps.md), this one 30.
constexpr int kEmbLdsFloats = 4096;     // per wave: v_total * dim floats

template <int NCG>                       // 64-column groups of the tables' columns side by side (host: ceil(n_tab * dim / 64))
__global__ __launch_bounds__(256) void k_embed_bwd_rows(EmbBwd a) {
  __shared__ float s_acc[4][kEmbLdsFloats];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* acc = s_acc[wave];
  const int img = a.v_total * a.dim;
  for (int e = lane; e < img; e += 64) acc[e] = 0.f;
  const int part = blockIdx.x * 4 + wave;
  const int64_t per = (a.n_rows + kSlices * 4 - 1) / (kSlices * 4);
  const int64_t r0 = part * per;
  int64_t r1 = r0 + per;
  if (r1 > a.n_rows) r1 = a.n_rows;
  const int width = a.n_tab * a.dim;
  // this lane's columns 64 g + lane: their table, its rows in the image, the column inside the table
  bool on[NCG];
  int cc[NCG], vb[NCG], vn[NCG], col[NCG];
  const int64_t* idx[NCG];
#pragma unroll
  for (int g = 0; g < NCG; ++g) {
    const int c = 64 * g + lane;
    on[g] = c < width;
    const int t = on[g] ? c / a.dim : 0;
    cc[g] = c - t * a.dim;
    idx[g] = a.idx[t];
    vb[g] = a.vbase[t];
    vn[g] = a.vbase[t + 1] - vb[g];
    col[g] = a.col0 + (on[g] ? c : 0);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  constexpr int RB = NCG <= 2 ? 8 : 4;                   // rows per trip: 2 * NCG * RB loads in flight
  for (int64_t base = r0; base < r1; base += RB) {
    float val[RB][NCG];
    int64_t id[RB][NCG];
#pragma unroll
    for (int u = 0; u < RB; ++u) {
      const int64_t r = base + u < r1 ? base + u : r1 - 1;
#pragma unroll
      for (int g = 0; g < NCG; ++g) {
        id[u][g] = idx[g][r];
        val[u][g] = a.dout[r * a.ld_dout + col[g]];
      }
    }
#pragma unroll
    for (int u = 0; u < RB; ++u)                           // rows in order; an id outside the table contributes nothing, as above
#pragma unroll
      for (int g = 0; g < NCG; ++g)
        if (on[g] && base + u < r1 && id[u][g] >= 0 && id[u][g] < vn[g]) acc[(vb[g] + static_cast<int>(id[u][g])) * a.dim + cc[g]] += val[u][g];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  float* dst = a.slab + static_cast<int64_t>(part) * a.v_pad * a.dim_pad;
  for (int e = lane; e < img; e += 64) dst[e] = acc[e];    // (dim_pad == dim: the slab rows are the image rows)
}

}  // namespace

extern "C" size_t agnn_embed_workspace_bytes(int32_t n_tab, const int32_t* vocab, int32_t dim) {
  if (n_tab <= 0 || n_tab > kMaxTab || !vocab || dim <= 0) return 0;
  int64_t v_total = 0;
  for (int t = 0; t < n_tab; ++t) v_total += vocab[t];
  return static_cast<size_t>(kSlices * 4) * static_cast<size_t>(v_total) * static_cast<size_t>((dim + 1) & ~1) * sizeof(float) + 256;
}

extern "C" int agnn_embed_cat_fwd_f32(const float* x, int64_t ld_x, int32_t in_x, int64_t n_rows, int32_t n_tab,
                                      const int64_t* const* idx, const float* const* tables, const int32_t* vocab,
                                      int32_t dim, float* out, int64_t ld_out, agnn_stream_t stream_) {
  using namespace agnn;
  if (n_tab < 0 || n_tab > kMaxTab) return fail(AGNN_EINVAL, "embed_cat: n_tab=%d not in [0,%d]", n_tab, kMaxTab);
  if (n_rows < 0 || in_x < 0 || dim <= 0) return fail(AGNN_EINVAL, "embed_cat: bad sizes n_rows=%lld in_x=%d dim=%d", (long long)n_rows, in_x, dim);
  if (ld_out < static_cast<int64_t>(in_x) + static_cast<int64_t>(n_tab) * dim || (in_x > 0 && ld_x < in_x))
    return fail(AGNN_EINVAL, "embed_cat: leading dimension smaller than the row");
  if (n_rows == 0) return AGNN_OK;
  if (!out || (in_x > 0 && !x) || (n_tab > 0 && (!idx || !tables || !vocab))) return fail(AGNN_EINVAL, "embed_cat: null argument");
  EmbFwd a{};
  a.x = x; a.ld_x = ld_x; a.in_x = in_x; a.n_tab = n_tab; a.dim = dim; a.n_rows = n_rows; a.out = out; a.ld_out = ld_out;
  for (int t = 0; t < n_tab; ++t) {
    if (!idx[t] || !tables[t] || vocab[t] <= 0) return fail(AGNN_EINVAL, "embed_cat: table %d: null pointer or empty table", t);
    a.idx[t] = idx[t]; a.tab[t] = tables[t]; a.vocab[t] = vocab[t];
  }
  const int64_t total = n_rows * ld_out;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(k_embed_cat_fwd, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, static_cast<hipStream_t>(stream_), a);
  return check_launch("embed_cat_fwd");
}

extern "C" int agnn_embed_cat_bwd_f32(const float* dout, int64_t ld_dout, int32_t col0, int64_t n_rows, int32_t n_tab,
                                      const int64_t* const* idx, const int32_t* vocab, int32_t dim, float* dtables,
                                      void* workspace, size_t workspace_bytes, agnn_stream_t stream_) {
  using namespace agnn;
  if (n_tab <= 0 || n_tab > kMaxTab) return fail(AGNN_EINVAL, "embed_bwd: n_tab=%d not in [1,%d]", n_tab, kMaxTab);
  if (n_rows < 0 || col0 < 0 || dim <= 0 || (dim & 1)) return fail(AGNN_EINVAL, "embed_bwd: bad sizes n_rows=%lld col0=%d dim=%d (dim must be even)", (long long)n_rows, col0, dim);
  if (!dout || !idx || !vocab || !dtables || !workspace) return fail(AGNN_EINVAL, "embed_bwd: null argument");
  if (ld_dout < static_cast<int64_t>(col0) + static_cast<int64_t>(n_tab) * dim) return fail(AGNN_EINVAL, "embed_bwd: ld_dout smaller than the row");
  if (reinterpret_cast<uintptr_t>(dtables) & 7u) return fail(AGNN_EALIGN, "embed_bwd: dtables must be 8-byte aligned");
  const size_t need = agnn_embed_workspace_bytes(n_tab, vocab, dim);
  if (need == 0 || workspace_bytes < need) return fail(AGNN_ENOMEM, "embed_bwd: workspace %zu < %zu bytes", workspace_bytes, need);
  EmbBwd a{};
  a.dout = dout; a.ld_dout = ld_dout; a.col0 = col0; a.n_tab = n_tab; a.dim = dim; a.dim_pad = dim; a.n_rows = n_rows;
  int vt = 0;
  for (int t = 0; t < n_tab; ++t) {
    if (!idx[t] || vocab[t] <= 0) return fail(AGNN_EINVAL, "embed_bwd: table %d: null index pointer or empty table", t);
    a.idx[t] = idx[t];
    a.vbase[t] = vt;
    vt += vocab[t];
  }
  a.vbase[n_tab] = vt;
  a.v_total = vt; a.v_pad = vt;
  a.slab = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~uintptr_t{255});
  hipStream_t s = static_cast<hipStream_t>(stream_);
  const int ncg = (n_tab * dim + 63) / 64;
  if (static_cast<int64_t>(vt) * dim <= kEmbLdsFloats && ncg <= 4) {
    if (ncg == 1) hipLaunchKernelGGL(k_embed_bwd_rows<1>, dim3(kSlices), dim3(256), 0, s, a);
    else if (ncg == 2) hipLaunchKernelGGL(k_embed_bwd_rows<2>, dim3(kSlices), dim3(256), 0, s, a);
    else if (ncg == 3) hipLaunchKernelGGL(k_embed_bwd_rows<3>, dim3(kSlices), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(k_embed_bwd_rows<4>, dim3(kSlices), dim3(256), 0, s, a);
    if (int rc = check_launch("embed_bwd(rows)")) return rc;
  } else {
    hipLaunchKernelGGL(k_embed_bwd, dim3(vt, kSlices), dim3(256), 0, s, a);
    if (int rc = check_launch("embed_bwd")) return rc;
  }
  return launch_slab_reduce(a.slab, nullptr, kSlices * 4, vt, dim, vt, dim, dtables, dim, nullptr, s);
}
