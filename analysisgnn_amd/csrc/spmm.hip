// Multi-relation segmented gather-reduce ("hetero SpMM") for gfx950.
//
// One 64-lane wavefront owns one output row; a lane owns 4 consecutive floats of every
// 256-float chunk of the row, so a 1 KiB row (H = 256) is ONE global_load_dwordx4 per neighbour
// and one global_store_dwordx4 per relation slot.  Everything wave-uniform (row, rowptr, column
// ids, loop control, row base addresses) is kept on the scalar unit; blocks are renumbered so that
// each XCD works on a contiguous slab of rows (neighbours are close in index -> L2 hits; measured:
// FETCH_SIZE = the source matrix once).  Sums run in CSR order: bitwise reproducible, no atomics.
// HBM/L2-bound integer + fp32-add work: no MFMA here on purpose.
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

// ------------------------------------------------------------------------------------------
// Generic kernel ("scalarised"): one wavefront per output row, and EVERYTHING that is the same for the 64
// lanes — row number, rowptr/rowend, column ids, edge weights, loop control, neighbour-row base
// addresses — lives on the scalar unit (s_load_* through the constant address space, SALU
// arithmetic).  The vector unit only issues what moves bytes: one global_load_dwordx4 per
// neighbour row chunk (scalar base + lane offset), 4 FMAs, one global_store_dwordx4 per slot.
// Handles every option of the C-ABI (trimming, edge weights, self-loop / column filters, any H % 4 == 0).
// History and measurements of the earlier variants: profiles/r01_spmm_kernel_study.md.
// ------------------------------------------------------------------------------------------
typedef const __attribute__((address_space(4))) int32_t* k_i32p;
typedef const __attribute__((address_space(4))) float* k_f32p;

template <int CH, bool SHARED, bool SELF>
__global__ __launch_bounds__(256) void k_spmm_s(RelTable t, SpmmArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int vb = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);   // XCD-contiguous row slabs
  const int row = vb * 4 + wave;                                            // wave-uniform (SGPR)
  if (row >= a.n_rows) return;
  const bool mean = (a.flags & AGNN_SPMM_MEAN) != 0;
  const bool skip_self = (a.flags & AGNN_SPMM_SKIP_SELF) != 0;
  const bool accum = (a.flags & AGNN_SPMM_ACCUM) != 0;
  bool lane_on[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) lane_on[c] = (c * 256 + lane * 4) < a.H;

  float4 selfv[SELF ? CH : 1];
  if (SELF) {
    const float4* sp = reinterpret_cast<const float4*>(a.self + static_cast<int64_t>(row) * a.ld_self);
#pragma unroll
    for (int c = 0; c < (SELF ? CH : 1); ++c) selfv[c] = lane_on[c] ? sp[c * 64 + lane] : f4_zero();
  }
  float4 tot[SHARED ? CH : 1];
#pragma unroll
  for (int c = 0; c < (SHARED ? CH : 1); ++c) tot[c] = f4_zero();

  for (int r = 0; r < t.n_rel; ++r) {
    const agnn_rel_t& R = t.r[r];
    const k_i32p rowptr = (k_i32p)R.rowptr;
    const k_i32p col = (k_i32p)R.col;
    const k_f32p ew = (k_f32p)R.ew;
    const k_f32p cs = (k_f32p)R.colscale;
    const float* src = R.src;
    const int64_t ld = R.ld_src;
    const int start = rowptr[row];
    const int end = (R.rowend != nullptr) ? ((k_i32p)R.rowend)[row] : rowptr[row + 1];
    float4 acc[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) acc[c] = f4_zero();
    int cnt = 0;
    for (int p = start; p < end; p += 4) {
      int ck[4];
      float wk[4];
      bool ok[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        ok[u] = false;
        ck[u] = 0;
        wk[u] = 1.f;
        if (p + u < end) {
          const int c0 = col[p + u];
          ok[u] = !(skip_self && c0 == row) && (c0 < a.col_limit) && (c0 >= 0);
          if (ok[u]) {
            ck[u] = c0;
            if (ew != nullptr) wk[u] = ew[p + u];
            if (cs != nullptr) wk[u] *= cs[c0];
          }
        }
      }
      float4 v[4][CH];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (ok[u]) {
          const float4* sp = reinterpret_cast<const float4*>(src + static_cast<int64_t>(ck[u]) * ld);
#pragma unroll
          for (int c = 0; c < CH; ++c) v[u][c] = lane_on[c] ? sp[c * 64 + lane] : f4_zero();
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (ok[u]) {
          ++cnt;
#pragma unroll
          for (int c = 0; c < CH; ++c) f4_fma(acc[c], wk[u], v[u][c]);
        }
      }
    }
    const float denom = static_cast<float>(cnt > 1 ? cnt : 1);
    if (a.inv_cnt != nullptr && lane == 0) a.inv_cnt[static_cast<int64_t>(r) * a.n_rows + row] = 1.f / denom;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      if (SELF) f4_add(acc[c], selfv[SELF ? c : 0]);
      if (mean) f4_div(acc[c], denom);
    }
    if (SHARED) {
#pragma unroll
      for (int c = 0; c < CH; ++c) f4_add(tot[SHARED ? c : 0], acc[c]);
    } else {
      float4* op = reinterpret_cast<float4*>(a.out + static_cast<int64_t>(row) * a.ld_out + static_cast<int64_t>(r) * a.rel_stride);
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        if (!lane_on[c]) continue;
        float4 o = acc[c];
        if (accum) f4_add(o, op[c * 64 + lane]);
        op[c * 64 + lane] = o;
      }
    }
  }
  if (SHARED) {
    float4* op = reinterpret_cast<float4*>(a.out + static_cast<int64_t>(row) * a.ld_out);
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      if (!lane_on[c]) continue;
      float4 o = tot[SHARED ? c : 0];
      if (accum) f4_add(o, op[c * 64 + lane]);
      op[c * 64 + lane] = o;
    }
  }
}


// ------------------------------------------------------------------------------------------
// Fast path.  PMC on the earlier variants (profiles/r01_spmm_kernel_study.md): ~1000 instructions per row,
// SQ_ACTIVE_INST_ANY ~ the whole kernel time — the aggregation was INSTRUCTION-ISSUE bound (rows have
// ~1.6 neighbours per relation), not bandwidth bound.  This kernel spends ~1/4 of the instructions:
//   * relations are processed four at a time with all four `rowptr` pairs fetched by back-to-back
//     scalar loads and all four column-id vectors by back-to-back coalesced vector loads, so a row
//     costs three dependent memory round trips, not 3R;
//   * column ids / weights of a segment sit in one VGPR each and are broadcast with v_readlane;
//     neighbour rows are fetched as scalar base + lane offset, two in flight per segment;
//   * no per-edge predicates: this path is taken only when there is no trimming (rowend), no
//     per-edge weight, no self-loop / column filter and H is exactly 256 or 512 (host-side check);
//   * 1/count is one v_rcp_f32 (<= 1 ulp) and four multiplies instead of four IEEE divisions.
// Everything else falls back to k_spmm_s above (same results up to that ulp).
// ------------------------------------------------------------------------------------------
template <int CH, bool HAS_CS, bool SHARED, bool SELF>
__global__ __launch_bounds__(256) void k_spmm_fast(RelTable t, SpmmArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int vb = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);   // XCD-contiguous row slabs
  const int row = vb * 4 + wave;
  if (row >= a.n_rows) return;
  const bool mean = (a.flags & AGNN_SPMM_MEAN) != 0;
  const bool accum = (a.flags & AGNN_SPMM_ACCUM) != 0;
  const uint32_t loff = static_cast<uint32_t>(lane) * 16u;

  float4 selfv[SELF ? CH : 1];
  if (SELF) {
    const char* sp = reinterpret_cast<const char*>(a.self + static_cast<int64_t>(row) * a.ld_self);
#pragma unroll
    for (int c = 0; c < (SELF ? CH : 1); ++c) selfv[c] = *reinterpret_cast<const float4*>(sp + (loff + c * 1024u));
  }
  float4 tot[SHARED ? CH : 1];
#pragma unroll
  for (int c = 0; c < (SHARED ? CH : 1); ++c) tot[c] = f4_zero();

  for (int r0 = 0; r0 < t.n_rel; r0 += 4) {
    int start[4], n[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      start[u] = 0;
      n[u] = 0;
      if (r0 + u < t.n_rel) {
        const k_i32p rp = (k_i32p)t.r[r0 + u].rowptr;
        start[u] = rp[row];
        n[u] = rp[row + 1] - start[u];
      }
    }
    int colv[4];
    float wv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      colv[u] = 0;
      if (n[u] > 0 && lane < n[u]) colv[u] = t.r[r0 + u].col[start[u] + lane];
    }
    if (HAS_CS) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        wv[u] = 0.f;
        if (n[u] > 0 && lane < n[u]) wv[u] = t.r[r0 + u].colscale[colv[u]];
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (r0 + u >= t.n_rel) break;
      const agnn_rel_t& R = t.r[r0 + u];
      const char* src = reinterpret_cast<const char*>(R.src);
      const int64_t ldb = R.ld_src * 4;
      float4 acc[CH];
#pragma unroll
      for (int c = 0; c < CH; ++c) acc[c] = f4_zero();
      const int nn = n[u];
      for (int base = 0; base < nn; base += 64) {
        int cv = colv[u];
        float wq = HAS_CS ? wv[u] : 1.f;
        if (base > 0) {                       // rows with more than 64 neighbours: next batch of ids
          cv = 0;
          wq = 0.f;
          if (lane < nn - base) {
            cv = R.col[start[u] + base + lane];
            if (HAS_CS) wq = R.colscale[cv];
          }
        }
        const int m = (nn - base) < 64 ? (nn - base) : 64;
        for (int k = 0; k < m; k += 2) {
          const bool two = k + 1 < m;
          const int c0 = __builtin_amdgcn_readlane(cv, k);
          const int c1 = __builtin_amdgcn_readlane(cv, two ? k + 1 : k);
          float w0 = 1.f, w1 = 1.f;
          if (HAS_CS) {
            w0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wq), k));
            w1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wq), two ? k + 1 : k));
          }
          const char* b0 = src + static_cast<int64_t>(c0) * ldb;
          const char* b1 = src + static_cast<int64_t>(c1) * ldb;
          float4 v0[CH], v1[CH];
#pragma unroll
          for (int c = 0; c < CH; ++c) v0[c] = *reinterpret_cast<const float4*>(b0 + (loff + c * 1024u));
          if (two) {
#pragma unroll
            for (int c = 0; c < CH; ++c) v1[c] = *reinterpret_cast<const float4*>(b1 + (loff + c * 1024u));
          }
#pragma unroll
          for (int c = 0; c < CH; ++c) f4_fma(acc[c], w0, v0[c]);
          if (two) {
#pragma unroll
            for (int c = 0; c < CH; ++c) f4_fma(acc[c], w1, v1[c]);
          }
        }
      }
      const float inv = __builtin_amdgcn_rcpf(static_cast<float>(nn > 1 ? nn : 1));
      if (a.inv_cnt != nullptr && lane == 0) a.inv_cnt[static_cast<int64_t>(r0 + u) * a.n_rows + row] = inv;
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        if (SELF) f4_add(acc[c], selfv[SELF ? c : 0]);
        if (mean) { acc[c].x *= inv; acc[c].y *= inv; acc[c].z *= inv; acc[c].w *= inv; }
      }
      if (SHARED) {
#pragma unroll
        for (int c = 0; c < CH; ++c) f4_add(tot[SHARED ? c : 0], acc[c]);
      } else {
        char* op = reinterpret_cast<char*>(a.out + static_cast<int64_t>(row) * a.ld_out + static_cast<int64_t>(r0 + u) * a.rel_stride);
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          float4 o = acc[c];
          float4* q = reinterpret_cast<float4*>(op + (loff + c * 1024u));
          if (accum) f4_add(o, *q);
          *q = o;
        }
      }
    }
  }
  if (SHARED) {
    char* op = reinterpret_cast<char*>(a.out + static_cast<int64_t>(row) * a.ld_out);
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      float4 o = tot[SHARED ? c : 0];
      float4* q = reinterpret_cast<float4*>(op + (loff + c * 1024u));
      if (accum) f4_add(o, *q);
      *q = o;
    }
  }
}

template <int CH>
void launch_fast(dim3 grid, hipStream_t stream, const RelTable& t, const SpmmArgs& a, bool has_cs) {
  const bool shared = a.rel_stride == 0, self = a.self != nullptr;
#define AGNN_FAST(CS, SH, SE) hipLaunchKernelGGL((k_spmm_fast<CH, CS, SH, SE>), grid, dim3(256), 0, stream, t, a)
  if (has_cs) {
    if (shared && self) AGNN_FAST(true, true, true); else if (shared) AGNN_FAST(true, true, false);
    else if (self) AGNN_FAST(true, false, true); else AGNN_FAST(true, false, false);
  } else {
    if (shared && self) AGNN_FAST(false, true, true); else if (shared) AGNN_FAST(false, true, false);
    else if (self) AGNN_FAST(false, false, true); else AGNN_FAST(false, false, false);
  }
#undef AGNN_FAST
}

template <int CH>
void launch_s(dim3 grid, hipStream_t stream, const RelTable& t, const SpmmArgs& a) {
  const bool shared = a.rel_stride == 0, self = a.self != nullptr;
  if (shared && self) hipLaunchKernelGGL((k_spmm_s<CH, true, true>), grid, dim3(256), 0, stream, t, a);
  else if (shared) hipLaunchKernelGGL((k_spmm_s<CH, true, false>), grid, dim3(256), 0, stream, t, a);
  else if (self) hipLaunchKernelGGL((k_spmm_s<CH, false, true>), grid, dim3(256), 0, stream, t, a);
  else hipLaunchKernelGGL((k_spmm_s<CH, false, false>), grid, dim3(256), 0, stream, t, a);
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
  int64_t blocks = ((n_rows + 3) / 4 + 7) & ~int64_t{7};   // multiple of 8: the XCD remap is a bijection
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (!(flags & AGNN_SPMM_GENERIC) && (H == 256 || H == 512) && !(flags & AGNN_SPMM_SKIP_SELF) &&
      col_limit == INT32_MAX) {
    bool plain = true, any_cs = false, all_cs = true;
    for (int r = 0; r < n_rel; ++r) {
      plain = plain && rels[r].rowend == nullptr && rels[r].ew == nullptr && rels[r].src != nullptr && rels[r].col != nullptr;
      any_cs = any_cs || rels[r].colscale != nullptr;
      all_cs = all_cs && rels[r].colscale != nullptr;
    }
    if (plain && any_cs == all_cs) {
      if (H == 256) launch_fast<1>(dim3(blocks), stream, t, a, all_cs);
      else launch_fast<2>(dim3(blocks), stream, t, a, all_cs);
      return check_launch("spmm(fast)");
    }
  }
  if (H <= 256) launch_s<1>(dim3(blocks), stream, t, a);
  else if (H <= 512) launch_s<2>(dim3(blocks), stream, t, a);
  else launch_s<4>(dim3(blocks), stream, t, a);
  return check_launch("spmm");
}
