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
  int32_t self_rows;   // rows of `self`; rows beyond take no self term (AGNN_SPMM_ROOT: the root slot may be shorter than the output)
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


// ------------------------------------------------------------------------------------------
// Fast path, straight-line index phase.  PMC on k_spmm_fast (profiles/r01_spmm_kernel_study.md, second part): at
// full occupancy (8 waves / SIMD) a wave lives ~6 us, 55 % of it in s_waitcnt, and the launch takes two such
// rounds — the kernel is bound by the LENGTH OF THE DEPENDENT-LOAD CHAIN of one row, not by bytes.  The chain in
// k_spmm_fast: per relation {pointer from the kernel-argument table -> rowptr[row] -> rowptr[row+1]} one after the
// other, then the column ids, then per relation {gather -> store}, with every later gather also waiting for the
// previous relation's store (vmcnt counts loads and stores in issue order).  Here:
//   * the table is a structure of arrays, so the four rowptr / rowend / col / src pointers come with one scalar load
//     each, and entries beyond n_rel repeat the last relation: no branch anywhere in the index phase;
//   * a row's end comes from its own pointer: the relation's `rowend` array when the batch is trimmed (PyG
//     trim_to_layer, what every training step of the reference passes: models/analysis.py:960-961, models/cadence.py:
//     167-173), `rowptr + 1` otherwise — the same two scalar loads either way, no branch;
//   * all four rowptr / rowend pairs are one batch of scalar loads, all four column-id vectors one batch of vector
//     loads, the first two neighbour rows of all four relations (8 loads; rows have ~1.6 neighbours per relation) one
//     batch, and the 1/deg column scales of the backward pass travel with that batch;
//   * nothing is stored before every gather of the row has been issued.
// A row therefore costs rowptr -> col -> gather.  Loads are unconditional: lanes beyond the neighbour count repeat
// the last neighbour (an L1 hit) or, for an empty segment, row 0, and what they fetch is dropped by a per-lane
// select (not a multiply by zero: a non-finite value must not leak).  Neighbours beyond the second (rare) take the
// two-at-a-time loop of k_spmm_fast.
// FILT adds the two per-edge predicates of the C-ABI (column < col_limit: the backward pass of a trimmed layer;
// column != row: the onset pooling, models/analysis.py:581-584).  They are evaluated once per column-id vector and kept
// as a 64-bit wave mask in SGPRs: a neighbour's "keep" bit is a scalar shift, the valid count one s_bcnt1.
// One row per wave: consecutive rows sharing one index phase measured slower (see launch_fast).
// ------------------------------------------------------------------------------------------
struct FastTable {
  const int32_t* rowptr[AGNN_MAX_SEG + 3];
  const int32_t* rowend[AGNN_MAX_SEG + 3];     // per-row ends: the relation's rowend array, or rowptr + 1
  const int32_t* col[AGNN_MAX_SEG + 3];
  const float* src[AGNN_MAX_SEG + 3];
  const float* colscale[AGNN_MAX_SEG + 3];
  uint32_t ldb[AGNN_MAX_SEG + 4];     // row stride in bytes (< 4 GiB, host-checked)
  int n_rel;
};

__device__ __forceinline__ int to_vgpr(int s) {
  int v;
  asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "s"(s));
  return v;
}
__device__ __forceinline__ float4 f4_keep(bool keep, const float4& v) {
  return make_float4(keep ? v.x : 0.f, keep ? v.y : 0.f, keep ? v.z : 0.f, keep ? v.w : 0.f);
}

template <int CH, bool HAS_CS, bool SHARED, bool SELF, bool FILT>
__global__ __launch_bounds__(256) void k_spmm_fast7(FastTable t, SpmmArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int vb = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);   // XCD-contiguous row slabs
  const int row = vb * 4 + wave;
  if (row >= a.n_rows) return;
  const bool mean = (a.flags & AGNN_SPMM_MEAN) != 0;
  const bool skip_self = FILT && (a.flags & AGNN_SPMM_SKIP_SELF) != 0;
  const uint32_t loff = static_cast<uint32_t>(lane) * 16u;

  // `self` is either the torch_scatter numerator term (added to every relation before the scale) or, with AGNN_SPMM_ROOT,
  // the root operand of a SAGE layer: an extra output slot behind the relations' (forward) / one more addend of the
  // shared sum for the rows it has (backward).  Uniform per launch: a scalar branch.
  const bool root = SELF && (a.flags & AGNN_SPMM_ROOT) != 0;
  float4 selfv[SELF ? CH : 1];
  if (SELF) {
    const bool has = row < a.self_rows;
    const char* sp = reinterpret_cast<const char*>(a.self + static_cast<int64_t>(has ? row : 0) * a.ld_self);
#pragma unroll
    for (int c = 0; c < (SELF ? CH : 1); ++c) selfv[c] = f4_keep(has, *reinterpret_cast<const float4*>(sp + (loff + c * 1024u)));
  }
  float4 tot[SHARED ? CH : 1];
#pragma unroll
  for (int c = 0; c < (SHARED ? CH : 1); ++c) tot[c] = f4_zero();

  for (int r0 = 0; r0 < t.n_rel; r0 += 4) {
    // ---- index phase: no control flow; the empty asm statements pin "all pointers, then all rowptr / rowend entries,
    //      then all column ids" (left alone, the compiler sinks each scalar load to its first use: one trip per relation)
    k_i32p rpp[4], rep[4];
    const int32_t* cpp[4];
    const char* srcp[4];
    const float* csp[4];
    uint32_t ldbv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      rpp[u] = (k_i32p)t.rowptr[r0 + u];
      rep[u] = (k_i32p)t.rowend[r0 + u];
      cpp[u] = t.col[r0 + u];
      srcp[u] = reinterpret_cast<const char*>(t.src[r0 + u]);
      csp[u] = HAS_CS ? t.colscale[r0 + u] : nullptr;
      ldbv[u] = t.ldb[r0 + u];
    }
    asm volatile("" ::"s"(rpp[0]), "s"(rpp[1]), "s"(rpp[2]), "s"(rpp[3]), "s"(cpp[0]), "s"(cpp[1]), "s"(cpp[2]), "s"(cpp[3]));
    asm volatile("" ::"s"(rep[0]), "s"(rep[1]), "s"(rep[2]), "s"(rep[3]));
    asm volatile("" ::"s"(srcp[0]), "s"(srcp[1]), "s"(srcp[2]), "s"(srcp[3]), "s"(ldbv[0]), "s"(ldbv[1]), "s"(ldbv[2]), "s"(ldbv[3]));
    if (HAS_CS) asm volatile("" ::"s"(csp[0]), "s"(csp[1]), "s"(csp[2]), "s"(csp[3]));
    int beg[4], fin[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      beg[u] = rpp[u][row];
      fin[u] = rep[u][row];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) asm volatile("" ::"s"(beg[u]), "s"(fin[u]));
    int n[4], cntf[4];
    int colv[4];
    float wv[4];
    uint64_t okm[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int live = r0 + u < t.n_rel ? -1 : 0;   // mask, not a branch: the tail entries of the table are valid duplicates
      n[u] = (fin[u] - beg[u]) & live;
      colv[u] = 0;
      if (lane < n[u]) colv[u] = cpp[u][beg[u] + lane];
    }
    if (FILT) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const bool ok = lane < n[u] && colv[u] < a.col_limit && !(skip_self && colv[u] == row);
        okm[u] = __ballot(ok);
        colv[u] = ok ? colv[u] : 0;               // dropped neighbours are fetched from row 0 and discarded
        cntf[u] = 0;
      }
    }
#define AGNN_OVF(u) (n[u] > 64)     /* segment not (entirely) among the 64 ids held in registers */
#define AGNN_BIT(u, k) static_cast<int>((okm[u] >> (k)) & 1u)
    // the first two neighbour rows of every relation
    float4 v0[4][CH], v1[4][CH];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint64_t ldb = ldbv[u];
      int step;                                    // 1 when the segment has >= 2 neighbours, else the second load repeats the first
      asm("s_sub_i32 %0, 1, %1\n\ts_lshr_b32 %0, %0, 31" : "=s"(step) : "s"(n[u]) : "scc");
      const uint32_t c0 = static_cast<uint32_t>(__builtin_amdgcn_readlane(colv[u], 0));
      const uint32_t c1 = static_cast<uint32_t>(__builtin_amdgcn_readlane(colv[u], step));
      const char* b0 = srcp[u] + c0 * ldb;         // column ids are non-negative: one s_mul + one s_mul_hi
      const char* b1 = srcp[u] + c1 * ldb;
#pragma unroll
      for (int c = 0; c < CH; ++c) v0[u][c] = *reinterpret_cast<const float4*>(b0 + (loff + c * 1024u));
#pragma unroll
      for (int c = 0; c < CH; ++c) v1[u][c] = *reinterpret_cast<const float4*>(b1 + (loff + c * 1024u));
      if (HAS_CS) {                                // the 1/deg scales travel with the gathers
        wv[u] = 0.f;
        if (lane < n[u]) wv[u] = csp[u][colv[u]];
      }
      __builtin_amdgcn_sched_barrier(0);           // every gather is issued before the first one is waited for, and
    }                                              // address registers are recycled from pair to pair
    // reduce (in place: v0 becomes the accumulator)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      int k0s, k1s;
      if (FILT) {
        k0s = (AGNN_OVF(u) || n[u] < 1) ? 0 : AGNN_BIT(u, 0);
        k1s = (AGNN_OVF(u) || n[u] < 2) ? 0 : AGNN_BIT(u, 1);
      } else {
        const int nvs = AGNN_OVF(u) ? 0 : n[u];
        k0s = nvs > 0 ? 1 : 0;
        k1s = nvs > 1 ? 1 : 0;
      }
      const int k0v = to_vgpr(k0s), k1v = to_vgpr(k1s);
      float w0 = 1.f, w1 = 1.f;
      if (HAS_CS) {
        w0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wv[u]), 0));
        w1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wv[u]), 1));
      }
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const float4 k0 = f4_keep(k0v != 0, v0[u][c]), k1 = f4_keep(k1v != 0, v1[u][c]);
        if (HAS_CS) {
          v0[u][c] = make_float4(w0 * k0.x, w0 * k0.y, w0 * k0.z, w0 * k0.w);
          f4_fma(v0[u][c], w1, k1);
        } else {
          v0[u][c] = make_float4(k0.x + k1.x, k0.y + k1.y, k0.z + k1.z, k0.w + k1.w);
        }
      }
    }
    // third and later neighbours (and segments whose ids are not in registers), two at a time
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int nn = n[u];
      if (nn > 2 || AGNN_OVF(u)) {
        const uint64_t ldb = ldbv[u];
        const bool in_regs = !AGNN_OVF(u);
        const k_i32p colg = (k_i32p)cpp[u] + beg[u];
        for (int k = in_regs ? 2 : 0; k < nn; k += 2) {
          const bool two = k + 1 < nn;
          const int ka = k, kb = two ? k + 1 : k;
          uint32_t c0, c1;
          float x0 = 1.f, x1 = 1.f;
          int keep0 = 1, keep1 = two ? 1 : 0;
          if (in_regs) {
            c0 = static_cast<uint32_t>(__builtin_amdgcn_readlane(colv[u], ka));
            c1 = static_cast<uint32_t>(__builtin_amdgcn_readlane(colv[u], kb));
            if (HAS_CS) {
              x0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wv[u]), ka));
              x1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wv[u]), kb));
            }
            if (FILT) {
              keep0 = AGNN_BIT(u, ka);
              keep1 &= AGNN_BIT(u, kb);
            }
          } else {                               // > 64 neighbours in this row of this relation: scalar loads
            c0 = static_cast<uint32_t>(colg[ka]);
            c1 = static_cast<uint32_t>(colg[kb]);
            if (FILT) {
              const int i0 = static_cast<int>(c0), i1 = static_cast<int>(c1);
              keep0 = (i0 < a.col_limit && !(skip_self && i0 == row)) ? 1 : 0;
              keep1 &= (i1 < a.col_limit && !(skip_self && i1 == row)) ? 1 : 0;
              c0 = keep0 ? c0 : 0u;
              c1 = keep1 ? c1 : 0u;
              cntf[u] += keep0 + keep1;
            }
            if (HAS_CS) {
              x0 = ((k_f32p)csp[u])[c0];
              x1 = ((k_f32p)csp[u])[c1];
            }
          }
          const char* b0 = srcp[u] + c0 * ldb;
          const char* b1 = srcp[u] + c1 * ldb;
          float4 q0[CH], q1[CH];
#pragma unroll
          for (int c = 0; c < CH; ++c) q0[c] = *reinterpret_cast<const float4*>(b0 + (loff + c * 1024u));
#pragma unroll
          for (int c = 0; c < CH; ++c) q1[c] = *reinterpret_cast<const float4*>(b1 + (loff + c * 1024u));
          const int k1v = to_vgpr(keep1);
          if (FILT) {
            const int k0v = to_vgpr(keep0);
#pragma unroll
            for (int c = 0; c < CH; ++c) {
              f4_fma(v0[u][c], x0, f4_keep(k0v != 0, q0[c]));
              f4_fma(v0[u][c], x1, f4_keep(k1v != 0, q1[c]));
            }
          } else {
#pragma unroll
            for (int c = 0; c < CH; ++c) {
              f4_fma(v0[u][c], x0, q0[c]);
              f4_fma(v0[u][c], x1, f4_keep(k1v != 0, q1[c]));
            }
          }
        }
      }
    }
    // scale and store.  1 / max(count, 1) of the four relations is computed once, in lanes 0..3 (v_rcp_f32, <= 1 ulp),
    // stored by those lanes with one instruction and broadcast back as scalars.
    int nvec = 1;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      int cnt = n[u];
      if (FILT) cnt = __builtin_amdgcn_readfirstlane(AGNN_OVF(u) ? cntf[u] : __builtin_popcountll(okm[u]));
      cnt = cnt > 1 ? cnt : 1;
      asm("v_writelane_b32 %0, %1, %2" : "+v"(nvec) : "s"(cnt), "n"(u));
    }
    const float invv = __builtin_amdgcn_rcpf(static_cast<float>(nvec));
    if (a.inv_cnt != nullptr && lane < 4 && r0 + lane < t.n_rel)
      a.inv_cnt[static_cast<int64_t>(r0 + lane) * a.n_rows + row] = invv;
    const float scalev = mean ? invv : 1.f;      // sums: x * 1.f is exact
    char* op = reinterpret_cast<char*>(a.out + static_cast<int64_t>(row) * a.ld_out + static_cast<int64_t>(r0) * a.rel_stride) + loff;
    const int64_t op_step = a.rel_stride * 4;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (r0 + u >= t.n_rel) break;
      const float inv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(scalev), u));
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        float4& o = v0[u][c];
        if (SELF && !root) f4_add(o, selfv[SELF ? c : 0]);
        o.x *= inv; o.y *= inv; o.z *= inv; o.w *= inv;
      }
      if (SHARED) {
#pragma unroll
        for (int c = 0; c < CH; ++c) f4_add(tot[SHARED ? c : 0], v0[u][c]);
      } else {
#pragma unroll
        for (int c = 0; c < CH; ++c) *reinterpret_cast<float4*>(op + c * 1024) = v0[u][c];
        op += op_step;
      }
    }
#undef AGNN_OVF
#undef AGNN_BIT
  }
  if (SHARED) {
    char* op = reinterpret_cast<char*>(a.out + static_cast<int64_t>(row) * a.ld_out);
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      float4 o = tot[SHARED ? c : 0];
      if (root) f4_add(o, selfv[SELF ? c : 0]);
      *reinterpret_cast<float4*>(op + (loff + c * 1024u)) = o;
    }
  } else if (root) {
    char* op = reinterpret_cast<char*>(a.out + static_cast<int64_t>(row) * a.ld_out + static_cast<int64_t>(t.n_rel) * a.rel_stride);
#pragma unroll
    for (int c = 0; c < CH; ++c) *reinterpret_cast<float4*>(op + (loff + c * 1024u)) = selfv[SELF ? c : 0];
  }
}

template <int CH>
void launch_fast(hipStream_t stream, const RelTable& t, const SpmmArgs& a, bool has_cs, bool filt) {
  const bool shared = a.rel_stride == 0, self = a.self != nullptr;
  const bool v4 = (a.flags & AGNN_SPMM_FAST_V4) != 0 || (a.flags & AGNN_SPMM_ACCUM) != 0;
  // Rows per wave: one.  More than one (consecutive rows sharing one index phase) measured SLOWER at the C2 shape, both
  // with the rows' gathers in flight together (95 VGPRs -> 5 waves / SIMD: forward 17.1 us vs 15.5 us) and one row after
  // the other on the same registers (16.5 us; backward 18.1 vs 15.5 us): vmcnt counts in issue order, so the second
  // row's gathers cannot be consumed before the first row's stores are acknowledged, whereas a wave that ends after
  // its stores hands its slot to a new wave at once (round 1, profiles/r01_spmm_kernel_study.md).
  const dim3 grid(static_cast<unsigned>(((a.n_rows + 3) / 4 + 7) & ~7));   // multiple of 8: the XCD remap is a bijection
  FastTable f{};
  f.n_rel = t.n_rel;
  for (int r = 0; r < AGNN_MAX_SEG + 3; ++r) {
    const agnn_rel_t& R = t.r[r < t.n_rel ? r : t.n_rel - 1];     // the tail repeats the last relation: loads stay valid
    f.rowptr[r] = R.rowptr;
    f.rowend[r] = R.rowend != nullptr ? R.rowend : R.rowptr + 1;   // untrimmed: row i ends where row i + 1 starts
    f.col[r] = R.col;
    f.src[r] = R.src;
    f.colscale[r] = R.colscale;
    f.ldb[r] = static_cast<uint32_t>(R.ld_src * 4);
  }
#define AGNN_FAST(CS, SH, SE)                                                                                     \
  do {                                                                                                            \
    if (v4) hipLaunchKernelGGL((k_spmm_fast<CH, CS, SH, SE>), grid, dim3(256), 0, stream, t, a);                  \
    else if (filt) hipLaunchKernelGGL((k_spmm_fast7<CH, CS, SH, SE, true>), grid, dim3(256), 0, stream, f, a);    \
    else hipLaunchKernelGGL((k_spmm_fast7<CH, CS, SH, SE, false>), grid, dim3(256), 0, stream, f, a);             \
  } while (0)
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

namespace {
__global__ __launch_bounds__(256) void k_spmm_self_grad(const float* __restrict__ dout, int64_t ld_dout, int64_t rel_stride, int n_rel,
                                                        const float* __restrict__ inv_cnt, int64_t ld_inv, int64_t n_rows, int H4,
                                                        float* __restrict__ out, int64_t ld_out, int accumulate) {
  const int64_t total = n_rows * H4;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; e < total; e += static_cast<int64_t>(gridDim.x) * 256) {
    const int64_t i = e / H4;
    const int c = static_cast<int>(e - i * H4) * 4;
    float4 acc = f4_zero();
    for (int r = 0; r < n_rel; ++r) {
      const float w = inv_cnt != nullptr ? inv_cnt[r * ld_inv + i] : 1.f;
      f4_fma(acc, w, *reinterpret_cast<const float4*>(dout + i * ld_dout + r * rel_stride + c));
    }
    float4* q = reinterpret_cast<float4*>(out + i * ld_out + c);
    if (accumulate) f4_add(acc, *q);
    *q = acc;
  }
}
}  // namespace

extern "C" int agnn_spmm_self_grad_f32(const float* dout, int64_t ld_dout, int64_t rel_stride, int32_t n_rel, const float* inv_cnt,
                                       int64_t ld_inv, int64_t n_rows, int32_t H, float* out, int64_t ld_out, int32_t accumulate,
                                       agnn_stream_t stream_) {
  using namespace agnn;
  if (n_rel <= 0 || n_rows < 0 || H <= 0 || (H & 3)) return fail(AGNN_EINVAL, "spmm_self_grad: n_rel=%d n_rows=%lld H=%d (H must be a multiple of 4)", n_rel, (long long)n_rows, H);
  if (n_rows == 0) return AGNN_OK;
  if (!dout || !out) return fail(AGNN_EINVAL, "spmm_self_grad: null argument");
  if (!aligned16(dout) || !aligned16(out) || (ld_dout & 3) || (ld_out & 3) || (rel_stride & 3) || ld_out < H || ld_dout < H)
    return fail(AGNN_EALIGN, "spmm_self_grad: operands must be 16-byte aligned with leading dimensions >= H");
  if (inv_cnt && ld_inv < n_rows) return fail(AGNN_EINVAL, "spmm_self_grad: ld_inv=%lld < n_rows", (long long)ld_inv);
  const int64_t total = n_rows * (H / 4);
  int64_t blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(k_spmm_self_grad, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, static_cast<hipStream_t>(stream_), dout, ld_dout,
                     rel_stride, n_rel, inv_cnt, ld_inv, n_rows, H / 4, out, ld_out, accumulate);
  return check_launch("spmm_self_grad");
}

namespace {
int spmm_entry(int n_rel, const agnn_rel_t* rels, int64_t n_rows, int32_t H, float* out, int64_t ld_out, int64_t rel_stride,
               const float* self, int64_t ld_self, int64_t self_rows, float* inv_cnt, int32_t col_limit, uint32_t flags,
               agnn_stream_t stream_);
}

extern "C" int agnn_spmm_f32(int n_rel, const agnn_rel_t* rels, int64_t n_rows, int32_t H, float* out,
                             int64_t ld_out, int64_t rel_stride, const float* self, int64_t ld_self,
                             float* inv_cnt, int32_t col_limit, uint32_t flags, agnn_stream_t stream_) {
  if (flags & AGNN_SPMM_ROOT) return agnn::fail(AGNN_EINVAL, "spmm: AGNN_SPMM_ROOT is agnn_spmm_root_f32's flag");
  return spmm_entry(n_rel, rels, n_rows, H, out, ld_out, rel_stride, self, ld_self, n_rows, inv_cnt, col_limit, flags, stream_);
}

extern "C" int agnn_spmm_root_f32(int n_rel, const agnn_rel_t* rels, int64_t n_rows, int32_t H, float* out, int64_t ld_out,
                                  int64_t rel_stride, const float* root, int64_t ld_root, int64_t root_rows, float* inv_cnt,
                                  int32_t col_limit, uint32_t flags, agnn_stream_t stream_) {
  using namespace agnn;
  if (!root || root_rows < 0) return fail(AGNN_EINVAL, "spmm_root: null root operand or root_rows=%lld", (long long)root_rows);
  if (rel_stride != 0 && root_rows < n_rows) return fail(AGNN_EINVAL, "spmm_root: the root slot needs a row per output row (root_rows=%lld < n_rows=%lld)", (long long)root_rows, (long long)n_rows);
  if (flags & (AGNN_SPMM_ACCUM | AGNN_SPMM_GENERIC | AGNN_SPMM_FAST_V4)) return fail(AGNN_EINVAL, "spmm_root: flags 0x%x not supported", flags);
  if (H != 256 && H != 512) return fail(AGNN_EINVAL, "spmm_root: H=%d (256 or 512: the widths the specialised kernel is built for)", H);
  if (rel_stride != 0 && ld_out < n_rel * rel_stride + H) return fail(AGNN_EINVAL, "spmm_root: ld_out=%lld has no room for the root slot", (long long)ld_out);
  for (int r = 0; rels && r < n_rel && r < AGNN_MAX_SEG; ++r)
    if (rels[r].ew != nullptr) return fail(AGNN_EINVAL, "spmm_root: per-edge weights are not supported");
  return spmm_entry(n_rel, rels, n_rows, H, out, ld_out, rel_stride, root, ld_root, root_rows < n_rows ? root_rows : n_rows, inv_cnt,
                    col_limit, flags | AGNN_SPMM_ROOT, stream_);
}

namespace {
int spmm_entry(int n_rel, const agnn_rel_t* rels, int64_t n_rows, int32_t H, float* out, int64_t ld_out, int64_t rel_stride,
               const float* self, int64_t ld_self, int64_t self_rows, float* inv_cnt, int32_t col_limit, uint32_t flags,
               agnn_stream_t stream_) {
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
  SpmmArgs a{static_cast<int32_t>(n_rows), H, out, ld_out, rel_stride, self, ld_self, inv_cnt, col_limit, flags, static_cast<int32_t>(self_rows)};
  int64_t blocks = ((n_rows + 3) / 4 + 7) & ~int64_t{7};   // multiple of 8: the XCD remap is a bijection
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (!(flags & AGNN_SPMM_GENERIC) && (H == 256 || H == 512)) {
    // fast path: everything except per-edge weights; trimmed rows (rowend) cost nothing extra, the per-edge predicates
    // (SKIP_SELF, col_limit) select the FILT instantiation; the older v4 kernel (ACCUM, A/B timing) knows neither
    const bool filt = (flags & AGNN_SPMM_SKIP_SELF) != 0 || col_limit != INT32_MAX;
    const bool v4 = (flags & (AGNN_SPMM_FAST_V4 | AGNN_SPMM_ACCUM)) != 0;
    bool plain = true, any_cs = false, all_cs = true;
    for (int r = 0; r < n_rel; ++r) {
      plain = plain && rels[r].ew == nullptr && rels[r].src != nullptr && rels[r].col != nullptr &&
              rels[r].ld_src < (int64_t{1} << 30) && !(v4 && rels[r].rowend != nullptr);
      any_cs = any_cs || rels[r].colscale != nullptr;
      all_cs = all_cs && rels[r].colscale != nullptr;
    }
    if (plain && any_cs == all_cs && !(v4 && filt)) {
      if (H == 256) launch_fast<1>(stream, t, a, all_cs, filt);
      else launch_fast<2>(stream, t, a, all_cs, filt);
      return check_launch("spmm(fast)");
    }
  }
  if (flags & AGNN_SPMM_ROOT) return fail(AGNN_EINVAL, "spmm_root: this launch is outside the specialised kernel (missing src / col, or mixed column scales)");
  if (H <= 256) launch_s<1>(dim3(blocks), stream, t, a);
  else if (H <= 512) launch_s<2>(dim3(blocks), stream, t, a);
  else launch_s<4>(dim3(blocks), stream, t, a);
  return check_launch("spmm");
}
}  // namespace
