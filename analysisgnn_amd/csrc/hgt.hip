// HGT edge-softmax attention (PyG HGTConv message / softmax / aggregate) for gfx950.
//
// Reference path: graphmuse `HybridHGT` -> torch_geometric `HGTConv(heads=4)` reached from
// analysisgnn/models/analysis.py:445-453 (no in-tree implementation).  Semantics restated in
// SURVEY.md App. A.4 and oracle/pyg_ref.py::hgt_conv: for destination row i and head h
//     s_e  = <q_i,h , k'_e,h> * p_rel[r(e),h] / sqrt(D)        over ALL incoming edges of all relations
//     a_e  = exp(s_e - max) / (sum_e exp(s_e - max) + 1e-16)
//     m_i,h = sum_e a_e v'_e,h
// where k' = k A_r^k, v' = v A_r^v are the relation-transformed keys / values of the source node
// (dense grouped GEMMs done by the caller).
//
// One wavefront per destination row, a lane owns 4 consecutive floats of every 256-float chunk,
// so the D = H/heads floats of a head sit in D/4 adjacent lanes and the per-head dot products are
// butterfly reductions inside those lane groups.  Single pass over the concatenated CSR segments
// with an online (running max / running sum) softmax: K' and V' rows are read exactly once, the
// [E, heads] score matrix is never materialised in the forward.  Backward = one pass by destination
// (recomputes the probabilities from the saved max / 1/sum, emits dq and per-edge alpha, ds) and
// one pass by source over the transposed CSR (dK', dV'): no atomics, bitwise reproducible.
// HBM/L2-bound gather work; no MFMA here.
#include <cmath>

#include "agnn_common.h"

namespace {

struct HgtTable {
  agnn_hgt_rel_t r[AGNN_MAX_SEG];
  int n_rel;
};

__device__ __forceinline__ float4 f4z() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float dot4(const float4& a, const float4& b) {
  return fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, a.w * b.w)));
}
// sum over the gl (power of two <= 64) adjacent lanes that hold one head; every lane gets the total.  Up to 16 lanes
// (D <= 64: a head sits inside one 16-lane DPP row) the butterfly runs on the DPP cross-lane paths — quad_perm [1,0,3,2],
// quad_perm [2,3,0,1], row_half_mirror, row_mirror — instead of ds_bpermute round trips through the LDS crossbar.
__device__ __forceinline__ float group_sum(float v, int gl) {
  if (gl >= 2) v += agnn::dpp_mov<0xB1>(v);
  if (gl >= 4) v += agnn::dpp_mov<0x4E>(v);
  if (gl >= 8) v += agnn::dpp_mov<0x141>(v);
  if (gl >= 16) v += agnn::dpp_mov<0x140>(v);
  for (int off = 16; off < gl; off <<= 1) v += __shfl_xor(v, off, 64);
  return v;
}
typedef const __attribute__((address_space(4))) int32_t* hk_i32p;      // wave-uniform index loads on the scalar unit

struct HgtArgs {
  const float* q;
  int64_t ld_q;
  int32_t n_rows;
  int32_t H;
  int32_t heads;
  int32_t col_limit;
};

template <int CH>
__device__ __forceinline__ void hgt_fwd_rows(const HgtTable& t, int r0, int r1, const HgtArgs& a, float* __restrict__ out, int64_t ld_out,
                                             float* __restrict__ m_out, float* __restrict__ linv_out, int block, int n_blocks) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int vb = (block & 7) * (n_blocks >> 3) + (block >> 3);   // XCD-contiguous row slabs
  const int row = vb * 4 + wave;
  if (row >= a.n_rows) return;
  const int D = a.H / a.heads;
  const int gl = D >> 2;
  bool on[CH];
  int head[CH];
  float4 qv[CH], acc[CH];
  float m[CH], l[CH];
  const float4* qp = reinterpret_cast<const float4*>(a.q + static_cast<int64_t>(row) * a.ld_q);
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int f = c * 256 + lane * 4;
    on[c] = f < a.H;
    head[c] = on[c] ? f / D : 0;
    qv[c] = on[c] ? qp[c * 64 + lane] : f4z();
    acc[c] = f4z();
    m[c] = -INFINITY;
    l[c] = 0.f;
  }
  // Index phase on the scalar unit (rowptr / rowend of a relation are wave-uniform), the column ids of a segment in ONE
  // vector register (broadcast with v_readlane), and TWO neighbours in flight: their four K' / V' rows are requested
  // before the first score is reduced, and one running-max update (three exponentials) covers both.
  for (int r = r0; r < r1; ++r) {
    const agnn_hgt_rel_t& R = t.r[r];
    const int start = ((hk_i32p)R.rowptr)[row];
    const int end = (R.rowend != nullptr) ? ((hk_i32p)R.rowend)[row] : ((hk_i32p)R.rowptr)[row + 1];
    const int n = end - start;
    if (n <= 0) continue;
    float ps[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) ps[c] = R.pscale[head[c]];
    for (int base = 0; base < n; base += 64) {
      const int mcnt = (n - base) < 64 ? (n - base) : 64;
      const int cv = lane < mcnt ? R.col[start + base + lane] : 0;
      for (int k = 0; k < mcnt; k += 2) {
        const bool two = k + 1 < mcnt;
        const int j0 = __builtin_amdgcn_readlane(cv, k);
        const int j1 = __builtin_amdgcn_readlane(cv, two ? k + 1 : k);
        const float4* kp0 = reinterpret_cast<const float4*>(R.k + static_cast<int64_t>(j0) * R.ld);
        const float4* vp0 = reinterpret_cast<const float4*>(R.v + static_cast<int64_t>(j0) * R.ld);
        const float4* kp1 = reinterpret_cast<const float4*>(R.k + static_cast<int64_t>(j1) * R.ld);
        const float4* vp1 = reinterpret_cast<const float4*>(R.v + static_cast<int64_t>(j1) * R.ld);
        float4 k0[CH], v0[CH], k1[CH], v1[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          k0[c] = on[c] ? kp0[c * 64 + lane] : f4z();
          k1[c] = on[c] ? kp1[c * 64 + lane] : f4z();
          v0[c] = on[c] ? vp0[c * 64 + lane] : f4z();
          v1[c] = on[c] ? vp1[c * 64 + lane] : f4z();
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          const float s0 = group_sum(dot4(qv[c], k0[c]), gl) * ps[c];
          const float s1 = two ? group_sum(dot4(qv[c], k1[c]), gl) * ps[c] : -INFINITY;
          const float mn = fmaxf(m[c], fmaxf(s0, s1));
          const float corr = __expf(m[c] - mn);     // exp(-inf) = 0 on the first edge
          const float p0 = __expf(s0 - mn), p1 = __expf(s1 - mn);
          l[c] = fmaf(l[c], corr, p0 + p1);
          acc[c].x = fmaf(acc[c].x, corr, fmaf(p0, v0[c].x, p1 * v1[c].x));
          acc[c].y = fmaf(acc[c].y, corr, fmaf(p0, v0[c].y, p1 * v1[c].y));
          acc[c].z = fmaf(acc[c].z, corr, fmaf(p0, v0[c].z, p1 * v1[c].z));
          acc[c].w = fmaf(acc[c].w, corr, fmaf(p0, v0[c].w, p1 * v1[c].w));
          m[c] = mn;
        }
      }
    }
  }
  float4* op = reinterpret_cast<float4*>(out + static_cast<int64_t>(row) * ld_out);
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    if (!on[c]) continue;
    const float linv = 1.f / (l[c] + 1e-16f);
    float4 o = acc[c];
    o.x *= linv; o.y *= linv; o.z *= linv; o.w *= linv;
    op[c * 64 + lane] = o;
    if (((c * 256 + lane * 4) % D) == 0) {
      m_out[static_cast<int64_t>(row) * a.heads + head[c]] = m[c];
      linv_out[static_cast<int64_t>(row) * a.heads + head[c]] = linv;
    }
  }
}

template <int CH>
__global__ __launch_bounds__(256) void k_hgt_fwd(HgtTable t, HgtArgs a, float* __restrict__ out, int64_t ld_out,
                                                 float* __restrict__ m_out, float* __restrict__ linv_out) {
  hgt_fwd_rows<CH>(t, 0, t.n_rel, a, out, ld_out, m_out, linv_out, blockIdx.x, gridDim.x);
}

// Several destination types in ONE launch (an HGT layer has one attention per destination type; the small types' launches —
// beats, measures — each cost a launch's latency on the layer's serial chain for a few microseconds of work): item i owns the
// blocks [blk0[i], blk0[i + 1]) and the relations [rel0[i], rel0[i + 1]) of the table.
struct HgtMulti {
  HgtArgs a[AGNN_HGT_MAX_DST];
  float* out[AGNN_HGT_MAX_DST];
  float* m_out[AGNN_HGT_MAX_DST];
  float* linv_out[AGNN_HGT_MAX_DST];
  int64_t ld_out[AGNN_HGT_MAX_DST];
  int32_t rel0[AGNN_HGT_MAX_DST + 1];
  int32_t blk0[AGNN_HGT_MAX_DST + 1];
  int32_t n_items;
};

template <int CH>
__global__ __launch_bounds__(256) void k_hgt_fwd_multi(HgtTable t, HgtMulti mm) {
  int i = 0;
  while (i + 1 < mm.n_items && static_cast<int>(blockIdx.x) >= mm.blk0[i + 1]) ++i;     // block-uniform
  hgt_fwd_rows<CH>(t, mm.rel0[i], mm.rel0[i + 1], mm.a[i], mm.out[i], mm.ld_out[i], mm.m_out[i], mm.linv_out[i],
                   static_cast<int>(blockIdx.x) - mm.blk0[i], mm.blk0[i + 1] - mm.blk0[i]);
}

// Pass by destination: dq, and per edge (indexed by the edge's COO position `perm`) alpha, gs = ds * pscale,
// tdot = ds * <q,k'> (for the gradient of p_rel).
template <int CH>
__global__ __launch_bounds__(256) void k_hgt_bwd_dst(HgtTable t, HgtArgs a, const float* __restrict__ dm, int64_t ld_dm,
                                                     const float* __restrict__ mo, int64_t ld_mo,
                                                     const float* __restrict__ m_in, const float* __restrict__ linv_in,
                                                     float* __restrict__ dq, int64_t ld_dq) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int vb = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int row = vb * 4 + wave;
  if (row >= a.n_rows) return;
  const int D = a.H / a.heads;
  const int gl = D >> 2;
  bool on[CH], lead[CH];
  int head[CH];
  float4 qv[CH], dmv[CH], dqa[CH];
  float mrow[CH], linv[CH], dsum[CH];
  const float4* qp = reinterpret_cast<const float4*>(a.q + static_cast<int64_t>(row) * a.ld_q);
  const float4* dp = reinterpret_cast<const float4*>(dm + static_cast<int64_t>(row) * ld_dm);
  const float4* mp = reinterpret_cast<const float4*>(mo + static_cast<int64_t>(row) * ld_mo);
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int f = c * 256 + lane * 4;
    on[c] = f < a.H;
    head[c] = on[c] ? f / D : 0;
    lead[c] = on[c] && (f % D) == 0;
    qv[c] = on[c] ? qp[c * 64 + lane] : f4z();
    dmv[c] = on[c] ? dp[c * 64 + lane] : f4z();
    const float4 mv = on[c] ? mp[c * 64 + lane] : f4z();
    dsum[c] = group_sum(dot4(dmv[c], mv), gl);                       // sum_e alpha_e dalpha_e = <dM, M>
    mrow[c] = m_in[static_cast<int64_t>(row) * a.heads + head[c]];
    linv[c] = linv_in[static_cast<int64_t>(row) * a.heads + head[c]];
    dqa[c] = f4z();
  }
  // same loop structure as the forward pass: scalar index phase, column ids / COO positions of a segment in one vector
  // register each, two neighbours' K' / V' rows in flight
  for (int r = 0; r < t.n_rel; ++r) {
    const agnn_hgt_rel_t& R = t.r[r];
    const int start = ((hk_i32p)R.rowptr)[row];
    const int end = (R.rowend != nullptr) ? ((hk_i32p)R.rowend)[row] : ((hk_i32p)R.rowptr)[row + 1];
    const int n = end - start;
    if (n <= 0) continue;
    float ps[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) ps[c] = R.pscale[head[c]];
    for (int base = 0; base < n; base += 64) {
      const int mcnt = (n - base) < 64 ? (n - base) : 64;
      const int cv = lane < mcnt ? R.col[start + base + lane] : 0;
      const int pv = lane < mcnt ? R.perm[start + base + lane] : 0;
      for (int k = 0; k < mcnt; k += 2) {
        const bool two = k + 1 < mcnt;
        const int k1i = two ? k + 1 : k;
        const int j0 = __builtin_amdgcn_readlane(cv, k), j1 = __builtin_amdgcn_readlane(cv, k1i);
        const int64_t e0 = __builtin_amdgcn_readlane(pv, k), e1 = __builtin_amdgcn_readlane(pv, k1i);
        const float4* kp0 = reinterpret_cast<const float4*>(R.k + static_cast<int64_t>(j0) * R.ld);
        const float4* vp0 = reinterpret_cast<const float4*>(R.v + static_cast<int64_t>(j0) * R.ld);
        const float4* kp1 = reinterpret_cast<const float4*>(R.k + static_cast<int64_t>(j1) * R.ld);
        const float4* vp1 = reinterpret_cast<const float4*>(R.v + static_cast<int64_t>(j1) * R.ld);
        float4 k0[CH], v0[CH], k1[CH], v1[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          k0[c] = on[c] ? kp0[c * 64 + lane] : f4z();
          k1[c] = on[c] ? kp1[c * 64 + lane] : f4z();
          v0[c] = on[c] ? vp0[c * 64 + lane] : f4z();
          v1[c] = on[c] ? vp1[c * 64 + lane] : f4z();
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          const float dot0 = group_sum(dot4(qv[c], k0[c]), gl), dot1 = group_sum(dot4(qv[c], k1[c]), gl);
          const float da0 = group_sum(dot4(dmv[c], v0[c]), gl), da1 = group_sum(dot4(dmv[c], v1[c]), gl);
          const float al0 = __expf(dot0 * ps[c] - mrow[c]) * linv[c];
          const float al1 = two ? __expf(dot1 * ps[c] - mrow[c]) * linv[c] : 0.f;
          const float ds0 = al0 * (da0 - dsum[c]), ds1 = al1 * (da1 - dsum[c]);
          const float g0 = ds0 * ps[c], g1 = ds1 * ps[c];
          dqa[c].x = fmaf(g0, k0[c].x, fmaf(g1, k1[c].x, dqa[c].x));
          dqa[c].y = fmaf(g0, k0[c].y, fmaf(g1, k1[c].y, dqa[c].y));
          dqa[c].z = fmaf(g0, k0[c].z, fmaf(g1, k1[c].z, dqa[c].z));
          dqa[c].w = fmaf(g0, k0[c].w, fmaf(g1, k1[c].w, dqa[c].w));
          if (lead[c]) {
            R.alpha[e0 * a.heads + head[c]] = al0;
            R.gs[e0 * a.heads + head[c]] = g0;
            R.tdot[R.ld_tdot > 0 ? head[c] * R.ld_tdot + e0 : e0 * a.heads + head[c]] = ds0 * dot0;
            if (two) {
              R.alpha[e1 * a.heads + head[c]] = al1;
              R.gs[e1 * a.heads + head[c]] = g1;
              R.tdot[R.ld_tdot > 0 ? head[c] * R.ld_tdot + e1 : e1 * a.heads + head[c]] = ds1 * dot1;
            }
          }
        }
      }
    }
  }
  float4* op = reinterpret_cast<float4*>(dq + static_cast<int64_t>(row) * ld_dq);
#pragma unroll
  for (int c = 0; c < CH; ++c)
    if (on[c]) op[c * 64 + lane] = dqa[c];
}

// Pass by source (one relation, transposed CSR: rows = source nodes, col = destination rows):
//   dV'[j] = sum_e alpha_e dM[i(e)]   ;   dK'[j] = sum_e gs_e q[i(e)]     (per head)
template <int CH>
__device__ __forceinline__ void hgt_bwd_src_rows(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ rowend,
                                                 const int32_t* __restrict__ col, const int32_t* __restrict__ perm,
                                                 const float* __restrict__ alpha, const float* __restrict__ gs, const HgtArgs& a,
                                                 const float* __restrict__ dm, int64_t ld_dm, float* __restrict__ dk,
                                                 float* __restrict__ dv, int64_t ld_o, const int block, const int n_blocks) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int vb = (block & 7) * (n_blocks >> 3) + (block >> 3);
  const int row = vb * 4 + wave;
  if (row >= a.n_rows) return;
  const int D = a.H / a.heads;
  bool on[CH];
  int head[CH];
  float4 dka[CH], dva[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int f = c * 256 + lane * 4;
    on[c] = f < a.H;
    head[c] = on[c] ? f / D : 0;
    dka[c] = f4z();
    dva[c] = f4z();
  }
  const int start = ((hk_i32p)rowptr)[row];
  const int end = (rowend != nullptr) ? ((hk_i32p)rowend)[row] : ((hk_i32p)rowptr)[row + 1];
  const int n = end - start;
  for (int base = 0; base < n; base += 64) {
    const int mcnt = (n - base) < 64 ? (n - base) : 64;
    const int cv = lane < mcnt ? col[start + base + lane] : 0;
    const int pv = lane < mcnt ? perm[start + base + lane] : 0;
    for (int k = 0; k < mcnt; k += 2) {
      const int k1i = k + 1 < mcnt ? k + 1 : k;
      const int i0 = __builtin_amdgcn_readlane(cv, k), i1 = __builtin_amdgcn_readlane(cv, k1i);
      const bool ok0 = i0 < a.col_limit, ok1 = (k + 1 < mcnt) && i1 < a.col_limit;      // wave-uniform
      const int is0 = ok0 ? i0 : 0, is1 = ok1 ? i1 : 0;                                   // dropped edges read row 0, weight 0
      const int64_t e0 = __builtin_amdgcn_readlane(pv, k), e1 = __builtin_amdgcn_readlane(pv, k1i);
      const float4* qp0 = reinterpret_cast<const float4*>(a.q + static_cast<int64_t>(is0) * a.ld_q);
      const float4* dp0 = reinterpret_cast<const float4*>(dm + static_cast<int64_t>(is0) * ld_dm);
      const float4* qp1 = reinterpret_cast<const float4*>(a.q + static_cast<int64_t>(is1) * a.ld_q);
      const float4* dp1 = reinterpret_cast<const float4*>(dm + static_cast<int64_t>(is1) * ld_dm);
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        if (!on[c]) continue;
        const float4 q0 = qp0[c * 64 + lane], d0 = dp0[c * 64 + lane], q1 = qp1[c * 64 + lane], d1 = dp1[c * 64 + lane];
        const float al0 = ok0 ? alpha[e0 * a.heads + head[c]] : 0.f, g0 = ok0 ? gs[e0 * a.heads + head[c]] : 0.f;
        const float al1 = ok1 ? alpha[e1 * a.heads + head[c]] : 0.f, g1 = ok1 ? gs[e1 * a.heads + head[c]] : 0.f;
        dva[c].x = fmaf(al0, d0.x, fmaf(al1, d1.x, dva[c].x)); dva[c].y = fmaf(al0, d0.y, fmaf(al1, d1.y, dva[c].y));
        dva[c].z = fmaf(al0, d0.z, fmaf(al1, d1.z, dva[c].z)); dva[c].w = fmaf(al0, d0.w, fmaf(al1, d1.w, dva[c].w));
        dka[c].x = fmaf(g0, q0.x, fmaf(g1, q1.x, dka[c].x)); dka[c].y = fmaf(g0, q0.y, fmaf(g1, q1.y, dka[c].y));
        dka[c].z = fmaf(g0, q0.z, fmaf(g1, q1.z, dka[c].z)); dka[c].w = fmaf(g0, q0.w, fmaf(g1, q1.w, dka[c].w));
      }
    }
  }
  float4* okp = reinterpret_cast<float4*>(dk + static_cast<int64_t>(row) * ld_o);
  float4* ovp = reinterpret_cast<float4*>(dv + static_cast<int64_t>(row) * ld_o);
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    if (!on[c]) continue;
    okp[c * 64 + lane] = dka[c];
    ovp[c * 64 + lane] = dva[c];
  }
}

template <int CH>
__global__ __launch_bounds__(256) void k_hgt_bwd_src(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ rowend,
                                                     const int32_t* __restrict__ col, const int32_t* __restrict__ perm,
                                                     const float* __restrict__ alpha, const float* __restrict__ gs, HgtArgs a,
                                                     const float* __restrict__ dm, int64_t ld_dm, float* __restrict__ dk,
                                                     float* __restrict__ dv, int64_t ld_o) {
  hgt_bwd_src_rows<CH>(rowptr, rowend, col, perm, alpha, gs, a, dm, ld_dm, dk, dv, ld_o, blockIdx.x, gridDim.x);
}

// All relations that end in one destination type in ONE launch (they share q and dM; each writes its own column block of its
// source type's dK' / dV'): as separate launches they were four ~13 us kernels in a row on the HGT stack's backward chain.
struct SrcBatch {
  agnn_hgt_src_item_t it[AGNN_MAX_SEG];
  int32_t first[AGNN_MAX_SEG + 1];       // first workgroup of item i (each item's count is a multiple of 8: XCD-contiguous slabs)
  const float* q;
  const float* dm;
  int64_t ld_q, ld_dm;
  int32_t n, H, heads;
};

template <int CH>
__global__ __launch_bounds__(256) void k_hgt_bwd_src_batch(SrcBatch b) {
  int i = 0;
  const int blk = blockIdx.x;
  while (i + 1 < b.n && blk >= b.first[i + 1]) ++i;              // workgroup-uniform
  const agnn_hgt_src_item_t& it = b.it[i];
  const HgtArgs a{b.q, b.ld_q, it.n_src_rows, b.H, b.heads, it.col_limit};
  hgt_bwd_src_rows<CH>(it.rowptr, it.rowend, it.col, it.perm, it.alpha, it.gs, a, b.dm, b.ld_dm, it.dk, it.dv, it.ld_o, blk - b.first[i],
                       b.first[i + 1] - b.first[i]);
}

int check_shape(const char* who, int64_t n_rows, int32_t H, int32_t heads) {
  using namespace agnn;
  if (n_rows < 0 || n_rows >= (int64_t{1} << 31)) return fail(AGNN_EINVAL, "%s: n_rows=%lld", who, (long long)n_rows);
  if (H <= 0 || (H & 3) || H > 1024) return fail(AGNN_EINVAL, "%s: H=%d must be a multiple of 4 in [4,1024]", who, H);
  if (heads <= 0 || H % heads) return fail(AGNN_EINVAL, "%s: heads=%d must divide H=%d", who, heads, H);
  const int D = H / heads;
  const int gl = D / 4;
  if ((D & 3) || gl > 64 || (gl & (gl - 1)) || (256 % D != 0 && D % 256 != 0))
    return fail(AGNN_EINVAL, "%s: head dim D=%d must be 4*2^k, at most 256, and must not straddle a 256-float chunk", who, D);
  return AGNN_OK;
}

inline unsigned grid_for(int64_t n_rows) { return static_cast<unsigned>((((n_rows + 3) / 4) + 7) & ~int64_t{7}); }

}  // namespace

extern "C" int agnn_hgt_attn_fwd_f32(int n_rel, const agnn_hgt_rel_t* rels, const float* q, int64_t ld_q, int64_t n_rows,
                                     int32_t H, int32_t heads, float* out, int64_t ld_out, float* m_out, float* linv_out,
                                     agnn_stream_t stream_) {
  using namespace agnn;
  if (int rc = check_shape("hgt_fwd", n_rows, H, heads)) return rc;
  if (n_rel < 0 || n_rel > AGNN_MAX_SEG) return fail(AGNN_EINVAL, "hgt_fwd: n_rel=%d", n_rel);
  if (n_rows == 0) return AGNN_OK;
  if (!q || !out || !m_out || !linv_out || (n_rel > 0 && !rels)) return fail(AGNN_EINVAL, "hgt_fwd: null argument");
  if (!aligned16(q) || !aligned16(out) || (ld_q & 3) || (ld_out & 3)) return fail(AGNN_EALIGN, "hgt_fwd: q/out misaligned");
  HgtTable t{};
  t.n_rel = n_rel;
  for (int r = 0; r < n_rel; ++r) {
    if (!rels[r].rowptr || !rels[r].pscale) return fail(AGNN_EINVAL, "hgt_fwd: relation %d incomplete", r);
    if ((rels[r].k && !aligned16(rels[r].k)) || (rels[r].v && !aligned16(rels[r].v)) || (rels[r].ld & 3)) return fail(AGNN_EALIGN, "hgt_fwd: relation %d k/v misaligned", r);
    t.r[r] = rels[r];
  }
  HgtArgs a{q, ld_q, static_cast<int32_t>(n_rows), H, heads, INT32_MAX};
  hipStream_t s = static_cast<hipStream_t>(stream_);
  const dim3 grid(grid_for(n_rows)), block(256);
  if (H <= 256) hipLaunchKernelGGL(k_hgt_fwd<1>, grid, block, 0, s, t, a, out, ld_out, m_out, linv_out);
  else if (H <= 512) hipLaunchKernelGGL(k_hgt_fwd<2>, grid, block, 0, s, t, a, out, ld_out, m_out, linv_out);
  else hipLaunchKernelGGL(k_hgt_fwd<4>, grid, block, 0, s, t, a, out, ld_out, m_out, linv_out);
  return check_launch("hgt_fwd");
}

extern "C" int agnn_hgt_attn_fwd_multi_f32(int32_t n_items, const agnn_hgt_dst_item_t* items, int32_t H, int32_t heads, agnn_stream_t stream_) {
  using namespace agnn;
  if (n_items < 0 || n_items > AGNN_HGT_MAX_DST || (n_items > 0 && !items)) return fail(AGNN_EINVAL, "hgt_fwd_multi: n_items=%d (0 .. %d)", n_items, AGNN_HGT_MAX_DST);
  HgtTable t{};
  HgtMulti mm{};
  int nr = 0, blocks = 0, k = 0;
  for (int i = 0; i < n_items; ++i) {
    const agnn_hgt_dst_item_t& it = items[i];
    if (int rc = check_shape("hgt_fwd_multi", it.n_rows, H, heads)) return rc;
    if (it.n_rel < 0 || nr + it.n_rel > AGNN_MAX_SEG) return fail(AGNN_EINVAL, "hgt_fwd_multi: %d relations in all (max %d)", nr + it.n_rel, AGNN_MAX_SEG);
    if (it.n_rows == 0) continue;
    if (!it.q || !it.out || !it.m_out || !it.linv_out || (it.n_rel > 0 && !it.rels)) return fail(AGNN_EINVAL, "hgt_fwd_multi: item %d: null argument", i);
    if (!aligned16(it.q) || !aligned16(it.out) || (it.ld_q & 3) || (it.ld_out & 3)) return fail(AGNN_EALIGN, "hgt_fwd_multi: item %d: q/out misaligned", i);
    mm.rel0[k] = nr;
    for (int r = 0; r < it.n_rel; ++r) {
      const agnn_hgt_rel_t& R = it.rels[r];
      if (!R.rowptr || !R.pscale) return fail(AGNN_EINVAL, "hgt_fwd_multi: item %d relation %d incomplete", i, r);
      if ((R.k && !aligned16(R.k)) || (R.v && !aligned16(R.v)) || (R.ld & 3)) return fail(AGNN_EALIGN, "hgt_fwd_multi: item %d relation %d k/v misaligned", i, r);
      t.r[nr++] = R;
    }
    mm.a[k] = HgtArgs{it.q, it.ld_q, static_cast<int32_t>(it.n_rows), H, heads, INT32_MAX};
    mm.out[k] = it.out; mm.m_out[k] = it.m_out; mm.linv_out[k] = it.linv_out; mm.ld_out[k] = it.ld_out;
    mm.blk0[k] = blocks;
    blocks += static_cast<int>(grid_for(it.n_rows));
    ++k;
  }
  if (k == 0) return AGNN_OK;
  mm.rel0[k] = nr; mm.blk0[k] = blocks; mm.n_items = k;
  t.n_rel = nr;
  hipStream_t s = static_cast<hipStream_t>(stream_);
  const dim3 grid(static_cast<unsigned>(blocks)), block(256);
  if (H <= 256) hipLaunchKernelGGL(k_hgt_fwd_multi<1>, grid, block, 0, s, t, mm);
  else if (H <= 512) hipLaunchKernelGGL(k_hgt_fwd_multi<2>, grid, block, 0, s, t, mm);
  else hipLaunchKernelGGL(k_hgt_fwd_multi<4>, grid, block, 0, s, t, mm);
  return check_launch("hgt_fwd_multi");
}

extern "C" int agnn_hgt_attn_bwd_dst_f32(int n_rel, const agnn_hgt_rel_t* rels, const float* q, int64_t ld_q, const float* dm,
                                         int64_t ld_dm, const float* out, int64_t ld_out, const float* m_in,
                                         const float* linv_in, int64_t n_rows, int32_t H, int32_t heads, float* dq,
                                         int64_t ld_dq, agnn_stream_t stream_) {
  using namespace agnn;
  if (int rc = check_shape("hgt_bwd_dst", n_rows, H, heads)) return rc;
  if (n_rel < 0 || n_rel > AGNN_MAX_SEG) return fail(AGNN_EINVAL, "hgt_bwd_dst: n_rel=%d", n_rel);
  if (n_rows == 0) return AGNN_OK;
  if (!q || !dm || !out || !m_in || !linv_in || !dq || (n_rel > 0 && !rels)) return fail(AGNN_EINVAL, "hgt_bwd_dst: null argument");
  if (!aligned16(q) || !aligned16(dm) || !aligned16(out) || !aligned16(dq) || (ld_q & 3) || (ld_dm & 3) || (ld_out & 3) || (ld_dq & 3))
    return fail(AGNN_EALIGN, "hgt_bwd_dst: misaligned matrix");
  HgtTable t{};
  t.n_rel = n_rel;
  for (int r = 0; r < n_rel; ++r) {
    if (!rels[r].rowptr || !rels[r].pscale || !rels[r].perm || !rels[r].alpha || !rels[r].gs || !rels[r].tdot)
      return fail(AGNN_EINVAL, "hgt_bwd_dst: relation %d incomplete", r);
    t.r[r] = rels[r];
  }
  HgtArgs a{q, ld_q, static_cast<int32_t>(n_rows), H, heads, INT32_MAX};
  hipStream_t s = static_cast<hipStream_t>(stream_);
  const dim3 grid(grid_for(n_rows)), block(256);
  if (H <= 256) hipLaunchKernelGGL(k_hgt_bwd_dst<1>, grid, block, 0, s, t, a, dm, ld_dm, out, ld_out, m_in, linv_in, dq, ld_dq);
  else if (H <= 512) hipLaunchKernelGGL(k_hgt_bwd_dst<2>, grid, block, 0, s, t, a, dm, ld_dm, out, ld_out, m_in, linv_in, dq, ld_dq);
  else hipLaunchKernelGGL(k_hgt_bwd_dst<4>, grid, block, 0, s, t, a, dm, ld_dm, out, ld_out, m_in, linv_in, dq, ld_dq);
  return check_launch("hgt_bwd_dst");
}

extern "C" int agnn_hgt_attn_bwd_src_f32(const int32_t* rowptr, const int32_t* rowend, const int32_t* col, const int32_t* perm,
                                         const float* alpha, const float* gs, const float* q, int64_t ld_q, const float* dm,
                                         int64_t ld_dm, int64_t n_src_rows, int32_t col_limit, int32_t H, int32_t heads,
                                         float* dk, float* dv, int64_t ld_o, agnn_stream_t stream_) {
  using namespace agnn;
  if (int rc = check_shape("hgt_bwd_src", n_src_rows, H, heads)) return rc;
  if (n_src_rows == 0) return AGNN_OK;
  if (!rowptr || !dk || !dv) return fail(AGNN_EINVAL, "hgt_bwd_src: null argument");
  if (!aligned16(dk) || !aligned16(dv) || (ld_o & 3) || (q && (!aligned16(q) || (ld_q & 3))) || (dm && (!aligned16(dm) || (ld_dm & 3))))
    return fail(AGNN_EALIGN, "hgt_bwd_src: misaligned matrix");
  HgtArgs a{q, ld_q, static_cast<int32_t>(n_src_rows), H, heads, col_limit};
  hipStream_t s = static_cast<hipStream_t>(stream_);
  const dim3 grid(grid_for(n_src_rows)), block(256);
  if (H <= 256) hipLaunchKernelGGL(k_hgt_bwd_src<1>, grid, block, 0, s, rowptr, rowend, col, perm, alpha, gs, a, dm, ld_dm, dk, dv, ld_o);
  else if (H <= 512) hipLaunchKernelGGL(k_hgt_bwd_src<2>, grid, block, 0, s, rowptr, rowend, col, perm, alpha, gs, a, dm, ld_dm, dk, dv, ld_o);
  else hipLaunchKernelGGL(k_hgt_bwd_src<4>, grid, block, 0, s, rowptr, rowend, col, perm, alpha, gs, a, dm, ld_dm, dk, dv, ld_o);
  return check_launch("hgt_bwd_src");
}

extern "C" int agnn_hgt_attn_bwd_src_batch_f32(int32_t n_items, const agnn_hgt_src_item_t* items, const float* q, int64_t ld_q,
                                               const float* dm, int64_t ld_dm, int32_t H, int32_t heads, agnn_stream_t stream_) {
  using namespace agnn;
  if (n_items < 0 || n_items > AGNN_MAX_SEG || (n_items > 0 && !items)) return fail(AGNN_EINVAL, "hgt_bwd_src_batch: n_items=%d (0 .. %d)", n_items, AGNN_MAX_SEG);
  if ((q && (!aligned16(q) || (ld_q & 3))) || (dm && (!aligned16(dm) || (ld_dm & 3)))) return fail(AGNN_EALIGN, "hgt_bwd_src_batch: misaligned matrix");
  SrcBatch b{};
  b.q = q; b.dm = dm; b.ld_q = ld_q; b.ld_dm = ld_dm; b.H = H; b.heads = heads;
  int blocks = 0;
  for (int i = 0; i < n_items; ++i) {
    const agnn_hgt_src_item_t& it = items[i];
    if (int rc = check_shape("hgt_bwd_src_batch", it.n_src_rows, H, heads)) return rc;
    if (it.n_src_rows == 0) continue;
    if (!it.rowptr || !it.dk || !it.dv) return fail(AGNN_EINVAL, "hgt_bwd_src_batch: item %d: null argument", i);
    if (!aligned16(it.dk) || !aligned16(it.dv) || (it.ld_o & 3)) return fail(AGNN_EALIGN, "hgt_bwd_src_batch: item %d: misaligned output", i);
    b.first[b.n] = blocks;
    b.it[b.n++] = it;
    blocks += static_cast<int>(grid_for(it.n_src_rows));
  }
  if (b.n == 0) return AGNN_OK;
  b.first[b.n] = blocks;
  hipStream_t s = static_cast<hipStream_t>(stream_);
  const dim3 grid(static_cast<unsigned>(blocks)), block(256);
  if (H <= 256) hipLaunchKernelGGL(k_hgt_bwd_src_batch<1>, grid, block, 0, s, b);
  else if (H <= 512) hipLaunchKernelGGL(k_hgt_bwd_src_batch<2>, grid, block, 0, s, b);
  else hipLaunchKernelGGL(k_hgt_bwd_src_batch<4>, grid, block, 0, s, b);
  return check_launch("hgt_bwd_src_batch");
}
