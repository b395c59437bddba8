/*
 * agnn.h — C-ABI of libagnn_hip.so: the MI355X (gfx950) kernels behind the AnalysisGNN
 * heterogeneous message-passing hot path.
 *
 * The reference (manoskary/analysisgnn) is 100 % Python and has no FFI of its own; its hot path
 * bottoms out in three third-party operator families.  Each entry point below replaces one of
 * them, and the Python host side (analysisgnn_amd/_lib.py, ctypes) is the only caller:
 *
 *   agnn_csr_build        the per-relation `edge_index[:, edge_type == r]` compaction and the
 *                         COO bookkeeping that torch_scatter / PyG redo on every call
 *                         (ref: analysisgnn/models/core/hgnn.py:137-139, :481-483)
 *   agnn_csr_rowend       PyG `trim_to_layer` edge narrowing (ref: models/cadence.py:167-173)
 *   agnn_spmm_f32         `h[edge_index[1]]` gather + `torch_scatter.scatter(..., reduce=sum|mean,
 *                         out=...)` (ref: core/gnn.py:70-74, :511, :539; core/hgnn.py:406-407;
 *                         models/analysis.py:580-586) and PyG `SAGEConv` mean aggregation under
 *                         `HeteroConv` (ref: models/cadence.py:147-159,174)
 *   agnn_gru_fwd/bwd_f32  `torch.nn.GRU` of the hybrid sequence branch (ref: models/cadence.py:249-285)
 *   agnn_gated_*          `ResGatedGraphConv` edge gate + scatter (ref: core/gnn.py:246-257)
 *   agnn_absdiff_*        `RelEdgeConv` per-row sums of h_j and |h_i - h_j| (ref: core/gnn.py:99-105)
 *   agnn_norm_act_*       LayerNorm / ReLU / Dropout chains between the projections (ref: models/analysis.py:429-443)
 *   agnn_wgrad_f32        weight/bias gradients of the dense projections (fp32 MFMA, split over N)
 *   agnn_pack_f32         per-relation parameter cat / sum / gradient fan-out of the fused HeteroConv (ref: models/cadence.py:147-159)
 *   agnn_embed_cat_*      note-input assembly `cat([x, pitch_embedding(ps), key_embedding(ks)])` and the tables' gradient
 *                         (ref: models/analysis.py:399-400, :574)
 *   agnn_gproj_*          the task heads' last Linear layers as one grouped projection (ref: models/analysis.py:486-496)
 *   agnn_adamw_f32        gradient clipping + `torch.optim.AdamW` step on the flat buffers (ref: models/analysis.py:1380-1381)
 *   agnn_multitask_ce_f32 the 21 per-task CrossEntropyLoss terms (ref: models/analysis.py:881-888)
 *   agnn_sample_hops      graphmuse `MuseNeighborLoader` batch assembly (ref: data/datamodules/analysis.py:270-293)
 *   agnn_relt_*           PyG `HGTConv` per-head relation transforms (k_rel / v_rel)
 *   agnn_hgt_attn_*       PyG `HGTConv` message/softmax/aggregate, reached through graphmuse
 *                         `HybridHGT` (ref: models/analysis.py:445-453)
 *
 * Conventions (all entry points)
 *   - plain pointers and sizes only; every `const T*`/`T*` marked (device) is device memory owned
 *     by the caller (PyTorch's caching allocator); arrays marked (host) are read during the call.
 *   - all work is enqueued on `stream`; no implicit synchronisation, no default-stream use, no
 *     allocation, no global mutable state: safe to call from several host threads and under
 *     hipGraph capture.
 *   - return 0 on success, a negative errno-style code otherwise (AGNN_E*); the message is
 *     available from agnn_last_error() (thread-local).  Nothing is printed, nothing throws.
 *   - feature matrices are fp32, row-major, rows 16-byte aligned, H % 4 == 0.
 *   - index arrays produced by the library are int32; COO inputs are int64 as PyTorch holds them.
 *   - sums run in a fixed order (CSR order = original edge order inside a row): results are
 *     bitwise reproducible; no float atomics anywhere.
 */
#ifndef AGNN_H_
#define AGNN_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* agnn_stream_t; /* hipStream_t */

#define AGNN_OK        0
#define AGNN_EINVAL  (-22) /* bad shape / size / flag combination */
#define AGNN_EALIGN  (-14) /* misaligned pointer or leading dimension */
#define AGNN_ENOMEM  (-12) /* workspace too small */
#define AGNN_ERUNTIME (-5) /* HIP runtime error */

#define AGNN_MAX_SEG 32    /* segments (relations x directions) per call */

const char* agnn_last_error(void);
/* library / ABI version: (major << 16) | minor */
int agnn_version(void);

/* ------------------------------------------------------------------------------------------
 * CSR construction.  A "segment" is one relation in one direction: edges (row_e, col_e),
 * e < n_edges, grouped by row.  If `etype` is non-NULL only edges with etype[e] == etype_code
 * take part (the in-tree layers' `edge_type == code` mask, hgnn.py:138).
 * Output (device):
 *   rowstart [total_rows + 1]  with total_rows = sum_s n_rows[s];  segment s owns the slice
 *            rowstart + rowbase[s] (n_rows[s] + 1 entries, rowbase = exclusive prefix of n_rows);
 *            entries are offsets into col/perm shared by all segments.
 *   col      [E_total]  column (gathered row) id of every kept edge, grouped by row,
 *   perm     [E_total]  original edge position e inside its segment's COO arrays.
 * Inside a row, edges keep their original order (stable) so perm is increasing there.
 * E_total = sum_s n_edges[s] is the capacity; slots past the number of kept edges are undefined.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  const int64_t* row;    /* (device) [n_edges] */
  const int64_t* col;    /* (device) [n_edges] */
  const int64_t* etype;  /* (device) [n_edges] or NULL */
  int64_t etype_code;
  int64_t n_edges;
  int64_t n_rows;
} agnn_coo_seg_t;

size_t agnn_csr_workspace_bytes(int64_t e_total, int64_t total_rows);

/* `status` (device int32[1], may be NULL; caller-owned, zero-initialised once, never reset by the library): incremented
 * for every edge whose position falls outside its row — impossible when the build's own counters are intact, so a non-zero
 * word means the workspace was disturbed while the build ran (round 1: a memset node of a captured graph that had not
 * cleared the counters).  Nothing is written out of range in that case; the index is incomplete and agnn_check_status
 * reports it. */
int agnn_csr_build(int n_seg, const agnn_coo_seg_t* segs /* (host) */,
                   int32_t* rowstart, int32_t* col, int32_t* perm,
                   void* workspace, size_t workspace_bytes, int32_t* status, agnn_stream_t stream);

/* Synchronises `stream`, reads the status word and returns AGNN_ERUNTIME (with a message) when it is non-zero.  The only
 * entry point that waits for the device: meant for tests, the end of a benchmark, or once per epoch. */
int agnn_check_status(const int32_t* status, agnn_stream_t stream);

/* rowend[i] = first position p in [rowptr[i], rowptr[i+1]) with perm[p] >= e_limit (or
 * rowptr[i+1]): the row's edges restricted to the COO prefix [0, e_limit) — PyG trim_to_layer. */
int agnn_csr_rowend(const int32_t* rowptr, const int32_t* perm, int64_t n_rows, int64_t e_limit,
                    int32_t* rowend, agnn_stream_t stream);

/* The same for up to AGNN_ROWEND_MAX_ITEMS (CSR, limit) pairs in ONE launch: a sampled batch needs the row ends of
 * every relation, in both directions, for every trimmed layer (reference models/cadence.py:165-173 trims per layer;
 * R = 4, L = 3: 16 items). */
#define AGNN_ROWEND_MAX_ITEMS 64
typedef struct {
  const int32_t* rowptr;   /* (device) [n_rows + 1] */
  const int32_t* perm;     /* (device) shared with rowptr's offsets */
  int32_t* rowend;         /* (device) [n_rows] out */
  int32_t n_rows;
  int32_t e_limit;
} agnn_rowend_item_t;
int agnn_csr_rowend_batch(int n_items, const agnn_rowend_item_t* items /* (host) */, agnn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Multi-relation segmented gather-reduce ("hetero SpMM").  For every output row i < n_rows and
 * relation r < n_rel:
 *     acc_r(i) = sum_{p in [rowptr_r[i], end_r(i))}  w_r(p) * src_r[ col_r[p], 0:H ]
 *     w_r(p)   = (ew_r ? ew_r[p] : 1) * (colscale_r ? colscale_r[col_r[p]] : 1) * valid(p)
 *     valid(p) = !(SKIP_SELF && col==i) && col < col_limit
 *     cnt_r(i) = number of valid p
 *     val_r(i) = (acc_r(i) + (self ? self[i] : 0)) * (MEAN ? 1 / max(cnt_r(i), 1) : 1)
 * and out[i*ld_out + r*rel_stride + 0:H] (=|+=) val_r(i); with rel_stride == 0 all relations
 * add into the same slot (first one obeys `accumulate`).  `self` reproduces torch_scatter's
 * `out=x.clone()` numerator.  inv_cnt (optional, [n_rel, n_rows]) receives
 * 1/max(cnt,1) for the backward pass.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  const float* src;        /* (device) [n_src, ld_src] */
  const int32_t* rowptr;   /* (device) [n_rows + 1] */
  const int32_t* rowend;   /* (device) [n_rows] or NULL */
  const int32_t* col;      /* (device) base shared with rowptr's offsets */
  const float* ew;         /* (device) per-position weights or NULL */
  const float* colscale;   /* (device) [n_src] or NULL */
  int64_t ld_src;
} agnn_rel_t;

#define AGNN_SPMM_MEAN      1u
#define AGNN_SPMM_SKIP_SELF 2u
#define AGNN_SPMM_ACCUM     4u
#define AGNN_SPMM_ROOT      8u  /* set by agnn_spmm_root_f32 (internal) */
#define AGNN_SPMM_GENERIC 1024u /* never take the specialised fast path (tests / A/B timing) */
#define AGNN_SPMM_FAST_V4 2048u /* fast path: the older one-relation-at-a-time kernel (A/B timing) */

int agnn_spmm_f32(int n_rel, const agnn_rel_t* rels /* (host) */, int64_t n_rows, int32_t H,
                  float* out, int64_t ld_out, int64_t rel_stride,
                  const float* self, int64_t ld_self,
                  float* inv_cnt, int32_t col_limit, uint32_t flags, agnn_stream_t stream);

/* The same aggregation with the ROOT operand of a SAGE layer riding along, so that PyG `SAGEConv`'s
 * `lin_l(mean_j x_j) + lin_r(x_i)` summed over the relations of a destination type (ref: models/cadence.py:147-159,174) is ONE
 * GEMM over [A_1 .. A_R | x_dst] and its backward ONE transposed launch (no slice gradient, no gradient add):
 *   rel_stride != 0 (forward):  out[i, r*rel_stride + 0:H] as agnn_spmm_f32, and out[i, n_rel*rel_stride + 0:H] = root[i]
 *                               (root_rows >= n_rows, ld_out >= n_rel*rel_stride + H);
 *   rel_stride == 0 (backward): out[i] = sum_r (...)  +  (i < root_rows ? root[i] : 0).
 * H must be 256 or 512, no per-edge weights; flags: AGNN_SPMM_MEAN, AGNN_SPMM_SKIP_SELF. */
int agnn_spmm_root_f32(int n_rel, const agnn_rel_t* rels /* (host) */, int64_t n_rows, int32_t H,
                       float* out, int64_t ld_out, int64_t rel_stride,
                       const float* root, int64_t ld_root, int64_t root_rows,
                       float* inv_cnt, int32_t col_limit, uint32_t flags, agnn_stream_t stream);

/* Gradient of agnn_spmm_f32 w.r.t. its `self` operand, optionally added onto an existing gradient:
 *     out[i, 0:H] (=|+=) sum_{r < n_rel} dout[i*ld_dout + r*rel_stride + 0:H] * (inv_cnt ? inv_cnt[r*ld_inv + i] : 1)
 * (rel_stride == 0: every relation read the same slot).  One launch instead of the reduce / multiply / add chain; with
 * `accumulate` it lands on the neighbour gradient the backward SpMM has just written when `self` and the source are the
 * same matrix (onset pooling, ref: models/analysis.py:580-587). */
int agnn_spmm_self_grad_f32(const float* dout, int64_t ld_dout, int64_t rel_stride, int32_t n_rel, const float* inv_cnt,
                            int64_t ld_inv, int64_t n_rows, int32_t H, float* out, int64_t ld_out, int32_t accumulate,
                            agnn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Persistent bidirectional GRU layer (one launch for all T steps), replacing `torch.nn.GRU` in the
 * hybrid branch (ref: models/cadence.py:249-285, models/analysis.py:527-537).  hidden must be 128
 * (the H = 256 configuration); other sizes return AGNN_EINVAL and the host falls back to the
 * library RNN.  Gate order r, z, n and equations are torch's.
 *   gi    [B, T, 2, 3*hidden]   x W_ih^T + b_ih for both directions (library GEMM, done by the caller)
 *   w_hh  [2, 3*hidden, hidden], b_hh [2, 3*hidden]
 *   y     [B, T, 2*hidden]      forward | reverse halves, as nn.GRU(batch_first) returns
 *   saved [B, T, 2, 4, hidden]  r, z, n, W_hn h + b_hn  (kept for the backward pass)
 * Backward: given dy emits dgi [B, T, 2, 3*hidden] (gradient w.r.t. gi) and dgh [B, T, 2, 3*hidden]
 * (gradient w.r.t. W_hh h + b_hh: the r and z blocks equal dgi's, the n block is d n_pre * r);
 * weight/input gradients are GEMMs over those.
 * ------------------------------------------------------------------------------------------ */
int agnn_gru_fwd_f32(const float* gi, const float* w_hh, const float* b_hh, int64_t B, int64_t T,
                     int32_t hidden, float* y, float* saved, const float* drop_scale, float* y_drop,
                     agnn_stream_t stream);
int agnn_gru_bwd_f32(const float* dy, const float* y, const float* saved, const float* w_hh,
                     int64_t B, int64_t T, int32_t hidden, float* dgi, float* dgh, const float* drop_scale, float* hprev,
                     agnn_stream_t stream);
/* hprev (optional) [B, T, 2, hidden]: the backward walk reads h_{t-1} anyway and leaves it here — the right operand of the
 * W_hh weight-gradient product (the same matrix agnn_gru_hprev_f32 builds in a pass of its own). */
/* Inter-layer dropout of `nn.GRU(dropout=p)` (ref: models/cadence.py:249-251) rides along: drop_scale [B, T, 2*hidden]
 * holds 0 or 1 / (1 - p) per element; forward additionally writes y_drop = y * drop_scale (the next layer's input),
 * backward takes dy = d(y_drop) and applies the same factor.  Both NULL: no dropout. */
/* hp [B, T, 2, hidden]: the previous hidden state of each direction (forward: y[b, t-1, :hidden], reverse:
 * y[b, t+1, hidden:], zero at the sequence ends) — the operand of the W_hh weight-gradient GEMMs, in one launch. */
int agnn_gru_hprev_f32(const float* y, int64_t B, int64_t T, int32_t hidden, float* hp, agnn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * HGT edge-softmax attention = message / softmax / aggregate of PyG `HGTConv`, reached through
 * graphmuse `HybridHGT` (ref: models/analysis.py:445-453; semantics SURVEY.md App. A.4).
 * For destination row i, head h (D = H / heads floats per head), over ALL incoming edges e of all
 * n_rel relations:   s_e = <q_i,h, k_r[col_e],h> * pscale_r[h]      (pscale = p_rel / sqrt(D))
 *                    a_e = exp(s_e - max) / (sum exp(s_e - max) + 1e-16),   out_i,h = sum_e a_e v_r[col_e],h
 * k_r / v_r are the relation-transformed keys / values of the source type (dense GEMMs by the caller).
 * m_out / linv_out [n_rows, heads] keep the row max and 1/(sum + 1e-16) for the backward pass.
 * D must be 4*2^k, <= 256.
 * Backward, pass by destination: dq plus per-edge alpha, gs = ds*pscale, tdot = ds*<q,k> written at the
 * edge's COO position perm[p] into [n_edges, heads] arrays of its relation.
 * Backward, pass by source (one relation, transposed CSR, col = destination row, rows >= col_limit
 * skipped): dv[j] = sum_e alpha_e dm[col_e], dk[j] = sum_e gs_e q[col_e].
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  const float* k;          /* (device) [n_src, ld] */
  const float* v;          /* (device) [n_src, ld] */
  const int32_t* rowptr;   /* (device) [n_rows + 1] */
  const int32_t* rowend;   /* (device) [n_rows] or NULL */
  const int32_t* col;      /* (device) */
  const int32_t* perm;     /* (device) COO position of every CSR position (backward only) */
  const float* pscale;     /* (device) [heads] */
  int64_t ld;
  float* alpha;            /* (device) [n_edges, heads]  backward outputs */
  float* gs;               /* (device) [n_edges, heads] */
  float* tdot;             /* (device) [n_edges, heads], or head-major [heads, ld_tdot] when ld_tdot > 0 */
  int64_t ld_tdot;         /* 0: edge-major tdot; > 0: tdot[h * ld_tdot + e] (a contiguous sum over the edges per head) */
} agnn_hgt_rel_t;

int agnn_hgt_attn_fwd_f32(int n_rel, const agnn_hgt_rel_t* rels /* (host) */, const float* q, int64_t ld_q,
                          int64_t n_rows, int32_t H, int32_t heads, float* out, int64_t ld_out,
                          float* m_out, float* linv_out, agnn_stream_t stream);
/* The attention of several DESTINATION types in one launch (an HGT layer has one per destination type; ref. as above): item i =
 * one agnn_hgt_attn_fwd_f32 call.  Items and their relation tables are read on the host during the call; at most
 * AGNN_HGT_MAX_DST items, AGNN_MAX_SEG relations in all. */
#define AGNN_HGT_MAX_DST 4
typedef struct {
  const agnn_hgt_rel_t* rels;   /* (host) n_rel relations ending in this destination type */
  int32_t n_rel;
  const float* q;
  int64_t ld_q;
  int64_t n_rows;
  float* out;
  int64_t ld_out;
  float* m_out;
  float* linv_out;
} agnn_hgt_dst_item_t;
int agnn_hgt_attn_fwd_multi_f32(int32_t n_items, const agnn_hgt_dst_item_t* items /* (host) */, int32_t H, int32_t heads,
                                agnn_stream_t stream);
int agnn_hgt_attn_bwd_dst_f32(int n_rel, const agnn_hgt_rel_t* rels /* (host) */, const float* q, int64_t ld_q,
                              const float* dm, int64_t ld_dm, const float* out, int64_t ld_out,
                              const float* m_in, const float* linv_in, int64_t n_rows, int32_t H,
                              int32_t heads, float* dq, int64_t ld_dq, agnn_stream_t stream);
int agnn_hgt_attn_bwd_src_f32(const int32_t* rowptr, const int32_t* rowend, const int32_t* col,
                              const int32_t* perm, const float* alpha, const float* gs, const float* q,
                              int64_t ld_q, const float* dm, int64_t ld_dm, int64_t n_src_rows,
                              int32_t col_limit, int32_t H, int32_t heads, float* dk, float* dv,
                              int64_t ld_o, agnn_stream_t stream);
/* The same for up to AGNN_MAX_SEG relations in one launch — all relations ending in one destination type share q and dm; item i
 * writes its own dk / dv (a column block of its source type's gradient).  Items are read on the host during the call. */
typedef struct {
  const int32_t* rowptr;
  const int32_t* rowend;      /* NULL: rowptr + 1 */
  const int32_t* col;
  const int32_t* perm;
  const float* alpha;
  const float* gs;
  float* dk;
  float* dv;
  int64_t ld_o;
  int32_t n_src_rows;
  int32_t col_limit;
} agnn_hgt_src_item_t;
int agnn_hgt_attn_bwd_src_batch_f32(int32_t n_items, const agnn_hgt_src_item_t* items /* (host) */, const float* q, int64_t ld_q,
                                    const float* dm, int64_t ld_dm, int32_t H, int32_t heads, agnn_stream_t stream);


/* ------------------------------------------------------------------------------------------
 * Edge-gated aggregation of the in-tree ResGatedGraphConv (ref: models/core/gnn.py:243-258):
 *     S_i = sum_{p in row i} sigmoid(a_i + b_col[p] [+ c_perm[p]]) * h_col[p]
 * a, b, h are [*, ld] fp32 matrices (W3 x, W4 x, W2 x), c an optional per-edge [E, ld_c] term in COO order.
 * fwd / bwd_dst take the CSR grouped by the aggregating row (a indexed by row; b, h by col);
 * bwd_src takes the TRANSPOSED CSR (b, h indexed by row; a and dS by col).  Same a/b/h/c pointers in all calls.
 *   bwd_dst: da_i = sum_e dS_i h_j z(1-z)   and, when dc != NULL, dc[perm[p]] = that summand
 *   bwd_src: db_j = sum_e dS_i h_j z(1-z),  dh_j = sum_e z dS_i
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  const int32_t* rowptr;  /* (device) [n_rows + 1] */
  const int32_t* col;     /* (device) */
  const int32_t* perm;    /* (device) COO position per CSR position (needed when c != NULL and in backward) */
  const float* a;
  const float* b;
  const float* h;
  const float* c;         /* (device) [n_edges, ld_c] or NULL */
  int64_t ld, ld_c;
  int64_t n_rows;
  int32_t H;
} agnn_gated_t;

int agnn_gated_fwd_f32(const agnn_gated_t* g /* (host) */, float* out, int64_t ld_out, agnn_stream_t stream);
int agnn_gated_bwd_dst_f32(const agnn_gated_t* g /* (host) */, const float* ds, int64_t ld_ds, float* da,
                           float* dc, agnn_stream_t stream);
int agnn_gated_bwd_src_f32(const agnn_gated_t* g /* (host) */, const float* ds, int64_t ld_ds, float* db,
                           float* dh, agnn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * fp32 projection GEMM on the matrix cores (v_mfma_f32_32x32x2_f32, exact fp32):  C[M, N] = A[M, K] * W[N, K]^T (+ bias[N])
 * — `nn.Linear` / PyG SAGEConv's lin_l, lin_r on a tall activation matrix (ref: models/cadence.py:147-159, core/gnn.py:65,75).
 * agnn_gemm_nt_f32: both operands K-contiguous (the forward product with the weight [out, in] as it lies in memory);
 * agnn_gemm_nn_f32: the second operand K-MAJOR, w[K, N] — C[M, N] = A[M, K] * w[K, N] (+ bias): the input-gradient product
 * dX = dY * W with the same weight [out = K, in = N], no transposed copy.  N % 64 == 0, K % 16 == 0, 16-byte aligned rows
 * (ld % 4 == 0, covering a row of their operand).  linear.HAND_GEMM sends the step's projections with >= 4 096 rows here.
 * ------------------------------------------------------------------------------------------ */
int agnn_gemm_nn_f32(const float* a, int64_t ld_a, const float* w /* [K, N] */, int64_t ld_w, const float* bias, int64_t M, int32_t N,
                     int32_t K, float* c, int64_t ld_c, agnn_stream_t stream);
int agnn_gemm_nt_f32(const float* a, int64_t ld_a, const float* w, int64_t ld_w, const float* bias, int64_t M, int32_t N,
                     int32_t K, float* c, int64_t ld_c, agnn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Aggregation side of the in-tree RelEdgeConv (ref: models/core/gnn.py:99-105: m_ij = W_e [h_j || |h_i - h_j|] + b_e scattered
 * with mean onto row i).  W_e is linear, so sum_j m_ij = W_e [S_i || D_i] + deg_i b_e with the two per-row aggregates
 *     S_i = sum_{p in row i} h_col[p] ,    D_i = sum_{p in row i} |h_i - h_col[p]|
 * computed here in one pass (either output may be NULL); the edge GEMM [E, 2F] x [2F, F] becomes a node GEMM.
 * Backward, one launch per CSR direction, the second with accumulate = 1 onto the first's output:
 *     by_col = 0 (the forward CSR):      out_r (+)= gd_r * sum_p sign(h_r - h_col[p])
 *     by_col = 1 (the transposed CSR):   out_r (+)= sum_p [ sign(h_r - h_col[p]) * gd_col[p] + gs_col[p] ]
 * (gd / gs: gradients of D / S, either may be NULL; sign(0) = 0 as torch.abs' backward.)  H % 4 == 0, H <= 1024.
 * ------------------------------------------------------------------------------------------ */
int agnn_absdiff_fwd_f32(const int32_t* rowptr, const int32_t* col, const float* h, int64_t ld, int64_t n_rows, int32_t H,
                         float* out_s, float* out_d, int64_t ld_out, agnn_stream_t stream);
int agnn_absdiff_bwd_f32(const int32_t* rowptr, const int32_t* col, const float* h, int64_t ld, int64_t n_rows, int32_t H,
                         const float* gd, const float* gs, int64_t ld_g, int32_t by_col, int32_t accumulate, float* out,
                         int64_t ld_out, agnn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Fused [ReLU ->] LayerNorm [-> ReLU] [-> dropout] over the rows of x [n, H] (the element-wise chains between the
 * projections, ref: models/analysis.py:429-443, :474-485; models/cadence.py:252-259).  Saves mean / rstd [n];
 * backward recomputes the ReLU masks from x and the dropout mask from the counter-based generator:
 * Philox-4x32-10(seed = rng_state[0], counter = (element, call_id, step = rng_state[1])), rng_state a DEVICE
 * int64[2] so a captured graph draws new masks when the caller bumps the step between replays.  The forward call copies
 * the (seed, step) pair it used into `rng_used` (DEVICE int64[2], may be NULL): hand THAT to the backward call as its
 * rng_state — the live counter may have been bumped by another forward in between (two batches with a joint backward,
 * gradient accumulation, the reference's memory replay: models/analysis.py:1064-1066).
 * `seg`: the statistics are taken over segments of `seg` consecutive floats of a row (seg = H: plain LayerNorm;
 * seg = 64 with H = T*64: the T task heads' LayerNorms of one note in one pass, gamma / beta being the [T*64]
 * concatenation of the per-task affines).  seg must be 4*2^k, divide 256 and divide H.  mean / rstd: [n, H/seg].
 * ------------------------------------------------------------------------------------------ */
#define AGNN_NA_PRE_RELU  1u
#define AGNN_NA_POST_RELU 2u
size_t agnn_norm_act_workspace_bytes(int32_t H);
int agnn_norm_act_fwd_f32(const float* x, int64_t ld_x, const float* gamma, const float* beta, int32_t seg,
                          int64_t n, int32_t H, float eps, float p, uint32_t flags, const int64_t* rng_state, uint32_t call_id, float* y,
                          int64_t ld_y, float* mean, float* rstd, int64_t* rng_used, agnn_stream_t stream);
int agnn_norm_act_bwd_f32(const float* x, int64_t ld_x, const float* gamma, const float* beta, int32_t seg,
                          int64_t n, int32_t H, float eps, float p, uint32_t flags, const int64_t* rng_state, uint32_t call_id,
                          const float* dy, int64_t ld_dy, const float* mean, const float* rstd, float* dx,
                          int64_t ld_dx, float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes,
                          agnn_stream_t stream);

/* agnn_norm_act_bwd_f32 with dgamma == dbeta == NULL leaves the per-block partial sums in `workspace` and launches only the
 * input-gradient kernel; this entry point turns them into dgamma / dbeta later (same n, H, workspace): the two parameter
 * gradients only feed the optimizer and need not sit on the backward pass's dependent chain. */
/* Several pending column sums (agnn_norm_act_colsum_f32) in one launch: they only feed the optimizer and are pending together at
 * the encoder's flush points.  At most 16 items. */
typedef struct {
  const void* workspace;   /* what agnn_norm_act_bwd_f32 (dgamma == dbeta == NULL) left */
  size_t workspace_bytes;
  int64_t n;
  int32_t H;
  float* dgamma;
  float* dbeta;
} agnn_colsum_item_t;
int agnn_norm_act_colsum_batch_f32(int32_t n_items, const agnn_colsum_item_t* items /* (host) */, agnn_stream_t stream);

/* HGT layer epilogue in one launch each way (round 3):  z = dropout_p( relu?( x + sigmoid(*skip) * (o - x) ) )  — PyG HGTConv's
 * learnable skip connection (alpha = sigmoid(skip[node type])) followed by the ReLU + dropout the encoders put between layers
 * (graphmuse HybridHGT via models/analysis.py:445-453).  x == skip == NULL: no skip connection, z = act(o).  flags bit 0 = ReLU.
 * Dropout as agnn_norm_act_* (counter-based masks; `rng_used` receives the (seed, step) this call drew from, backward takes it).
 * Backward recomputes both masks, writes d o (= alpha g), d x (= (1 - alpha) g, optional) and d skip (optional; needs
 * `workspace`: agnn_skip_act_workspace_bytes(), 256-byte aligned, ZERO-FILLED once by the caller — every call leaves it so). */
size_t agnn_skip_act_workspace_bytes(void);
int agnn_skip_act_fwd_f32(const float* x, int64_t ld_x, const float* o, int64_t ld_o, const float* skip, int64_t n, int32_t H, float p,
                          uint32_t flags, const int64_t* rng_state, uint32_t call_id, float* z, int64_t ld_z, int64_t* rng_used,
                          agnn_stream_t stream);
int agnn_skip_act_bwd_f32(const float* x, int64_t ld_x, const float* o, int64_t ld_o, const float* skip, int64_t n, int32_t H, float p,
                          uint32_t flags, const int64_t* rng_state, uint32_t call_id, const float* dz, int64_t ld_dz, float* dx,
                          int64_t ld_dx, float* dout, int64_t ld_do, float* dskip, void* workspace, size_t workspace_bytes,
                          agnn_stream_t stream);
int agnn_norm_act_colsum_f32(const void* workspace, size_t workspace_bytes, int64_t n, int32_t H, float* dgamma, float* dbeta,
                             agnn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Weight-gradient GEMM of a projection layer y = x W^T + b over N rows (N large, out/in small):
 *     dw[out, in] = sum_n dy[n, out] * x[n, in]        db[out] = sum_n dy[n, out]   (db may be NULL)
 * fp32-input MFMA, split along N into slabs in `workspace` (agnn_wgrad_workspace_bytes), summed in a fixed
 * order.  Replaces the backward of the `nn.Linear` / PyG `lin_l`, `lin_r` projections on the path
 * (ref: models/analysis.py:429-443,474-485; models/cadence.py:147-159).  out, in and all leading dimensions
 * must be even, pointers 8-byte aligned.
 * ------------------------------------------------------------------------------------------ */
size_t agnn_wgrad_workspace_bytes(int64_t n, int32_t out_f, int32_t in_f);
int agnn_wgrad_f32(const float* dy, int64_t ld_dy, const float* x, int64_t ld_x, int64_t n, int32_t out_f,
                   int32_t in_f, float* dw, int64_t ld_dw, float* db, void* workspace, size_t workspace_bytes,
                   agnn_stream_t stream);
/* Several weight gradients in ONE launch pair (product kernel + slab reduction): the items' workgroups fill the chip together, so
 * every item gets by with a few row slices instead of up to 64 (fewer slabs written and read back, longer K loops) and the
 * step's ~20 reduction launches become one per group.  Same arithmetic per item as agnn_wgrad_f32 with that slice count
 * (deterministic; the slice count — hence the summation order — depends on the group's composition).  At most 16 items. */
typedef struct {
  const float* dy;   /* (device) [n, out_f], leading dimension ld_dy */
  const float* x;    /* (device) [n, in_f],  leading dimension ld_x  */
  float* dw;         /* (device) out [out_f, in_f], leading dimension ld_dw */
  float* db;         /* (device) out [out_f] or NULL */
  int64_t ld_dy, ld_x, ld_dw, n;
  int32_t out_f, in_f;
} agnn_wgrad_item_t;
size_t agnn_wgrad_batch_workspace_bytes(int32_t n_items, const agnn_wgrad_item_t* items /* (host) */);
int agnn_wgrad_batch_f32(int32_t n_items, const agnn_wgrad_item_t* items /* (host) */, void* workspace, size_t workspace_bytes,
                         agnn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Batched small 2-D gather / sum, one launch for many parameter-sized pieces:
 *     dst_i[r, c] = sum_{k < n_src_i} src_{i,k}[r, c]        r < rows_i, c < cols_i
 * Used to present per-relation parameters as one GEMM operand (PyG HeteroConv over SAGEConv: the lin_l weights side by
 * side, lin_r weights and biases summed — ref: models/cadence.py:147-159,174) and to split / fan out the operand's
 * gradient again.  All sources of an item share ld_src.  Destinations of different items must not overlap.
 * n_src_i = 0 zero-fills the piece (ld_src ignored).
 * `vec_ok` is filled in by the library.  Items are read on the host during the call.
 * ------------------------------------------------------------------------------------------ */
#define AGNN_PACK_MAX_SRC   8
#define AGNN_PACK_MAX_ITEMS 24
typedef struct {
  float* dst;                               /* (device) */
  const float* src[AGNN_PACK_MAX_SRC];      /* (device) */
  int64_t ld_dst, ld_src;
  int32_t rows, cols, n_src, vec_ok;
} agnn_pack_item_t;
int agnn_pack_f32(int32_t n_items, const agnn_pack_item_t* items /* (host) */, agnn_stream_t stream);

/* Measurement aid (scripts/step_stamps.py): one single-lane kernel that stores the constant-rate 100 MHz counter
 * (s_memrealtime) into *slot when the stream reaches it — a time stamp that can be captured into a hipGraph and read back
 * after the replay, without a profiler attached.  Not used by the product path. */
int agnn_debug_stamp(uint64_t* slot /* (device) */, agnn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Note-input assembly (ref: models/analysis.py:574 `torch.cat([x_dict["note"], self.pitch_embedding(pitch_spelling),
 * self.key_embedding(key_signature)], dim=-1)`; tables models/analysis.py:399-400) and the gradient of the tables.
 *   forward   out[n, :] = [ x[n, 0:in_x] | tables[0][idx[0][n], :] | ... | tables[n_tab-1][idx[n_tab-1][n], :] | 0 ... ]
 *             x [n_rows, in_x] (ld_x), idx[t] DEVICE int64 [n_rows] (ids outside [0, vocab[t]) are clamped — torch raises
 *             a device assert there), tables[t] DEVICE [vocab[t], dim] contiguous, out [n_rows, ld_out] with
 *             ld_out >= in_x + n_tab*dim; the columns behind the last table are zero-filled.
 *   backward  dtables[vbase_t + v, :] = sum_{n: idx[t][n] == v} dout[n, col0 + t*dim : col0 + (t+1)*dim]  with
 *             vbase_t = vocab[0] + ... + vocab[t-1]; dtables DEVICE [sum vocab, dim] contiguous; dim even.  Rows are
 *             added in row order inside 32 row slices that are summed in a fixed order (no atomics, reproducible).
 * `idx`, `tables`, `vocab` are HOST arrays (of device pointers / sizes) read during the call.
 * ------------------------------------------------------------------------------------------ */
#define AGNN_EMBED_MAX_TABLES 4
int agnn_embed_cat_fwd_f32(const float* x, int64_t ld_x, int32_t in_x, int64_t n_rows, int32_t n_tab,
                           const int64_t* const* idx, const float* const* tables, const int32_t* vocab, int32_t dim,
                           float* out, int64_t ld_out, agnn_stream_t stream);
size_t agnn_embed_workspace_bytes(int32_t n_tab, const int32_t* vocab, int32_t dim);
int agnn_embed_cat_bwd_f32(const float* dout, int64_t ld_dout, int32_t col0, int64_t n_rows, int32_t n_tab,
                           const int64_t* const* idx, const int32_t* vocab, int32_t dim, float* dtables,
                           void* workspace, size_t workspace_bytes, agnn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * The task-head block in one launch (ref: models/analysis.py:486-496 `clf_dict[task] = Linear(o, o/2) -> ReLU ->
 * LayerNorm(o/2) -> Linear(o/2, C_t)`, :546-548): for every task t
 *     z_t = x W1_t^T + b1_t;   y_t = LayerNorm_t(ReLU(z_t));   logits[:, offs[t]:offs[t+1]] = y_t W2_t^T + b2_t
 *   x [n_rows, in_f] (ld_x), w1 [T*hidden, in_f] / b1, gamma, beta [T*hidden] = the tasks' parameters stacked along rows,
 *   w2 [sum_c, hidden] / b2 [sum_c] (b2 may be NULL), offs_dev DEVICE int32 [T+1], offs_host the same on the HOST (it sizes
 *   the grid and deals the tasks to workgroups).  Written for the backward pass (agnn_norm_act_bwd_f32 with seg = hidden,
 *   agnn_gproj_bwd_f32): z, y [n_rows, T*hidden] (ld_h), mean / rstd [n_rows, T].  This build: in_f = 128, hidden = 64, T <= 64.
 * ------------------------------------------------------------------------------------------ */
int agnn_heads_fwd_f32(const float* x, int64_t ld_x, int64_t n_rows, int32_t in_f, int32_t hidden, int32_t T, const float* w1,
                       const float* b1, const float* gamma, const float* beta, float eps, const float* w2, const float* b2,
                       const int32_t* offs_dev, const int32_t* offs_host, float* z, float* y, int64_t ld_h, float* mean,
                       float* rstd, float* logits, int64_t ld_o, agnn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Grouped projection: the last Linear(h2 -> C_t) of all task heads in one launch per direction
 * (ref: models/analysis.py:486-496 `clf_dict[task]`, :546-548).  Group g reads columns [g*K, (g+1)*K) of a and
 * owns output columns [seg_off[g], seg_off[g+1]) (the same side-by-side logits layout agnn_multitask_ce_f32 takes):
 *     out[n, seg_off[g] + c] = b[seg_off[g] + c] + sum_k a[n, g*K + k] * w[seg_off[g] + c, k]
 *   a [n_rows, n_groups*K] (ld_a), w [sum_c, K] = the per-task weights stacked along rows, b [sum_c] or NULL,
 *   seg_off DEVICE int32 [n_groups + 1], n_tiles32 = sum_g ceil(C_g / 32) (host-side count that sizes the grid),
 *   K in {32, 64, 128}.  Backward: da [n_rows, n_groups*K] (NULL to skip), dw [sum_c, K] and db [sum_c] (dw NULL to
 *   skip both; db may be NULL); dw/db are summed over row slices held in `workspace` in a fixed order.
 * ------------------------------------------------------------------------------------------ */
int agnn_gproj_fwd_f32(const float* a, int64_t ld_a, const float* w, const float* b, const int32_t* seg_off,
                       int32_t n_groups, int32_t K, int32_t n_tiles32, int64_t n_rows, float* out, int64_t ld_out,
                       agnn_stream_t stream);
size_t agnn_gproj_workspace_bytes(int64_t n_rows, int32_t sum_c, int32_t K, int32_t n_tiles32);
int agnn_gproj_bwd_f32(const float* dout, int64_t ld_dout, const float* a, int64_t ld_a, const float* w,
                       const int32_t* seg_off, int32_t n_groups, int32_t K, int32_t n_tiles32, int32_t sum_c,
                       int64_t n_rows, float* da, int64_t ld_da, float* dw, float* db, void* workspace,
                       size_t workspace_bytes, agnn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Fused multi-task cross entropy (label smoothing, ignore index) over column segments of one logits
 * matrix: replaces the per-task `nn.CrossEntropyLoss(ignore_index=-1, label_smoothing=0.1)` terms
 * (ref: models/analysis.py:881-888, models/chord.py:39-49).  Task t owns columns [seg_off[t], seg_off[t+1]).
 *   labels    int64 [n_tasks, n_rows]
 *   row_loss  [n_tasks, n_rows]  per-row loss terms (0 for ignored rows)                      (scratch output)
 *   dlogits   [n_rows, ld]       d loss_row / d logits, NOT yet divided by the task's row count (ignored rows: 0;
 *                                columns outside the segments untouched)
 *   loss      [n_tasks]          mean loss per task = sum_n row_loss[t][n] / max(count_t, 1)
 *   inv_count [n_tasks]          1 / max(count_t, 1), count_t = rows with label != ignore_index
 * agnn_multitask_ce_scale_f32: out[n, c] = dlogits[n, c] * scale[task(c)] for c < n_cols (columns outside every
 * segment are copied) — the backward pass with scale[t] = inv_count[t] * (incoming gradient of loss[t]).
 * ------------------------------------------------------------------------------------------ */
int agnn_multitask_ce_f32(const float* logits, int64_t ld, const int32_t* seg_off, int32_t n_tasks,
                          const int64_t* labels, int64_t n_rows, float label_smoothing,
                          int64_t ignore_index, float* row_loss, float* dlogits, float* loss, float* inv_count,
                          agnn_stream_t stream);
int agnn_multitask_ce_scale_f32(const float* dlogits, int64_t ld, const int32_t* seg_off, int32_t n_tasks,
                                int64_t n_rows, int32_t n_cols, const float* scale, float* out, int64_t ld_out,
                                agnn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Training loss of one step in two launches, its gradient in one:
 *     total = ce_scale * sum_t (w_t * loss[t] + reg_t) + lambda_feat * mean(feat^2)
 *     task_param == NULL: w_t = 1, reg_t = 0;  else p = task_param[t]: w_t = 0.5 / p^2, reg_t = log(1 + p^2)
 * i.e. the per-task label-smoothed cross entropies above, combined as the reference's MultiTaskLoss does (learned
 * uncertainty weights, ref models/chord.py:39-49; selected by the CLI default --mt_strategy wloss, models/analysis.py:
 * 899-908), divided by the number of tasks (ce_scale = 1/T: models/analysis.py:1036), plus the feature-norm term
 * (models/analysis.py:984 `feature_loss = x.pow(2).mean()`, :1072 `... + feature_loss * self.lambda_featl`, :910 default
 * 0.1).  Same per-task outputs as agnn_multitask_ce_f32, plus `total` (device scalar), `wscale[t]` = ce_scale * w_t *
 * inv_count[t] (the per-task gradient scale: hand it to agnn_train_loss_bwd_f32 as its `inv_count`) and `dparam[t]` =
 * d total / d p_t = ce_scale * (2 p / (1 + p^2) - loss[t] / p^3) (0 without task_param); wscale / dparam may be NULL.
 * A task without any valid row contributes loss 0 (torch: NaN).  feat [n_rows, feat_cols] (ld_feat) may be NULL (no
 * feature term).  `workspace` (agnn_train_loss_workspace_bytes(), 256-byte aligned) must be ZERO-FILLED before its first
 * use; every call leaves its ticket zero again (it orders the fixed-order final sum).  Backward, given g =
 * d(objective)/d(total) as a DEVICE scalar:
 *     out[n, c]   = dlogits[n, c] * g * inv_count[task(c)]          (columns outside the segments: 0)
 *     dfeat[n, c] = g * 2 * lambda_feat / (n_rows * feat_cols) * feat[n, c]        (dfeat NULL to skip)
 * ------------------------------------------------------------------------------------------ */
size_t agnn_train_loss_workspace_bytes(void);
int agnn_train_loss_f32(const float* logits, int64_t ld, const int32_t* seg_off, int32_t n_tasks, const int64_t* labels,
                        int64_t n_rows, float label_smoothing, int64_t ignore_index, const float* feat, int64_t ld_feat,
                        int32_t feat_cols, float lambda_feat, const float* task_param, float ce_scale, float* row_loss,
                        float* dlogits, float* loss, float* inv_count, float* total, float* wscale, float* dparam,
                        void* workspace, size_t workspace_bytes, agnn_stream_t stream);
/* The same objective with the gradients FINISHED in the forward launches (for an incoming gradient of 1, which is what
 * `total.backward()` passes): wscale[t] is computed first (a count of the task's valid labels: one small launch), the
 * cross-entropy kernel then writes dlogits[n, c] = wscale[task(c)] * (p - target) directly, and the launch that reduces the
 * losses also writes dfeat[n, c] = 2 * lambda_feat / (n_rows * feat_cols) * feat[n, c] (dfeat may be NULL).  The backward pass
 * then has no launch of its own (agnn_train_loss_bwd_f32's scale pass re-read and re-wrote the [N, C] matrix: 23 us at C2 on
 * the step's serial stretch); a caller with another incoming gradient g multiplies the three gradients by g.  Same
 * reference lines, outputs, workspace contract and error behaviour as agnn_train_loss_f32; wscale is required here. */
int agnn_train_loss_final_f32(const float* logits, int64_t ld, const int32_t* seg_off, int32_t n_tasks, const int64_t* labels,
                              int64_t n_rows, float label_smoothing, int64_t ignore_index, const float* feat, int64_t ld_feat,
                              int32_t feat_cols, float lambda_feat, const float* task_param, float ce_scale, float* row_loss,
                              float* dlogits, float* loss, float* inv_count, float* total, float* wscale, float* dparam,
                              float* dfeat, int64_t ld_dfeat, void* workspace, size_t workspace_bytes, agnn_stream_t stream);
int agnn_train_loss_bwd_f32(const float* dlogits, int64_t ld, const int32_t* seg_off, int32_t n_tasks, int64_t n_rows,
                            int32_t n_cols, const float* inv_count, const float* g, float* out, int64_t ld_out,
                            const float* feat, int64_t ld_feat, int32_t feat_cols, float lambda_feat, float* dfeat,
                            int64_t ld_dfeat, agnn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Device-side batch assembly: neighbour sampling of note windows out of score graphs resident in device memory
 * (replaces graphmuse's `MuseNeighborLoader` + collation + host-to-device copy, ref data/datamodules/analysis.py:270-293:
 * subgraph_size target notes per window, num_neighbors per hop, batch_size windows; PyG NeighborLoader layout consumed at
 * models/analysis.py:948-961).  One launch fills buffers of STATIC shape (hop blocks padded to capacity), so the sampled
 * batch — and with it the whole training step — is a fixed launch sequence a hipGraph can replay:
 *   nodes  [ n_sub*n_targets | n_sub*cap[0] | n_sub*cap[1] | ... ]   node_gid = global note id, -1 in unused slots; a
 *          subgraph's new nodes of a hop take its slots in ascending global id;
 *   edges[r]  int64 [2, e_cap] (row 0 = source, row 1 = target, batch-local ids; (-1, -1) in unused slots, which
 *          agnn_csr_build drops): hop h owns n_sub * F_h * fan[h] slots, F_1 = n_targets, F_h = cap[h-2]; slot
 *          ((s*F_h + i)*fan[h] + k) = k-th sampled in-neighbour of frontier node i of subgraph s.
 * rowptr / col: CSR by DESTINATION of every relation over all notes of all scores (int32).  All in-neighbours are taken
 * when there are at most fan[h]; otherwise fan[h] of them without replacement (selection sampling on Philox-4x32-10 keyed
 * by rng = (seed, step) and (destination, relation, hop)).  Hop h+1 expands only the nodes hop h added (PyG semantics).
 * `drops` (optional device int32[1], caller-owned, only ever incremented) counts sources dropped because a hop proposed
 * more than cap[h] new nodes — a STATISTIC, expected on crowded scores: the cap[h] smallest global ids are kept, whatever
 * their number, and a dropped source stays dropped in later hops.  `status` (optional, the agnn_csr_build word) is bumped
 * only by an ERROR: a subgraph proposed more distinct out-of-window sources than the kernel's table holds (4096); the
 * sources that found no place are left out of the batch and agnn_check_status reports it.
 * agnn_gather_rows_f32 / agnn_gather_i64 fetch the batch's feature rows / integer attributes by node_gid (0 / `fill` for
 * padding slots).
 * ------------------------------------------------------------------------------------------ */
#define AGNN_SAMPLER_MAX_REL 8
#define AGNN_SAMPLER_MAX_HOPS 4
#define AGNN_SAMPLER_MAX_FAN 16
#define AGNN_SAMPLER_MAX_CAP 256
typedef struct {
  int32_t n_rel;
  const int32_t* rowptr[AGNN_SAMPLER_MAX_REL];   /* (device) [n_notes_total + 1] */
  const int32_t* col[AGNN_SAMPLER_MAX_REL];      /* (device) source note of every in-edge */
  const int32_t* win_start;                      /* (device) [n_sub] global id of each window's first target note */
  int32_t n_sub, n_targets, n_hops;
  int32_t fan[AGNN_SAMPLER_MAX_HOPS];
  int32_t cap[AGNN_SAMPLER_MAX_HOPS];
  const int64_t* rng;                            /* (device) int64[2]: seed, step */
  int32_t* node_gid;                             /* (device) out [agnn_sampler_num_nodes] */
  int64_t* edges[AGNN_SAMPLER_MAX_REL];          /* (device) out, per relation int64 [2, e_cap] */
  int64_t e_cap;                                 /* = agnn_sampler_edge_capacity */
  int32_t* status;                               /* (device) int32[1] or NULL: errors (agnn_check_status) */
  int32_t* drops;                                /* (device) int32[1] or NULL: sources cut by cap[h] (statistic) */
  int32_t* kept;                                 /* (device) int32[n_hops * n_sub] or NULL: nodes kept per (hop, subgraph) */
} agnn_sampler_t;
int64_t agnn_sampler_num_nodes(const agnn_sampler_t* cfg /* (host) */);
int64_t agnn_sampler_edge_capacity(const agnn_sampler_t* cfg /* (host) */);
int agnn_sample_hops(const agnn_sampler_t* cfg /* (host) */, agnn_stream_t stream);
/* Metrical nodes of a sampled batch (the reference's graphs carry beat and measure nodes unless `remove_beats/measures` is set,
 * ref data/datamodules/analysis.py:213-225; every note belongs to one beat and one measure, hgraph edge type
 * (note, connects, beat|measure)).  `group_of` (device int32 [n_notes_total]): the global id of the note's group, non-decreasing
 * inside a score (notes are sorted by onset), so the groups of a window's target notes are ONE contiguous range
 * [group_of[w], group_of[w + n_targets - 1]] — no set, no sort.  After agnn_sample_hops (same layout arguments):
 *   group_gid [n_sub * cap_g]   subgraph s owns slots [s*cap_g, (s+1)*cap_g): its range in ascending id, -1 in unused slots
 *                               (`drops`, optional, counts groups beyond the capacity);
 *   edges int64 [2, n_nodes]    slot i = batch note i: (i, batch-local group slot) when the note's group lies in its subgraph's
 *                               range, else (-1, -1).  Slot order = node order, so the per-hop edge counts that drive
 *                               trim_to_layer are [n_sub*(n_targets + cap[0]), n_sub*cap[1], ...]: an edge is trimmed with
 *                               its source note's hop block; the group nodes themselves are all hop-0 (never trimmed). */
int agnn_sample_members(const int32_t* node_gid, int64_t n_nodes, const int64_t* batch /* NULL, or (pool layout) subgraph id per slot */,
                        const int32_t* group_of, const int32_t* win_start, int32_t n_sub, int32_t n_targets, int32_t n_hops,
                        const int32_t* cap /* (host) [n_hops] */, int32_t cap_g, int32_t* group_gid, int64_t* edges, int32_t* drops,
                        agnn_stream_t stream);
/* The padded hop blocks [n_sub x cap[h]] squeezed into batch-wide POOLS of pool[h] <= n_sub * cap[h] slots (a subgraph rarely
 * fills its capacity: at C2 170 of 2 048 slots are used): subgraph s's kept nodes of hop h move to
 * pool_base[h] + sum_{s' < s} kept[h][s'] + rank — still hop-ordered, still static shapes (trim counts = the pool sizes), a
 * subgraph's nodes still contiguous.  Run after agnn_sample_hops with the same configuration (`kept` set): writes the node ids
 * in pool layout (agnn_sample_compact_nodes slots) and the subgraph id of every slot, and rewrites the endpoints of every edge
 * slot IN PLACE; nodes past a pool's end are dropped with their edges and counted in `drops`. */
int64_t agnn_sample_compact_nodes(const agnn_sampler_t* cfg /* (host) */, const int32_t* pool /* (host) [n_hops] */);
int agnn_sample_compact(const agnn_sampler_t* cfg /* (host) */, const int32_t* pool /* (host) [n_hops] */, int32_t* node_gid_out,
                        int64_t* batch_out, agnn_stream_t stream);
int agnn_gather_rows_f32(const float* src, int64_t ld_src, const int32_t* gid, int64_t n, int32_t H, float* out,
                         int64_t ld_out, agnn_stream_t stream);
int agnn_gather_i64(const int64_t* src, int64_t ld_src, const int32_t* gid, int64_t n, int32_t n_vec, int64_t fill,
                    int64_t* out, int64_t ld_out, agnn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Per-head relation transforms of HGTConv (PyG `k_rel` / `v_rel`: one D x D matrix per (edge type, head); reached through
 * graphmuse's HybridHGT, ref models/analysis.py:445-453 `heads=4`).  For the n_rel relations leaving one node type:
 *     forward   y[n, (r*heads + h)*D + j] = sum_i x[n, h*D + i] * w[((r*heads + h)*D + i)*D + j]
 *     backward  dx[n, h*D + i]            = sum_r sum_j dy[n, (r*heads + h)*D + j] * wt[((r*heads + h)*D + j)*D + i]
 *     dw        dw[((r*heads + h)*D + i)*D + j] = sum_n x[n, h*D + i] * dy[n, (r*heads + h)*D + j]
 * i.e. n_rel*heads independent [n, D] x [D, D] products on the fp32-input MFMA instead of one dense GEMM against a
 * block-diagonal weight (heads x the useful FLOPs).  D = 64.  Up to AGNN_RELT_MAX_ITEMS operands of the same shape
 * (K and V) per launch.  Item fields per entry point:
 *     fwd: x = x [n, heads*D] (ld_x), w = blocks [n_rel*heads][D][D], y = y [n, n_rel*heads*D] (ld_y)
 *     bwd: x = dy (ld_x), w = the TRANSPOSED blocks, y = dx [n, heads*D] (ld_y)
 *     dw:  x = x (ld_x), w = dy (ld_y), y = dw blocks [n_rel*heads][D][D] (contiguous); row slices are summed in a
 *          fixed order through `workspace` (agnn_relt_dw_workspace_bytes).
 * ------------------------------------------------------------------------------------------ */
#define AGNN_RELT_MAX_ITEMS 4
typedef struct {
  const float* x;
  const float* w;
  float* y;
  int64_t ld_x, ld_y;
} agnn_relt_item_t;
int agnn_relt_fwd_f32(int n_items, const agnn_relt_item_t* items /* (host) */, int32_t n_rel, int32_t heads, int32_t D,
                      int64_t n_rows, agnn_stream_t stream);
int agnn_relt_bwd_f32(int n_items, const agnn_relt_item_t* items /* (host) */, int32_t n_rel, int32_t heads, int32_t D,
                      int64_t n_rows, agnn_stream_t stream);
size_t agnn_relt_dw_workspace_bytes(int n_items, int32_t n_rel, int32_t heads, int32_t D, int64_t n_rows);
int agnn_relt_dw_f32(int n_items, const agnn_relt_item_t* items /* (host) */, int32_t n_rel, int32_t heads, int32_t D,
                     int64_t n_rows, void* workspace, size_t workspace_bytes, agnn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Global-norm gradient clipping + AdamW over flat fp32 buffers (ref optimizer: models/analysis.py:1380-1381
 * `torch.optim.AdamW`; clipping as Lightning's gradient_clip_val does before the step):
 *     g' = g * min(max_norm / (||g||_2 + 1e-6), 1)            (max_norm <= 0: no clipping)
 *     p  = p (1 - lr wd) - lr * (m' / (1 - b1^t)) / (sqrt(v') / sqrt(1 - b2^t) + eps),  m' = b1 m + (1-b1) g', v' = b2 v + (1-b2) g'^2
 * `step` is a DEVICE float holding t-1; the call increments it (so a captured hipGraph advances it on replay).
 * norm_out (device float, may be NULL) receives ||g||_2 before clipping.  write_clipped_grad != 0 stores g' back.
 * Two launches, sums in a fixed order.  Buffers 16-byte aligned.
 * ------------------------------------------------------------------------------------------ */
size_t agnn_adamw_workspace_bytes(void);
int agnn_adamw_f32(float* p, float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                   float weight_decay, float max_norm, float* step, float* norm_out, int32_t write_clipped_grad,
                   void* workspace, size_t workspace_bytes, agnn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* AGNN_H_ */
