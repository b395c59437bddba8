/*
 * agnn_oracle.c — plain-C CPU restatement of the kernel-level contracts in include/agnn.h.
 *
 * TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).
 * Never linked into or called from analysisgnn_amd/.
 *
 * What it restates, with the reference lines each loop follows:
 *   oracle_csr_build  : `edge_index[:, edge_type == code]` (analysisgnn/models/core/hgnn.py:137-139)
 *                       followed by grouping edges by their row index, keeping edge order.
 *   oracle_spmm_f32   : `he = h[edge_index[1]]` + `scatter(he, edge_index[0], out=x.clone(),
 *                       reduce='mean'|'sum')` (core/gnn.py:70-74), the zero-initialised scatter
 *                       sums (core/gnn.py:511,539; core/hgnn.py:406-407) and the onset pool
 *                       (models/analysis.py:580-586: drop self loops, both ends < batch_size).
 * torch_scatter semantics are those of SURVEY.md App. A.1 (sum then divide the whole `out` by
 * max(count,1)).  Sequential, one thread, fp32 accumulation in edge order with fmaf — the same
 * operation order as the HIP kernel, so results are expected to agree to the last bit for sums
 * and to <= 1 ulp after the mean division.
 * Pinned by tests/test_oracle_c.py against oracle/scatter_ref.py and the golden fixtures.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  const int64_t* row;
  const int64_t* col;
  const int64_t* etype;
  int64_t etype_code;
  int64_t n_edges;
  int64_t n_rows;
} oracle_coo_seg_t;

/* same output contract as agnn_csr_build (include/agnn.h) */
int oracle_csr_build(int n_seg, const oracle_coo_seg_t* segs, int32_t* rowstart, int32_t* col, int32_t* perm) {
  int64_t total_rows = 0;
  for (int s = 0; s < n_seg; ++s) total_rows += segs[s].n_rows;
  int32_t* cnt = (int32_t*)calloc((size_t)total_rows + 1, sizeof(int32_t));
  if (!cnt) return -12;
  int64_t rbase = 0;
  for (int s = 0; s < n_seg; ++s) {
    for (int64_t e = 0; e < segs[s].n_edges; ++e) {
      int64_t r = segs[s].row[e];
      if (r < 0 || r >= segs[s].n_rows) continue;
      if (segs[s].etype && segs[s].etype[e] != segs[s].etype_code) continue;
      cnt[rbase + r + 1]++;
    }
    rbase += segs[s].n_rows;
  }
  rowstart[0] = 0;
  for (int64_t q = 0; q < total_rows; ++q) rowstart[q + 1] = rowstart[q] + cnt[q + 1];
  int32_t* cur = (int32_t*)malloc(((size_t)total_rows + 1) * sizeof(int32_t));
  if (!cur) { free(cnt); return -12; }
  memcpy(cur, rowstart, ((size_t)total_rows + 1) * sizeof(int32_t));
  rbase = 0;
  for (int s = 0; s < n_seg; ++s) {
    for (int64_t e = 0; e < segs[s].n_edges; ++e) {
      int64_t r = segs[s].row[e];
      if (r < 0 || r >= segs[s].n_rows) continue;
      if (segs[s].etype && segs[s].etype[e] != segs[s].etype_code) continue;
      int32_t p = cur[rbase + r]++;
      col[p] = (int32_t)segs[s].col[e];
      perm[p] = (int32_t)e;
    }
    rbase += segs[s].n_rows;
  }
  free(cnt);
  free(cur);
  return 0;
}

int oracle_csr_rowend(const int32_t* rowptr, const int32_t* perm, int64_t n_rows, int64_t e_limit, int32_t* rowend) {
  for (int64_t i = 0; i < n_rows; ++i) {
    int32_t p = rowptr[i];
    while (p < rowptr[i + 1] && perm[p] < e_limit) ++p;
    rowend[i] = p;
  }
  return 0;
}

typedef struct {
  const float* src;
  const int32_t* rowptr;
  const int32_t* rowend;
  const int32_t* col;
  const float* ew;
  const float* colscale;
  int64_t ld_src;
} oracle_rel_t;

#define ORACLE_SPMM_MEAN 1u
#define ORACLE_SPMM_SKIP_SELF 2u
#define ORACLE_SPMM_ACCUM 4u

/* same contract as agnn_spmm_f32 (include/agnn.h) */
int oracle_spmm_f32(int n_rel, const oracle_rel_t* rels, int64_t n_rows, int32_t H, float* out, int64_t ld_out,
                    int64_t rel_stride, const float* self, int64_t ld_self, float* inv_cnt, int32_t col_limit,
                    uint32_t flags) {
  float* acc = (float*)malloc((size_t)H * sizeof(float));
  float* tot = (float*)malloc((size_t)H * sizeof(float));
  if (!acc || !tot) return -12;
  for (int64_t i = 0; i < n_rows; ++i) {
    for (int h = 0; h < H; ++h) tot[h] = 0.f;
    for (int r = 0; r < n_rel; ++r) {
      const oracle_rel_t* R = &rels[r];
      int32_t start = R->rowptr[i];
      int32_t end = R->rowend ? R->rowend[i] : R->rowptr[i + 1];
      int cnt = 0;
      for (int h = 0; h < H; ++h) acc[h] = 0.f;
      for (int32_t p = start; p < end; ++p) {
        int32_t c = R->col[p];
        if (c < 0 || c >= col_limit) continue;
        if ((flags & ORACLE_SPMM_SKIP_SELF) && c == i) continue;
        float w = 1.f;
        if (R->ew) w *= R->ew[p];
        if (R->colscale) w *= R->colscale[c];
        const float* srow = R->src + (int64_t)c * R->ld_src;
        for (int h = 0; h < H; ++h) acc[h] = fmaf(w, srow[h], acc[h]);
        ++cnt;
      }
      float denom = (float)(cnt > 1 ? cnt : 1);
      if (inv_cnt) inv_cnt[(int64_t)r * n_rows + i] = 1.f / denom;
      for (int h = 0; h < H; ++h) {
        if (self) acc[h] += self[i * ld_self + h];
        if (flags & ORACLE_SPMM_MEAN) acc[h] /= denom;
      }
      if (rel_stride == 0) {
        for (int h = 0; h < H; ++h) tot[h] += acc[h];
      } else {
        float* o = out + i * ld_out + (int64_t)r * rel_stride;
        for (int h = 0; h < H; ++h) o[h] = (flags & ORACLE_SPMM_ACCUM) ? acc[h] + o[h] : acc[h];
      }
    }
    if (rel_stride == 0) {
      float* o = out + i * ld_out;
      for (int h = 0; h < H; ++h) o[h] = (flags & ORACLE_SPMM_ACCUM) ? tot[h] + o[h] : tot[h];
    }
  }
  free(acc);
  free(tot);
  return 0;
}
