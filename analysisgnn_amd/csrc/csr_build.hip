// COO (int64, unsorted) -> CSR (int32, stable) for up to AGNN_MAX_SEG relation-directions in
// ONE pass: a single key space (segment row base + row), one stable radix sort, one boundary
// scan.  Replaces the reference's per-relation boolean-mask compaction
// (analysisgnn/models/core/hgnn.py:137-139, :481-483) and the unsorted-index scatter inside
// torch_scatter.  Pure integer work, HBM/L2-bound; no float math here.
#include <hipcub/hipcub.hpp>

#include "agnn_common.h"

namespace {

struct SegTable {
  const int64_t* row[AGNN_MAX_SEG];
  const int64_t* col[AGNN_MAX_SEG];
  const int64_t* etype[AGNN_MAX_SEG];
  int64_t code[AGNN_MAX_SEG];
  int32_t ebase[AGNN_MAX_SEG + 1];  // exclusive prefix of n_edges
  int32_t rbase[AGNN_MAX_SEG + 1];  // exclusive prefix of n_rows
  int32_t n_seg;
};

__device__ __forceinline__ int find_seg(const SegTable& t, int32_t e) {
  int s = 0;
#pragma unroll 1
  while (s + 1 < t.n_seg && e >= t.ebase[s + 1]) ++s;
  return s;
}

// key = global row id (sentinel total_rows for masked-out / out-of-range edges), val = global edge slot
__global__ void k_make_keys(SegTable t, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  const int32_t e_total = t.ebase[t.n_seg];
  const uint32_t sentinel = static_cast<uint32_t>(t.rbase[t.n_seg]);
  for (int32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < e_total; e += gridDim.x * blockDim.x) {
    const int s = find_seg(t, e);
    const int32_t le = e - t.ebase[s];
    const int64_t r = t.row[s][le];
    const int32_t nrows = t.rbase[s + 1] - t.rbase[s];
    bool keep = r >= 0 && r < nrows;
    if (t.etype[s] != nullptr) keep = keep && (t.etype[s][le] == t.code[s]);
    keys[e] = keep ? static_cast<uint32_t>(t.rbase[s] + static_cast<int32_t>(r)) : sentinel;
    vals[e] = static_cast<uint32_t>(e);
  }
}

// rowstart[q] = first sorted position whose key >= q, for q in [0, total_rows]
__global__ void k_rowstart(const uint32_t* __restrict__ keys, int32_t e_total, int32_t total_rows,
                           int32_t* __restrict__ rowstart) {
  for (int32_t p = blockIdx.x * blockDim.x + threadIdx.x; p <= e_total; p += gridDim.x * blockDim.x) {
    const int64_t k_prev = (p == 0) ? -1 : static_cast<int64_t>(keys[p - 1]);
    const int64_t k_here = (p == e_total) ? static_cast<int64_t>(total_rows)
                                          : static_cast<int64_t>(keys[p]);
    int64_t hi = k_here < total_rows ? k_here : total_rows;
    for (int64_t q = k_prev + 1; q <= hi; ++q) rowstart[q] = p;
  }
}

__global__ void k_gather_col(SegTable t, const uint32_t* __restrict__ keys,
                             const uint32_t* __restrict__ vals, int32_t* __restrict__ col,
                             int32_t* __restrict__ perm) {
  const int32_t e_total = t.ebase[t.n_seg];
  const uint32_t sentinel = static_cast<uint32_t>(t.rbase[t.n_seg]);
  for (int32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < e_total; p += gridDim.x * blockDim.x) {
    if (keys[p] >= sentinel) {  // masked-out tail: defined but never referenced
      col[p] = 0;
      perm[p] = 0;
      continue;
    }
    const int32_t e = static_cast<int32_t>(vals[p]);
    const int s = find_seg(t, e);
    const int32_t le = e - t.ebase[s];
    col[p] = static_cast<int32_t>(t.col[s][le]);
    perm[p] = le;
  }
}

__global__ void k_rowend(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ perm,
                         int32_t n_rows, int32_t e_limit, int32_t* __restrict__ rowend) {
  for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_rows; i += gridDim.x * blockDim.x) {
    int32_t lo = rowptr[i], hi = rowptr[i + 1];
    while (lo < hi) {  // perm is increasing inside a row (stable sort)
      const int32_t mid = lo + ((hi - lo) >> 1);
      if (perm[mid] < e_limit) lo = mid + 1; else hi = mid;
    }
    rowend[i] = lo;
  }
}

inline size_t align_up(size_t x) { return (x + 255) & ~static_cast<size_t>(255); }

inline int bits_for(int64_t n) {  // number of key bits needed for values in [0, n]
  int b = 1;
  while ((int64_t{1} << b) <= n) ++b;
  return b;
}

size_t sort_temp_bytes(int64_t e_total, int64_t total_rows) {
  size_t bytes = 0;
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, static_cast<const uint32_t*>(nullptr),
                                     static_cast<uint32_t*>(nullptr),
                                     static_cast<const uint32_t*>(nullptr),
                                     static_cast<uint32_t*>(nullptr), static_cast<int>(e_total), 0,
                                     bits_for(total_rows), nullptr);
  return bytes;
}

}  // namespace

extern "C" size_t agnn_csr_workspace_bytes(int64_t e_total, int64_t total_rows) {
  if (e_total < 0 || total_rows < 0) return 0;
  const size_t arr = align_up(static_cast<size_t>(e_total > 0 ? e_total : 1) * sizeof(uint32_t));
  return 4 * arr + align_up(sort_temp_bytes(e_total > 0 ? e_total : 1, total_rows)) + 256;
}

extern "C" int agnn_csr_build(int n_seg, const agnn_coo_seg_t* segs, int32_t* rowstart, int32_t* col,
                              int32_t* perm, void* workspace, size_t workspace_bytes,
                              agnn_stream_t stream_) {
  using namespace agnn;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (n_seg <= 0 || n_seg > AGNN_MAX_SEG) return fail(AGNN_EINVAL, "csr_build: n_seg=%d not in [1,%d]", n_seg, AGNN_MAX_SEG);
  if (!segs || !rowstart) return fail(AGNN_EINVAL, "csr_build: null argument");
  SegTable t{};
  t.n_seg = n_seg;
  int64_t e_total = 0, r_total = 0;
  for (int s = 0; s < n_seg; ++s) {
    if (segs[s].n_edges < 0 || segs[s].n_rows < 0) return fail(AGNN_EINVAL, "csr_build: negative size in segment %d", s);
    if (segs[s].n_edges > 0 && (!segs[s].row || !segs[s].col)) return fail(AGNN_EINVAL, "csr_build: null COO pointer in segment %d", s);
    t.row[s] = segs[s].row;
    t.col[s] = segs[s].col;
    t.etype[s] = segs[s].etype;
    t.code[s] = segs[s].etype_code;
    t.ebase[s] = static_cast<int32_t>(e_total);
    t.rbase[s] = static_cast<int32_t>(r_total);
    e_total += segs[s].n_edges;
    r_total += segs[s].n_rows;
  }
  if (e_total >= (int64_t{1} << 31) - 1 || r_total >= (int64_t{1} << 31) - 1) return fail(AGNN_EINVAL, "csr_build: sizes exceed int32 (E=%lld rows=%lld)", (long long)e_total, (long long)r_total);
  t.ebase[n_seg] = static_cast<int32_t>(e_total);
  t.rbase[n_seg] = static_cast<int32_t>(r_total);
  if (e_total > 0 && (!col || !perm)) return fail(AGNN_EINVAL, "csr_build: null output");

  const int threads = 256;
  if (e_total == 0) {
    const int blocks = static_cast<int>((r_total + 1 + threads - 1) / threads);
    hipLaunchKernelGGL(k_rowstart, dim3(blocks), dim3(threads), 0, stream, nullptr, 0,
                       static_cast<int32_t>(r_total), rowstart);
    return check_launch("csr_build/rowstart");
  }
  const size_t need = agnn_csr_workspace_bytes(e_total, r_total);
  if (!workspace || workspace_bytes < need) return fail(AGNN_ENOMEM, "csr_build: workspace %zu < %zu bytes", workspace_bytes, need);
  const size_t arr = align_up(static_cast<size_t>(e_total) * sizeof(uint32_t));
  char* ws = static_cast<char*>(workspace);
  ws = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(ws) + 255) & ~uintptr_t{255});
  uint32_t* keys_in = reinterpret_cast<uint32_t*>(ws);
  uint32_t* keys_out = reinterpret_cast<uint32_t*>(ws + arr);
  uint32_t* vals_in = reinterpret_cast<uint32_t*>(ws + 2 * arr);
  uint32_t* vals_out = reinterpret_cast<uint32_t*>(ws + 3 * arr);
  void* temp = ws + 4 * arr;
  size_t temp_bytes = sort_temp_bytes(e_total, r_total);

  int blocks = static_cast<int>((e_total + threads - 1) / threads);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(k_make_keys, dim3(blocks), dim3(threads), 0, stream, t, keys_in, vals_in);
  if (int rc = check_launch("csr_build/make_keys")) return rc;
  hipError_t e = hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out,
                                                    static_cast<int>(e_total), 0, bits_for(r_total), stream);
  if (e != hipSuccess) return fail(AGNN_ERUNTIME, "csr_build/sort: %s", hipGetErrorString(e));
  int blocks1 = static_cast<int>((e_total + 1 + threads - 1) / threads);
  if (blocks1 > 4096) blocks1 = 4096;
  hipLaunchKernelGGL(k_rowstart, dim3(blocks1), dim3(threads), 0, stream, keys_out,
                     static_cast<int32_t>(e_total), static_cast<int32_t>(r_total), rowstart);
  if (int rc = check_launch("csr_build/rowstart")) return rc;
  hipLaunchKernelGGL(k_gather_col, dim3(blocks), dim3(threads), 0, stream, t, keys_out, vals_out, col, perm);
  return check_launch("csr_build/gather_col");
}

extern "C" int agnn_csr_rowend(const int32_t* rowptr, const int32_t* perm, int64_t n_rows, int64_t e_limit,
                               int32_t* rowend, agnn_stream_t stream_) {
  using namespace agnn;
  if (n_rows < 0 || n_rows >= (int64_t{1} << 31)) return fail(AGNN_EINVAL, "csr_rowend: n_rows=%lld", (long long)n_rows);
  if (n_rows == 0) return AGNN_OK;
  if (!rowptr || !rowend) return fail(AGNN_EINVAL, "csr_rowend: null argument");
  if (e_limit > INT32_MAX) e_limit = INT32_MAX;
  if (e_limit < 0) e_limit = 0;
  const int threads = 256;
  int blocks = static_cast<int>((n_rows + threads - 1) / threads);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(k_rowend, dim3(blocks), dim3(threads), 0, static_cast<hipStream_t>(stream_), rowptr,
                     perm, static_cast<int32_t>(n_rows), static_cast<int32_t>(e_limit), rowend);
  return check_launch("csr_rowend");
}
