// COO (int64, unsorted) -> CSR (int32, stable) for up to AGNN_MAX_SEG relation-directions in
// ONE pass over a single key space (segment row base + row).  Replaces the reference's per-relation boolean-mask
// compaction (analysisgnn/models/core/hgnn.py:137-139, :481-483) and the unsorted-index scatter inside
// torch_scatter.  Pure integer work, HBM/L2-bound; no float math here.
//
// A counting sort with a per-row fix-up instead of a library sort: for ~2 x 10^5 edges the library's stable sort is
// a chain of ~20 tiny merge kernels (~0.2 ms on the critical path of every step, profiles/r01_q); rows of a score
// graph hold a handful of edges, so
//   k_count    key per edge, integer atomicAdd into the per-row counters
//   scan       exclusive sum of the counters = rowstart (library scan)
//   k_scatter  edge id -> rowstart[key] + slot, slot handed out by atomicSub on the same counters (which are zero
//              again afterwards); the order INSIDE a row is whatever the atomics gave ...
//   k_rows     ... and is made the stable one here: one thread per row sorts its <= 8 edge ids in registers; longer
//              rows go to a list
//   k_emit     col / perm of those rows, one thread per edge position (coalesced)
//   k_heavy    one wavefront per listed row: rank of every id among the row's ids (O(d^2 / 64)), col / perm written at
//              the rank; also zero-fills the slots behind the kept edges
// Integer atomics only decide intermediate positions; the result is the unique stable order (bit-identical to the
// oracle's), whatever the interleaving.
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

// key = global row id (sentinel total_rows for masked-out / out-of-range edges); counts the kept edges of every row
__global__ void k_count(SegTable t, uint32_t* __restrict__ keys, uint32_t* __restrict__ cnt) {
  const int32_t e_total = t.ebase[t.n_seg];
  const uint32_t sentinel = static_cast<uint32_t>(t.rbase[t.n_seg]);
  for (int32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < e_total; e += gridDim.x * blockDim.x) {
    const int s = find_seg(t, e);
    const int32_t le = e - t.ebase[s];
    const int64_t r = t.row[s][le];
    const int32_t nrows = t.rbase[s + 1] - t.rbase[s];
    bool keep = r >= 0 && r < nrows;
    if (t.etype[s] != nullptr) keep = keep && (t.etype[s][le] == t.code[s]);
    const uint32_t key = keep ? static_cast<uint32_t>(t.rbase[s] + static_cast<int32_t>(r)) : sentinel;
    keys[e] = key;
    if (keep) atomicAdd(cnt + key, 1u);
  }
}

// counters and the heavy-row count start at zero.  A kernel, not hipMemsetAsync: a memset node inside a captured
// single-stream graph was followed by an out-of-range write of k_scatter (stale counters) on ROCm 7.2.
__global__ void k_zero_u32(uint32_t* __restrict__ p, int64_t n) {
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * blockDim.x) p[i] = 0u;
}

// rowstart for an edge-less build
__global__ void k_rowstart_zero(int32_t total_rows, int32_t* __restrict__ rowstart) {
  for (int32_t q = blockIdx.x * blockDim.x + threadIdx.x; q <= total_rows; q += gridDim.x * blockDim.x) rowstart[q] = 0;
}

// vals[rowstart[key] + slot] = edge id; the counters return to zero
// With clean counters every position is inside the row (slot < count, rowstart from the scan of the same counts).  A
// position outside the edge array means the counters were NOT what k_count left (round 1: a memset node that had not
// zeroed them before k_count ran): nothing is written out of range, and the caller's status word is bumped so that the
// build is reported as failed (agnn_check_status) instead of passing on an index with missing edges.
__global__ void k_scatter(const uint32_t* __restrict__ keys, int32_t e_total, uint32_t sentinel,
                          const int32_t* __restrict__ rowstart, uint32_t* __restrict__ cnt, uint32_t* __restrict__ vals,
                          uint32_t* __restrict__ rowof, int32_t* __restrict__ status) {
  for (int32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < e_total; e += gridDim.x * blockDim.x) {
    const uint32_t key = keys[e];
    if (key >= sentinel) continue;
    const uint32_t before = atomicSub(cnt + key, 1u);
    const uint32_t slot = before - 1u;
    const uint32_t pos = static_cast<uint32_t>(rowstart[key]) + slot;
    const bool inside = before != 0u && pos < static_cast<uint32_t>(e_total) && pos < static_cast<uint32_t>(rowstart[key + 1]);
    if (inside) {
      vals[pos] = static_cast<uint32_t>(e);
      rowof[pos] = key;
    } else if (status != nullptr) {
      atomicAdd(status, 1);
    }
  }
}

constexpr int kSmallRow = 8;

// one thread per row: rows of 2 .. kSmallRow (= 8) edges are sorted in registers — eight predicated loads in flight, a
// 19-exchange sorting network, predicated stores; an insertion sort on global memory is a chain of dependent round trips
// (110 us for the 18-edge measure rows of the C3 graph).  Longer rows are appended to `heavy` (one wavefront each).
// col / perm of the short rows are written by k_emit.
__device__ __forceinline__ void cex(uint32_t& a, uint32_t& b) {
  const uint32_t lo = a < b ? a : b, hi = a < b ? b : a;
  a = lo;
  b = hi;
}

__global__ void k_rows(SegTable t, const int32_t* __restrict__ rowstart, uint32_t* __restrict__ vals,
                       uint32_t* __restrict__ heavy_n, uint32_t* __restrict__ heavy) {
  static_assert(kSmallRow == 8, "the sorting network below is for 8 keys");
  const int32_t r_total = t.rbase[t.n_seg];
  for (int32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < r_total; q += gridDim.x * blockDim.x) {
    const int32_t s0 = rowstart[q], d = rowstart[q + 1] - s0;
    if (d <= 1) continue;
    if (d > kSmallRow) {
      heavy[atomicAdd(heavy_n, 1u)] = static_cast<uint32_t>(q);
      continue;
    }
    uint32_t* v = vals + s0;
    uint32_t a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = i < d ? v[i] : 0xffffffffu;      // padding sorts to the end
    // Batcher's odd-even merge sort for 8 keys
    cex(a[0], a[1]); cex(a[2], a[3]); cex(a[4], a[5]); cex(a[6], a[7]);
    cex(a[0], a[2]); cex(a[1], a[3]); cex(a[4], a[6]); cex(a[5], a[7]);
    cex(a[1], a[2]); cex(a[5], a[6]);
    cex(a[0], a[4]); cex(a[1], a[5]); cex(a[2], a[6]); cex(a[3], a[7]);
    cex(a[2], a[4]); cex(a[3], a[5]);
    cex(a[1], a[2]); cex(a[3], a[4]); cex(a[5], a[6]);
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (i < d) v[i] = a[i];
  }
}

// one thread per kept edge position (coalesced): col / perm of every row of <= kSmallRow edges
__global__ void k_emit(SegTable t, const int32_t* __restrict__ rowstart, const uint32_t* __restrict__ vals,
                       const uint32_t* __restrict__ rowof, int32_t* __restrict__ col, int32_t* __restrict__ perm) {
  const int32_t kept = rowstart[t.rbase[t.n_seg]];
  for (int32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < kept; p += gridDim.x * blockDim.x) {
    const int32_t q = static_cast<int32_t>(rowof[p]);
    if (rowstart[q + 1] - rowstart[q] > kSmallRow) continue;        // emitted by k_heavy
    int sg = 0;
#pragma unroll 1
    while (sg + 1 < t.n_seg && q >= t.rbase[sg + 1]) ++sg;
    const int32_t le = static_cast<int32_t>(vals[p]) - t.ebase[sg];
    col[p] = static_cast<int32_t>(t.col[sg][le]);
    perm[p] = le;
  }
}

// one wavefront per heavy row (rank sort), then the slots behind the kept edges are zero-filled
__global__ __launch_bounds__(256) void k_heavy(SegTable t, const int32_t* __restrict__ rowstart, const uint32_t* __restrict__ vals,
                                               int32_t* __restrict__ col, int32_t* __restrict__ perm,
                                               const uint32_t* __restrict__ heavy_n, const uint32_t* __restrict__ heavy) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int n_waves = gridDim.x * (blockDim.x >> 6);
  const uint32_t n_heavy = *heavy_n;
  for (uint32_t h = wave; h < n_heavy; h += n_waves) {
    const int32_t q = static_cast<int32_t>(heavy[h]);
    const int32_t s0 = rowstart[q], d = rowstart[q + 1] - s0;
    int sg = 0;
#pragma unroll 1
    while (sg + 1 < t.n_seg && q >= t.rbase[sg + 1]) ++sg;
    const int64_t* cs = t.col[sg];
    const int32_t eb = t.ebase[sg];
    for (int32_t c0 = 0; c0 < d; c0 += 64) {
      const bool mine = c0 + lane < d;
      const uint32_t x = mine ? vals[s0 + c0 + lane] : 0xffffffffu;
      int32_t rank = 0;
      for (int32_t j0 = 0; j0 < d; j0 += 64) {
        const uint32_t y = (j0 + lane < d) ? vals[s0 + j0 + lane] : 0xffffffffu;
        const int m = (d - j0) < 64 ? (d - j0) : 64;
        for (int b = 0; b < m; ++b) rank += (static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(y), b)) < x) ? 1 : 0;
      }
      if (mine) {                                   // edge ids are distinct: ranks are a permutation of 0 .. d-1
        const int32_t le = static_cast<int32_t>(x) - eb;
        col[s0 + rank] = static_cast<int32_t>(cs[le]);
        perm[s0 + rank] = le;
      }
    }
  }
  const int32_t e_total = t.ebase[t.n_seg];
  const int32_t kept = rowstart[t.rbase[t.n_seg]];
  for (int32_t p = kept + blockIdx.x * blockDim.x + threadIdx.x; p < e_total; p += gridDim.x * blockDim.x) {
    col[p] = 0;                                     // masked-out tail: defined but never referenced
    perm[p] = 0;
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

struct RowendBatch {
  agnn_rowend_item_t it[AGNN_ROWEND_MAX_ITEMS];
};

// blockIdx.y = item; the items of a batch (relations x directions x trimmed layers of one sampled batch) differ in size
// by small factors only, so surplus blocks of the shorter ones just leave
__global__ __launch_bounds__(256) void k_rowend_batch(RowendBatch b) {
  const agnn_rowend_item_t& I = b.it[blockIdx.y];
  for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < I.n_rows; i += gridDim.x * blockDim.x) {
    int32_t lo = I.rowptr[i], hi = I.rowptr[i + 1];
    while (lo < hi) {  // perm is increasing inside a row (stable sort)
      const int32_t mid = lo + ((hi - lo) >> 1);
      if (I.perm[mid] < I.e_limit) lo = mid + 1; else hi = mid;
    }
    I.rowend[i] = lo;
  }
}

inline size_t align_up(size_t x) { return (x + 255) & ~static_cast<size_t>(255); }

size_t scan_temp_bytes(int64_t total_rows) {
  size_t bytes = 0;
  (void)hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, static_cast<const uint32_t*>(nullptr), static_cast<int32_t*>(nullptr),
                                         static_cast<int>(total_rows + 1), nullptr);
  return bytes;
}

struct Layout {
  size_t keys, vals, rowof, cnt, heavy, temp, temp_bytes, total;
};

Layout make_layout(int64_t e_total, int64_t total_rows) {
  Layout l;
  const size_t e = static_cast<size_t>(e_total > 0 ? e_total : 1), r = static_cast<size_t>(total_rows);
  size_t off = 0;
  l.keys = off; off += align_up(e * sizeof(uint32_t));
  l.vals = off; off += align_up(e * sizeof(uint32_t));
  l.rowof = off; off += align_up(e * sizeof(uint32_t));
  l.cnt = off; off += align_up((r + 2) * sizeof(uint32_t));              // counters [total_rows + 1], then the heavy-row count
  l.heavy = off; off += align_up((e / (kSmallRow + 1) + 1) * sizeof(uint32_t));
  l.temp = off;
  l.temp_bytes = scan_temp_bytes(total_rows);
  off += align_up(l.temp_bytes);
  l.total = off;
  return l;
}

}  // namespace

extern "C" size_t agnn_csr_workspace_bytes(int64_t e_total, int64_t total_rows) {
  if (e_total < 0 || total_rows < 0) return 0;
  return make_layout(e_total, total_rows).total + 256;
}

extern "C" int agnn_csr_build(int n_seg, const agnn_coo_seg_t* segs, int32_t* rowstart, int32_t* col,
                              int32_t* perm, void* workspace, size_t workspace_bytes, int32_t* status,
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
    hipLaunchKernelGGL(k_rowstart_zero, dim3(blocks > 4096 ? 4096 : blocks), dim3(threads), 0, stream, static_cast<int32_t>(r_total), rowstart);
    return check_launch("csr_build/rowstart");
  }
  const size_t need = agnn_csr_workspace_bytes(e_total, r_total);
  if (!workspace || workspace_bytes < need) return fail(AGNN_ENOMEM, "csr_build: workspace %zu < %zu bytes", workspace_bytes, need);
  const Layout l = make_layout(e_total, r_total);
  char* ws = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~uintptr_t{255});
  uint32_t* keys = reinterpret_cast<uint32_t*>(ws + l.keys);
  uint32_t* vals = reinterpret_cast<uint32_t*>(ws + l.vals);
  uint32_t* rowof = reinterpret_cast<uint32_t*>(ws + l.rowof);
  uint32_t* cnt = reinterpret_cast<uint32_t*>(ws + l.cnt);
  uint32_t* heavy_n = cnt + r_total + 1;
  uint32_t* heavy = reinterpret_cast<uint32_t*>(ws + l.heavy);
  void* temp = ws + l.temp;
  size_t temp_bytes = l.temp_bytes;

  hipError_t e = hipSuccess;
  int blocks_z = static_cast<int>((r_total + 2 + threads - 1) / threads);
  if (blocks_z > 2048) blocks_z = 2048;
  hipLaunchKernelGGL(k_zero_u32, dim3(blocks_z), dim3(threads), 0, stream, cnt, r_total + 2);
  if (int rc = check_launch("csr_build/zero")) return rc;
  int blocks = static_cast<int>((e_total + threads - 1) / threads);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(k_count, dim3(blocks), dim3(threads), 0, stream, t, keys, cnt);
  if (int rc = check_launch("csr_build/count")) return rc;
  e = hipcub::DeviceScan::ExclusiveSum(temp, temp_bytes, cnt, rowstart, static_cast<int>(r_total + 1), stream);
  if (e != hipSuccess) return fail(AGNN_ERUNTIME, "csr_build/scan: %s", hipGetErrorString(e));
  hipLaunchKernelGGL(k_scatter, dim3(blocks), dim3(threads), 0, stream, keys, static_cast<int32_t>(e_total),
                     static_cast<uint32_t>(r_total), rowstart, cnt, vals, rowof, status);
  if (int rc = check_launch("csr_build/scatter")) return rc;
  int blocks_r = static_cast<int>((r_total + threads - 1) / threads);
  if (blocks_r > 4096) blocks_r = 4096;
  if (blocks_r < 1) blocks_r = 1;
  hipLaunchKernelGGL(k_rows, dim3(blocks_r), dim3(threads), 0, stream, t, rowstart, vals, heavy_n, heavy);
  if (int rc = check_launch("csr_build/rows")) return rc;
  hipLaunchKernelGGL(k_emit, dim3(blocks), dim3(threads), 0, stream, t, rowstart, vals, rowof, col, perm);
  if (int rc = check_launch("csr_build/emit")) return rc;
  hipLaunchKernelGGL(k_heavy, dim3(1024), dim3(256), 0, stream, t, rowstart, vals, col, perm, heavy_n, heavy);
  return check_launch("csr_build/heavy");
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

extern "C" int agnn_csr_rowend_batch(int n_items, const agnn_rowend_item_t* items, agnn_stream_t stream_) {
  using namespace agnn;
  if (n_items < 0 || n_items > AGNN_ROWEND_MAX_ITEMS) return fail(AGNN_EINVAL, "csr_rowend_batch: n_items=%d not in [0,%d]", n_items, AGNN_ROWEND_MAX_ITEMS);
  if (n_items == 0) return AGNN_OK;
  if (!items) return fail(AGNN_EINVAL, "csr_rowend_batch: null argument");
  RowendBatch b{};
  int32_t max_rows = 0;
  for (int i = 0; i < n_items; ++i) {
    b.it[i] = items[i];
    if (items[i].n_rows < 0) return fail(AGNN_EINVAL, "csr_rowend_batch: item %d has n_rows=%d", i, items[i].n_rows);
    if (items[i].n_rows > 0 && (!items[i].rowptr || !items[i].perm || !items[i].rowend)) return fail(AGNN_EINVAL, "csr_rowend_batch: item %d has a null pointer", i);
    if (b.it[i].e_limit < 0) b.it[i].e_limit = 0;
    if (items[i].n_rows > max_rows) max_rows = items[i].n_rows;
  }
  if (max_rows == 0) return AGNN_OK;
  int bx = (max_rows + 255) / 256;
  if (bx > 1024) bx = 1024;
  hipLaunchKernelGGL(k_rowend_batch, dim3(bx, n_items), dim3(256), 0, static_cast<hipStream_t>(stream_), b);
  return check_launch("csr_rowend_batch");
}
