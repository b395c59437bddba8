// Per-head relation transforms of HGTConv (PyG >= 2.3: `k_rel` / `v_rel` = HeteroLinear with one D x D matrix per
// (edge type, head); reached through graphmuse's HybridHGT, reference analysisgnn/models/analysis.py:445-453):
//     k'[n, r, h, :] = k[n, h, :] @ A[r, h]          for every relation r leaving the node type, every head h
// Round 1 ran them as ONE dense library GEMM against a block-diagonal [H, R*H] weight: 4x the useful FLOPs at heads = 4
// (12.6 GFLOP instead of 3.1 per operand and layer at the C3 shape) plus the launches that assemble the weight.
// Here they are what they are: R*heads independent [N, D] x [D, D] products on the fp32-input MFMA
// (v_mfma_f32_32x32x2_f32: exact fp32, 64 FLOP/clk/SIMD), D = 64.
//   k_relt<false>  forward: one workgroup = 128 rows x one head; a wave keeps its 32 x 64 slice of k in registers (the A
//                  operand of all R products) and walks the relations; A[r, h] (16 KB) is staged through LDS, double
//                  buffered, once per workgroup; the 32 x 64 result of every relation is written as whole 128-byte lines.
//   k_relt<true>   input gradient: dk[n, h, :] = sum_r dk'[n, r, h, :] @ A[r, h]^T — the same loop with the roles
//                  swapped (the A operand changes per relation, ONE accumulator is carried across the relations); the
//                  caller passes the transposed blocks.
//   k_relt_dw      weight gradient: dA[r, h] = k[:, h, :]^T dk'[:, r, h, :], a 64 x 64 output with the reduction over N:
//                  one wave = one (relation, head, row slice), operands straight from global memory (already "k-major":
//                  one MFMA k-step = two consecutive rows, as in wgrad.hip), slices summed in a fixed order by
//                  agnn::launch_slab_reduce (no atomics).
// K and V (and anything else that shares the shape) go through ONE launch: up to 4 items per call.
#include "agnn_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kD = 64;

struct ReltItem {
  const float* x;     // fwd: [n, heads*D] (ld_x);  bwd: dy [n, n_rel*heads*D] (ld_x);  dw: x [n, heads*D]
  const float* w;     // fwd: A blocks [n_rel*heads][D][D];  bwd: the transposed blocks;  dw: dy [n, n_rel*heads*D] (ld_y)
  float* y;           // fwd: [n, n_rel*heads*D] (ld_y);  bwd: dx [n, heads*D] (ld_y);  dw: unused
  int64_t ld_x, ld_y;
};

struct ReltArgs {
  ReltItem it[AGNN_RELT_MAX_ITEMS];
  int32_t n_rel, heads;
  int64_t n_rows;
  // dw only
  float* slab;              // [items][S][n_rel*heads*D][D]
  int32_t S, rows_per_slice;
};

template <bool BWD>
__global__ __launch_bounds__(256) void k_relt(ReltArgs p) {
  constexpr int D = kD;
  __shared__ __attribute__((aligned(16))) float sW[2][D * D];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c32 = lane & 31, kk = lane >> 5;
  const int h = blockIdx.y % p.heads, item = blockIdx.y / p.heads;
  const ReltItem& I = p.it[item];
  const int64_t row0 = static_cast<int64_t>(blockIdx.x) * 128 + wave * 32;
  int64_t rowc = row0 + c32;
  if (rowc > p.n_rows - 1) rowc = p.n_rows - 1;
  const int H = p.heads * D;

  float4 wr[4];
  auto fetch_w = [&](int r) {
    const float4* src = reinterpret_cast<const float4*>(I.w + static_cast<size_t>(r * p.heads + h) * D * D);
#pragma unroll
    for (int u = 0; u < 4; ++u) wr[u] = src[tid + 256 * u];
  };
  auto put_w = [&](int buf) {
    float4* dst = reinterpret_cast<float4*>(sW[buf]);
#pragma unroll
    for (int u = 0; u < 4; ++u) dst[tid + 256 * u] = wr[u];
  };
  // A operand of the MFMA: lane (row c32, half kk) holds the 32 consecutive inputs k = kk*32 .. kk*32 + 31 of its row (the k
  // order inside a product is free as long as the B operand uses the same one): eight 16-byte loads.
  float a[32];
  auto load_a = [&](int colbase) {
    const float4* src = reinterpret_cast<const float4*>(I.x + rowc * I.ld_x + colbase + kk * 32);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float4 v = src[u];
      a[4 * u] = v.x; a[4 * u + 1] = v.y; a[4 * u + 2] = v.z; a[4 * u + 3] = v.w;
    }
  };
  // C/D layout of the 32x32 MFMA: lane l, register q -> row (q&3) + 8*(q>>2) + 4*(l>>5), column l&31
  auto store = [&](const f32x16& acc0, const f32x16& acc1, int colbase) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int64_t row = row0 + (q & 3) + 8 * (q >> 2) + 4 * kk;
      if (row < p.n_rows) {
        float* o = I.y + row * I.ld_y + colbase + c32;
        o[0] = acc0[q];
        o[32] = acc1[q];
      }
    }
  };

  fetch_w(0);
  put_w(0);
  if (!BWD) load_a(h * D);
  f32x16 acc0 = {0}, acc1 = {0};
  __syncthreads();
  for (int r = 0; r < p.n_rel; ++r) {
    if (r + 1 < p.n_rel) fetch_w(r + 1);                 // in flight behind the 64 MFMAs below
    if (BWD) load_a((r * p.heads + h) * D);
    if (!BWD) {
      acc0 = f32x16{0};
      acc1 = f32x16{0};
    }
    const float* sw = sW[r & 1] + kk * 32 * D + c32;     // B operand: B[k = kk*32 + s][j = c32 (+32)]
#pragma unroll
    for (int s = 0; s < 32; ++s) {
      const float b0 = sw[s * D], b1 = sw[s * D + 32];
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b0, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b1, acc1, 0, 0, 0);
    }
    if (!BWD) store(acc0, acc1, (r * p.heads + h) * D);
    if (r + 1 < p.n_rel) put_w((r + 1) & 1);             // that buffer was last read in iteration r - 1 (barrier below)
    __syncthreads();
  }
  if (BWD) store(acc0, acc1, h * D);
  (void)H;
}

// dA[r, h][i][j] = sum_n x[n, h*D + i] * dy[n, (r*heads + h)*D + j].  One wave = one (item, relation, head, row slice):
// a 64 x 64 output as four 32 x 32 accumulators; a lane loads TWO adjacent columns of both operands per row (8 bytes:
// columns 2*c32, 2*c32 + 1 feed the two tiles of that operand), the lane halves take the two rows of a k-step.
__global__ __launch_bounds__(256) void k_relt_dw(ReltArgs p) {
  constexpr int D = kD;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c32 = lane & 31, kk = lane >> 5;
  const int g = blockIdx.x;                               // (item, relation, head)
  const int groups = p.n_rel * p.heads;
  const int item = g / groups, rh = g - item * groups, h = rh % p.heads;
  const ReltItem& I = p.it[item];
  const int slice = blockIdx.y * 4 + wave;
  const int64_t r0 = static_cast<int64_t>(slice) * p.rows_per_slice;
  int64_t r1 = r0 + p.rows_per_slice;
  if (r1 > p.n_rows) r1 = p.n_rows;
  f32x16 acc00 = {0}, acc01 = {0}, acc10 = {0}, acc11 = {0};
  if (r0 < r1) {
    const float* xs = I.x + h * D + 2 * c32;              // operand "A": rows of the 64 x 64 output = input feature i
    const float* ys = I.w + rh * D + 2 * c32;             // operand "B": columns = output feature j
    constexpr int CH = 8;
    float2 av[2][CH], bv[2][CH];
    auto fetch = [&](int64_t base, float2* ao, float2* bo) {
#pragma unroll
      for (int u = 0; u < CH; ++u) {
        int64_t row = base + 2 * u + kk;
        if (row > p.n_rows - 1) row = p.n_rows - 1;
        ao[u] = *reinterpret_cast<const float2*>(xs + row * I.ld_x);
        bo[u] = *reinterpret_cast<const float2*>(ys + row * I.ld_y);
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    auto mma = [&](int64_t base, const float2* ao, const float2* bo) {
#pragma unroll
      for (int u = 0; u < CH; ++u) {
        const float m = (base + 2 * u + kk < r1) ? 1.f : 0.f;      // rows past the slice contribute nothing
        const float ax = ao[u].x * m, ay = ao[u].y * m;
        acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(ax, bo[u].x, acc00, 0, 0, 0);
        acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(ax, bo[u].y, acc01, 0, 0, 0);
        acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(ay, bo[u].x, acc10, 0, 0, 0);
        acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(ay, bo[u].y, acc11, 0, 0, 0);
      }
    };
    fetch(r0, av[0], bv[0]);
    int cur = 0;
    for (int64_t base = r0; base < r1; base += 2 * CH) {
      if (base + 2 * CH < r1) fetch(base + 2 * CH, av[cur ^ 1], bv[cur ^ 1]);
      mma(base, av[cur], bv[cur]);
      cur ^= 1;
    }
  }
  // accXY: rows = input features 2*i + X, columns = output features 2*j + Y (the stride-2 split of the float2 loads)
  float* slab = p.slab + ((static_cast<size_t>(item) * p.S + slice) * groups + rh) * D * D;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int i = (q & 3) + 8 * (q >> 2) + 4 * kk;
    float* o = slab + (2 * i) * D + 2 * c32;
    *reinterpret_cast<float2*>(o) = make_float2(acc00[q], acc01[q]);
    *reinterpret_cast<float2*>(o + D) = make_float2(acc10[q], acc11[q]);
  }
}

int relt_check(const char* who, int n_items, const agnn_relt_item_t* items, int n_rel, int heads, int D, int64_t n_rows) {
  using namespace agnn;
  if (n_items <= 0 || n_items > AGNN_RELT_MAX_ITEMS || !items) return fail(AGNN_EINVAL, "%s: n_items=%d not in [1,%d]", who, n_items, AGNN_RELT_MAX_ITEMS);
  if (D != kD) return fail(AGNN_EINVAL, "%s: D=%d (built for D = %d)", who, D, kD);
  if (n_rel <= 0 || n_rel > 64 || heads <= 0 || heads > 64) return fail(AGNN_EINVAL, "%s: n_rel=%d heads=%d", who, n_rel, heads);
  if (n_rows < 0 || n_rows >= (int64_t{1} << 31)) return fail(AGNN_EINVAL, "%s: n_rows=%lld", who, (long long)n_rows);
  return AGNN_OK;
}

}  // namespace

extern "C" int agnn_relt_fwd_f32(int n_items, const agnn_relt_item_t* items, int32_t n_rel, int32_t heads, int32_t D, int64_t n_rows,
                                 agnn_stream_t stream_) {
  using namespace agnn;
  if (int rc = relt_check("relt_fwd", n_items, items, n_rel, heads, D, n_rows)) return rc;
  if (n_rows == 0) return AGNN_OK;
  ReltArgs p{};
  p.n_rel = n_rel; p.heads = heads; p.n_rows = n_rows;
  const int64_t H = static_cast<int64_t>(heads) * D;
  for (int i = 0; i < n_items; ++i) {
    const agnn_relt_item_t& t = items[i];
    if (!t.x || !t.w || !t.y) return fail(AGNN_EINVAL, "relt_fwd: item %d has a null pointer", i);
    if (!aligned16(t.x) || !aligned16(t.w) || (t.ld_x & 3) || t.ld_x < H || t.ld_y < H * n_rel)
      return fail(AGNN_EALIGN, "relt_fwd: item %d: x / w must be 16-byte aligned, ld_x %% 4 == 0, ld_x >= heads*D, ld_y >= n_rel*heads*D", i);
    p.it[i] = ReltItem{t.x, t.w, t.y, t.ld_x, t.ld_y};
  }
  const dim3 grid(static_cast<unsigned>((n_rows + 127) / 128), static_cast<unsigned>(heads * n_items));
  hipLaunchKernelGGL(k_relt<false>, grid, dim3(256), 0, static_cast<hipStream_t>(stream_), p);
  return check_launch("relt_fwd");
}

extern "C" int agnn_relt_bwd_f32(int n_items, const agnn_relt_item_t* items, int32_t n_rel, int32_t heads, int32_t D, int64_t n_rows,
                                 agnn_stream_t stream_) {
  using namespace agnn;
  if (int rc = relt_check("relt_bwd", n_items, items, n_rel, heads, D, n_rows)) return rc;
  if (n_rows == 0) return AGNN_OK;
  ReltArgs p{};
  p.n_rel = n_rel; p.heads = heads; p.n_rows = n_rows;
  const int64_t H = static_cast<int64_t>(heads) * D;
  for (int i = 0; i < n_items; ++i) {
    const agnn_relt_item_t& t = items[i];          // x = dy [n, n_rel*H], w = transposed blocks, y = dx [n, H]
    if (!t.x || !t.w || !t.y) return fail(AGNN_EINVAL, "relt_bwd: item %d has a null pointer", i);
    if (!aligned16(t.x) || !aligned16(t.w) || (t.ld_x & 3) || t.ld_x < H * n_rel || t.ld_y < H)
      return fail(AGNN_EALIGN, "relt_bwd: item %d: dy / wt must be 16-byte aligned, ld_dy %% 4 == 0, ld_dy >= n_rel*heads*D, ld_dx >= heads*D", i);
    p.it[i] = ReltItem{t.x, t.w, t.y, t.ld_x, t.ld_y};
  }
  const dim3 grid(static_cast<unsigned>((n_rows + 127) / 128), static_cast<unsigned>(heads * n_items));
  hipLaunchKernelGGL(k_relt<true>, grid, dim3(256), 0, static_cast<hipStream_t>(stream_), p);
  return check_launch("relt_bwd");
}

namespace {
struct DwPlan { int S; int rows_per_slice; };
DwPlan relt_dw_plan(int64_t n_rows) {
  // ~32 row slices (a multiple of the 4 waves of a workgroup), at least 128 rows each, an even number of rows per slice
  int S = static_cast<int>((n_rows + 511) / 512);
  if (S > 32) S = 32;
  if (S < 4) S = 4;
  S = (S + 3) & ~3;
  int rps = static_cast<int>((n_rows + S - 1) / S);
  rps = (rps + 1) & ~1;
  if (rps < 2) rps = 2;
  return DwPlan{S, rps};
}
}  // namespace

extern "C" size_t agnn_relt_dw_workspace_bytes(int n_items, int32_t n_rel, int32_t heads, int32_t D, int64_t n_rows) {
  if (n_items <= 0 || n_rel <= 0 || heads <= 0 || D <= 0 || n_rows <= 0) return 0;
  const DwPlan pl = relt_dw_plan(n_rows);
  return static_cast<size_t>(n_items) * pl.S * n_rel * heads * D * D * sizeof(float) + 256;
}

extern "C" int agnn_relt_dw_f32(int n_items, const agnn_relt_item_t* items, int32_t n_rel, int32_t heads, int32_t D, int64_t n_rows,
                                void* workspace, size_t workspace_bytes, agnn_stream_t stream_) {
  using namespace agnn;
  if (int rc = relt_check("relt_dw", n_items, items, n_rel, heads, D, n_rows)) return rc;
  const int64_t H = static_cast<int64_t>(heads) * D;
  const int groups = n_rel * heads;
  hipStream_t s = static_cast<hipStream_t>(stream_);
  if (n_rows == 0) {
    for (int i = 0; i < n_items; ++i) {
      if (!items[i].y) return fail(AGNN_EINVAL, "relt_dw: item %d has a null output", i);
      hipError_t e = hipMemsetAsync(items[i].y, 0, static_cast<size_t>(groups) * D * D * sizeof(float), s);
      if (e != hipSuccess) return fail(AGNN_ERUNTIME, "relt_dw: %s", hipGetErrorString(e));
    }
    return AGNN_OK;
  }
  const size_t need = agnn_relt_dw_workspace_bytes(n_items, n_rel, heads, D, n_rows);
  if (!workspace || workspace_bytes < need) return fail(AGNN_ENOMEM, "relt_dw: workspace %zu < %zu bytes", workspace_bytes, need);
  const DwPlan pl = relt_dw_plan(n_rows);
  ReltArgs p{};
  p.n_rel = n_rel; p.heads = heads; p.n_rows = n_rows; p.S = pl.S; p.rows_per_slice = pl.rows_per_slice;
  p.slab = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~uintptr_t{255});
  for (int i = 0; i < n_items; ++i) {
    const agnn_relt_item_t& t = items[i];          // x = x [n, H] (ld_x), w = dy [n, n_rel*H] (ld_y), y = dA blocks [groups][D][D]
    if (!t.x || !t.w || !t.y) return fail(AGNN_EINVAL, "relt_dw: item %d has a null pointer", i);
    if ((reinterpret_cast<uintptr_t>(t.x) & 7u) || (reinterpret_cast<uintptr_t>(t.w) & 7u) || (t.ld_x & 1) || (t.ld_y & 1) || t.ld_x < H || t.ld_y < H * n_rel)
      return fail(AGNN_EALIGN, "relt_dw: item %d: x / dy must be 8-byte aligned with even leading dimensions", i);
    p.it[i] = ReltItem{t.x, t.w, t.y, t.ld_x, t.ld_y};
  }
  const dim3 grid(static_cast<unsigned>(groups * n_items), static_cast<unsigned>(pl.S / 4));
  hipLaunchKernelGGL(k_relt_dw, grid, dim3(256), 0, s, p);
  if (int rc = check_launch("relt_dw")) return rc;
  for (int i = 0; i < n_items; ++i) {
    const float* slab = p.slab + static_cast<size_t>(i) * pl.S * groups * D * D;
    if (int rc = launch_slab_reduce(slab, nullptr, pl.S, groups * D, D, groups * D, D, items[i].y, D, nullptr, s)) return rc;
  }
  return AGNN_OK;
}
