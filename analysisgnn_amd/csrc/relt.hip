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

  // next relation's weights on their way global -> registers -> LDS: four named registers, not an array (an array captured
  // by the two lambdas stayed in scratch memory: `scratch_store` right behind the loads, i.e. an s_waitcnt on them and on
  // every store issued before them)
  float4 w0 = make_float4(0.f, 0.f, 0.f, 0.f), w1 = w0, w2 = w0, w3 = w0;
  auto fetch_w = [&](int r) {
    const float4* src = reinterpret_cast<const float4*>(I.w + static_cast<size_t>(r * p.heads + h) * D * D) + tid;
    w0 = src[0];
    w1 = src[256];
    w2 = src[512];
    w3 = src[768];
  };
  auto put_w = [&](int buf) {
    float4* dst = reinterpret_cast<float4*>(sW[buf]) + tid;
    dst[0] = w0;
    dst[256] = w1;
    dst[512] = w2;
    dst[768] = w3;
  };
  // A operand of the MFMA: lane (row c32, half kk) holds the 32 consecutive inputs k = kk*32 .. kk*32 + 31 of its row (the k
  // order inside a product is free as long as the B operand uses the same one): eight 16-byte loads.
  const float* xrow = I.x + rowc * I.ld_x + kk * 32;
  auto load_a = [&](float4 (&dst)[8], int colbase) {
    const float4* src = reinterpret_cast<const float4*>(xrow + colbase);
#pragma unroll
    for (int u = 0; u < 8; ++u) dst[u] = src[u];
  };
  // C/D layout of the 32x32 MFMA: lane l, register q -> row (q&3) + 8*(q>>2) + 4*(l>>5), column l&31
  const bool full = row0 + 32 <= p.n_rows;               // wave-uniform: all 32 rows of this wave exist (the usual case)
  float* const ybase = I.y + (row0 + 4 * kk) * I.ld_y + c32;
  auto store = [&](const f32x16& acc0, const f32x16& acc1, int colbase) {
    float* o = ybase + colbase;
    if (full) {                                          // straight-line: 32 stores, no per-row branch
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        float* oq = o + ((q & 3) + 8 * (q >> 2)) * I.ld_y;
        oq[0] = acc0[q];
        oq[32] = acc1[q];
      }
    } else {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        if (row0 + (q & 3) + 8 * (q >> 2) + 4 * kk < p.n_rows) {
          float* oq = o + ((q & 3) + 8 * (q >> 2)) * I.ld_y;
          oq[0] = acc0[q];
          oq[32] = acc1[q];
        }
      }
    }
  };
  // 64 MFMAs of one relation.  The B operand B[k = kk*32 + s][j = c32 (+32)] comes from LDS in chunks of 8 k-steps, the
  // next chunk's 16 reads issued BEFORE the 16 MFMAs of the current one (left to the compiler, every k-step was
  // `ds_read2 -> s_waitcnt lgkmcnt(0) -> 2 MFMAs` through one register pair: the LDS latency of every read exposed).
  auto product = [&](const float4 (&av)[8], int buf, f32x16& acc0, f32x16& acc1) {
    const float* sw = sW[buf] + kk * 32 * D + c32;
    float b0[2][8], b1[2][8];
    auto rd = [&](int c, int slot) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        b0[slot][u] = sw[(8 * c + u) * D];
        b1[slot][u] = sw[(8 * c + u) * D + 32];
      }
    };
    rd(0, 0);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (c + 1 < 4) rd(c + 1, (c + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float4 q = av[2 * c + (u >> 2)];
        const float as = (u & 3) == 0 ? q.x : (u & 3) == 1 ? q.y : (u & 3) == 2 ? q.z : q.w;
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(as, b0[c & 1][u], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(as, b1[c & 1][u], acc1, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  fetch_w(0);
  put_w(0);
  float4 aA[8], aB[8];                                   // BWD: the A operand of relation r and of relation r + 1 (prefetched)
  load_a(aA, h * D);                                     // BWD: relation 0's block is (0*heads + h)*D = h*D as well
  // The A operand is waited for HERE, outside the relation loop: left to the first MFMA that uses it, the compiler puts an
  // `s_waitcnt vmcnt(8)` at the top of EVERY iteration (it cannot see that a previous iteration already waited), which
  // also waits for the previous relation's 32 stores — one HBM write latency per relation, the MFMA pipe idle meanwhile.
#pragma unroll
  for (int u = 0; u < 8; ++u) asm volatile("" ::"v"(aA[u].x), "v"(aA[u].y), "v"(aA[u].z), "v"(aA[u].w));
  f32x16 acc0 = {0}, acc1 = {0};
  __syncthreads();
  // one relation: [prefetch the next weights (global -> registers) and, for the input gradient, the next A operand]
  // -> 64 MFMAs -> [store] -> [weights registers -> the other LDS buffer] -> barrier
  auto relation = [&](int r, const float4 (&cur)[8], float4 (&nxt)[8]) {
    if (r + 1 < p.n_rel) {
      fetch_w(r + 1);
      if (BWD) load_a(nxt, ((r + 1) * p.heads + h) * D);
    }
    if (!BWD) {
      acc0 = f32x16{0};
      acc1 = f32x16{0};
    }
    product(cur, r & 1, acc0, acc1);
    if (!BWD) store(acc0, acc1, (r * p.heads + h) * D);
    if (r + 1 < p.n_rel) put_w((r + 1) & 1);             // that buffer was last read in iteration r - 1 (barrier below)
    __syncthreads();
  };
  if (BWD) {
    for (int r = 0; r < p.n_rel; r += 2) {
      relation(r, aA, aB);
      if (r + 1 < p.n_rel) relation(r + 1, aB, aA);
    }
    store(acc0, acc1, h * D);
  } else {
    for (int r = 0; r < p.n_rel; ++r) relation(r, aA, aB);
  }
}

// dA[r, h][i][j] = sum_n x[n, h*D + i] * dy[n, (r*heads + h)*D + j].  One workgroup = (item, relation, row slice), one
// wave per head (the waves of a workgroup read ADJACENT 256-byte pieces of the same rows: whole 1 KiB row segments of x
// and of dy per workgroup); a wave's 64 x 64 output is four 32 x 32 accumulators; a lane loads TWO adjacent columns of
// both operands per row (8 bytes: columns 2*c32, 2*c32 + 1 feed the two tiles of that operand), the lane halves take the
// two rows of a k-step.  Three register stages in rotation (two chunks of loads in flight), as in wgrad.hip.
__global__ __launch_bounds__(256) void k_relt_dw(ReltArgs p) {
  constexpr int D = kD;
  const int lane = threadIdx.x & 63, h = blockIdx.z * 4 + (threadIdx.x >> 6);      // up to four heads per workgroup
  if (h >= p.heads) return;
  const int c32 = lane & 31, kk = lane >> 5;
  const int item = blockIdx.x / p.n_rel, r = blockIdx.x - item * p.n_rel;
  const int rh = r * p.heads + h;
  const ReltItem& I = p.it[item];
  const int slice = blockIdx.y;
  const int64_t r0 = static_cast<int64_t>(slice) * p.rows_per_slice;
  int64_t r1 = r0 + p.rows_per_slice;
  if (r1 > p.n_rows) r1 = p.n_rows;
  f32x16 acc00 = {0}, acc01 = {0}, acc10 = {0}, acc11 = {0};
  if (r0 < r1) {
    const float* xs = I.x + h * D + 2 * c32;              // operand "A": rows of the 64 x 64 output = input feature i
    const float* ys = I.w + rh * D + 2 * c32;             // operand "B": columns = output feature j
    constexpr int CH = 8;
    constexpr int STEP = 2 * CH;
    float2 a0[CH], b0[CH], a1[CH], b1[CH], a2[CH], b2[CH];
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
    auto mma = [&](const float2* ao, const float2* bo) {
#pragma unroll
      for (int u = 0; u < CH; ++u) {
        acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(ao[u].x, bo[u].x, acc00, 0, 0, 0);
        acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(ao[u].x, bo[u].y, acc01, 0, 0, 0);
        acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(ao[u].y, bo[u].x, acc10, 0, 0, 0);
        acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(ao[u].y, bo[u].y, acc11, 0, 0, 0);
      }
    };
    auto mma_tail = [&](int64_t base, const float2* ao, const float2* bo) {      // rows >= r1 contribute nothing
#pragma unroll
      for (int u = 0; u < CH; ++u) {
        const float m = (base + 2 * u + kk < r1) ? 1.f : 0.f;
        const float ax = ao[u].x * m, ay = ao[u].y * m;
        acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(ax, bo[u].x, acc00, 0, 0, 0);
        acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(ax, bo[u].y, acc01, 0, 0, 0);
        acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(ay, bo[u].x, acc10, 0, 0, 0);
        acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(ay, bo[u].y, acc11, 0, 0, 0);
      }
    };
    const int64_t nfull = (r1 - r0) / STEP, nchunks = (r1 - r0 + STEP - 1) / STEP;
    fetch(r0, a0, b0);
    fetch(r0 + STEP, a1, b1);
    int64_t c = 0;
    for (; c + 3 <= nfull; c += 3) {
      fetch(r0 + (c + 2) * STEP, a2, b2);
      mma(a0, b0);
      fetch(r0 + (c + 3) * STEP, a0, b0);
      mma(a1, b1);
      fetch(r0 + (c + 4) * STEP, a1, b1);
      mma(a2, b2);
    }
    fetch(r0 + (c + 2) * STEP, a2, b2);
    if (c < nchunks) mma_tail(r0 + c * STEP, a0, b0);
    if (c + 1 < nchunks) mma_tail(r0 + (c + 1) * STEP, a1, b1);
    if (c + 2 < nchunks) mma_tail(r0 + (c + 2) * STEP, a2, b2);
  }
  // accXY: rows = input features 2*i + X, columns = output features 2*j + Y (the stride-2 split of the float2 loads)
  const int groups = p.n_rel * p.heads;
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
  // up to 32 row slices of at least ~512 rows, an even number of rows per slice
  int S = static_cast<int>((n_rows + 511) / 512);
  if (S > 32) S = 32;
  if (S < 1) S = 1;
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
  const dim3 grid(static_cast<unsigned>(n_rel * n_items), static_cast<unsigned>(pl.S), static_cast<unsigned>((heads + 3) / 4));
  hipLaunchKernelGGL(k_relt_dw, grid, dim3(static_cast<unsigned>(64 * (heads < 4 ? heads : 4))), 0, s, p);
  if (int rc = check_launch("relt_dw")) return rc;
  for (int i = 0; i < n_items; ++i) {
    const float* slab = p.slab + static_cast<size_t>(i) * pl.S * groups * D * D;
    if (int rc = launch_slab_reduce(slab, nullptr, pl.S, groups * D, D, groups * D, D, items[i].y, D, nullptr, s)) return rc;
  }
  return AGNN_OK;
}
