// Weight-gradient GEMM for the projection layers:  dW[out, in] = sum_n dY[n, out] * X[n, in]  (+ db[out] = sum_n dY[n, out])
//
// Every projection of the encoder sees N = (#subgraphs x 500) rows, so its weight gradient is a GEMM with a SMALL
// output (256 x 256 ... 256 x 1024) and a HUGE reduction dimension (N = 16 000).  The library picks a
// 32 x 32-tile kernel without a K split for that shape — a few dozen workgroups on a 256-CU chip: 88 us per
// launch whatever the size, 19 launches per step (profiles/r01_c).  This kernel is built for the shape:
//   * fp32-input MFMA (v_mfma_f32_32x32x2_f32: exact fp32, 64 FLOP/clk/SIMD), no LDS at all — both operands are
//     already "K-major": one MFMA k-step needs dY[n, 32 cols] and X[n, 32 cols] for two consecutive n, which is
//     exactly one coalesced 8-byte-per-lane load of each (lane = column pair, lane half = n parity);
//   * a 128 x 128 block tile, 4 waves in 2 x 2, each wave 64 x 64 = four accumulators; output rows/cols are taken
//     with stride 2 inside a wave so the two floats a lane loads feed two different MFMA tiles without any shuffle;
//   * N is cut into S slices so that (#tiles x S) ~ 2 workgroups per CU; each workgroup writes its partial tile to
//     a slab and a second tiny kernel sums the S slabs in a fixed order (deterministic, no atomics);
//   * the bias gradient rides along (column sums of the dY values already in registers).
#include "agnn_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct WgArgs {
  const float* dy;
  const float* x;
  int64_t ld_dy, ld_x;
  int32_t n, out_f, in_f;
  int32_t rows_per_slice;   // even
  int32_t tiles_in;         // number of 128-wide tiles along `in`
  float* slab;              // [S][out_pad][in_pad]
  float* slab_b;            // [S][out_pad] or nullptr
  int32_t out_pad, in_pad;
};

__device__ __forceinline__ void wgrad_tile(const WgArgs& a, const int tile, const int slice) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int to = tile / a.tiles_in, ti = tile - to * a.tiles_in;
  const int kk = lane >> 5, c32 = lane & 31;
  const int ocol = to * 128 + wm * 64 + 2 * c32;     // this lane's two output-feature columns of dY
  const int icol = ti * 128 + wn * 64 + 2 * c32;     // this lane's two input-feature columns of X
  const bool o_ok = ocol < a.out_f;                  // widths are even: a float2 is inside or outside as a whole
  const bool i_ok = icol < a.in_f;
  const int r0 = slice * a.rows_per_slice;
  int r1 = r0 + a.rows_per_slice;
  if (r1 > a.n) r1 = a.n;

  f32x16 acc00 = {0}, acc01 = {0}, acc10 = {0}, acc11 = {0};
  float bs0 = 0.f, bs1 = 0.f;
  // Loads are unconditional so the loop stays straight-line code and the compiler can wait on "all but the last two
  // chunks" (s_waitcnt vmcnt(N)) instead of draining the queue behind a branch: lanes whose columns lie past the
  // matrix read column 0 (their accumulator rows / columns are never read back), rows past the end of the matrix
  // are clamped, and rows past the end of the SLICE only occur in the last chunks, which multiply by a 0/1 mask.
  // Addresses: a per-slice base pointer (uniform -> SGPR pair) plus a 32-bit byte offset per lane (the C-ABI entry
  // point checks rows_per_slice * ld * 4 < 2^32), i.e. the "saddr + voffset" form: one multiply-add per load.
  const float* dy_s = a.dy + static_cast<int64_t>(r0) * a.ld_dy;
  const float* x_s = a.x + static_cast<int64_t>(r0) * a.ld_x;
  const uint32_t ldb_dy = static_cast<uint32_t>(a.ld_dy) * 4u, ldb_x = static_cast<uint32_t>(a.ld_x) * 4u;
  const uint32_t cb_dy = (o_ok ? ocol : 0) * 4u, cb_x = (i_ok ? icol : 0) * 4u;
  const int rel_last = a.n - 1 - r0;
  constexpr int CH = 8;                              // k-steps (row pairs) per chunk
  constexpr int STEP = 2 * CH;                       // rows per chunk
  // Three register stages in rotation: the loads of chunk c+2 are issued before the MFMAs of chunk c, so two whole
  // chunks of matrix work (2 x 32 MFMAs = 4096 cycles of one wave) cover an L2/HBM round trip even when a SIMD
  // holds a single wave (256 x 256 outputs give one workgroup per CU).  The scheduling barrier after each fetch
  // keeps the compiler from hoisting the bias-sum adds of a chunk right behind its loads (which drained the queue).
  float2 a0[CH], b0[CH], a1[CH], b1[CH], a2[CH], b2[CH];
  // Row offsets advance by two rows per k-step: running 32-bit byte offsets and ONE add per load.  (Computed from the row number
  // each time, every load cost a clamp and a 64-bit multiply-add — v_mad_u64_u32, quarter rate: 32 of them per 32 MFMAs, ~700 cycles
  // of address arithmetic per 2 048 cycles of matrix work in a wave that cannot issue an MFMA meanwhile.)  Chunks are fetched in
  // order, so the offsets only move forward; the last chunks, which may reach past the end of the MATRIX, take the clamped path.
  uint32_t o_dy = static_cast<uint32_t>(kk) * ldb_dy + cb_dy, o_x = static_cast<uint32_t>(kk) * ldb_x + cb_x;
  const uint32_t s_dy = 2u * ldb_dy, s_x = 2u * ldb_x;
  auto fetch_fast = [&](float2* ao, float2* bo) {     // the next chunk in order, entirely inside the matrix
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      ao[u] = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(dy_s) + o_dy);
      bo[u] = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(x_s) + o_x);
      o_dy += s_dy;
      o_x += s_x;
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto fetch = [&](int chunk, float2* ao, float2* bo) {
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      const uint32_t rel = static_cast<uint32_t>(min(chunk * STEP + 2 * u + kk, rel_last));
      ao[u] = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(dy_s) + (rel * ldb_dy + cb_dy));
      bo[u] = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(x_s) + (rel * ldb_x + cb_x));
    }
    o_dy += CH * s_dy;                               // keeps the running offsets in step (not used past this point in practice)
    o_x += CH * s_x;
    __builtin_amdgcn_sched_barrier(0);
  };
  auto mma = [&](const float2* av, const float2* bv) {
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].x, bv[u].x, acc00, 0, 0, 0);
      acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].x, bv[u].y, acc01, 0, 0, 0);
      acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].y, bv[u].x, acc10, 0, 0, 0);
      acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].y, bv[u].y, acc11, 0, 0, 0);
      // volatile asm: pins the bias-sum adds to this phase (plain adds were hoisted behind the loads of the chunk)
      asm volatile("v_add_f32 %0, %0, %1" : "+v"(bs0) : "v"(av[u].x));
      asm volatile("v_add_f32 %0, %0, %1" : "+v"(bs1) : "v"(av[u].y));
    }
  };
  auto mma_tail = [&](int chunk, const float2* av, const float2* bv) {     // rows >= r1 contribute nothing
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      const float m = (r0 + chunk * STEP + 2 * u + kk < r1) ? 1.f : 0.f;
      const float ax = av[u].x * m, ay = av[u].y * m;
      acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(ax, bv[u].x, acc00, 0, 0, 0);
      acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(ax, bv[u].y, acc01, 0, 0, 0);
      acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(ay, bv[u].x, acc10, 0, 0, 0);
      acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(ay, bv[u].y, acc11, 0, 0, 0);
      bs0 += ax;
      bs1 += ay;
    }
  };
  const int nfull = (r1 - r0) / STEP;                // chunks that lie entirely inside the slice
  const int nchunks = (r1 - r0 + STEP - 1) / STEP;
  fetch(0, a0, b0);
  fetch(1, a1, b1);
  int c = 0;
  const int safe = (rel_last + 1) / STEP;            // chunks [0, safe) lie entirely inside the matrix: no clamp needed
  for (; c + 3 <= nfull && c + 5 <= safe; c += 3) {  // (the trip fetches chunks c + 2, c + 3, c + 4)
    fetch_fast(a2, b2);
    mma(a0, b0);
    fetch_fast(a0, b0);
    mma(a1, b1);
    fetch_fast(a1, b1);
    mma(a2, b2);
  }
  for (; c + 3 <= nfull; c += 3) {
    fetch(c + 2, a2, b2);
    mma(a0, b0);
    fetch(c + 3, a0, b0);
    mma(a1, b1);
    fetch(c + 4, a1, b1);
    mma(a2, b2);
  }
  fetch(c + 2, a2, b2);
  if (c < nchunks) mma_tail(c, a0, b0);
  if (c + 1 < nchunks) mma_tail(c + 1, a1, b1);
  if (c + 2 < nchunks) mma_tail(c + 2, a2, b2);
  // C/D layout of 32x32 MFMA: lane l, register r -> row i = (r&3) + 8*(r>>2) + 4*(l>>5), column j = l&31
  float* slab = a.slab + (static_cast<int64_t>(slice) * a.out_pad) * a.in_pad;
  const int ob = to * 128 + wm * 64, ib = ti * 128 + wn * 64;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int i = (r & 3) + 8 * (r >> 2) + 4 * kk;
    float* row0 = slab + static_cast<int64_t>(ob + 2 * i) * a.in_pad + ib + 2 * c32;          // output feature ob+2i   (tile c = 0)
    float* row1 = row0 + a.in_pad;                                                            // output feature ob+2i+1 (tile c = 1)
    *reinterpret_cast<float2*>(row0) = make_float2(acc00[r], acc01[r]);
    *reinterpret_cast<float2*>(row1) = make_float2(acc10[r], acc11[r]);
  }
  if (a.slab_b != nullptr && wn == 0) {
    bs0 += __shfl_xor(bs0, 32, 64);                  // the two lane halves hold the two n parities
    bs1 += __shfl_xor(bs1, 32, 64);
    if (kk == 0 && ti == 0)
      *reinterpret_cast<float2*>(a.slab_b + static_cast<int64_t>(slice) * a.out_pad + ocol) = make_float2(bs0, bs1);
  }
}

__global__ __launch_bounds__(256) void k_wgrad(WgArgs a) { wgrad_tile(a, blockIdx.x, blockIdx.y); }

// ---- several weight gradients in ONE launch (round 3) ------------------------------------------------------------------------
// A training step computes ~20 of these products, most of them small (256 x 256, 384 x 128, 128 x 128 ... outputs): alone, each
// needs up to 64 row slices to put a workgroup on every CU — 64 slabs written and read back per output, a 250-row K loop per
// workgroup, and a reduction launch of its own (20 x ~11 us per step).  The deferred weight gradients (dp.defer_weight_grads) are
// pending TOGETHER at the encoder's flush points, so they are issued together: the workgroups of all items fill the chip, every
// item gets by with the same few slices (384 / total tiles: 5 at the GNN stack's flush), K loops of thousands of rows, and ONE
// reduction launch serves all outputs.
constexpr int kWgBatchMax = 16;

struct WgBatch {
  WgArgs it[kWgBatchMax];
  int32_t first[kWgBatchMax + 1];        // first workgroup of item i (prefix sums of tiles_i * S_i)
  int32_t tiles[kWgBatchMax];
  int32_t n;
};

__global__ __launch_bounds__(256) void k_wgrad_batch(WgBatch b) {
  int i = 0;
  const int blk = blockIdx.x;
  while (i + 1 < b.n && blk >= b.first[i + 1]) ++i;              // workgroup-uniform
  const int local = blk - b.first[i];
  const int tiles = b.tiles[i];
  const int slice = local / tiles;
  wgrad_tile(b.it[i], local - slice * tiles, slice);
}


// dw[o][i] = sum_s slab[s][o][i] ;  db[o] = sum_s slab_b[s][o]   (fixed order: 8 interleaved partial sums, then a
// fixed tree).  A block = 32 float2 columns x 8 slab groups, so the S slab reads of one output element are
// spread over 8 threads with independent loads in flight instead of one thread walking S strided lines.
__device__ __forceinline__ void wgrad_reduce_blocks(const float* __restrict__ slab, const float* __restrict__ slab_b, int S, int out_f, int in_f,
                                                    int out_pad, int in_pad, float* __restrict__ dw, int64_t ld_dw, float* __restrict__ db,
                                                    const int block, const int n_blocks) {
  __shared__ float2 part[8][32];
  const int col = threadIdx.x & 31, sg = threadIdx.x >> 5;
  const int half_in = in_f >> 1;
  const int64_t total = static_cast<int64_t>(out_f) * half_in;
  const int64_t plane = static_cast<int64_t>(out_pad) * in_pad;
  for (int64_t e0 = static_cast<int64_t>(block) * 32; e0 < total; e0 += static_cast<int64_t>(n_blocks) * 32) {
    const int64_t e = e0 + col;
    float2 s = make_float2(0.f, 0.f);
    int o = 0, i = 0;
    if (e < total) {
      o = static_cast<int>(e / half_in);
      i = static_cast<int>(e - static_cast<int64_t>(o) * half_in) * 2;
      const float* p = slab + static_cast<int64_t>(o) * in_pad + i;
      for (int k = sg; k < S; k += 8) {
        const float2 v = *reinterpret_cast<const float2*>(p + k * plane);
        s.x += v.x;
        s.y += v.y;
      }
    }
    part[sg][col] = s;
    __syncthreads();
    if (sg == 0 && e < total) {
      float2 t = part[0][col];
#pragma unroll
      for (int g = 1; g < 8; ++g) { t.x += part[g][col].x; t.y += part[g][col].y; }
      *reinterpret_cast<float2*>(dw + static_cast<int64_t>(o) * ld_dw + i) = t;
    }
    __syncthreads();
  }
  if (db != nullptr) {
    // same 32 x 8 decomposition for the bias slabs (a serial walk over S slabs cost ~19 us of load latency)
    __shared__ float partb[8][32];
    for (int e0 = block * 32; e0 < out_f; e0 += n_blocks * 32) {
      const int e = e0 + col;
      float s = 0.f;
      if (e < out_f)
        for (int k = sg; k < S; k += 8) s += slab_b[static_cast<int64_t>(k) * out_pad + e];
      partb[sg][col] = s;
      __syncthreads();
      if (sg == 0 && e < out_f) {
        float t = partb[0][col];
#pragma unroll
        for (int g = 1; g < 8; ++g) t += partb[g][col];
        db[e] = t;
      }
      __syncthreads();
    }
  }
}

__global__ __launch_bounds__(256) void k_wgrad_reduce(const float* __restrict__ slab, const float* __restrict__ slab_b, int S,
                                                      int out_f, int in_f, int out_pad, int in_pad, float* __restrict__ dw,
                                                      int64_t ld_dw, float* __restrict__ db) {
  wgrad_reduce_blocks(slab, slab_b, S, out_f, in_f, out_pad, in_pad, dw, ld_dw, db, blockIdx.x, gridDim.x);
}

struct RedItem {
  const float* slab;
  const float* slab_b;
  float* dw;
  float* db;
  int64_t ld_dw;
  int32_t S, out_f, in_f, out_pad, in_pad;
};
struct RedBatch {
  RedItem it[kWgBatchMax];
  int32_t first[kWgBatchMax + 1];
  int32_t n;
};

__global__ __launch_bounds__(256) void k_wgrad_reduce_batch(RedBatch b) {
  int i = 0;
  const int blk = blockIdx.x;
  while (i + 1 < b.n && blk >= b.first[i + 1]) ++i;
  const RedItem& r = b.it[i];
  wgrad_reduce_blocks(r.slab, r.slab_b, r.S, r.out_f, r.in_f, r.out_pad, r.in_pad, r.dw, r.ld_dw, r.db, blk - b.first[i], b.first[i + 1] - b.first[i]);
}

struct Plan {
  int tiles_out, tiles_in, S, rows_per_slice, out_pad, in_pad;
};

Plan make_plan(int64_t n, int out_f, int in_f) {
  Plan p;
  p.tiles_out = (out_f + 127) / 128;
  p.tiles_in = (in_f + 127) / 128;
  p.out_pad = p.tiles_out * 128;
  p.in_pad = p.tiles_in * 128;
  const int tiles = p.tiles_out * p.tiles_in;
  // two workgroups per CU, and never a few more than that: rounding the slice count UP gave 520 workgroups for 256 x 1280
  // (20 tiles x 26 slices) — a third round of 8 workgroups behind two full ones: 147 us instead of the ~110 its FLOPs take
  int64_t S = 512 / tiles;
  const int64_t max_s = (n + 63) / 64;              // at least 64 rows per slice
  if (S > max_s) S = max_s;
  if (S > 64) S = 64;                               // slab traffic: S * out * in * 4 B written and read back
  if (S < 1) S = 1;
  int64_t rps = (n + S - 1) / S;
  rps = (rps + 1) & ~int64_t{1};
  p.rows_per_slice = static_cast<int>(rps);
  p.S = static_cast<int>((n + rps - 1) / rps);
  return p;
}

}  // namespace

// dw[o][i] = sum_s slab[s][o][i] (in_f even), db[o] = sum_s slab_b[s][o]; shared with gproj.hip
int agnn::launch_slab_reduce(const float* slab, const float* slab_b, int S, int out_f, int in_f, int out_pad, int in_pad,
                             float* dw, int64_t ld_dw, float* db, hipStream_t s) {
  const int64_t total = static_cast<int64_t>(out_f) * (in_f >> 1);
  int blocks = static_cast<int>((total + 31) / 32);
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(k_wgrad_reduce, dim3(blocks), dim3(256), 0, s, slab, slab_b, S, out_f, in_f, out_pad, in_pad, dw, ld_dw, db);
  return check_launch("slab_reduce");
}

extern "C" size_t agnn_wgrad_workspace_bytes(int64_t n, int32_t out_f, int32_t in_f) {
  if (n <= 0 || out_f <= 0 || in_f <= 0) return 0;
  const Plan p = make_plan(n, out_f, in_f);
  return (static_cast<size_t>(p.S) * p.out_pad * p.in_pad + static_cast<size_t>(p.S) * p.out_pad) * sizeof(float) + 256;
}

extern "C" int agnn_wgrad_f32(const float* dy, int64_t ld_dy, const float* x, int64_t ld_x, int64_t n, int32_t out_f,
                              int32_t in_f, float* dw, int64_t ld_dw, float* db, void* workspace, size_t workspace_bytes,
                              agnn_stream_t stream_) {
  using namespace agnn;
  if (n <= 0 || n >= (int64_t{1} << 31) || out_f <= 0 || in_f <= 0) return fail(AGNN_EINVAL, "wgrad: bad sizes n=%lld out=%d in=%d", (long long)n, out_f, in_f);
  if ((out_f & 1) || (in_f & 1) || (ld_dy & 1) || (ld_x & 1) || (ld_dw & 1)) return fail(AGNN_EALIGN, "wgrad: widths and leading dimensions must be even");
  if (!dy || !x || !dw || !workspace) return fail(AGNN_EINVAL, "wgrad: null argument");
  if ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dw) | reinterpret_cast<uintptr_t>(db)) & 7u)
    return fail(AGNN_EALIGN, "wgrad: pointers must be 8-byte aligned");
  if (ld_dy < out_f || ld_x < in_f || ld_dw < in_f) return fail(AGNN_EINVAL, "wgrad: leading dimension smaller than the width");
  const size_t need = agnn_wgrad_workspace_bytes(n, out_f, in_f);
  if (workspace_bytes < need) return fail(AGNN_ENOMEM, "wgrad: workspace %zu < %zu bytes", workspace_bytes, need);
  const Plan p = make_plan(n, out_f, in_f);
  if ((static_cast<int64_t>(p.rows_per_slice) + 128) * (ld_dy > ld_x ? ld_dy : ld_x) * 4 >= (int64_t{1} << 32))
    return fail(AGNN_EINVAL, "wgrad: a row slice (%d rows x ld) exceeds the kernel's 32-bit byte offsets", p.rows_per_slice);
  char* ws = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~uintptr_t{255});
  float* slab = reinterpret_cast<float*>(ws);
  float* slab_b = slab + static_cast<size_t>(p.S) * p.out_pad * p.in_pad;
  WgArgs a{dy, x, ld_dy, ld_x, static_cast<int32_t>(n), out_f, in_f, p.rows_per_slice, p.tiles_in, slab,
           db ? slab_b : nullptr, p.out_pad, p.in_pad};
  hipStream_t s = static_cast<hipStream_t>(stream_);
  // (an LDS-staged variant of the tile, round 2, was ~5 % faster alone and slower inside the training step, next to the
  // recurrence kernels whose workgroups hold most of a CU's LDS: profiles/r01_wgrad_mfma.md; it is not in the library)
  hipLaunchKernelGGL(k_wgrad, dim3(p.tiles_out * p.tiles_in, p.S), dim3(256), 0, s, a);
  if (int rc = check_launch("wgrad")) return rc;
  return launch_slab_reduce(slab, db ? slab_b : nullptr, p.S, out_f, in_f, p.out_pad, p.in_pad, dw, ld_dw, db, s);
}

namespace {
struct BatchPlan {
  int n;
  int tiles_out[kWgBatchMax], tiles_in[kWgBatchMax], S[kWgBatchMax], rps[kWgBatchMax];
  size_t slab_off[kWgBatchMax], slabb_off[kWgBatchMax], total_floats;
};

// one slice count for the whole batch: ~1.5 workgroups per CU over all items together, K loops of >= 64 rows.  The target was swept
// inside the training step (c2s, two runs each on one box): 256 -> 3.39 ms, 320 -> 3.27, 384 -> 3.19, 448 -> 3.28, 512 -> 3.29,
// 640 -> 3.21, 896 -> 3.22, 1280 -> 3.26 (one launch pair per product, as in round 2: 3.34)
int batch_plan(int n_items, const agnn_wgrad_item_t* items, BatchPlan& bp) {
  using namespace agnn;
  if (n_items <= 0 || n_items > kWgBatchMax || !items) return fail(AGNN_EINVAL, "wgrad_batch: %d items (1 .. %d)", n_items, kWgBatchMax);
  bp.n = n_items;
  int64_t tiles = 0;
  for (int i = 0; i < n_items; ++i) {
    const agnn_wgrad_item_t& it = items[i];
    if (it.n <= 0 || it.n >= (int64_t{1} << 31) || it.out_f <= 0 || it.in_f <= 0) return fail(AGNN_EINVAL, "wgrad_batch: item %d: bad sizes", i);
    bp.tiles_out[i] = (it.out_f + 127) / 128;
    bp.tiles_in[i] = (it.in_f + 127) / 128;
    tiles += static_cast<int64_t>(bp.tiles_out[i]) * bp.tiles_in[i];
  }
  int64_t S = 384 / tiles;
  if (S > 64) S = 64;
  if (S < 1) S = 1;
  size_t off = 0;
  for (int i = 0; i < n_items; ++i) {
    const agnn_wgrad_item_t& it = items[i];
    int64_t s_i = S;
    const int64_t max_s = (it.n + 63) / 64;
    if (s_i > max_s) s_i = max_s;
    int64_t rps = (it.n + s_i - 1) / s_i;
    rps = (rps + 1) & ~int64_t{1};
    bp.rps[i] = static_cast<int>(rps);
    bp.S[i] = static_cast<int>((it.n + rps - 1) / rps);
    const size_t plane = static_cast<size_t>(bp.tiles_out[i]) * 128 * bp.tiles_in[i] * 128;
    bp.slab_off[i] = off;
    off += static_cast<size_t>(bp.S[i]) * plane;
    bp.slabb_off[i] = off;
    off += static_cast<size_t>(bp.S[i]) * bp.tiles_out[i] * 128;
    off = (off + 63) & ~size_t{63};                 // every item's slabs start on a 256-byte boundary
  }
  bp.total_floats = off;
  return AGNN_OK;
}
}  // namespace

extern "C" size_t agnn_wgrad_batch_workspace_bytes(int32_t n_items, const agnn_wgrad_item_t* items) {
  BatchPlan bp;
  if (batch_plan(n_items, items, bp) != AGNN_OK) return 0;
  return bp.total_floats * sizeof(float) + 256;
}

extern "C" int agnn_wgrad_batch_f32(int32_t n_items, const agnn_wgrad_item_t* items, void* workspace, size_t workspace_bytes,
                                    agnn_stream_t stream_) {
  using namespace agnn;
  BatchPlan bp;
  if (int rc = batch_plan(n_items, items, bp)) return rc;
  if (!workspace || workspace_bytes < bp.total_floats * sizeof(float) + 256) return fail(AGNN_ENOMEM, "wgrad_batch: workspace too small");
  float* ws = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~uintptr_t{255});
  WgBatch wb{};
  RedBatch rb{};
  wb.n = rb.n = n_items;
  int wg = 0, red = 0;
  for (int i = 0; i < n_items; ++i) {
    const agnn_wgrad_item_t& it = items[i];
    if ((it.out_f & 1) || (it.in_f & 1) || (it.ld_dy & 1) || (it.ld_x & 1) || (it.ld_dw & 1)) return fail(AGNN_EALIGN, "wgrad_batch: item %d: widths and leading dimensions must be even", i);
    if (!it.dy || !it.x || !it.dw) return fail(AGNN_EINVAL, "wgrad_batch: item %d: null argument", i);
    if ((reinterpret_cast<uintptr_t>(it.dy) | reinterpret_cast<uintptr_t>(it.x) | reinterpret_cast<uintptr_t>(it.dw) | reinterpret_cast<uintptr_t>(it.db)) & 7u)
      return fail(AGNN_EALIGN, "wgrad_batch: item %d: pointers must be 8-byte aligned", i);
    if (it.ld_dy < it.out_f || it.ld_x < it.in_f || it.ld_dw < it.in_f) return fail(AGNN_EINVAL, "wgrad_batch: item %d: leading dimension smaller than the width", i);
    if ((static_cast<int64_t>(bp.rps[i]) + 128) * (it.ld_dy > it.ld_x ? it.ld_dy : it.ld_x) * 4 >= (int64_t{1} << 32))
      return fail(AGNN_EINVAL, "wgrad_batch: item %d: a row slice (%d rows x ld) exceeds the kernel's 32-bit byte offsets", i, bp.rps[i]);
    const int out_pad = bp.tiles_out[i] * 128, in_pad = bp.tiles_in[i] * 128;
    float* slab = ws + bp.slab_off[i];
    float* slab_b = ws + bp.slabb_off[i];
    wb.it[i] = WgArgs{it.dy, it.x, it.ld_dy, it.ld_x, static_cast<int32_t>(it.n), it.out_f, it.in_f, bp.rps[i], bp.tiles_in[i], slab,
                      it.db ? slab_b : nullptr, out_pad, in_pad};
    wb.tiles[i] = bp.tiles_out[i] * bp.tiles_in[i];
    wb.first[i] = wg;
    wg += wb.tiles[i] * bp.S[i];
    rb.it[i] = RedItem{slab, it.db ? slab_b : nullptr, it.dw, it.db, it.ld_dw, bp.S[i], it.out_f, it.in_f, out_pad, in_pad};
    rb.first[i] = red;
    int64_t blocks = (static_cast<int64_t>(it.out_f) * (it.in_f >> 1) + 31) / 32;
    if (blocks > 1024) blocks = 1024;
    red += static_cast<int>(blocks);
  }
  wb.first[n_items] = wg;
  rb.first[n_items] = red;
  hipStream_t s = static_cast<hipStream_t>(stream_);
  hipLaunchKernelGGL(k_wgrad_batch, dim3(static_cast<unsigned>(wg)), dim3(256), 0, s, wb);
  if (int rc = check_launch("wgrad_batch")) return rc;
  hipLaunchKernelGGL(k_wgrad_reduce_batch, dim3(static_cast<unsigned>(red)), dim3(256), 0, s, rb);
  return check_launch("wgrad_batch(reduce)");
}
