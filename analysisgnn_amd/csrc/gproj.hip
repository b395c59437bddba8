// Grouped projection: the last Linear(h2 -> C_t) of all T task heads as ONE launch per direction.
//
//   out[n, off_g + c] = b[off_g + c] + sum_k a[n, g*K + k] * w[off_g + c, k]        c < C_g = off_{g+1} - off_g
//
// The reference runs T = 21 separate nn.Linear(64, C_t) (analysisgnn/models/analysis.py:486-496, :546-548).  As one
// library GEMM against the block-diagonal [sum C, T*64] weight it costs 21x the useful FLOPs (311 + 216 + 243 us
// for forward / dX / dW per step, profiles/r01_h).  The useful work is tiny (K = 64) and HBM-bound: read a
// [N, T*64] once, write [N, sum C] once.  Kernels (fp32-input MFMA 32x32x2, one wave = one 32-row x one group item):
//   k_gproj_fwd  one workgroup = 128 rows x one group; the `a` tile and each 32-class weight tile go through LDS (coalesced
//                16-byte loads, weights fetched once per workgroup); A operand = K/2 floats per lane (the two lane halves
//                take the two halves of the K range), B operand likewise; K/2 MFMAs per 32-class tile.
//   k_gproj_dx   da[n, g*K + k] = sum_c dout[n, off_g + c] w[off_g + c, k]; the class range is split between the lane halves.
//   k_gproj_dw   dw[off_g + c, k] = sum_n dout[n, off_g + c] a[n, g*K + k] and db: both operands "n-major", row pairs
//                per MFMA step, N cut into S slices -> slabs -> agnn::launch_slab_reduce (fixed order, no atomics).
// Lanes past a group's class count read the group's last class (valid memory) and their results are never stored.
#include <cstdlib>

#include "agnn_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GpArgs {
  const float* a;          // [n, G*K]
  const float* w;          // [sumC, K]
  const float* b;          // [sumC] or null
  const float* dout;       // [n, sumC] (ld_out)
  const int32_t* seg_off;  // device [G+1]
  float* out;              // fwd: [n, sumC]
  float* da;               // dx: [n, G*K]
  float* slab;             // dw: [S][sumC][K]
  float* slab_b;           // dw: [S][sumC]
  int64_t ld_a, ld_out, ld_da, n_rows;
  int32_t G, n_row_tiles, n_tiles32, S, rows_per_slice, sum_c;
};

__device__ __forceinline__ int d_row(int r, int kk) { return (r & 3) + 8 * (r >> 2) + 4 * kk; }   // C/D layout of 32x32 MFMA

// LDS-staged forward: one workgroup = 128 rows x ONE group.  The 128 x K tile of `a` is fetched with coalesced 16-byte
// loads (16 lanes per 256-byte row piece) and the 32 x K weight tile of each 32-class step once per workgroup (all four
// waves use it), instead of every lane gathering its own 128-byte row piece from global memory (profiles/r01_gproj_pmc.md:
// 40 % of the wave cycles waited on those gathers).  Row stride K + 4 floats keeps the 16-byte LDS reads at 2-way conflicts.
template <int K>
__global__ __launch_bounds__(256) void k_gproj_fwd(GpArgs p) {
  constexpr int KH = K / 2, LDA = K + 4;
  __shared__ __attribute__((aligned(16))) float sA[128 * LDA];
  __shared__ __attribute__((aligned(16))) float sW[2][32 * LDA];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rb = blockIdx.x / p.G, g = blockIdx.x - rb * p.G;
  const int j = lane & 31, kk = lane >> 5;
  const int off = p.seg_off[g], C = p.seg_off[g + 1] - off;
  if (C <= 0) return;
  constexpr int F4 = K / 4;                           // float4 per row
  constexpr int RPP = 256 / F4;                       // rows per pass of the whole workgroup
  const int lr = tid / F4, lc = tid - lr * F4;
#pragma unroll
  for (int r0 = 0; r0 < 128; r0 += RPP) {
    int64_t row = static_cast<int64_t>(rb) * 128 + r0 + lr;
    if (row > p.n_rows - 1) row = p.n_rows - 1;
    const float4 v = *reinterpret_cast<const float4*>(p.a + row * p.ld_a + g * K + 4 * lc);
    *reinterpret_cast<float4*>(&sA[(r0 + lr) * LDA + 4 * lc]) = v;
  }
  auto load_w = [&](int ct, int buf) {
#pragma unroll
    for (int r0 = 0; r0 < 32; r0 += RPP) {
      const int c = ct * 32 + r0 + lr;
      const int cc = c < C ? c : C - 1;
      const float4 v = *reinterpret_cast<const float4*>(p.w + static_cast<int64_t>(off + cc) * K + 4 * lc);
      *reinterpret_cast<float4*>(&sW[buf][(r0 + lr) * LDA + 4 * lc]) = v;
    }
  };
  const int n_ct = (C + 31) >> 5;
  load_w(0, 0);
  __syncthreads();
  float av[KH];
  {
    const float4* ap = reinterpret_cast<const float4*>(&sA[(wave * 32 + j) * LDA + kk * KH]);
#pragma unroll
    for (int q = 0; q < KH / 4; ++q) {
      const float4 v = ap[q];
      av[4 * q] = v.x; av[4 * q + 1] = v.y; av[4 * q + 2] = v.z; av[4 * q + 3] = v.w;
    }
  }
  for (int ct = 0; ct < n_ct; ++ct) {
    const int buf = ct & 1;
    if (ct + 1 < n_ct) load_w(ct + 1, buf ^ 1);       // next weight tile into the other buffer while this one is used
    float bv[KH];
    const float4* wp = reinterpret_cast<const float4*>(&sW[buf][j * LDA + kk * KH]);
#pragma unroll
    for (int q = 0; q < KH / 4; ++q) {
      const float4 v = wp[q];
      bv[4 * q] = v.x; bv[4 * q + 1] = v.y; bv[4 * q + 2] = v.z; bv[4 * q + 3] = v.w;
    }
    f32x16 acc = {0};
#pragma unroll
    for (int s = 0; s < KH; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bv[s], acc, 0, 0, 0);
    const int col = ct * 32 + j;
    if (col < C) {
      const float bias = p.b != nullptr ? p.b[off + col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t ro = static_cast<int64_t>(rb) * 128 + wave * 32 + d_row(r, kk);
        if (ro < p.n_rows) p.out[ro * p.ld_out + off + col] = acc[r] + bias;
      }
    }
    __syncthreads();                                   // buffer `buf` free for tile ct + 2, tile ct + 1 visible
  }
}

// ------------------------------------------------------------------------------------------
// Whole-row kernels (K = 64): a logits row leaves ONE CU.  profiles/r01_gproj_pmc.md: in the kernel above a row is written
// by 21 workgroups on different XCDs (one per group) in 8 ... 740-byte strips, ~20 of its 60 us go to those stores.  The
// tile scheme the whole-row kernels share (the first of them, a 64-row workgroup per launch, was replaced by the persistent
// kernel below: profiles/r02_gproj.md):
//   * `a` streams through an LDS image in chunks of groups, one wave instruction reading 1 KB of ONE row;
//   * v_mfma_f32_16x16x4_f32 tiles, 16 rows x 16 classes: classes are padded to 16 instead of 32 (848 instead of 1088
//     padded classes for the 21 heads).  Lane (i = lane & 15, q = lane >> 4) owns k = 16u + 4q + e (u, e < 4) of BOTH
//     operands — any k order serves as long as A and B agree — so the four float4 of a lane ARE its 16 operand registers;
//   * the 16-class tiles of a chunk's groups go round-robin to the multiplier waves; a weight tile (16 classes x 64 = 4 KB,
//     L2) is fetched while the wave's previous tile is computed;
//   * two accumulators per row tile (even / odd k-steps): the 16-step chain is issue-bound (32 cycles), not latency-bound (40).
// D layout of the 16x16 tile: column = lane & 15 (class), rows 4q + r in register r.
// ------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kGpMaxTiles16 = 128;      // capacity of the flat tile list (host: n_tiles32 <= 64, a 32-class tile is at most two)
constexpr int kGpLdO = 308;             // floats per LDS row of a chunk of logits (4 * 308 = 16 mod 64 banks)

struct GpWTile {
  float4 w[4];
  float bias;
  int t, off, C, gc;                    // wave-uniform: tile index in its group, the group's column range, group index in the chunk
  bool ok;                              // (persistent kernel) the slot holds a tile
};

__device__ __forceinline__ void gp_rowtile16(const GpArgs& p, const float4 (&af)[4], const GpWTile& r, float bias_on, int i, int q,
                                             int64_t row0, int rt16, float* so, int chunk_off) {
  const float av[16] = {af[0].x, af[0].y, af[0].z, af[0].w, af[1].x, af[1].y, af[1].z, af[1].w,
                        af[2].x, af[2].y, af[2].z, af[2].w, af[3].x, af[3].y, af[3].z, af[3].w};
  const float bv[16] = {r.w[0].x, r.w[0].y, r.w[0].z, r.w[0].w, r.w[1].x, r.w[1].y, r.w[1].z, r.w[1].w,
                        r.w[2].x, r.w[2].y, r.w[2].z, r.w[2].w, r.w[3].x, r.w[3].y, r.w[3].z, r.w[3].w};
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < 16; s += 2) {
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv[s], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s + 1], bv[s + 1], acc1, 0, 0, 0);
  }
  const int c = 16 * r.t + i;
  if (so != nullptr) {
    // the chunk's logits are assembled in LDS (row stride kGpLdO: the four lane groups land 16 banks apart) and leave
    // the CU as whole row pieces afterwards
    if (c < r.C) {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) so[(rt16 + 4 * q + rr) * kGpLdO + r.off - chunk_off + c] = acc0[rr] + acc1[rr] + bias_on * r.bias;
    }
    return;
  }
  // A chunk wider than the LDS image: straight to global memory, 64-byte pieces.  Unconditional: a lane past the group's
  // classes computed class C - 1 again (its weight row was clamped) and a lane past the last row computed the last row
  // again, so writing to the clamped address repeats a correct value.
  const int cs = c < r.C ? c : r.C - 1;
  float* op = p.out + r.off + cs;
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    int64_t ro = row0 + rt16 + 4 * q + rr;
    ro = ro < p.n_rows ? ro : p.n_rows - 1;
    op[ro * p.ld_out] = acc0[rr] + acc1[rr] + bias_on * r.bias;
  }
}

// ------------------------------------------------------------------------------------------
// Whole-row forward, persistent and wave-specialised: one workgroup of 12 waves per CU walks 32-row blocks; the groups are
// cut into ceil(G / 16) equal chunks (21 heads: 11 + 10 groups, 704 / 640 columns of `a`).  Same tiles and operand layout
// as above.  What changes, and why (in-kernel stamps, scripts/gproj_stamps.py, profiles/r02_gproj.md):
//   * vmcnt retires in program order, so in a wave that has the next chunk of `a` in flight (HBM, microseconds) every
//     weight tile requested later (L2, a few hundred ns) is usable only after the whole chunk has landed: the tile loop of
//     the kernel above runs at HBM latency once per chunk.  Here waves 8..11 only move `a` (8 rows each: load the next
//     stage's chunk into registers, write it into the image between the two barriers of a stage) and waves 0..7 only
//     fetch weight tiles (one tile ahead, two register sets), multiply and store; neither queue waits for the other;
//   * tile rounds: with four groups per chunk the 53 16-class tiles of the 21 heads take 11 rounds of 8 waves; with two
//     chunks they take 8 (6.6 would be a perfect deal).  The chunk image is 32 x 1028 floats, so a weight tile serves two
//     row tiles and the logits go straight to global memory in 64-byte pieces;
//   * the NEXT stage is the other chunk of this row block or the first chunk of the workgroup's next row block: only the
//     very first chunk load of a workgroup is exposed.
// A wave's tile slots of a chunk are the flat tiles chunk_begin + wave + 8t, t < n_slots (even); an empty slot repeats the
// chunk's first tile (same values to the same addresses), which keeps the loop free of branches around memory operations.
// ------------------------------------------------------------------------------------------
constexpr int kGpLdA16 = 1028;          // floats per LDS row of a chunk (1024 + 4: 16-byte fragments of 16 rows, 16 bank groups)


template <int NQ>                         // 256-column quarters per chunk (host: ceil(groups per chunk / 4))
__global__ __launch_bounds__(768) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_gproj_fwd_rows32(GpArgs p, int CG, int n_chunks) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool mover = wave >= 8;
  const int i = lane & 15, q = lane >> 4;
  __shared__ __attribute__((aligned(16))) float sA[32 * kGpLdA16];
  __shared__ uint32_t s_tiles[8][kGpMaxTiles16];
  const int width = p.G * 64;
  const int n_rb = static_cast<int>((p.n_rows + 31) >> 5);
  if (static_cast<int>(blockIdx.x) >= n_rb) return;
  // ---- movers: wave 8 + m owns rows 8m .. 8m + 7 of the block, lane l columns 4l .. 4l + 3 of each 256-column quarter j
  // (named registers: an array indexed inside unrolled loops stayed in scratch here)
#define GP_MV_DECL(r) float4 mv##r##0, mv##r##1, mv##r##2, mv##r##3;
  GP_MV_DECL(0) GP_MV_DECL(1) GP_MV_DECL(2) GP_MV_DECL(3) GP_MV_DECL(4) GP_MV_DECL(5) GP_MV_DECL(6) GP_MV_DECL(7)
#define GP_MV_LOAD1(r, j, rb, c)                                                                  \
  if (NQ > (j)) {                                                                                 \
    int64_t row_ = static_cast<int64_t>(rb) * 32 + 8 * (wave - 8) + (r);                          \
    row_ = row_ < p.n_rows ? row_ : p.n_rows - 1;                                                 \
    const int end_ = (c) + 1 < n_chunks ? ((c) + 1) * CG * 64 : width;                            \
    const int col_ = (c) * CG * 64 + 256 * (j) + 4 * lane;                                        \
    const int cc_ = col_ < end_ ? col_ : end_ - 4;  /* past the chunk: repeat a valid piece */     \
    mv##r##j = *reinterpret_cast<const float4*>(p.a + row_ * p.ld_a + cc_);                       \
  }
#define GP_MV_WRITE1(r, j, rb, c) if (NQ > (j)) *reinterpret_cast<float4*>(&sA[(8 * (wave - 8) + (r)) * kGpLdA16 + 256 * (j) + 4 * lane]) = mv##r##j;
#define GP_MV_ROW(M, r, rb, c) M(r, 0, rb, c) M(r, 1, rb, c) M(r, 2, rb, c) M(r, 3, rb, c)
#define GP_MV_ALL(M, rb, c) { GP_MV_ROW(M, 0, rb, c) GP_MV_ROW(M, 1, rb, c) GP_MV_ROW(M, 2, rb, c) GP_MV_ROW(M, 3, rb, c) GP_MV_ROW(M, 4, rb, c) GP_MV_ROW(M, 5, rb, c) GP_MV_ROW(M, 6, rb, c) GP_MV_ROW(M, 7, rb, c) }
#define GP_MOVE_LOAD(rb, c) GP_MV_ALL(GP_MV_LOAD1, rb, c)
#define GP_MOVE_WRITE() GP_MV_ALL(GP_MV_WRITE1, 0, 0)
  // Two loops, one per role, each with the same barriers (s_barrier counts waves, not program addresses): the registers of
  // the two roles then share one allocation instead of adding up.
  if (mover) {
    int rb = blockIdx.x;
    GP_MOVE_LOAD(rb, 0)
    GP_MOVE_WRITE()
    __syncthreads();
    for (; rb < n_rb; rb += gridDim.x) {
      for (int c = 0; c < n_chunks; ++c) {
        const bool more = c + 1 < n_chunks || rb + static_cast<int>(gridDim.x) < n_rb;
        if (!more) break;
        const int nx_rb = c + 1 < n_chunks ? rb : rb + static_cast<int>(gridDim.x), nx_c = c + 1 < n_chunks ? c + 1 : 0;
        GP_MOVE_LOAD(nx_rb, nx_c)
        __syncthreads();                               // every tile of the chunk has read its fragments
        GP_MOVE_WRITE()
        __syncthreads();
      }
    }
    return;
  }
  // ---- multipliers: the tile table
  const int gl = lane < p.G ? lane : p.G - 1;
  const int c0 = p.seg_off[gl], c1 = p.seg_off[gl + 1];
  const int my_nt = lane < p.G ? ((c1 - c0 + 15) >> 4) : 0;
  int start = 0, T = 0;
  for (int g = 0; g < p.G; ++g) {
    start = lane == g ? T : start;
    T += __builtin_amdgcn_readlane(my_nt, g);
  }
  start = lane < p.G ? start : T;
  for (int t = 0; t < my_nt; ++t) s_tiles[wave][start + t] = static_cast<uint32_t>(lane) | (static_cast<uint32_t>(t) << 8);
  const float* bias_p = p.b != nullptr ? p.b : p.w;
  const float bias_on = p.b != nullptr ? 1.f : 0.f;
  // ---- multipliers: weight tiles, two register sets
  GpWTile wA, wB;
#define GP_W_FETCH(WT, CH, SL)                                                                       \
  {                                                                                               \
    const int cb_ = __builtin_amdgcn_readlane(start, CG * (CH) < 63 ? CG * (CH) : 63);              \
    const int ce_ = __builtin_amdgcn_readlane(start, CG * ((CH) + 1) < 63 ? CG * ((CH) + 1) : 63);  \
    const int n_ = cb_ + wave + 8 * (SL);                                                          \
    const uint32_t ds_ = s_tiles[wave][n_ < ce_ ? n_ : (cb_ < T ? cb_ : T - 1)];                  \
    const int g_ = __builtin_amdgcn_readfirstlane(static_cast<int>(ds_ & 0xffu));                 \
    WT.t = __builtin_amdgcn_readfirstlane(static_cast<int>(ds_ >> 8));                             \
    WT.off = __builtin_amdgcn_readlane(c0, g_);                                                    \
    WT.C = __builtin_amdgcn_readlane(c1, g_) - WT.off;                                              \
    WT.gc = g_ - (CH) * CG;                                                                         \
    const int c_ = 16 * WT.t + i;                                                                  \
    const int cc_ = c_ < WT.C ? c_ : WT.C - 1;                                                      \
    const float4* wp_ = reinterpret_cast<const float4*>(p.w + static_cast<int64_t>(WT.off + cc_) * 64 + 4 * q); \
    WT.w[0] = wp_[0]; WT.w[1] = wp_[4]; WT.w[2] = wp_[8]; WT.w[3] = wp_[12];                          \
    WT.bias = bias_p[WT.off + cc_];                                                                 \
  }
#define GP_TILE(WT)                                                                                \
  {                                                                                               \
    float4 f_[2][4];                                                                              \
    _Pragma("unroll") for (int rt = 0; rt < 2; ++rt)                                              \
      _Pragma("unroll") for (int u = 0; u < 4; ++u)                                               \
        f_[rt][u] = *reinterpret_cast<const float4*>(&sA[(16 * rt + i) * kGpLdA16 + WT.gc * 64 + 16 * u + 4 * q]); \
    _Pragma("unroll") for (int rt = 0; rt < 2; ++rt) gp_rowtile16(p, f_[rt], WT, bias_on, i, q, row0, 16 * rt, nullptr, 0); \
  }
  int rb = blockIdx.x;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // this wave's tile list is read back below
  __builtin_amdgcn_wave_barrier();
  GP_W_FETCH(wA, 0, 0)
  __syncthreads();
  for (; rb < n_rb; rb += gridDim.x) {
    const int64_t row0 = static_cast<int64_t>(rb) * 32;
    for (int c = 0; c < n_chunks; ++c) {
      const bool more = c + 1 < n_chunks || rb + static_cast<int>(gridDim.x) < n_rb;
      const int nx_c = c + 1 < n_chunks ? c + 1 : 0;
      const int cb = __builtin_amdgcn_readlane(start, CG * c < 63 ? CG * c : 63);
      const int ce = __builtin_amdgcn_readlane(start, CG * (c + 1) < 63 ? CG * (c + 1) : 63);
      const int n_slots = ((ce - cb + 15) >> 4) << 1;    // tiles per wave, rounded up to the two register sets
      for (int t = 0; t < n_slots; t += 2) {
        GP_W_FETCH(wB, c, t + 1)
        GP_TILE(wA)
        if (t + 2 < n_slots) GP_W_FETCH(wA, c, t + 2)
        else GP_W_FETCH(wA, nx_c, 0)                     // the first tile of the next stage (weights do not depend on the rows)
        GP_TILE(wB)
      }
      if (n_slots == 0) GP_W_FETCH(wA, nx_c, 0)           // (a chunk of class-less groups)
      if (more) {
        __syncthreads();
        __syncthreads();                               // the movers have rewritten the image
      }
    }
  }
#undef GP_MV_DECL
#undef GP_MV_LOAD1
#undef GP_MV_WRITE1
#undef GP_MV_ROW
#undef GP_MV_ALL
#undef GP_MOVE_LOAD
#undef GP_MOVE_WRITE
#undef GP_W_FETCH
#undef GP_TILE
}

template <int K>
__global__ __launch_bounds__(256) void k_gproj_dx(GpArgs p) {
  constexpr int NT = K / 32;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t item = static_cast<int64_t>(blockIdx.x) * 4 + wave;
  const int rt = static_cast<int>(item / p.G), g = static_cast<int>(item - static_cast<int64_t>(rt) * p.G);
  if (rt >= p.n_row_tiles) return;
  const int j = lane & 31, kk = lane >> 5;
  const int off = p.seg_off[g], C = p.seg_off[g + 1] - off;
  const int Ch = (C + 1) >> 1;                      // classes [0, Ch) on lane half 0, [Ch, C) on lane half 1
  int64_t row = static_cast<int64_t>(rt) * 32 + j;
  if (row > p.n_rows - 1) row = p.n_rows - 1;
  const float* dp = p.dout + row * p.ld_out + off;
  const float* wp = p.w + static_cast<int64_t>(off) * K + j;
  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = f32x16{0};
  for (int s0 = 0; s0 < Ch; s0 += 8) {
    float avv[8], bvv[8][NT];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int s = s0 + u, c = kk * Ch + s;
      const float m = (s < Ch && c < C) ? 1.f : 0.f;
      const int cc = c < C ? c : C - 1;
      avv[u] = dp[cc] * m;
#pragma unroll
      for (int t = 0; t < NT; ++t) bvv[u][t] = wp[static_cast<int64_t>(cc) * K + t * 32];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(avv[u], bvv[u][t], acc[t], 0, 0, 0);
  }
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t ro = static_cast<int64_t>(rt) * 32 + d_row(r, kk);
      if (ro < p.n_rows) p.da[ro * p.ld_da + g * K + t * 32 + j] = acc[t][r];
    }
}

template <int K>
__global__ __launch_bounds__(256) void k_gproj_dw(GpArgs p) {
  constexpr int NT = K / 32;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t id = static_cast<int64_t>(blockIdx.x) * 4 + wave;
  const int slice = static_cast<int>(id / p.n_tiles32), tile = static_cast<int>(id - static_cast<int64_t>(slice) * p.n_tiles32);
  if (slice >= p.S) return;
  int g = -1, ct = 0, off = 0, C = 0;
  for (int q = 0, base = 0; q < p.G; ++q) {          // wave-uniform scan of the (<= 32) groups
    const int o = p.seg_off[q], cq = p.seg_off[q + 1] - o, nt = (cq + 31) >> 5;
    if (tile < base + nt) { g = q; ct = tile - base; off = o; C = cq; break; }
    base += nt;
  }
  if (g < 0) return;
  const int j = lane & 31, kk = lane >> 5;
  const int r0 = slice * p.rows_per_slice;
  int r1 = r0 + p.rows_per_slice;
  if (r1 > p.n_rows) r1 = static_cast<int>(p.n_rows);
  const int col = ct * 32 + j;
  const int c = col < C ? col : C - 1;
  const float* dp = p.dout + off + c;
  const float* ap = p.a + g * K + j;
  const int n_last = static_cast<int>(p.n_rows) - 1;
  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = f32x16{0};
  float bs = 0.f;
  auto fetch = [&](int base, float (&avv)[8], float (&bvv)[8][NT]) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int n = base + 2 * u + kk;
      const float m = n < r1 ? 1.f : 0.f;
      const int64_t nn = n < n_last ? n : n_last;
      avv[u] = dp[nn * p.ld_out] * m;
#pragma unroll
      for (int t = 0; t < NT; ++t) bvv[u][t] = ap[nn * p.ld_a + t * 32];
    }
  };
  auto mma = [&](const float (&avv)[8], const float (&bvv)[8][NT]) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(avv[u], bvv[u][t], acc[t], 0, 0, 0);
      bs += avv[u];
    }
  };
  // three register sets in rotation: the 32 rows after the current 16 are in flight during its MFMAs (two waves per SIMD and
  // 1024 cycles of MFMA per set: one set ahead covered half of an HBM round trip).  A fetch past the slice reads the last
  // row with a zero mask: no branch around the loads.
  if (r0 < r1) {
    float a0[8], b0[8][NT], a1[8], b1[8][NT], a2[8], b2[8][NT];
    fetch(r0, a0, b0);
    fetch(r0 + 16, a1, b1);
    for (int base = r0; base < r1; base += 48) {
      fetch(base + 32, a2, b2);
      mma(a0, b0);
      if (base + 16 >= r1) break;
      fetch(base + 48, a0, b0);
      mma(a1, b1);
      if (base + 32 >= r1) break;
      fetch(base + 64, a1, b1);
      mma(a2, b2);
    }
  }
  float* slab = p.slab + (static_cast<int64_t>(slice) * p.sum_c + off) * K;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ci = ct * 32 + d_row(r, kk);          // D row = class index (A operand), D column = k
      if (ci < C) slab[static_cast<int64_t>(ci) * K + t * 32 + j] = acc[t][r];
    }
  if (p.slab_b != nullptr) {
    bs += __shfl_xor(bs, 32, 64);                     // the two lane halves hold the two row parities
    if (kk == 0 && col < C) p.slab_b[static_cast<int64_t>(slice) * p.sum_c + off + col] = bs;
  }
}

// ------------------------------------------------------------------------------------------
// Whole-row input gradient (K = 64), persistent and wave-specialised like the forward above: one workgroup of 12 waves per
// CU walks 32-row blocks.  da[n, g*64 + k] = sum_c dout[n, off_g + c] w[off_g + c, k].
//   * waves 8..11 move `dout`: 8 rows each, whole rows (lane l takes columns 64j + l: 256 contiguous bytes per wave
//     instruction, no alignment asked of the rows), held in registers while the block before is multiplied, then written
//     into a 32 x ld_img image in which every group's classes start at a multiple of 16 floats and are zero-padded to a
//     multiple of 16 (the image is zeroed once; the padding is never written).  A lane's columns — and so their places in
//     the image — are the same for every row and block.
//   * waves 0..7 multiply.  A unit of work is (group, 16-row tile), a step 16 of its classes: one ds_read_b128 gives lane
//     (i = lane & 15, q = lane >> 4) the classes 4q .. 4q + 3 of row i, four 16-byte loads give it w[class 4q + e][4i .. 4i+3],
//     and MFMA (e, t) multiplies "k = class 4q + e" with "column 4i + t": the 16 MFMAs of a step cover all 64 columns, and
//     the four accumulators of a lane are four CONSECUTIVE columns, so the tile leaves as 16-byte stores, 256 contiguous
//     bytes per row.  Units are handed out largest group first from an LDS counter (the class counts are ragged: 1 .. 12
//     steps, 13 of the 21 heads have one).  A wave's steps form ONE stream across its units: the weights of a step are
//     requested two steps ahead (three register sets, handed on by register moves), the next unit is taken from the counter
//     when the last step of the one before is requested — a unit start costs no memory latency.
// The kernel moves 40 MB in and 86 MB out for 1.7 GFLOP: it is bound by the HBM stream, not by the MFMAs.
// ------------------------------------------------------------------------------------------
constexpr int kGpDxLd = 1092;           // floats per image row (32 x 1092 x 4 = 139 776 bytes; 273 16-byte slots: odd).  A constant, so
                                        // that the movers' 8 rows are immediate offsets of one address register per column piece

struct GdStep {                           // wave-uniform description of one step, travelling with its weight registers
  int g, rt, s, C, po;                    // group, row tile, step, classes of the group, place of the group in the image
  bool last, valid;
};

template <int NJ>                         // 64-column pieces per row of dout (host: ceil(sum_c / 64))
__global__ __launch_bounds__(768) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_gproj_dx_rows(GpArgs p) {
  constexpr int ld_img = kGpDxLd;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool mover = wave >= 8;
  __shared__ __attribute__((aligned(16))) float sD[32 * kGpDxLd];
  __shared__ uint32_t s_units[64];
  __shared__ int s_next;
  __shared__ int s_place[1024];                           // column of dout -> its place in an image row
  const int n_rb = static_cast<int>((p.n_rows + 31) >> 5);
  if (static_cast<int>(blockIdx.x) >= n_rb) return;
  // ---- movers' registers: row r (of the wave's 8), piece j: column 64j + lane
#define GD_DECL(r) float mv##r##0, mv##r##1, mv##r##2, mv##r##3, mv##r##4, mv##r##5, mv##r##6, mv##r##7, mv##r##8, mv##r##9, \
                         mv##r##10, mv##r##11, mv##r##12, mv##r##13, mv##r##14, mv##r##15;
  GD_DECL(0) GD_DECL(1) GD_DECL(2) GD_DECL(3) GD_DECL(4) GD_DECL(5) GD_DECL(6) GD_DECL(7)
#define GD_LOAD1(r, j, rb)                                                                        \
  if (NJ > (j)) {                                                                                 \
    int64_t row_ = static_cast<int64_t>(rb) * 32 + 8 * (wave - 8) + (r);                          \
    row_ = row_ < p.n_rows ? row_ : p.n_rows - 1;                                                 \
    const float* rp_ = p.dout + row_ * p.ld_out;          /* uniform base + one 32-bit lane offset (+ immediate) */ \
    mv##r##j = (j) + 1 < NJ ? rp_[ulane + 64u * (j)] : rp_[ulast];                                \
  }
#define GD_WRITE1(r, j, rb) if (NJ > (j)) sDw[(r) * ld_img + pc##j] = mv##r##j;   /* no per-lane condition: see GD_PC */
#define GD_ROW(M, r, rb) M(r, 0, rb) M(r, 1, rb) M(r, 2, rb) M(r, 3, rb) M(r, 4, rb) M(r, 5, rb) M(r, 6, rb) M(r, 7, rb) \
                         M(r, 8, rb) M(r, 9, rb) M(r, 10, rb) M(r, 11, rb) M(r, 12, rb) M(r, 13, rb) M(r, 14, rb) M(r, 15, rb)
#define GD_ALL(M, rb) { GD_ROW(M, 0, rb) GD_ROW(M, 1, rb) GD_ROW(M, 2, rb) GD_ROW(M, 3, rb) GD_ROW(M, 4, rb) GD_ROW(M, 5, rb) GD_ROW(M, 6, rb) GD_ROW(M, 7, rb) }
  const unsigned ulane = lane;                             // only the last piece of a row can run past its end
  const unsigned ulast = 64 * (NJ - 1) + lane < p.sum_c ? 64u * (NJ - 1) + ulane : static_cast<unsigned>(p.sum_c - 1);
  // The movers do nothing but move: their first block is on its way while the other eight waves prepare the image and the
  // tables (anything more here and the compiler parks the rows in scratch, one wait per load).
  if (mover) {
    GD_ALL(GD_LOAD1, blockIdx.x)
    float* sDw = &sD[8 * (wave - 8) * ld_img];            // this wave's 8 rows
    __syncthreads();                                     // the image is zero, the places are known
    // this lane's columns 64j + lane; one past the row: the last float of the image row, which nothing reads
#define GD_PC(j)                                                                                  \
  int pc##j = ld_img - 1;                                                                         \
  if (NJ > (j)) {                                                                                 \
    pc##j = 64 * (j) + lane < p.sum_c ? s_place[64 * (j) + lane] : ld_img - 1;                    \
    __builtin_assume(pc##j >= 0 && pc##j < ld_img);        /* (lets the row offsets become immediate offsets of the LDS writes) */ \
  }
    GD_PC(0) GD_PC(1) GD_PC(2) GD_PC(3) GD_PC(4) GD_PC(5) GD_PC(6) GD_PC(7)
    GD_PC(8) GD_PC(9) GD_PC(10) GD_PC(11) GD_PC(12) GD_PC(13) GD_PC(14) GD_PC(15)
    GD_ALL(GD_WRITE1, 0)
    __syncthreads();                                     // the first block is in the image
    for (int rb = blockIdx.x; rb + static_cast<int>(gridDim.x) < n_rb; rb += gridDim.x) {
      GD_ALL(GD_LOAD1, rb + gridDim.x)
      __syncthreads();                                   // every unit of the block has been multiplied
      GD_ALL(GD_WRITE1, 0)
      if (wave == 8 && lane == 0) s_next = 0;
      __syncthreads();
    }
    return;
  }
  // ---- zero the image once (the padding of every group stays zero)
  for (int e = threadIdx.x * 4; e < 32 * ld_img; e += 512 * 4) *reinterpret_cast<float4*>(&sD[e]) = make_float4(0.f, 0.f, 0.f, 0.f);
  // ---- group table in lanes: lane g holds group g's class range, its 16-class steps, its place in the image
  const int gl = lane < p.G ? lane : p.G - 1;
  const int c0 = p.seg_off[gl], c1 = p.seg_off[gl + 1];
  const int my_steps = lane < p.G ? ((c1 - c0 + 15) >> 4) : 0;
  int poff = 0, tot = 0, rank = 0;
  for (int g = 0; g < p.G; ++g) {
    const int sg = __builtin_amdgcn_readlane(my_steps, g);
    poff = lane == g ? tot : poff;
    tot += 16 * sg;
    rank += (sg > my_steps || (sg == my_steps && g < lane)) ? 1 : 0;
  }
  // ---- every column's place in the image row, worked out once by the 512 threads
  // (a column's place = the column + the padding of every group that ends at or before it)
  const int my_pad = 16 * my_steps - (lane < p.G ? c1 - c0 : 0);
  // (uniform trip count: readlane of a lane that sits out a divergent loop is undefined)
  for (int base = 0; base < p.sum_c; base += 512) {
    const int col = base + static_cast<int>(threadIdx.x);
    int pl = col;
    for (int h = 0; h < p.G; ++h) pl += __builtin_amdgcn_readlane(c1, h) <= col ? __builtin_amdgcn_readlane(my_pad, h) : 0;
    if (col < p.sum_c) s_place[col] = pl;
  }
  // ---- multipliers
  const int i = lane & 15, q = lane >> 4;
  const int n_units = 2 * p.G;
  if (wave == 0 && lane == 0) s_next = 0;
  if (wave == 0 && lane < p.G) {                         // largest group first
    s_units[2 * rank] = static_cast<uint32_t>(lane);
    s_units[2 * rank + 1] = static_cast<uint32_t>(lane) | 256u;
  }
  __syncthreads();
  __syncthreads();
  for (int rb = blockIdx.x; rb < n_rb; rb += gridDim.x) {
    const int64_t row0 = static_cast<int64_t>(rb) * 32;
    // the request stream: unit (fg, frt) with fsteps steps, fs the next one to request
    int fg = 0, frt = 0, foff = 0, fC = 0, fpo = 0, fsteps = 1, fs = 0;
    bool fvalid = false;
#define GD_GRAB()                                                                                 \
  {                                                                                               \
    int u_ = 0;                                                                                   \
    if (lane == 0) u_ = atomicAdd(&s_next, 1);                                                    \
    u_ = __builtin_amdgcn_readfirstlane(u_);                                                      \
    fvalid = u_ < n_units;                                                                        \
    const uint32_t ds_ = s_units[fvalid ? u_ : 0];                                                \
    fg = __builtin_amdgcn_readfirstlane(static_cast<int>(ds_ & 0xffu));                           \
    frt = __builtin_amdgcn_readfirstlane(static_cast<int>(ds_ >> 8));                             \
    foff = __builtin_amdgcn_readlane(c0, fg);                                                     \
    fC = __builtin_amdgcn_readlane(c1, fg) - foff;                                                \
    fpo = __builtin_amdgcn_readlane(poff, fg);                                                    \
    fsteps = (fC + 15) >> 4;                                                                      \
    fsteps = fsteps > 0 ? fsteps : 1;                      /* a class-less group still writes its zeros */ \
    fs = 0;                                                                                       \
  }
    // request the weights of the stream's next step into (D, B0 .. B3); then move the stream on.  Always loads (from a
    // clamped, valid row): no branch around memory operations.
#define GD_NEXT(D, B)                                                                             \
  {                                                                                               \
    D.g = fg; D.rt = frt; D.s = fs; D.C = fC; D.po = fpo; D.last = fs + 1 == fsteps; D.valid = fvalid; \
    const int k_ = 16 * fs + 4 * q;                                                               \
    const int hi_ = fC > 0 ? fC - 1 : 0;                                                          \
    const float* wg_ = p.w + 4 * i;                                                               \
    int r0_ = foff + (k_ < hi_ ? k_ : hi_), r1_ = foff + (k_ + 1 < hi_ ? k_ + 1 : hi_);         \
    int r2_ = foff + (k_ + 2 < hi_ ? k_ + 2 : hi_), r3_ = foff + (k_ + 3 < hi_ ? k_ + 3 : hi_); \
    r0_ = r0_ < p.sum_c ? r0_ : p.sum_c - 1; r1_ = r1_ < p.sum_c ? r1_ : p.sum_c - 1;             \
    r2_ = r2_ < p.sum_c ? r2_ : p.sum_c - 1; r3_ = r3_ < p.sum_c ? r3_ : p.sum_c - 1;             \
    B##0 = *reinterpret_cast<const float4*>(wg_ + static_cast<int64_t>(r0_) * 64);                \
    B##1 = *reinterpret_cast<const float4*>(wg_ + static_cast<int64_t>(r1_) * 64);                \
    B##2 = *reinterpret_cast<const float4*>(wg_ + static_cast<int64_t>(r2_) * 64);                \
    B##3 = *reinterpret_cast<const float4*>(wg_ + static_cast<int64_t>(r3_) * 64);                \
    ++fs;                                                                                         \
    if (fvalid && fs == fsteps) GD_GRAB()                                                         \
  }
#define GD_MMA1(av, B)                                                                            \
  acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, B.x, acc0, 0, 0, 0);                            \
  acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, B.y, acc1, 0, 0, 0);                            \
  acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, B.z, acc2, 0, 0, 0);                            \
  acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, B.w, acc3, 0, 0, 0);
    GdStep d0, d1, d2;
    float4 x0, x1, x2, x3, y0, y1, y2, y3, z0, z1, z2, z3;   // three weight register sets in rotation (no register moves: a
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0, acc3 = acc0;   // move would wait for the newest request)
    // the fragment of `dout` one step ahead (a class-less group's place in the image is the next group's: zero instead)
#define GD_FRAG(D) (D.C > 0 ? *reinterpret_cast<const float4*>(&sD[(16 * D.rt + i) * ld_img + D.po + 16 * D.s + 4 * q]) : make_float4(0.f, 0.f, 0.f, 0.f))
    // one step: multiply with (DC, BC); (DN, ..) is the step after it; the step after that is requested into (DF, BF), the
    // set the step before this one used
#define GD_STEP(DC, BC, DN, DF, BF)                                                               \
  {                                                                                               \
    GD_NEXT(DF, BF)                                                                               \
    const float4 a = an;                                                                          \
    an = GD_FRAG(DN);                                                                             \
    if (DC.s == 0) { acc0 = f32x4{0.f, 0.f, 0.f, 0.f}; acc1 = acc0; acc2 = acc0; acc3 = acc0; }    \
    GD_MMA1(a.x, BC##0) GD_MMA1(a.y, BC##1) GD_MMA1(a.z, BC##2) GD_MMA1(a.w, BC##3)                \
    if (DC.last) {                                                                                \
      /* lane (i, q): rows 4q + r of the tile, columns 4i .. 4i + 3 of the group.  A row past the end was a copy of the */ \
      /* last row in the image: the same values go to the last row again. */                       \
      float* dp = p.da + DC.g * 64 + 4 * i;                                                       \
      _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                             \
        int64_t ro = row0 + 16 * DC.rt + 4 * q + r;                                               \
        ro = ro < p.n_rows ? ro : p.n_rows - 1;                                                   \
        GD_STORE(reinterpret_cast<float4*>(dp + ro * p.ld_da), make_float4(acc0[r], acc1[r], acc2[r], acc3[r])) \
      }                                                                                           \
    }                                                                                             \
  }
#define GD_STORE(ptr, v) *(ptr) = (v);
    GD_GRAB()
    GD_NEXT(d0, x)
    GD_NEXT(d1, y)
    float4 an = GD_FRAG(d0);
    for (;;) {
      if (!d0.valid) break;
      GD_STEP(d0, x, d1, d2, z)
      if (!d1.valid) break;
      GD_STEP(d1, y, d2, d0, x)
      if (!d2.valid) break;
      GD_STEP(d2, z, d0, d1, y)
    }
    if (rb + static_cast<int>(gridDim.x) < n_rb) {
      __syncthreads();
      __syncthreads();                                   // the movers have rewritten the image and reset the counter
    }
  }
#undef GD_DECL
#undef GD_LOAD1
#undef GD_WRITE1
#undef GD_ROW
#undef GD_ALL
#undef GD_PC
#undef GD_GRAB
#undef GD_NEXT
#undef GD_MMA1
#undef GD_FRAG
#undef GD_STEP
#undef GD_STORE
}

int cu_count() {                          // compute units of the current device (one persistent workgroup each)
  static const int n = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    return v;
  }();
  return n;
}

struct Plan { int S, rows_per_slice; };

Plan make_plan(int64_t n, int n_tiles32) {
  int64_t S = 2048 / n_tiles32;                     // 2 waves per SIMD over the whole chip, rounded DOWN: a 513th workgroup
                                                    // would be a third on its CU (its SIMDs 3 waves deep: +50 % on the critical path)
  const int64_t max_s = (n + 63) / 64;
  if (S > max_s) S = max_s;
  if (S > 64) S = 64;
  if (S < 1) S = 1;
  int64_t rps = (n + S - 1) / S;
  rps = (rps + 1) & ~int64_t{1};
  Plan p;
  p.rows_per_slice = static_cast<int>(rps);
  p.S = static_cast<int>((n + rps - 1) / rps);
  return p;
}

int check_common(const char* what, const void* a, int64_t ld_a, const void* w, const int32_t* seg_off, int32_t G, int32_t K,
                 int32_t n_tiles32, int64_t n_rows) {
  using namespace agnn;
  if (n_rows < 0 || n_rows >= (int64_t{1} << 31) || G <= 0 || G > AGNN_MAX_SEG || n_tiles32 < 0)
    return fail(AGNN_EINVAL, "%s: bad sizes n=%lld groups=%d tiles=%d", what, (long long)n_rows, G, n_tiles32);
  if (K != 32 && K != 64 && K != 128) return fail(AGNN_EINVAL, "%s: K=%d (32, 64 or 128)", what, K);
  if (!a || !w || !seg_off) return fail(AGNN_EINVAL, "%s: null argument", what);
  if (!aligned16(a) || !aligned16(w) || (ld_a & 3)) return fail(AGNN_EALIGN, "%s: a / w must be 16-byte aligned, ld_a %% 4 == 0", what);
  if (ld_a < static_cast<int64_t>(G) * K) return fail(AGNN_EINVAL, "%s: ld_a smaller than groups*K", what);
  return AGNN_OK;
}

}  // namespace


extern "C" int agnn_gproj_fwd_f32(const float* a, int64_t ld_a, const float* w, const float* b, const int32_t* seg_off,
                                  int32_t n_groups, int32_t K, int32_t n_tiles32, int64_t n_rows, float* out, int64_t ld_out,
                                  agnn_stream_t stream_) {
  using namespace agnn;
  if (int rc = check_common("gproj_fwd", a, ld_a, w, seg_off, n_groups, K, n_tiles32, n_rows)) return rc;
  if (!out) return fail(AGNN_EINVAL, "gproj_fwd: null output");
  if (n_rows == 0 || n_tiles32 == 0) return AGNN_OK;
  GpArgs p{};
  p.a = a; p.w = w; p.b = b; p.seg_off = seg_off; p.out = out;
  p.ld_a = ld_a; p.ld_out = ld_out; p.n_rows = n_rows; p.G = n_groups;
  p.n_row_tiles = static_cast<int32_t>((n_rows + 31) / 32);
  hipStream_t s = static_cast<hipStream_t>(stream_);
  if (K == 64 && n_groups <= 32 && n_tiles32 <= kGpMaxTiles16 / 2) {   // whole-row kernel (other shapes: one group per workgroup)
    {
      const int n_chunks = (n_groups + 15) / 16, CG = (n_groups + n_chunks - 1) / n_chunks, NQ = (CG + 3) / 4;
      const int64_t n_rb = (n_rows + 31) / 32;
      const dim3 grid(static_cast<unsigned>(n_rb < cu_count() ? n_rb : cu_count()));
      if (NQ == 1) hipLaunchKernelGGL(k_gproj_fwd_rows32<1>, grid, dim3(768), 0, s, p, CG, n_chunks);
      else if (NQ == 2) hipLaunchKernelGGL(k_gproj_fwd_rows32<2>, grid, dim3(768), 0, s, p, CG, n_chunks);
      else if (NQ == 3) hipLaunchKernelGGL(k_gproj_fwd_rows32<3>, grid, dim3(768), 0, s, p, CG, n_chunks);
      else hipLaunchKernelGGL(k_gproj_fwd_rows32<4>, grid, dim3(768), 0, s, p, CG, n_chunks);
      return check_launch("gproj_fwd(rows32)");
    }
  }
  const int64_t row_blocks = (n_rows + 127) / 128;
  const dim3 grid(static_cast<unsigned>(row_blocks * n_groups));
  if (K == 32) hipLaunchKernelGGL(k_gproj_fwd<32>, grid, dim3(256), 0, s, p);
  else if (K == 64) hipLaunchKernelGGL(k_gproj_fwd<64>, grid, dim3(256), 0, s, p);
  else hipLaunchKernelGGL(k_gproj_fwd<128>, grid, dim3(256), 0, s, p);
  return check_launch("gproj_fwd");
}

extern "C" size_t agnn_gproj_workspace_bytes(int64_t n_rows, int32_t sum_c, int32_t K, int32_t n_tiles32) {
  if (n_rows <= 0 || sum_c <= 0 || K <= 0 || n_tiles32 <= 0) return 0;
  const Plan pl = make_plan(n_rows, n_tiles32);
  return static_cast<size_t>(pl.S) * sum_c * (static_cast<size_t>(K) + 1) * sizeof(float) + 256;
}

extern "C" int agnn_gproj_bwd_f32(const float* dout, int64_t ld_dout, const float* a, int64_t ld_a, const float* w,
                                  const int32_t* seg_off, int32_t n_groups, int32_t K, int32_t n_tiles32, int32_t sum_c,
                                  int64_t n_rows, float* da, int64_t ld_da, float* dw, float* db, void* workspace,
                                  size_t workspace_bytes, agnn_stream_t stream_) {
  using namespace agnn;
  if (int rc = check_common("gproj_bwd", a, ld_a, w, seg_off, n_groups, K, n_tiles32, n_rows)) return rc;
  if (!dout || sum_c < 0 || ld_dout < sum_c) return fail(AGNN_EINVAL, "gproj_bwd: bad dout / sum_c");
  if (n_rows == 0 || n_tiles32 == 0 || sum_c == 0) return AGNN_OK;
  hipStream_t s = static_cast<hipStream_t>(stream_);
  GpArgs p{};
  p.a = a; p.w = w; p.dout = dout; p.seg_off = seg_off;
  p.ld_a = ld_a; p.ld_out = ld_dout; p.n_rows = n_rows; p.G = n_groups; p.sum_c = sum_c; p.n_tiles32 = n_tiles32;
  p.n_row_tiles = static_cast<int32_t>((n_rows + 31) / 32);
  if (da != nullptr) {
    if (ld_da < static_cast<int64_t>(n_groups) * K) return fail(AGNN_EINVAL, "gproj_bwd: ld_da smaller than groups*K");
    p.da = da; p.ld_da = ld_da;
    const int ld_img = sum_c + 15 * n_groups + 4;          // at most: every group padded to 16 classes, + the dump slot
    const bool rows_ok = K == 64 && n_groups <= 32 && sum_c >= 1 && sum_c <= 1024 && ld_img <= kGpDxLd && aligned16(da) && (ld_da & 3) == 0;
    if (rows_ok) {                                       // whole-row kernel (other shapes: one wave per (row tile, group))
      const int64_t n_rb = (n_rows + 31) / 32;
      const dim3 grid(static_cast<unsigned>(n_rb < cu_count() ? n_rb : cu_count()));
      switch ((sum_c + 63) / 64) {
#define GD_CASE(n) case n: hipLaunchKernelGGL(k_gproj_dx_rows<n>, grid, dim3(768), 0, s, p); break;
        GD_CASE(1) GD_CASE(2) GD_CASE(3) GD_CASE(4) GD_CASE(5) GD_CASE(6) GD_CASE(7) GD_CASE(8) GD_CASE(9) GD_CASE(10)
        GD_CASE(11) GD_CASE(12) GD_CASE(13) GD_CASE(14) GD_CASE(15)
        default: hipLaunchKernelGGL(k_gproj_dx_rows<16>, grid, dim3(768), 0, s, p); break;
#undef GD_CASE
      }
      if (int rc = check_launch("gproj_dx(rows)")) return rc;
    } else {
    const int64_t items = static_cast<int64_t>(p.n_row_tiles) * n_groups;
    const dim3 grid(static_cast<unsigned>((items + 3) / 4));
    if (K == 32) hipLaunchKernelGGL(k_gproj_dx<32>, grid, dim3(256), 0, s, p);
    else if (K == 64) hipLaunchKernelGGL(k_gproj_dx<64>, grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL(k_gproj_dx<128>, grid, dim3(256), 0, s, p);
    if (int rc = check_launch("gproj_dx")) return rc;
    }
  }
  if (dw != nullptr) {
    const size_t need = agnn_gproj_workspace_bytes(n_rows, sum_c, K, n_tiles32);
    if (!workspace || workspace_bytes < need) return fail(AGNN_ENOMEM, "gproj_bwd: workspace %zu < %zu bytes", workspace_bytes, need);
    const Plan pl = make_plan(n_rows, n_tiles32);
    char* ws = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~uintptr_t{255});
    p.slab = reinterpret_cast<float*>(ws);
    p.slab_b = db != nullptr ? p.slab + static_cast<size_t>(pl.S) * sum_c * K : nullptr;
    p.S = pl.S;
    p.rows_per_slice = pl.rows_per_slice;
    const int64_t waves = static_cast<int64_t>(pl.S) * n_tiles32;
    const dim3 grid(static_cast<unsigned>((waves + 3) / 4));
    if (K == 32) hipLaunchKernelGGL(k_gproj_dw<32>, grid, dim3(256), 0, s, p);
    else if (K == 64) hipLaunchKernelGGL(k_gproj_dw<64>, grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL(k_gproj_dw<128>, grid, dim3(256), 0, s, p);
    if (int rc = check_launch("gproj_dw")) return rc;
    return launch_slab_reduce(p.slab, p.slab_b, pl.S, sum_c, K, sum_c, K, dw, K, db, s);
  }
  return AGNN_OK;
}
