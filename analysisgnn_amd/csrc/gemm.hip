// fp32 projection GEMM on the matrix cores:  C[M, N] = A[M, K] * W[N, K]^T (+ bias[N])          (gfx950)
//
// The dense per-relation feature projections of the encoders (reference: PyG SAGEConv lin_l / lin_r behind
// analysisgnn/models/cadence.py:147-159; core/gnn.py:65,75) and every other `nn.Linear` of the path are products of a TALL
// activation matrix (M = 16 000 ... 18 000 notes) with a small weight (N, K <= 1 344): both operands are K-contiguous ("NT").
// Exact fp32 on v_mfma_f32_32x32x2_f32 (64 FLOP / clk / SIMD: 157 TFLOP/s on the chip).
//   * block tile 128 x 128, 4 waves in 2 x 2, a wave owns 64 x 64 = four 32 x 32 accumulators (64 VGPRs): two workgroups
//     (8 waves) per CU, so every SIMD has a second wave to issue MFMAs from while the first waits for LDS;
//   * K in steps of 16 through LDS, double buffered; the global loads run TWO steps ahead in two register sets, the staging
//     stores of step k + 1 sit between the two MFMA blocks of step k, and the LDS fragments of the next 8 k are requested
//     before the MFMAs of the current 8 (round 3, by ablation: with the loads one step ahead at the top of the step and the
//     staging stores at its end the loop ran at 93 TFLOP/s, its MFMAs alone at 127 — now 111 - 118, the library's rate on the
//     SAGE shape); ONE barrier per step, which the compiler sinks into the second MFMA block;
//   * LDS rows are padded to 20 floats: the 16-byte fragment reads of 8 consecutive rows fall into 8 different bank groups;
//   * fragments: lane l reads 4 consecutive k of row (l % 32) at k offset 4 * (l / 32) with one ds_read_b128 — MFMA j of the
//     four that follow multiplies k pair (j, 4 + j).  Any pairing serves as long as both operands use the same one, so no
//     shuffle is needed between the 16-byte LDS read and the MFMA's one-k-per-lane-half operand layout;
//   * workgroup -> tile mapping keeps the N tiles of one row block on ONE XCD (ids b, b + 8, ...): the second read of an
//     A tile is an L2 hit, HBM sees the activation matrix once.
// D layout of the 32 x 32 tile: lane l, register r -> row (r & 3) + 8 (r >> 2) + 4 (l >> 5), column l & 31.
#include "agnn_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BK = 16, LDT = BK + 4;

struct GemmArgs {
  const float* a;
  const float* w;
  const float* bias;
  float* c;
  int64_t ld_a, ld_w, ld_c;
  int32_t M, N, K;
  int32_t tiles_n;
};

// BN = 128: a wave owns 64 x 64 (four accumulators); BN = 64: 64 x 32 (two) — twice the workgroups, for shapes whose 128 x 128 tiling
// gives one workgroup per CU (M = 16 000, N = 256: 250 tiles): a SIMD with ONE wave has nobody to issue MFMAs while that wave waits
// for its LDS fragments, the barrier or the staging stores
template <int BN>
__global__ __launch_bounds__(256, 2) void k_gemm_nt(GemmArgs g) {
  constexpr int NJ = BN / 64;              // 32-column accumulators per wave
  __shared__ __attribute__((aligned(16))) float sA[2][BM * LDT];
  __shared__ __attribute__((aligned(16))) float sB[2][BN * LDT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware tile order: consecutive block ids go round-robin over the 8 XCDs; ids b, b + 8, ... (one XCD) take the N tiles
  // of the same row block one after the other
  const int b = blockIdx.x;
  const int group = b / (8 * g.tiles_n), in_group = b - group * 8 * g.tiles_n;
  const int tile_m = group * 8 + (in_group & 7), tile_n = in_group >> 3;
  const int row0 = tile_m * BM, col0 = tile_n * BN;
  if (row0 >= g.M) return;

  // global -> LDS staging: thread t moves rows (t / 4) and 64 + (t / 4), k offset 4 * (t % 4), of both operands
  const int sr = tid >> 2, sk = 4 * (tid & 3);
  const float* pa0 = g.a + static_cast<int64_t>(min(row0 + sr, g.M - 1)) * g.ld_a + sk;          // rows past M: clamped, never stored
  const float* pa1 = g.a + static_cast<int64_t>(min(row0 + 64 + sr, g.M - 1)) * g.ld_a + sk;
  const float* pw0 = g.w + static_cast<int64_t>(col0 + sr) * g.ld_w + sk;
  const float* pw1 = g.w + static_cast<int64_t>(col0 + (NJ == 2 ? 64 : 0) + sr) * g.ld_w + sk;      // (BN = 64: unused)
  const int so0 = sr * LDT + sk, so1 = (64 + sr) * LDT + sk;

  f32x16 acc[2][NJ];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f32x16{0};

  const int nk = g.K / BK;
  typedef float f32x4 __attribute__((ext_vector_type(4)));
#define LD4(p) (*reinterpret_cast<const f32x4*>(p))
#define ST4(p, v) (*reinterpret_cast<f32x4*>(p) = (v))
  // two staging register sets (X: even steps, Y: odd steps), requested TWO steps ahead; fragment sets F (kk = 0) and G (kk = 8)
  f32x4 xa0, xa1, xw0, xw1, ya0, ya1, yw0, yw1;
  f32x4 fa0, fa1, fb0, fb1, ga0, ga1, gb0, gb1;
  const int fr = lane & 31, fk = 4 * (lane >> 5);
  const int fa = (64 * wm + fr) * LDT + fk, fb = (32 * NJ * wn + fr) * LDT + fk;
  const int last = (nk - 1) * BK;
  xa0 = LD4(pa0); xa1 = LD4(pa1); xw0 = LD4(pw0); xw1 = LD4(pw1);
  {
    const int k1 = min(BK, last);
    ya0 = LD4(pa0 + k1); ya1 = LD4(pa1 + k1); yw0 = LD4(pw0 + k1); yw1 = LD4(pw1 + k1);
  }
  ST4(&sA[0][so0], xa0); ST4(&sA[0][so1], xa1); ST4(&sB[0][so0], xw0);
  if (NJ == 2) ST4(&sB[0][so1], xw1);
  {
    const int k2 = min(2 * BK, last);
    xa0 = LD4(pa0 + k2); xa1 = LD4(pa1 + k2); xw0 = LD4(pw0 + k2); xw1 = LD4(pw1 + k2);
  }
  __syncthreads();
  fa0 = LD4(&sA[0][fa]); fa1 = LD4(&sA[0][fa + 32 * LDT]); fb0 = LD4(&sB[0][fb]); fb1 = NJ == 2 ? LD4(&sB[0][fb + 32 * LDT]) : fb0;

#define MFMA_BLOCK(A0, A1, B0, B1)                                                                         \
  _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                          \
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0[j], B0[j], acc[0][0], 0, 0, 0);                    \
    if (NJ == 2) acc[0][NJ - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0[j], B1[j], acc[0][NJ - 1], 0, 0, 0); \
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1[j], B0[j], acc[1][0], 0, 0, 0);                    \
    if (NJ == 2) acc[1][NJ - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1[j], B1[j], acc[1][NJ - 1], 0, 0, 0); \
  }
  // one K step out of LDS buffer CUR; the staging set S* (the NEXT step's operands, requested two steps ago) goes to the other
  // buffer between the two MFMA blocks and is re-requested for step KB + 3 (indices clamped to the last step: no branch around a
  // memory operation — a conditional load costs a full vmcnt drain at the join)
#define STEP(CUR, KB, SA0, SA1, SW0, SW1)                                                                  \
  {                                                                                                        \
    ga0 = LD4(&sA[CUR][fa + 8]); ga1 = LD4(&sA[CUR][fa + 32 * LDT + 8]); gb0 = LD4(&sB[CUR][fb + 8]);      \
    gb1 = NJ == 2 ? LD4(&sB[CUR][fb + 32 * LDT + 8]) : gb0;                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    MFMA_BLOCK(fa0, fa1, fb0, fb1)                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    ST4(&sA[(CUR) ^ 1][so0], SA0); ST4(&sA[(CUR) ^ 1][so1], SA1); ST4(&sB[(CUR) ^ 1][so0], SW0);           \
    if (NJ == 2) ST4(&sB[(CUR) ^ 1][so1], SW1);                                                            \
    {                                                                                                      \
      const int k3 = min(((KB) + 3) * BK, last);                                                           \
      SA0 = LD4(pa0 + k3); SA1 = LD4(pa1 + k3); SW0 = LD4(pw0 + k3); SW1 = LD4(pw1 + k3);                  \
    }                                                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    MFMA_BLOCK(ga0, ga1, gb0, gb1)                                                                         \
    __syncthreads();                                                                                       \
    fa0 = LD4(&sA[(CUR) ^ 1][fa]); fa1 = LD4(&sA[(CUR) ^ 1][fa + 32 * LDT]); fb0 = LD4(&sB[(CUR) ^ 1][fb]); \
    fb1 = NJ == 2 ? LD4(&sB[(CUR) ^ 1][fb + 32 * LDT]) : fb0;                                              \
  }
  int kb = 0;
  for (; kb + 2 <= nk; kb += 2) {
    STEP(0, kb, ya0, ya1, yw0, yw1)
    STEP(1, kb + 1, xa0, xa1, xw0, xw1)
  }
  if (kb < nk) STEP(0, kb, ya0, ya1, yw0, yw1)
#undef STEP
#undef MFMA_BLOCK
#undef LD4
#undef ST4

  // epilogue: + bias, 128-byte row pieces (lanes 0..31 = 32 consecutive columns)
  const int cl = lane & 31, rh = 4 * (lane >> 5);
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int col = col0 + 32 * NJ * wn + 32 * j + cl;
    const float bj = g.bias != nullptr ? g.bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row0 + 64 * wm + 32 * i + (r & 3) + 8 * (r >> 2) + rh;
        if (row < g.M) g.c[static_cast<int64_t>(row) * g.ld_c + col] = acc[i][j][r] + bj;
      }
    }
  }
}

}  // namespace

extern "C" int agnn_gemm_nt_f32(const float* a, int64_t ld_a, const float* w, int64_t ld_w, const float* bias, int64_t M, int32_t N,
                                int32_t K, float* c, int64_t ld_c, agnn_stream_t stream_) {
  using namespace agnn;
  if (M < 0 || M >= (int64_t{1} << 31) || N <= 0 || K <= 0) return fail(AGNN_EINVAL, "gemm_nt: bad sizes M=%lld N=%d K=%d", (long long)M, N, K);
  if ((N % 64) || (K % BK)) return fail(AGNN_EINVAL, "gemm_nt: N=%d must be a multiple of 64 and K=%d of %d", N, K, BK);
  if (M == 0) return AGNN_OK;
  if (!a || !w || !c) return fail(AGNN_EINVAL, "gemm_nt: null argument");
  if (!aligned16(a) || !aligned16(w) || (ld_a & 3) || (ld_w & 3) || ld_a < K || ld_w < K || ld_c < N)
    return fail(AGNN_EALIGN, "gemm_nt: operands must be 16-byte aligned with leading dimensions that are multiples of 4 and >= K (C: >= N)");
  const int64_t tiles_m = (M + BM - 1) / BM;
  const int64_t groups = (tiles_m + 7) / 8;
  // 64-wide column tiles when 128-wide ones would leave the chip with fewer than two workgroups per CU (or N is not a multiple of 128)
  const bool narrow = (N % 128) != 0 || tiles_m * (N / 128) < 512;
  GemmArgs g{a, w, bias, c, ld_a, ld_w, ld_c, static_cast<int32_t>(M), N, K, narrow ? N / 64 : N / 128};
  const dim3 grid(static_cast<unsigned>(groups * 8 * g.tiles_n));
  if (narrow) hipLaunchKernelGGL(k_gemm_nt<64>, grid, dim3(256), 0, static_cast<hipStream_t>(stream_), g);
  else hipLaunchKernelGGL(k_gemm_nt<128>, grid, dim3(256), 0, static_cast<hipStream_t>(stream_), g);
  return check_launch("gemm_nt");
}
