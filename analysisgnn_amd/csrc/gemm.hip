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
// NN = true: the second operand is K-MAJOR, w[K, N] (C = A w + b) — the input-gradient product dX = dY W with the weight as it
// lies in memory.  Its tile sits in LDS as it is read, [k][n] (row stride BN + 8: the two lane halves, 4 k rows apart, fall into
// different bank halves), and a lane's four k values of a fragment come from four 4-byte reads (32 consecutive n per half: no conflict)
template <int BN, bool NN>
__global__ __launch_bounds__(256, 2) void k_gemm_nt(GemmArgs g) {
  constexpr int NJ = BN / 64;              // 32-column accumulators per wave
  constexpr int LDN = BN + 8;
  __shared__ __attribute__((aligned(16))) float sA[2][BM * LDT];
  __shared__ __attribute__((aligned(16))) float sB[2][NN ? BK * LDN : BN * LDT];
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
  const int so0 = sr * LDT + sk, so1 = (64 + sr) * LDT + sk;
  // second operand: NT — rows col0 + (t / 4) (+ 64) of w[N, K], as the first; NN — k rows (t / (BN / 4)) (+ 8) of w[K, N], 16 bytes of n each
  const int nr = NN ? tid / (BN / 4) : 0, nc = NN ? 4 * (tid % (BN / 4)) : 0;
  const float* pw0 = NN ? g.w + static_cast<int64_t>(nr) * g.ld_w + col0 + nc : g.w + static_cast<int64_t>(col0 + sr) * g.ld_w + sk;
  const float* pw1 = NN ? g.w + static_cast<int64_t>(nr + (NJ == 2 ? 8 : 0)) * g.ld_w + col0 + nc
                        : g.w + static_cast<int64_t>(col0 + (NJ == 2 ? 64 : 0) + sr) * g.ld_w + sk;      // (BN = 64: unused)
  const int sbo0 = NN ? nr * LDN + nc : so0, sbo1 = NN ? (nr + 8) * LDN + nc : so1;
  const int64_t wmul = NN ? g.ld_w : 1;       // elements between two k of the second operand

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
  const int fa = (64 * wm + fr) * LDT + fk;
  const int fb = NN ? fk * LDN + 32 * NJ * wn + fr : (32 * NJ * wn + fr) * LDT + fk;
  // the four k values (kk + fk .. + 3) of a lane's B fragment, columns fr (+ 32)
#define BFRAG(BUF, KK, D0, D1)                                                                             \
  if (NN) {                                                                                                \
    const float* q_ = &sB[BUF][(KK) * LDN + fb];                                                           \
    D0 = f32x4{q_[0], q_[LDN], q_[2 * LDN], q_[3 * LDN]};                                                  \
    D1 = NJ == 2 ? f32x4{q_[32], q_[LDN + 32], q_[2 * LDN + 32], q_[3 * LDN + 32]} : D0;                   \
  } else {                                                                                                 \
    D0 = LD4(&sB[BUF][fb + (KK)]);                                                                         \
    D1 = NJ == 2 ? LD4(&sB[BUF][fb + 32 * LDT + (KK)]) : D0;                                               \
  }
  const int last = (nk - 1) * BK;
  xa0 = LD4(pa0); xa1 = LD4(pa1); xw0 = LD4(pw0); xw1 = LD4(pw1);
  {
    const int k1 = min(BK, last);
    ya0 = LD4(pa0 + k1); ya1 = LD4(pa1 + k1); yw0 = LD4(pw0 + k1 * wmul); yw1 = LD4(pw1 + k1 * wmul);
  }
  ST4(&sA[0][so0], xa0); ST4(&sA[0][so1], xa1); ST4(&sB[0][sbo0], xw0);
  if (NJ == 2) ST4(&sB[0][sbo1], xw1);
  {
    const int k2 = min(2 * BK, last);
    xa0 = LD4(pa0 + k2); xa1 = LD4(pa1 + k2); xw0 = LD4(pw0 + k2 * wmul); xw1 = LD4(pw1 + k2 * wmul);
  }
  __syncthreads();
  fa0 = LD4(&sA[0][fa]); fa1 = LD4(&sA[0][fa + 32 * LDT]);
  BFRAG(0, 0, fb0, fb1)

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
    ga0 = LD4(&sA[CUR][fa + 8]); ga1 = LD4(&sA[CUR][fa + 32 * LDT + 8]);                                   \
    BFRAG(CUR, 8, gb0, gb1)                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    MFMA_BLOCK(fa0, fa1, fb0, fb1)                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    ST4(&sA[(CUR) ^ 1][so0], SA0); ST4(&sA[(CUR) ^ 1][so1], SA1); ST4(&sB[(CUR) ^ 1][sbo0], SW0);          \
    if (NJ == 2) ST4(&sB[(CUR) ^ 1][sbo1], SW1);                                                           \
    {                                                                                                      \
      const int k3 = min(((KB) + 3) * BK, last);                                                           \
      SA0 = LD4(pa0 + k3); SA1 = LD4(pa1 + k3); SW0 = LD4(pw0 + k3 * wmul); SW1 = LD4(pw1 + k3 * wmul);    \
    }                                                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    MFMA_BLOCK(ga0, ga1, gb0, gb1)                                                                         \
    __syncthreads();                                                                                       \
    fa0 = LD4(&sA[(CUR) ^ 1][fa]); fa1 = LD4(&sA[(CUR) ^ 1][fa + 32 * LDT]);                               \
    BFRAG((CUR) ^ 1, 0, fb0, fb1)                                                                          \
  }
  int kb = 0;
  for (; kb + 2 <= nk; kb += 2) {
    STEP(0, kb, ya0, ya1, yw0, yw1)
    STEP(1, kb + 1, xa0, xa1, xw0, xw1)
  }
  if (kb < nk) STEP(0, kb, ya0, ya1, yw0, yw1)
#undef STEP
#undef BFRAG
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

namespace {
int launch_gemm(bool nn, const float* a, int64_t ld_a, const float* w, int64_t ld_w, const float* bias, int64_t M, int32_t N, int32_t K,
                float* c, int64_t ld_c, agnn_stream_t stream_, const char* who) {
  using namespace agnn;
  if (M < 0 || M >= (int64_t{1} << 31) || N <= 0 || K <= 0) return fail(AGNN_EINVAL, "%s: bad sizes M=%lld N=%d K=%d", who, (long long)M, N, K);
  if ((N % 64) || (K % BK)) return fail(AGNN_EINVAL, "%s: N=%d must be a multiple of 64 and K=%d of %d", who, N, K, BK);
  if (M == 0) return AGNN_OK;
  if (!a || !w || !c) return fail(AGNN_EINVAL, "%s: null argument", who);
  if (!aligned16(a) || !aligned16(w) || (ld_a & 3) || (ld_w & 3) || ld_a < K || ld_w < (nn ? N : K) || ld_c < N)
    return fail(AGNN_EALIGN, "%s: operands must be 16-byte aligned with leading dimensions that are multiples of 4 and cover a row (C: >= N)", who);
  const int64_t tiles_m = (M + BM - 1) / BM;
  const int64_t groups = (tiles_m + 7) / 8;
  // 64-wide column tiles when 128-wide ones would leave the chip with fewer than two workgroups per CU (or N is not a multiple of 128)
  const bool narrow = (N % 128) != 0 || tiles_m * (N / 128) < 512;
  GemmArgs g{a, w, bias, c, ld_a, ld_w, ld_c, static_cast<int32_t>(M), N, K, narrow ? N / 64 : N / 128};
  const dim3 grid(static_cast<unsigned>(groups * 8 * g.tiles_n));
  hipStream_t s = static_cast<hipStream_t>(stream_);
  if (nn) {
    if (narrow) hipLaunchKernelGGL((k_gemm_nt<64, true>), grid, dim3(256), 0, s, g);
    else hipLaunchKernelGGL((k_gemm_nt<128, true>), grid, dim3(256), 0, s, g);
  } else {
    if (narrow) hipLaunchKernelGGL((k_gemm_nt<64, false>), grid, dim3(256), 0, s, g);
    else hipLaunchKernelGGL((k_gemm_nt<128, false>), grid, dim3(256), 0, s, g);
  }
  return check_launch(who);
}
}  // namespace

extern "C" int agnn_gemm_nt_f32(const float* a, int64_t ld_a, const float* w, int64_t ld_w, const float* bias, int64_t M, int32_t N,
                                int32_t K, float* c, int64_t ld_c, agnn_stream_t stream_) {
  return launch_gemm(false, a, ld_a, w, ld_w, bias, M, N, K, c, ld_c, stream_, "gemm_nt");
}

extern "C" int agnn_gemm_nn_f32(const float* a, int64_t ld_a, const float* w, int64_t ld_w, const float* bias, int64_t M, int32_t N,
                                int32_t K, float* c, int64_t ld_c, agnn_stream_t stream_) {
  return launch_gemm(true, a, ld_a, w, ld_w, bias, M, N, K, c, ld_c, stream_, "gemm_nn");
}
