// fp32 projection GEMM on the matrix cores:  C[M, N] = A[M, K] * W[N, K]^T (+ bias[N])          (gfx950)
//
// The dense per-relation feature projections of the encoders (reference: PyG SAGEConv lin_l / lin_r behind
// analysisgnn/models/cadence.py:147-159; core/gnn.py:65,75) and every other `nn.Linear` of the path are products of a TALL
// activation matrix (M = 16 000 ... 18 000 notes) with a small weight (N, K <= 1 344): both operands are K-contiguous ("NT").
// Exact fp32 on v_mfma_f32_32x32x2_f32 (64 FLOP / clk / SIMD: 157 TFLOP/s on the chip).
//   * block tile 128 x 128, 4 waves in 2 x 2, a wave owns 64 x 64 = four 32 x 32 accumulators (64 VGPRs): two workgroups
//     (8 waves) per CU, so every SIMD has a second wave to issue MFMAs from while the first waits for LDS;
//   * K in steps of 16 through LDS, double buffered: the global loads of step k + 1 (one 16-byte load per thread and
//     operand, a row's 64 bytes by 4 adjacent threads) are in flight while step k is multiplied; ONE barrier per step;
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

constexpr int BM = 128, BN = 128, BK = 16, LDT = BK + 4;

struct GemmArgs {
  const float* a;
  const float* w;
  const float* bias;
  float* c;
  int64_t ld_a, ld_w, ld_c;
  int32_t M, N, K;
  int32_t tiles_n;
};

__global__ __launch_bounds__(256, 2) void k_gemm_nt(GemmArgs g) {
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
  const float* pw1 = g.w + static_cast<int64_t>(col0 + 64 + sr) * g.ld_w + sk;
  const int so0 = sr * LDT + sk, so1 = (64 + sr) * LDT + sk;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x16{0};

  const int nk = g.K / BK;
  // One step ahead: while step kb is multiplied out of LDS buffer kb & 1, the operands of step kb + 1 are requested into
  // registers and written to the other buffer at the end of the step.  Straight-line code on named registers: the same loop
  // written with a staging struct passed to lambdas put the struct in scratch memory and ran at 68 TFLOP/s instead of 96, as did
  // a two-steps-ahead variant (profiles/r03_gemm.md).
  float4 ra0 = *reinterpret_cast<const float4*>(pa0), ra1 = *reinterpret_cast<const float4*>(pa1);
  float4 rw0 = *reinterpret_cast<const float4*>(pw0), rw1 = *reinterpret_cast<const float4*>(pw1);
  *reinterpret_cast<float4*>(&sA[0][so0]) = ra0;
  *reinterpret_cast<float4*>(&sA[0][so1]) = ra1;
  *reinterpret_cast<float4*>(&sB[0][so0]) = rw0;
  *reinterpret_cast<float4*>(&sB[0][so1]) = rw1;
  __syncthreads();

  const int fr = lane & 31, fk = 4 * (lane >> 5);
  const int fa = (64 * wm + fr) * LDT + fk, fb = (64 * wn + fr) * LDT + fk;
  for (int kb = 0; kb < nk; ++kb) {
    const int cur = kb & 1;
    const bool more = kb + 1 < nk;
    if (more) {
      const int ko = (kb + 1) * BK;
      ra0 = *reinterpret_cast<const float4*>(pa0 + ko);
      ra1 = *reinterpret_cast<const float4*>(pa1 + ko);
      rw0 = *reinterpret_cast<const float4*>(pw0 + ko);
      rw1 = *reinterpret_cast<const float4*>(pw1 + ko);
    }
    const float* A = sA[cur];
    const float* B = sB[cur];
#pragma unroll
    for (int kk = 0; kk < BK; kk += 8) {
      const float4 a0 = *reinterpret_cast<const float4*>(&A[fa + kk]);
      const float4 a1 = *reinterpret_cast<const float4*>(&A[fa + 32 * LDT + kk]);
      const float4 b0 = *reinterpret_cast<const float4*>(&B[fb + kk]);
      const float4 b1 = *reinterpret_cast<const float4*>(&B[fb + 32 * LDT + kk]);
      const float av0[4] = {a0.x, a0.y, a0.z, a0.w}, av1[4] = {a1.x, a1.y, a1.z, a1.w};
      const float bv0[4] = {b0.x, b0.y, b0.z, b0.w}, bv1[4] = {b1.x, b1.y, b1.z, b1.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[j], bv0[j], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[j], bv1[j], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[j], bv0[j], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[j], bv1[j], acc[1][1], 0, 0, 0);
      }
    }
    if (more) {
      const int nx = cur ^ 1;
      *reinterpret_cast<float4*>(&sA[nx][so0]) = ra0;
      *reinterpret_cast<float4*>(&sA[nx][so1]) = ra1;
      *reinterpret_cast<float4*>(&sB[nx][so0]) = rw0;
      *reinterpret_cast<float4*>(&sB[nx][so1]) = rw1;
    }
    __syncthreads();
  }

  // epilogue: + bias, 128-byte row pieces (lanes 0..31 = 32 consecutive columns)
  const int cl = lane & 31, rh = 4 * (lane >> 5);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = col0 + 64 * wn + 32 * j + cl;
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
  if ((N % BN) || (K % BK)) return fail(AGNN_EINVAL, "gemm_nt: N=%d must be a multiple of %d and K=%d of %d", N, BN, K, BK);
  if (M == 0) return AGNN_OK;
  if (!a || !w || !c) return fail(AGNN_EINVAL, "gemm_nt: null argument");
  if (!aligned16(a) || !aligned16(w) || (ld_a & 3) || (ld_w & 3) || ld_a < K || ld_w < K || ld_c < N)
    return fail(AGNN_EALIGN, "gemm_nt: operands must be 16-byte aligned with leading dimensions that are multiples of 4 and >= K (C: >= N)");
  GemmArgs g{a, w, bias, c, ld_a, ld_w, ld_c, static_cast<int32_t>(M), N, K, N / BN};
  const int64_t tiles_m = (M + BM - 1) / BM;
  const int64_t groups = (tiles_m + 7) / 8;
  hipLaunchKernelGGL(k_gemm_nt, dim3(static_cast<unsigned>(groups * 8 * g.tiles_n)), dim3(256), 0, static_cast<hipStream_t>(stream_), g);
  return check_launch("gemm_nt");
}
