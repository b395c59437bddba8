// The task-head block in one pass over the notes:   logits_t = LayerNorm_t(ReLU(x W1_t^T + b1_t)) W2_t^T + b2_t,  t = 1..T   (gfx950)
//
// Reference: T independent `Linear(o, o/2) -> ReLU -> LayerNorm(o/2) -> Linear(o/2, C_t)` heads on the encoder output
// (analysisgnn/models/analysis.py:486-496, :546-548).  As separate launches (library GEMM [N, o] x [o, T*64], segmented ReLU +
// LayerNorm, grouped projection) the [N, T*64] hidden matrix is written once and read twice between them, on the serial stretch
// of the step between the forward and the backward pass (profiles/r02_step_stamps.md).  Here a workgroup keeps its 128 notes'
// encoder rows in registers as MFMA A-fragments and walks a share of the tasks:
//   * first projection: the task's W1_t [64, 128] staged in LDS (rows padded to 132 floats: the 16-byte fragment reads of 8
//     consecutive rows fall into 8 different bank groups), 128 v_mfma_f32_32x32x2_f32 per wave (32 notes x 64 hidden units);
//   * bias, ReLU, LayerNorm on the accumulators: a note's 64 hidden units sit in 32 lanes x 2 tiles, so mean / variance are 4 DPP
//     steps + one v_permlane16_swap per value; the pre-activation z (the backward pass's ReLU mask and LayerNorm input), the
//     normalised y (left operand of the W2 weight gradient) and mean / rstd are written for the backward pass;
//   * y goes through a wave-private LDS tile to change from the accumulator layout (hidden unit = lane) to the A-operand layout
//     (hidden unit = k) and is multiplied with W2_t [C_t, 64] straight from L2 (B-fragments: 16 bytes of a class row per lane),
//     32 MFMAs per 32 classes; logits + b2 are written side by side [N, sum C].
// Tasks are dealt to `S` workgroups per row block (longest first, host side) so that ~2 workgroups per CU exist at C2.
// MFMA-bound: 2 N (128 * T*64 + 64 * sum of C_t rounded up to 32) FLOP; HBM traffic (x + z + y + logits) stays under it.
// D layout of a 32 x 32 tile: lane l, register r -> row (r & 3) + 8 (r >> 2) + 4 (l >> 5), column l & 31.
#include <cstddef>

#include "agnn_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));     // staging copies as native vectors (a float4 struct copy becomes a memcpy through scratch)

constexpr int HK = 128;            // encoder output width (contraction of the first projection)
constexpr int HJ = 64;             // hidden width of a head
constexpr int LD1 = HK + 4;        // LDS row stride of W1_t
constexpr int LDY = HJ + 4;        // LDS row stride of a wave's y tile
constexpr int RB = 128;            // notes per workgroup (4 waves x 32)
constexpr int MAXT = 64;
constexpr int MAXS = 8;

struct HeadsFwd {
  const float* x;
  int64_t ld_x;
  const float* w1;
  const float* b1;
  const float* gamma;
  const float* beta;
  const float* w2;
  const float* b2;
  const int32_t* offs;
  float* z;
  float* y;
  int64_t ld_h;
  float* mean;
  float* rstd;
  float* logits;
  int64_t ld_o;
  int32_t N, T, S, row_blocks, sum_c;
  float eps;
  // the plan, read with scalar loads from the kernel-argument segment (dwords: gfx950 has no scalar byte loads)
  int32_t order[MAXT];
  int32_t begin[MAXS + 1];
  int32_t offs_k[MAXT + 1];
};

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// CHECK: the row block may reach past the last note (the partial block at the end gets its own small launch) — the full blocks'
// stores carry no per-note execution masks.
// Vector-memory waits: loads and stores share one in-order counter (vmcnt) on gfx950, so waiting for a load that was requested
// before a burst of stores waits for (most of) those stores to reach memory.  Every load of a task is therefore requested at the
// START of its first projection (128 MFMAs = 8 192 cycles later it has landed) and waited for right after it, BEFORE the
// task's stores are issued; the next task's W1 goes to LDS at once (all waves are past the barrier), the second projection's
// later class tiles are requested one tile ahead and waited for before that tile's logit stores.
// K order of the fragments: lane (lr, hh) holds k = 64 hh + 4 s + {0..3} at step s of the first projection (32 hh + 4 s + .. of
// the second): any order serves as long as both operands use the same one, and this one makes a lane's 8 (16) fragments one
// contiguous 128 (256) byte piece of its row — for the global loads and for the z / y stores.
template <bool CHECK>
__global__ __launch_bounds__(256, 2) void k_heads_fwd(HeadsFwd g, int first_block) {
  __shared__ __attribute__((aligned(16))) float sW1[HJ * LD1];
  __shared__ __attribute__((aligned(16))) float sZ[4 * 32 * LDY];
  __shared__ __attribute__((aligned(16))) float sGB[2][2 * HJ];          // gamma | beta of the current / the next task
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, hh = lane >> 5;
  // the S workgroups of one row block sit on one XCD (ids b, b + 8, ...): the block's x rows come out of that L2
  const int per = 8 * g.S;
  const int grp = blockIdx.x / per, in = blockIdx.x - grp * per;
  const int rb = first_block + grp * 8 + (in & 7), split = in >> 3;
  if (rb >= g.row_blocks || (!CHECK && (rb + 1) * RB > g.N)) return;        // (a partial last block belongs to the CHECK launch)
  const int row0 = rb * RB + 32 * wave;
  // the task plan is indexed with run-time values: scalar loads from the kernel-argument segment itself (a by-value struct
  // member indexed dynamically would be copied to scratch memory first)
  const auto* ka = (const __attribute__((address_space(4))) uint8_t*)__builtin_amdgcn_kernarg_segment_ptr();
  const auto* k_order = (const __attribute__((address_space(4))) int32_t*)(ka + offsetof(HeadsFwd, order));
  const auto* k_begin = (const __attribute__((address_space(4))) int32_t*)(ka + offsetof(HeadsFwd, begin));
  const auto* k_offs = (const __attribute__((address_space(4))) int32_t*)(ka + offsetof(HeadsFwd, offs_k));
  const int t_begin = k_begin[split], t_end = k_begin[split + 1];
  if (t_begin >= t_end) return;

  // this wave's 32 notes as A-fragments: lane (lr, hh) holds x[row0 + lr][64 hh + 4 s .. + 3], s = 0..15
  const int64_t my_row = CHECK ? min(row0 + lr, g.N - 1) : row0 + lr;
  const bool row_live = !CHECK || row0 + lr < g.N;
  float4 xa[16];
  {
    const float* xp = g.x + my_row * g.ld_x + 64 * hh;
#pragma unroll
    for (int s = 0; s < 16; ++s) xa[s] = *reinterpret_cast<const float4*>(xp + 4 * s);
  }
  float* sZw = sZ + wave * 32 * LDY;

  // W1, gamma, beta of the first task -> LDS (thread: float4 number tid + 256 q of the [64, 128] matrix)
  {
    f32x4 w0[8];
    const int t0 = k_order[t_begin];
    const f32x4* p = reinterpret_cast<const f32x4*>(g.w1 + static_cast<size_t>(t0) * HJ * HK);
#pragma unroll
    for (int q = 0; q < 8; ++q) w0[q] = p[tid + 256 * q];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int f = tid + 256 * q;
      *reinterpret_cast<f32x4*>(&sW1[(f >> 5) * LD1 + 4 * (f & 31)]) = w0[q];
    }
    if (tid < 2 * HJ) sGB[0][tid] = tid < HJ ? g.gamma[t0 * HJ + tid] : g.beta[t0 * HJ + tid - HJ];
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0): the A-fragments too — no wait for them is left inside the task loop
  __syncthreads();

  for (int ti = t_begin; ti < t_end; ++ti) {
    const int t = k_order[ti];
    const int cur = (ti - t_begin) & 1;
    // ---- everything this task reads from memory, requested now ----
    const int tn = k_order[ti + 1 < t_end ? ti + 1 : ti];          // the next task (the last task: itself again — branch-free)
    f32x4 wn[8];
    {
      const f32x4* p = reinterpret_cast<const f32x4*>(g.w1 + static_cast<size_t>(tn) * HJ * HK);
#pragma unroll
      for (int q = 0; q < 8; ++q) wn[q] = p[tid + 256 * q];
    }
    const float gbn = tid < 2 * HJ ? (tid < HJ ? g.gamma[tn * HJ + tid] : g.beta[tn * HJ + tid - HJ]) : 0.f;
    const int c_lo = k_offs[t], C = k_offs[t + 1] - c_lo;
    const int tiles = (C + 31) >> 5;
    float4 bw[8];                                      // second projection, first class tile: lane = (class lr, k 32 hh + 4 s ..)
    {
      const int64_t cr = min(c_lo + lr, g.sum_c - 1);
      const float* wp = g.w2 + cr * HJ + 32 * hh;
#pragma unroll
      for (int s = 0; s < 8; ++s) bw[s] = *reinterpret_cast<const float4*>(wp + 4 * s);
    }
    const float bb0 = g.b1[t * HJ + lr], bb1 = g.b1[t * HJ + 32 + lr];
    __builtin_amdgcn_sched_barrier(0);

    // ---- first projection: 32 notes x 64 hidden units, K = 128 ----
    f32x16 a0 = {0}, a1 = {0};
    {
      const float* B0 = sW1 + lr * LD1 + 64 * hh;
      const float* B1 = B0 + 32 * LD1;
      float4 b0 = *reinterpret_cast<const float4*>(B0), b1 = *reinterpret_cast<const float4*>(B1);
#pragma unroll
      for (int s = 0; s < 16; ++s) {                   // the fragments of step s + 1 are read while step s is multiplied
        float4 n0 = b0, n1 = b1;
        if (s + 1 < 16) {
          n0 = *reinterpret_cast<const float4*>(B0 + 4 * (s + 1));
          n1 = *reinterpret_cast<const float4*>(B1 + 4 * (s + 1));
        }
        a0 = MFMA(xa[s].x, b0.x, a0);
        a1 = MFMA(xa[s].x, b1.x, a1);
        a0 = MFMA(xa[s].y, b0.y, a0);
        a1 = MFMA(xa[s].y, b1.y, a1);
        a0 = MFMA(xa[s].z, b0.z, a0);
        a1 = MFMA(xa[s].z, b1.z, a1);
        a0 = MFMA(xa[s].w, b0.w, a0);
        a1 = MFMA(xa[s].w, b1.w, a1);
        __builtin_amdgcn_sched_barrier(0);
        b0 = n0;
        b1 = n1;
      }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0) only (expcnt 7, lgkmcnt 15 = don't care): the requests above
    __syncthreads();                                   // every wave is through with W1_t: the next task's is written over it
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int f = tid + 256 * q;
      *reinterpret_cast<f32x4*>(&sW1[(f >> 5) * LD1 + 4 * (f & 31)]) = wn[q];
    }
    if (tid < 2 * HJ) sGB[cur ^ 1][tid] = gbn;

    // ---- + bias; accumulator layout (hidden unit = lane) -> row layout (note = lane) through the wave's own LDS tile ----
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rl = (r & 3) + 8 * (r >> 2) + 4 * hh;
      sZw[rl * LDY + lr] = a0[r] + bb0;
      sZw[rl * LDY + 32 + lr] = a1[r] + bb1;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");          // same wave: LDS operations complete in order
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ---- ReLU, LayerNorm: a lane holds 32 of its note's 64 hidden units, the other half of the wave the other 32 ----
    float4 ya[8];
    {
      const float* zp = &sZw[lr * LDY + 32 * hh];
      float* zo = g.z + my_row * g.ld_h + t * HJ + 32 * hh;
      float sum = 0.f;
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const float4 zv = *reinterpret_cast<const float4*>(zp + 4 * s);
        if (row_live) *reinterpret_cast<float4*>(zo + 4 * s) = zv;                 // the pre-activation: the backward pass's ReLU mask
        ya[s] = make_float4(agnn::relu_nan(zv.x), agnn::relu_nan(zv.y), agnn::relu_nan(zv.z), agnn::relu_nan(zv.w));
        sum += (ya[s].x + ya[s].y) + (ya[s].z + ya[s].w);
        if (s & 1) __builtin_amdgcn_sched_barrier(0);
      }
      {
        const auto p = __builtin_amdgcn_permlane32_swap(__float_as_uint(sum), __float_as_uint(sum), false, false);
        sum = __uint_as_float(p[0]) + __uint_as_float(p[1]);
      }
      const float m = sum * (1.f / HJ);
      float q = 0.f;
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        ya[s].x -= m; ya[s].y -= m; ya[s].z -= m; ya[s].w -= m;
        q += (ya[s].x * ya[s].x + ya[s].y * ya[s].y) + (ya[s].z * ya[s].z + ya[s].w * ya[s].w);
      }
      {
        const auto p = __builtin_amdgcn_permlane32_swap(__float_as_uint(q), __float_as_uint(q), false, false);
        q = __uint_as_float(p[0]) + __uint_as_float(p[1]);
      }
      const float rs = __builtin_amdgcn_rsqf(q * (1.f / HJ) + g.eps);               // v_rsq_f32: 1 ulp
      float* yo = g.y + my_row * g.ld_h + t * HJ + 32 * hh;
      const float* gp = &sGB[cur][32 * hh];
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const float4 ga = *reinterpret_cast<const float4*>(gp + 4 * s);
        const float4 be = *reinterpret_cast<const float4*>(gp + HJ + 4 * s);
        ya[s] = make_float4(ya[s].x * rs * ga.x + be.x, ya[s].y * rs * ga.y + be.y, ya[s].z * rs * ga.z + be.z, ya[s].w * rs * ga.w + be.w);
        if (row_live) *reinterpret_cast<float4*>(yo + 4 * s) = ya[s];
        if (s & 1) __builtin_amdgcn_sched_barrier(0);  // gamma / beta fragments two steps ahead at most (registers)
      }
      if (row_live && hh == 0) {
        g.mean[my_row * g.T + t] = m;
        g.rstd[my_row * g.T + t] = rs;
      }
    }

    // ---- second projection: 32 notes x C_t classes, K = 64, 32 classes at a time; y stays in registers as the A operand ----
    for (int tile = 0; tile < tiles; ++tile) {
      float4 bn[8];
      const bool nxt = tile + 1 < tiles;
      if (nxt) {                                       // the next class tile's B-fragments: requested before this tile's MFMAs
        const int64_t cr = min(c_lo + 32 * (tile + 1) + lr, g.sum_c - 1);
        const float* wp = g.w2 + cr * HJ + 32 * hh;
#pragma unroll
        for (int s = 0; s < 8; ++s) bn[s] = *reinterpret_cast<const float4*>(wp + 4 * s);
      }
      const int cc = 32 * tile + lr;
      const float bias = g.b2 != nullptr ? g.b2[min(c_lo + cc, g.sum_c - 1)] : 0.f;
      __builtin_amdgcn_sched_barrier(0);
      f32x16 c0 = {0}, c1 = {0};                       // two chains: a dependent MFMA does not wait for the one before it
#pragma unroll
      for (int s = 0; s < 8; s += 2) {
        c0 = MFMA(ya[s].x, bw[s].x, c0);
        c1 = MFMA(ya[s + 1].x, bw[s + 1].x, c1);
        c0 = MFMA(ya[s].y, bw[s].y, c0);
        c1 = MFMA(ya[s + 1].y, bw[s + 1].y, c1);
        c0 = MFMA(ya[s].z, bw[s].z, c0);
        c1 = MFMA(ya[s + 1].z, bw[s + 1].z, c1);
        c0 = MFMA(ya[s].w, bw[s].w, c0);
        c1 = MFMA(ya[s + 1].w, bw[s + 1].w, c1);
      }
      __builtin_amdgcn_s_waitcnt(0x0F70);              // vmcnt(0): the requests above, BEFORE the stores below are issued
      if (cc < C) {
        // scalar pointer (base + class column + row-in-tile * ld) + one 32-bit lane-varying byte offset (4 N ld_o < 2^32: host check)
        const uint32_t vo = 4u * (static_cast<uint32_t>(row0 + 4 * hh) * static_cast<uint32_t>(g.ld_o) + lr);
        char* ot = reinterpret_cast<char*>(g.logits + c_lo + 32 * tile);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rq = (r & 3) + 8 * (r >> 2);
          if (!CHECK || row0 + rq + 4 * hh < g.N) *reinterpret_cast<float*>(ot + static_cast<int64_t>(rq) * g.ld_o * 4 + vo) = (c0[r] + c1[r]) + bias;
        }
      }
      if (nxt) {
#pragma unroll
        for (int s = 0; s < 8; ++s) bw[s] = bn[s];
      }
    }
    __syncthreads();                                   // the next task's W1 / gamma / beta are in LDS for every wave
  }
}

}  // namespace

extern "C" int agnn_heads_fwd_f32(const float* x, int64_t ld_x, int64_t N, int32_t in_f, int32_t hidden, int32_t T, const float* w1,
                                  const float* b1, const float* gamma, const float* beta, float eps, const float* w2, const float* b2,
                                  const int32_t* offs_dev, const int32_t* offs_host, float* z, float* y, int64_t ld_h, float* mean,
                                  float* rstd, float* logits, int64_t ld_o, agnn_stream_t stream_) {
  using namespace agnn;
  if (in_f != HK || hidden != HJ) return fail(AGNN_EINVAL, "heads_fwd: built for in=%d hidden=%d, got in=%d hidden=%d", HK, HJ, in_f, hidden);
  if (N < 0 || N >= (int64_t{1} << 31) || T <= 0 || T > MAXT) return fail(AGNN_EINVAL, "heads_fwd: bad N=%lld T=%d (T <= %d)", (long long)N, T, MAXT);
  if (N == 0) return AGNN_OK;
  if (!x || !w1 || !b1 || !gamma || !beta || !w2 || !offs_dev || !offs_host || !z || !y || !mean || !rstd || !logits)
    return fail(AGNN_EINVAL, "heads_fwd: null argument");
  if (!aligned16(x) || !aligned16(w1) || !aligned16(w2) || !aligned16(z) || !aligned16(y) || (ld_x & 3) || ld_x < HK || (ld_h & 3) ||
      ld_h < static_cast<int64_t>(T) * HJ)
    return fail(AGNN_EALIGN, "heads_fwd: x, w1, w2, z, y must be 16-byte aligned, ld_x / ld_h multiples of 4, ld_x >= %d, ld_h >= T*%d", HK, HJ);
  if (offs_host[0] != 0) return fail(AGNN_EINVAL, "heads_fwd: offs[0] must be 0");
  for (int t = 0; t < T; ++t)
    if (offs_host[t + 1] <= offs_host[t]) return fail(AGNN_EINVAL, "heads_fwd: offs must increase (task %d has no classes)", t);
  const int sum_c = offs_host[T];
  if (ld_o < sum_c) return fail(AGNN_EINVAL, "heads_fwd: ld_o=%lld < sum of classes %d", (long long)ld_o, sum_c);
  if ((N + RB) * ld_h >= (int64_t{1} << 30) || (N + RB) * ld_o >= (int64_t{1} << 30))
    return fail(AGNN_EINVAL, "heads_fwd: N * ld_h and N * ld_o must stay below 2^30 elements (32-bit byte offsets)");

  HeadsFwd g{};
  g.x = x; g.ld_x = ld_x; g.w1 = w1; g.b1 = b1; g.gamma = gamma; g.beta = beta; g.w2 = w2; g.b2 = b2; g.offs = offs_dev;
  g.z = z; g.y = y; g.ld_h = ld_h; g.mean = mean; g.rstd = rstd; g.logits = logits; g.ld_o = ld_o;
  g.N = static_cast<int32_t>(N); g.T = T; g.sum_c = sum_c; g.eps = eps;
  g.row_blocks = static_cast<int32_t>((N + RB - 1) / RB);
  // ~2 workgroups per CU: S shares of the task list per row block, dealt longest-first by MFMA count (128 + 32 per class tile)
  int S = (512 + g.row_blocks - 1) / g.row_blocks;
  S = S < 1 ? 1 : (S > MAXS ? MAXS : S);
  if (S > T) S = T;
  g.S = S;
  int cost[MAXT], idx[MAXT];
  for (int t = 0; t < T; ++t) {
    cost[t] = 128 + 32 * ((offs_host[t + 1] - offs_host[t] + 31) / 32);
    idx[t] = t;
  }
  for (int i = 1; i < T; ++i) {                          // insertion sort, descending cost (stable: ties keep task order)
    const int v = idx[i];
    int j = i - 1;
    while (j >= 0 && cost[idx[j]] < cost[v]) { idx[j + 1] = idx[j]; --j; }
    idx[j + 1] = v;
  }
  int load[MAXS] = {0}, count[MAXS] = {0};
  int32_t bins[MAXS][MAXT];
  for (int i = 0; i < T; ++i) {
    int best = 0;
    for (int s = 1; s < S; ++s)
      if (load[s] < load[best]) best = s;
    bins[best][count[best]++] = idx[i];
    load[best] += cost[idx[i]];
  }
  int n = 0;
  for (int s = 0; s < S; ++s) {
    g.begin[s] = n;
    for (int i = 0; i < count[s]; ++i) g.order[n++] = bins[s][i];
  }
  g.begin[S] = n;
  for (int t = 0; t <= T; ++t) g.offs_k[t] = offs_host[t];
  const int full = static_cast<int>(N / RB);             // row blocks without a partial tail: no per-note store masks
  if (full > 0) {
    const int64_t groups = (full + 7) / 8;
    hipLaunchKernelGGL(k_heads_fwd<false>, dim3(static_cast<unsigned>(groups * 8 * S)), dim3(256), 0, static_cast<hipStream_t>(stream_), g, 0);
  }
  if (full < g.row_blocks)
    hipLaunchKernelGGL(k_heads_fwd<true>, dim3(static_cast<unsigned>(8 * S)), dim3(256), 0, static_cast<hipStream_t>(stream_), g, full);
  return check_launch("heads_fwd");
}
