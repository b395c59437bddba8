// Persistent bidirectional GRU layer for the hybrid (sequence) branch — gfx950.
//
// The reference runs `nn.GRU(H, H/2, num_layers=2, bidirectional=True)` over each subgraph's
// padded target-note sequence (analysisgnn/models/cadence.py:249-285, models/analysis.py:527-537).
// The library RNN path launches several tiny kernels per time step (T = 500 steps x 2 layers):
// ~24 000 launches per training step, >90 % of the step.  Here one 512-thread workgroup owns one
// (sequence, direction) chain for ALL T steps: W_hh (3*HH x HH fp32 = 192 KiB at HH = 128) lives in
// registers (96 per thread), h_{t-1} in LDS.  The input projection x W_ih^T (+b_ih) is a plain library
// GEMM done beforehand for all steps at once; the backward kernel walks the chain in reverse with
// W_hh^T in registers and emits dGI / dHN, weight gradients are GEMMs over those (gru.py).
// A time step is two phases with an LDS-only barrier after each (see the kernels).
// fp32 throughout (v_exp/v_rcp gate functions, ~3 ulp), no atomics, bitwise reproducible run to run.
#include "agnn_common.h"

// The recurrence kernels are latency bound on VALU issue, and a wave that streams MFMAs starves the VALU instructions of the other
// waves on its SIMD (scripts/mfma_valu_overlap_probe.py: 3.5 x slower) — in the training step the weight-gradient and library
// GEMMs of the other branch run beside the recurrence and land on the same CUs.  Declaring v255 used makes every wave allocate 256
// VGPRs: the workgroup's 8 waves (2 per SIMD) then own the whole register file of their CU and no other kernel's waves fit beside
// them; the GEMMs take the other 192 CUs.  C2 step 3.215 -> 3.185 ms (profiles/r03_heads.md, two alternating pairs on one box).
#define CLAIM_SIMD_REGISTERS() asm volatile("" ::: "v255")

namespace {

// hidden size per direction: template parameter HH of the kernels, 128 (H = 256 models: the C2 / C3 configurations) or 64
// (H = 128).  256 (H = 512) does not fit: W_hh = 768 x 256 fp32 = 768 KB against 512 KB of registers + 160 KB of LDS per CU.
constexpr int NT = 512;        // threads per chain: 8 waves = 8 chunks of the reduction dimension
constexpr int KC = 8;
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Workgroup barrier that orders LDS traffic only.  `__syncthreads()` also waits for every outstanding
// global load/store (s_waitcnt vmcnt(0)), which would put one HBM store round trip and the prefetch of
// the next step's inputs on the critical path of EVERY time step of the recurrence.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Gate non-linearities on the hardware transcendental units: v_exp_f32 and v_rcp_f32 are each good to
// ~1 ulp, so sigmoid is ~3 ulp (3e-7 relative) — far inside the 1e-4 parity budget — at 5 instructions
// instead of ~40 for expf + IEEE division.  This block is on the serial critical path of every step.
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return 2.f * __builtin_amdgcn_rcpf(1.f + __expf(-2.f * x)) - 1.f; }

// gi   [B, T, 2, 3*HH]  x W_ih^T + b_ih, gate order r,z,n (torch)
// w_hh [2, 3*HH, HH], b_hh [2, 3*HH]
// y    [B, T, 2*HH]
// saved[B, T, 2, 4, HH]  = r, z, n, (W_hn h + b_hn)
// ---- forward ---------------------------------------------------------------------------------------------------------
// A one-phase version (every lane: mat-vec slice, 3-step DPP butterfly, gates, stores) was bound by instruction issue:
// ~250 instructions per wave per step, of which 48 packed FMAs are the mat-vec — the gate math, address arithmetic, loads
// and stores were repeated by all 8 lanes sharing a unit pair (0.78 us per step).  Now (0.61 us per step):
//   phase 1 (all 8 waves; wave = one 16-wide k chunk of W_hh, lane = six of its 384 rows): the wave's h chunk from LDS
//            (one address for all lanes), 48 packed FMAs, six partial sums to LDS (part[kc][gate][unit], no cross-lane
//            reduction at all);
//   phase 2 (waves 0 and 1, one lane per hidden unit): add the 8 partials per gate in a fixed order, gates, h_t to LDS.
//            Round 3: the unit lanes issue no global load — the step's operands arrive in an LDS ring filled by a LOADER
//            wave pair (see "roles" in the kernel); in the backward kernel the results also leave through LDS and a STORER pair.
// Every wave's loop is straight-line code with unconditional loads / stores and counted s_waitcnt (a branch around a
// memory operation makes the compiler wait for ALL outstanding operations at the join, which put one memory round trip into
// every step).
// Tried, slower: two EXTRA waves for phase 2 (640 threads) with their stores deferred behind the second barrier.

// Inter-layer dropout rides along (nn.GRU(dropout=p), models/cadence.py:249-251): `drop` [B, T, 2*HH] holds 0 or 1 / (1 - p)
// per element and `y2` receives y * drop — what the next layer reads — while `y` stays the layer's own state; without
// dropout the host passes drop = any readable matrix of that size, use = 0 and y2 = y (the loads / stores stay
// unconditional: a branch around a memory operation costs a full wait in this loop).  One launch less on the recurrence
// chain in each direction (the backward kernel applies the same factor to dy).
template <int HH>
__global__ __launch_bounds__(NT) void k_gru_fwd(const float* __restrict__ gi, const float* __restrict__ w_hh,
                                                 const float* __restrict__ b_hh, int T, float* __restrict__ y,
                                                 float* __restrict__ saved, const float* __restrict__ drop, float use,
                                                 float* __restrict__ y2) {
  constexpr int PSTR = 3 * HH;           // part[ks][gate*HH + unit]
  // the 8 waves split the mat-vec KS ways along k and RS ways along the rows.  HH = 128: 4 x 2 — four partial sums per row
  // instead of eight: phase 2 (the two waves that carry the serial chain) reads 12 LDS words per unit instead of 24, phase 1
  // writes 3 per lane instead of 6 (8 x 1 before: 301 -> 2xx us per launch); HH = 64: 8 x 1 (3 rows per lane either way)
  constexpr int KS = HH == 128 ? 4 : 8, RS = KC / KS;
  constexpr int RPL = 3 * HH / (64 * RS);  // rows of W_hh per lane (3)
  constexpr int CPW = HH / KS;             // columns of W_hh per wave (32 / 8)
  __shared__ __attribute__((aligned(16))) float hbuf[2][HH];
  __shared__ __attribute__((aligned(16))) float part[KS * PSTR];
  __shared__ float s_in[4][4][HH];        // ring of 4 steps: gi_r, gi_z, gi_n, dropout factor — written by the helper lanes
  CLAIM_SIMD_REGISTERS();
  const int b = blockIdx.x >> 1, d = blockIdx.x & 1;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kc = wv % KS;                                        // this WAVE's k chunk: its h values are wave-uniform
  const int row0 = (wv / KS) * (3 * HH / RS) + lane;             // ... and its block of rows
  const float* W = w_hh + static_cast<size_t>(d) * 3 * HH * HH;
  // lane -> the RPL rows {row0 + 64 i} of W_hh (row = gate*HH + unit), columns [CPW kc, CPW kc + CPW)
  f32x2 w[RPL][CPW / 2];
#pragma unroll
  for (int i = 0; i < RPL; ++i) {
    const float4* src = reinterpret_cast<const float4*>(W + static_cast<size_t>(row0 + 64 * i) * HH + CPW * kc);
#pragma unroll
    for (int v = 0; v < CPW / 4; ++v) {
      const float4 t4 = src[v];
      w[i][2 * v] = f32x2{t4.x, t4.y};
      w[i][2 * v + 1] = f32x2{t4.z, t4.w};
    }
  }
  if (tid < HH) { hbuf[0][tid] = 0.f; hbuf[1][tid] = 0.f; }
  // (the one __syncthreads before the time loop sits in each role's branch, behind the helper lanes' first LDS writes)

  auto phase1 = [&](int cur) {
    f32x2 hk[CPW / 2];
    const float4* hp = reinterpret_cast<const float4*>(&hbuf[cur][CPW * kc]);     // same address in every lane: LDS broadcast
#pragma unroll
    for (int v = 0; v < CPW / 4; ++v) {
      const float4 t4 = hp[v];
      hk[2 * v] = f32x2{t4.x, t4.y};
      hk[2 * v + 1] = f32x2{t4.z, t4.w};
    }
#pragma unroll
    for (int i = 0; i < RPL; ++i) {
      f32x2 a = {0.f, 0.f};
#pragma unroll
      for (int k = 0; k < CPW / 2; ++k) a = __builtin_elementwise_fma(w[i][k], hk[k], a);
      part[kc * PSTR + row0 + 64 * i] = a.x + a.y;             // consecutive lanes, consecutive words: conflict-free
    }
  };

  // ---- roles.  The unit lanes (tid < HH: waves 0 and 1) issue NO global load: the step's operands are brought by loader lanes
  // (tid 128 .. 128 + HH = waves 2 and 3, SIMDs that idle during phase 2) in the window where those would wait at the barrier:
  // memory -> loader registers (four sets, requested 6 steps ahead) -> LDS ring `s_in` (2 steps ahead) -> phase 2.  With the
  // loads, stores and their address arithmetic inside phase 2 the step took 1 474 cycles, without them 1 310 (ablation,
  // profiles/r03_gru.md) — and beside another kernel's memory traffic the unit lanes additionally waited for their stores to be
  // acknowledged before a prefetched operand could be used (loads and stores share one in-order counter).  Without loads of
  // their own the unit lanes' stores are never waited for, so the five outputs leave from phase 2 directly (through LDS and a
  // storer wave pair, as in the backward kernel, the hand-over cost as much as the stores: 301 vs 298 us per launch).
  const bool loader = tid >= 128 && tid < 128 + HH;         // waves 2, 3: operands in
  const int hu = tid - 128;
  const size_t row3 = static_cast<size_t>(2) * 3 * HH;
  auto load_in = [&](int s_, float (&g3)[4]) {
    const int sc = s_ < T ? s_ : T - 1;
    const int t_ = d ? T - 1 - sc : sc;
    const float* p = gi + (static_cast<size_t>(b) * T + t_) * row3 + static_cast<size_t>(d) * 3 * HH + hu;
    g3[0] = p[0];
    g3[1] = p[HH];
    g3[2] = p[2 * HH];
    g3[3] = drop[(static_cast<size_t>(b) * T + t_) * 2 * HH + d * HH + hu];       // raw: arithmetic on a loaded value waits until put_in
  };
  auto put_in = [&](int s_, const float (&g3)[4]) {
    float* q = &s_in[s_ & 3][0][hu];
    q[0] = g3[0];
    q[HH] = g3[1];
    q[2 * HH] = g3[2];
    q[3 * HH] = use != 0.f ? g3[3] : 1.f;          // the factor itself.  A SELECT: without dropout `drop` aliases the output buffer
                                                    // (uninitialised ahead of the walk), and 0 * NaN is NaN
  };
  // window s = while the unit lanes run phase 2 of step s
  if (loader) {                          // step s + 2's operands -> LDS, step s + 6's requested: four register sets in flight,
    float g0[4], g1[4], g2[4], g3[4], g4[4], g5[4], g6[4], g7[4];    // eight register sets: ~4.4 us between request and use
    load_in(0, g0);
    load_in(1, g1);
    put_in(0, g0);
    put_in(1, g1);
    load_in(2, g2); load_in(3, g3); load_in(4, g4); load_in(5, g5); load_in(6, g6); load_in(7, g7);
    load_in(8, g0); load_in(9, g1);
    __syncthreads();
    int s = 0;
#define GRU_FWD_WINDOW(I, G) phase1((I) & 1); lds_barrier(); put_in(s + 2 + (I), G); load_in(s + 10 + (I), G); lds_barrier();
    for (; s + 8 <= T; s += 8) {
      GRU_FWD_WINDOW(0, g2) GRU_FWD_WINDOW(1, g3) GRU_FWD_WINDOW(2, g4) GRU_FWD_WINDOW(3, g5)
      GRU_FWD_WINDOW(4, g6) GRU_FWD_WINDOW(5, g7) GRU_FWD_WINDOW(6, g0) GRU_FWD_WINDOW(7, g1)
    }
#undef GRU_FWD_WINDOW
#define GRU_FWD_LAST(I, G) if (s + (I) < T) { phase1((I) & 1); lds_barrier(); put_in(s + 2 + (I), G); lds_barrier(); }
    GRU_FWD_LAST(0, g2) GRU_FWD_LAST(1, g3) GRU_FWD_LAST(2, g4) GRU_FWD_LAST(3, g5) GRU_FWD_LAST(4, g6) GRU_FWD_LAST(5, g7) GRU_FWD_LAST(6, g0)
#undef GRU_FWD_LAST
    return;
  }
  if (tid >= HH) {                       // the other waves: mat-vec only
    __syncthreads();
    int s = 0;
    for (; s + 2 <= T; s += 2) {
      phase1(0); lds_barrier(); lds_barrier();
      phase1(1); lds_barrier(); lds_barrier();
    }
    if (s < T) { phase1(0); lds_barrier(); lds_barrier(); }
    return;
  }

  // waves 0, 1: lane = hidden unit
  const int u = tid;
  const float bh0 = b_hh[d * 3 * HH + u], bh1 = b_hh[d * 3 * HH + HH + u], bh2 = b_hh[d * 3 * HH + 2 * HH + u];
  float hprev = 0.f;
  // running output pointers: time step 0 of this direction, then one row forward (reverse direction: backward) per step
  const int64_t t0 = d ? T - 1 : 0;
  const int64_t y_step = (d ? -1 : 1) * static_cast<int64_t>(2 * HH), sv_step = (d ? -1 : 1) * static_cast<int64_t>(2 * 4 * HH);
  float* y_p = y + (static_cast<size_t>(b) * T + t0) * 2 * HH + d * HH + u;
  float* y2_p = y2 + (static_cast<size_t>(b) * T + t0) * 2 * HH + d * HH + u;
  float* sv_p = saved + ((static_cast<size_t>(b) * T + t0) * 2 + d) * 4 * HH + u;
  auto phase2 = [&](int cur, int slot) {         // cur = s & 1, slot = s & 3: constants in the unrolled loop below
    float p0[KS], p1[KS], p2[KS];
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      p0[k] = part[k * PSTR + u];
      p1[k] = part[k * PSTR + HH + u];
      p2[k] = part[k * PSTR + 2 * HH + u];
    }
    const float* in = &s_in[slot][0][u];
    const float gi0 = in[0], gi1 = in[HH], gi2 = in[2 * HH], fac = in[3 * HH];
    // fixed-shape tree (same order every run: bitwise reproducible)
    float s0, s1, s2;
    if (KS == 4) {
      s0 = ((p0[0] + p0[1]) + (p0[2] + p0[3])) + bh0;
      s1 = ((p1[0] + p1[1]) + (p1[2] + p1[3])) + bh1;
      s2 = ((p2[0] + p2[1]) + (p2[2] + p2[3])) + bh2;
    } else {
      s0 = (((p0[0] + p0[1]) + (p0[2] + p0[3])) + ((p0[4 % KS] + p0[5 % KS]) + (p0[6 % KS] + p0[7 % KS]))) + bh0;
      s1 = (((p1[0] + p1[1]) + (p1[2] + p1[3])) + ((p1[4 % KS] + p1[5 % KS]) + (p1[6 % KS] + p1[7 % KS]))) + bh1;
      s2 = (((p2[0] + p2[1]) + (p2[2] + p2[3])) + ((p2[4 % KS] + p2[5 % KS]) + (p2[6 % KS] + p2[7 % KS]))) + bh2;
    }
    const float rr = sigmoidf_(gi0 + s0);
    const float zz = sigmoidf_(gi1 + s1);
    const float nn = tanhf_(gi2 + rr * s2);
    const float hnew = (1.f - zz) * nn + zz * hprev;
    hprev = hnew;
    hbuf[cur ^ 1][u] = hnew;
    // the unit lanes issue no load at all, so their stores are never waited for (vmcnt only matters to a load's consumer): the
    // five per-step outputs leave from here (through LDS and a storer wave pair the hand-over cost as much as these stores)
    y_p[0] = hnew;
    y2_p[0] = hnew * fac;
    sv_p[0] = rr;
    sv_p[HH] = zz;
    sv_p[2 * HH] = nn;
    sv_p[3 * HH] = s2;
    y_p += y_step;
    y2_p += y_step;
    sv_p += sv_step;
  };
  __syncthreads();
  int s = 0;
  for (; s + 4 <= T; s += 4) {
    phase1(0); lds_barrier(); phase2(0, 0); lds_barrier();
    phase1(1); lds_barrier(); phase2(1, 1); lds_barrier();
    phase1(0); lds_barrier(); phase2(0, 2); lds_barrier();
    phase1(1); lds_barrier(); phase2(1, 3); lds_barrier();
  }
  if (s < T)     { phase1(0); lds_barrier(); phase2(0, 0); lds_barrier(); }
  if (s + 1 < T) { phase1(1); lds_barrier(); phase2(1, 1); lds_barrier(); }
  if (s + 2 < T) { phase1(0); lds_barrier(); phase2(0, 2); lds_barrier(); }
}

// dy [B,T,2*HH]; y, saved from the forward; outputs dgi [B,T,2,3*HH] (d r_pre, d z_pre, d n_pre) and
// dgh [B,T,2,3*HH] (gradient w.r.t. W_h h + b_h: the r and z blocks of dgi again, and d n_pre * r in the n block —
// written by the kernel so that the weight-gradient GEMM for W_hh needs no 49 MB concatenation).
// ---- two-phase backward (same split as k_gru_fwd) ------------------------------------------------------------------
//   phase A (waves 0 and 1, one lane per hidden unit): carry = dh_{t+1} z_{t+1} + the 8 partial sums of W_hh^T dgh left
//            by phase B of the previous step; gate gradients; dgi / dgh to HBM; the three dgh vectors to LDS;
//   phase B (all 8 waves; wave = 48 rows of W_hh, lane = two hidden units): the wave's 48 dgh values from LDS (one
//            address for all lanes), 48 packed FMAs, two partial sums to LDS.
template <int HH, bool HP>
__global__ __launch_bounds__(NT) void k_gru_bwd(const float* __restrict__ dy, const float* __restrict__ y,
                                                 const float* __restrict__ saved, const float* __restrict__ w_hh,
                                                 int T, float* __restrict__ dgi, float* __restrict__ dgh_out,
                                                 const float* __restrict__ drop, float use, float* __restrict__ hp_out) {
  __shared__ __attribute__((aligned(16))) float s_out[2][4][HH];      // per step parity: d r_pre | d z_pre | d n_pre * r | d n_pre
  __shared__ __attribute__((aligned(16))) float cpart[KC * HH];
  __shared__ float s_in[4][6][HH];                                     // ring of 4 steps: dy * factor, r, z, n, W_hn h + b_hn, h_{t-1}
  constexpr int CSTR = HH;
  CLAIM_SIMD_REGISTERS();
  const int b = blockIdx.x >> 1, d = blockIdx.x & 1;
  const int tid = threadIdx.x;
  const int kc = __builtin_amdgcn_readfirstlane(tid >> 6);       // this WAVE's chunk of 48 rows j of W_hh: dgh[j] is wave-uniform
  const int u_raw = 2 * (tid & 63);                              // lane -> hidden units u0, u0 + 1
  const bool u_on = u_raw < HH;                                  // HH = 64: lanes 32..63 carry no units (they repeat a pair, store nothing)
  const int u0 = u_on ? u_raw : HH - 2;
  const float* W = w_hh + static_cast<size_t>(d) * 3 * HH * HH;
  constexpr int JC = 3 * HH / KC;   // 48 rows of W per wave
  f32x2 wa[JC / 2], wb[JC / 2];     // unit u0 / u0+1: (W[j][u], W[j+1][u]) pairs over the chunk's rows
#pragma unroll
  for (int j = 0; j < JC; j += 2) {
    const float2 v0 = *reinterpret_cast<const float2*>(W + static_cast<size_t>(JC * kc + j) * HH + u0);
    const float2 v1 = *reinterpret_cast<const float2*>(W + static_cast<size_t>(JC * kc + j + 1) * HH + u0);
    wa[j / 2] = f32x2{v0.x, v1.x};
    wb[j / 2] = f32x2{v0.y, v1.y};
  }
  for (int i = tid; i < KC * CSTR; i += NT) cpart[i] = 0.f;

  auto phaseB = [&](int cur) {
    f32x2 a0 = {0.f, 0.f}, a1 = {0.f, 0.f};
    const float4* gp = reinterpret_cast<const float4*>(&s_out[cur][0][0] + JC * kc);      // the first 3 HH floats: the dgh vector
#pragma unroll
    for (int v = 0; v < JC / 4; ++v) {
      const float4 t4 = gp[v];
      const f32x2 lo = {t4.x, t4.y}, hi = {t4.z, t4.w};
      a0 = __builtin_elementwise_fma(wa[2 * v], lo, a0);
      a1 = __builtin_elementwise_fma(wb[2 * v], lo, a1);
      a0 = __builtin_elementwise_fma(wa[2 * v + 1], hi, a0);
      a1 = __builtin_elementwise_fma(wb[2 * v + 1], hi, a1);
    }
    if (u_on) *reinterpret_cast<float2*>(&cpart[kc * CSTR + u0]) = make_float2(a0.x + a0.y, a1.x + a1.y);
  };

  // ---- roles: as in k_gru_fwd.  Unit lanes (tid < HH) touch LDS only; helper lanes (tid 128 .. 128 + HH) move the operands
  // memory -> registers (requested 4 steps ahead) -> `s_in` (2 steps ahead) and the gate gradients `s_out` -> memory one step
  // later, in the window where they would wait for phase A.  h_{t-1} (the W_hh weight gradient's right operand) is copied to
  // `hp_out` by the helper on the way in.  Step time 1 430 cycles with the 6 loads + 7 stores in phase A, 1 160 without.
  const bool loader = tid >= 128 && tid < 128 + HH;         // waves 2, 3: operands in (and h_{t-1} out)
  const bool storer = tid >= 384 && tid < 384 + HH;         // waves 6, 7: gate gradients out
  const int hu = tid - (loader ? 128 : 384);
  struct StepIn { float dyv, dr, r, z, n, q, hp; };      // raw loaded values: arithmetic on them waits until put_in
  auto load_in = [&](int s_, StepIn& o) {
    const int sc = s_ < T ? s_ : T - 1;
    const int t_ = d ? sc : T - 1 - sc;          // reverse of the forward walk
    const int tp_ = d ? t_ + 1 : t_ - 1;
    const size_t bt_ = static_cast<size_t>(b) * T + t_;
    o.dyv = dy[bt_ * 2 * HH + d * HH + hu];
    o.dr = drop[bt_ * 2 * HH + d * HH + hu];
    const float* sv = saved + (bt_ * 2 + d) * 4 * HH + hu;
    o.r = sv[0];
    o.z = sv[HH];
    o.n = sv[2 * HH];
    o.q = sv[3 * HH];
    const bool has_prev = tp_ >= 0 && tp_ < T;
    const int tpc = has_prev ? tp_ : t_;
    o.hp = y[(static_cast<size_t>(b) * T + tpc) * 2 * HH + d * HH + hu];
  };
  auto put_in = [&](int s_, const StepIn& o) {
    float* q = &s_in[s_ & 3][0][hu];
    const int sp = s_ < T ? s_ : T - 1;
    const int tq = d ? sp + 1 : T - 2 - sp;      // the time step h_{t-1} came from: outside the sequence at its first step
    const float hp = (tq >= 0 && tq < T) ? o.hp : 0.f;
    q[0] = o.dyv * (use != 0.f ? o.dr : 1.f);    // dy is d(y * drop); a select, not use * (...): the stand-in operand is not a factor
    q[HH] = o.r;
    q[2 * HH] = o.z;
    q[3 * HH] = o.n;
    q[4 * HH] = o.q;
    q[5 * HH] = hp;
    if (HP) {                                    // compile-time: no branch around a memory operation in the loop; a step index
      const int sc = s_ < T ? s_ : T - 1;        // past the end repeats the last step's (identical) store
      const int t_ = d ? sc : T - 1 - sc;
      hp_out[((static_cast<size_t>(b) * T + t_) * 2 + d) * HH + hu] = hp;
    }
  };
  auto store_out = [&](int s_) {
    const int t = d ? s_ : T - 1 - s_;
    const size_t bt = static_cast<size_t>(b) * T + t;
    const float* o = &s_out[s_ & 1][0][hu];
    const float drp = o[0], dzp = o[HH], dq = o[2 * HH], dnp = o[3 * HH];
    float* go = dgi + (bt * 2 + d) * 3 * HH + hu;
    go[0] = drp;
    go[HH] = dzp;
    go[2 * HH] = dnp;
    float* ho = dgh_out + (bt * 2 + d) * 3 * HH + hu;
    ho[0] = drp;
    ho[HH] = dzp;
    ho[2 * HH] = dq;
  };

  // window s = while the unit lanes run phase A of step s
  if (loader) {                          // eight register sets in flight: requested 10 steps ahead, to LDS 2 steps ahead
    StepIn i0, i1, i2, i3, i4, i5, i6, i7;
    load_in(0, i0);
    load_in(1, i1);
    put_in(0, i0);
    put_in(1, i1);
    load_in(2, i2); load_in(3, i3); load_in(4, i4); load_in(5, i5); load_in(6, i6); load_in(7, i7);
    load_in(8, i0); load_in(9, i1);
    __syncthreads();
    int s = 0;
#define GRU_BWD_WINDOW(I, G) put_in(s + 2 + (I), G); load_in(s + 10 + (I), G); lds_barrier(); phaseB((I) & 1); lds_barrier();
    for (; s + 8 <= T; s += 8) {
      GRU_BWD_WINDOW(0, i2) GRU_BWD_WINDOW(1, i3) GRU_BWD_WINDOW(2, i4) GRU_BWD_WINDOW(3, i5)
      GRU_BWD_WINDOW(4, i6) GRU_BWD_WINDOW(5, i7) GRU_BWD_WINDOW(6, i0) GRU_BWD_WINDOW(7, i1)
    }
#undef GRU_BWD_WINDOW
#define GRU_BWD_LAST(I, G) if (s + (I) < T) { put_in(s + 2 + (I), G); lds_barrier(); phaseB((I) & 1); lds_barrier(); }
    GRU_BWD_LAST(0, i2) GRU_BWD_LAST(1, i3) GRU_BWD_LAST(2, i4) GRU_BWD_LAST(3, i5) GRU_BWD_LAST(4, i6) GRU_BWD_LAST(5, i7) GRU_BWD_LAST(6, i0)
#undef GRU_BWD_LAST
    return;
  }
  if (storer) {
    __syncthreads();
    int s = 0;
    for (; s + 2 <= T; s += 2) {
      store_out(s > 0 ? s - 1 : 0); lds_barrier(); phaseB(0); lds_barrier();
      store_out(s);                 lds_barrier(); phaseB(1); lds_barrier();
    }
    if (s < T) { store_out(s > 0 ? s - 1 : 0); lds_barrier(); phaseB(0); lds_barrier(); }
    store_out(T - 1);
    return;
  }
  if (tid >= HH) {                       // the other waves: mat-vec only
    __syncthreads();
    int s = 0;
    for (; s + 2 <= T; s += 2) {
      lds_barrier(); phaseB(0); lds_barrier();
      lds_barrier(); phaseB(1); lds_barrier();
    }
    if (s < T) { lds_barrier(); phaseB(0); lds_barrier(); }
    return;
  }

  const int u = tid;
  float dhz = 0.f;
  auto phaseA = [&](int cur, int slot) {         // cur = s & 1, slot = s & 3: constants in the unrolled loop below
    float cp[KC];
#pragma unroll
    for (int k = 0; k < KC; ++k) cp[k] = cpart[k * CSTR + u];
    const float* in = &s_in[slot][0][u];
    const float dyv = in[0], r = in[HH], z = in[2 * HH], n = in[3 * HH], q = in[4 * HH], hp = in[5 * HH];
    const float carry = (((cp[0] + cp[1]) + (cp[2] + cp[3])) + ((cp[4] + cp[5]) + (cp[6] + cp[7]))) + dhz;   // fixed-shape tree
    const float dh = dyv + carry;
    const float dn = dh * (1.f - z);
    const float dz = dh * (hp - n);
    const float dnp = dn * (1.f - n * n);
    const float dr = dnp * q;
    const float dq = dnp * r;
    const float dzp = dz * z * (1.f - z);
    const float drp = dr * r * (1.f - r);
    dhz = dh * z;
    float* o = &s_out[cur][0][u];
    o[0] = drp;
    o[HH] = dzp;
    o[2 * HH] = dq;
    o[3 * HH] = dnp;
  };
  __syncthreads();
  int s = 0;
  for (; s + 4 <= T; s += 4) {
    phaseA(0, 0); lds_barrier(); phaseB(0); lds_barrier();
    phaseA(1, 1); lds_barrier(); phaseB(1); lds_barrier();
    phaseA(0, 2); lds_barrier(); phaseB(0); lds_barrier();
    phaseA(1, 3); lds_barrier(); phaseB(1); lds_barrier();
  }
  if (s < T)     { phaseA(0, 0); lds_barrier(); phaseB(0); lds_barrier(); }
  if (s + 1 < T) { phaseA(1, 1); lds_barrier(); phaseB(1); lds_barrier(); }
  if (s + 2 < T) { phaseA(0, 2); lds_barrier(); phaseB(0); lds_barrier(); }
}

// h_{t-1} of both directions as one [B*T, 2*HH] matrix — the left operand of the W_hh weight-gradient GEMMs:
// forward half = y[b, t-1, :HH] (zero at t = 0), reverse half = y[b, t+1, HH:] (zero at t = T-1).  One pass, 16-byte lanes.
template <int HH>
__global__ __launch_bounds__(256) void k_gru_hprev(const float* __restrict__ y, int T, int64_t total4, float* __restrict__ hp) {
  constexpr int Q = 2 * HH / 4;          // float4 per row
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; e < total4; e += static_cast<int64_t>(gridDim.x) * 256) {
    const int64_t row = e / Q;
    const int q = static_cast<int>(e - row * Q);
    const int t = static_cast<int>(row % T);
    const bool rev = q >= Q / 2;
    const bool has = rev ? (t + 1 < T) : (t > 0);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (has) v = reinterpret_cast<const float4*>(y)[(rev ? row + 1 : row - 1) * Q + q];
    reinterpret_cast<float4*>(hp)[e] = v;
  }
}

}  // namespace

extern "C" int agnn_gru_hprev_f32(const float* y, int64_t B, int64_t T, int32_t hidden, float* hp, agnn_stream_t stream_) {
  using namespace agnn;
  if (hidden != 128 && hidden != 64) return fail(AGNN_EINVAL, "gru_hprev: hidden=%d unsupported (this build: 64, 128)", hidden);
  const int HH = hidden;
  if (B < 0 || T < 0 || T >= (int64_t{1} << 31)) return fail(AGNN_EINVAL, "gru_hprev: bad B=%lld T=%lld", (long long)B, (long long)T);
  if (B == 0 || T == 0) return AGNN_OK;
  if (!y || !hp) return fail(AGNN_EINVAL, "gru_hprev: null argument");
  if (!aligned16(y) || !aligned16(hp)) return fail(AGNN_EALIGN, "gru_hprev: pointers must be 16-byte aligned");
  const int64_t total4 = B * T * (2 * HH / 4);
  int64_t blocks = (total4 + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  if (hidden == 128) hipLaunchKernelGGL(k_gru_hprev<128>, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, static_cast<hipStream_t>(stream_), y, static_cast<int>(T), total4, hp);
  else hipLaunchKernelGGL(k_gru_hprev<64>, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, static_cast<hipStream_t>(stream_), y, static_cast<int>(T), total4, hp);
  return check_launch("gru_hprev");
}

extern "C" int agnn_gru_fwd_f32(const float* gi, const float* w_hh, const float* b_hh, int64_t B, int64_t T,
                                int32_t hidden, float* y, float* saved, const float* drop_scale, float* y_drop,
                                agnn_stream_t stream_) {
  using namespace agnn;
  if (hidden != 128 && hidden != 64) return fail(AGNN_EINVAL, "gru_fwd: hidden=%d unsupported (this build: 64, 128)", hidden);
  if (B < 0 || T < 0 || B * 2 >= (int64_t{1} << 31) || T >= (int64_t{1} << 31)) return fail(AGNN_EINVAL, "gru_fwd: bad B=%lld T=%lld", (long long)B, (long long)T);
  if (B == 0 || T == 0) return AGNN_OK;
  if (!gi || !w_hh || !b_hh || !y || !saved) return fail(AGNN_EINVAL, "gru_fwd: null argument");
  if (!aligned16(gi) || !aligned16(w_hh) || !aligned16(y) || !aligned16(saved)) return fail(AGNN_EALIGN, "gru_fwd: pointers must be 16-byte aligned");
  if ((drop_scale == nullptr) != (y_drop == nullptr)) return fail(AGNN_EINVAL, "gru_fwd: drop_scale and y_drop go together");
  if (hidden == 128)
    hipLaunchKernelGGL(k_gru_fwd<128>, dim3(static_cast<unsigned>(B * 2)), dim3(NT), 0, static_cast<hipStream_t>(stream_), gi, w_hh, b_hh,
                       static_cast<int>(T), y, saved, drop_scale ? drop_scale : y, drop_scale ? 1.f : 0.f, y_drop ? y_drop : y);
  else
    hipLaunchKernelGGL(k_gru_fwd<64>, dim3(static_cast<unsigned>(B * 2)), dim3(NT), 0, static_cast<hipStream_t>(stream_), gi, w_hh, b_hh,
                       static_cast<int>(T), y, saved, drop_scale ? drop_scale : y, drop_scale ? 1.f : 0.f, y_drop ? y_drop : y);
  return check_launch("gru_fwd");
}

extern "C" int agnn_gru_bwd_f32(const float* dy, const float* y, const float* saved, const float* w_hh, int64_t B,
                                int64_t T, int32_t hidden, float* dgi, float* dgh, const float* drop_scale, float* hprev,
                                agnn_stream_t stream_) {
  using namespace agnn;
  if (hidden != 128 && hidden != 64) return fail(AGNN_EINVAL, "gru_bwd: hidden=%d unsupported (this build: 64, 128)", hidden);
  if (B < 0 || T < 0 || B * 2 >= (int64_t{1} << 31) || T >= (int64_t{1} << 31)) return fail(AGNN_EINVAL, "gru_bwd: bad B=%lld T=%lld", (long long)B, (long long)T);
  if (B == 0 || T == 0) return AGNN_OK;
  if (!dy || !y || !saved || !w_hh || !dgi || !dgh) return fail(AGNN_EINVAL, "gru_bwd: null argument");
  if (!aligned16(dy) || !aligned16(y) || !aligned16(saved) || !aligned16(w_hh) || !aligned16(dgi) || !aligned16(dgh)) return fail(AGNN_EALIGN, "gru_bwd: pointers must be 16-byte aligned");
  const dim3 grid(static_cast<unsigned>(B * 2)), block(NT);
  const hipStream_t st = static_cast<hipStream_t>(stream_);
  const float* dr = drop_scale ? drop_scale : y;
  const float use = drop_scale ? 1.f : 0.f;
  const int Ti = static_cast<int>(T);
  if (hidden == 128 && hprev) hipLaunchKernelGGL((k_gru_bwd<128, true>), grid, block, 0, st, dy, y, saved, w_hh, Ti, dgi, dgh, dr, use, hprev);
  else if (hidden == 128) hipLaunchKernelGGL((k_gru_bwd<128, false>), grid, block, 0, st, dy, y, saved, w_hh, Ti, dgi, dgh, dr, use, hprev);
  else if (hprev) hipLaunchKernelGGL((k_gru_bwd<64, true>), grid, block, 0, st, dy, y, saved, w_hh, Ti, dgi, dgh, dr, use, hprev);
  else hipLaunchKernelGGL((k_gru_bwd<64, false>), grid, block, 0, st, dy, y, saved, w_hh, Ti, dgi, dgh, dr, use, hprev);
  return check_launch("gru_bwd");
}
