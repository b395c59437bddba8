// Fused  [ReLU ->] LayerNorm [-> ReLU] [-> dropout]  over the rows of a [N, H] fp32 matrix, forward and backward.
//
// The reference's wrapper and the encoder spec interleave these element-wise ops between every projection
// (analysisgnn/models/analysis.py:429-443, :474-485; models/cadence.py:252-259): as separate library launches they
// are ~60 launches and ~1.3 ms of the 7.7 ms step (profiles/r01_g), each reading and writing the full activation.
// Here one wavefront owns one row (a lane owns 4 floats of every 256-float chunk): the row is read once, mean /
// variance are butterfly reductions in registers, and the result is written once.  Nothing but the input, the
// row statistics and the (seed, step, call) triple is kept for backward: ReLU masks are recomputed from the saved
// input, the dropout mask from the counter-based generator (Philox-4x32-10 keyed by the seed, counter = element
// index / call id / training step read from a device buffer, so a captured hipGraph draws fresh masks per replay).
// gamma / beta gradients: each wave accumulates its rows in registers and writes one partial row; a second small
// kernel sums the partials in a fixed order (no atomics, reproducible).  HBM-bound.
#include <cmath>

#include "agnn_common.h"

namespace {

struct NaArgs {
  const float* x;
  int64_t ld_x;
  const float* gamma;
  const float* beta;
  int64_t n;
  int32_t H;
  float eps;
  float p;                 // dropout probability (0 = no dropout)
  uint32_t flags;          // AGNN_NA_*
  const int64_t* rng;      // device [2]: seed, step
  uint32_t call_id;
  int32_t seg;             // statistics over segments of `seg` floats of the row (seg == H: plain LayerNorm)
};

__device__ __forceinline__ uint32_t mulhi32(uint32_t a, uint32_t b) { return __umulhi(a, b); }

// Philox-4x32-10 (Salmon et al. 2011): 4 x 32 random bits for counter (c0..c3), key (k0,k1)
__device__ __forceinline__ uint4 philox(uint4 c, uint2 k) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t h0 = mulhi32(0xD2511F53u, c.x), l0 = 0xD2511F53u * c.x;
    const uint32_t h1 = mulhi32(0xCD9E8D57u, c.z), l1 = 0xCD9E8D57u * c.z;
    c = make_uint4(h1 ^ c.y ^ k.x, l1, h0 ^ c.w ^ k.y, l0);
    k.x += 0x9E3779B9u;
    k.y += 0xBB67AE85u;
  }
  return c;
}

template <class Args>
__device__ __forceinline__ float4 keep_mask(const Args& a, int64_t row, int chunk_lane, float scale) {
  const uint64_t seed = static_cast<uint64_t>(a.rng[0]), step = static_cast<uint64_t>(a.rng[1]);
  const uint64_t idx = static_cast<uint64_t>(row) * 256u + static_cast<uint32_t>(chunk_lane);     // one counter per float4
  const uint4 r = philox(make_uint4(static_cast<uint32_t>(idx), static_cast<uint32_t>(idx >> 32), a.call_id, static_cast<uint32_t>(step)),
                         make_uint2(static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32) ^ static_cast<uint32_t>(step >> 32)));
  const uint32_t thr = static_cast<uint32_t>(static_cast<double>(a.p) * 4294967296.0);            // drop when r < thr
  return make_float4(r.x >= thr ? scale : 0.f, r.y >= thr ? scale : 0.f, r.z >= thr ? scale : 0.f, r.w >= thr ? scale : 0.f);
}

__device__ __forceinline__ float wsum(float v) { return agnn::wave_sum_dpp(v); }
// sum over the gl (power of two <= 64) adjacent lanes that hold one segment (DPP inside a 16-lane row)
__device__ __forceinline__ float gsum(float v, int gl) {
  if (gl >= 2) v += agnn::dpp_mov<0xB1>(v);
  if (gl >= 4) v += agnn::dpp_mov<0x4E>(v);
  if (gl >= 8) v += agnn::dpp_mov<0x141>(v);
  if (gl >= 16) v += agnn::dpp_mov<0x140>(v);
  for (int o = 16; o < gl; o <<= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <int CH>
__global__ __launch_bounds__(256) void k_na_fwd(NaArgs a, float* __restrict__ y, int64_t ld_y, float* __restrict__ mean_out,
                                                float* __restrict__ rstd_out, int64_t* __restrict__ rng_used) {
  const int lane = threadIdx.x & 63;
  const int64_t row = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  // The (seed, step) this call draws its masks from is part of what it saves for backward: the live counter may have
  // moved on by then (a second training forward before this one's backward).
  if (rng_used != nullptr && a.rng != nullptr && blockIdx.x == 0 && threadIdx.x == 0) {
    rng_used[0] = a.rng[0];
    rng_used[1] = a.rng[1];
  }
  if (row >= a.n) return;
  const bool pre = a.flags & AGNN_NA_PRE_RELU, post = a.flags & AGNN_NA_POST_RELU, drop = a.p > 0.f;
  bool on[CH];
  float4 v[CH];
  const float4* xp = reinterpret_cast<const float4*>(a.x + row * a.ld_x);
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    on[c] = (c * 256 + lane * 4) < a.H;
    v[c] = on[c] ? xp[c * 64 + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
    if (pre) { v[c].x = agnn::relu_nan(v[c].x); v[c].y = agnn::relu_nan(v[c].y); v[c].z = agnn::relu_nan(v[c].z); v[c].w = agnn::relu_nan(v[c].w); }
    s += (v[c].x + v[c].y) + (v[c].z + v[c].w);
  }
  // statistics: one segment = the whole row (plain LayerNorm) or `seg` floats held by seg/4 adjacent lanes of one chunk
  const bool whole = a.seg >= a.H;
  const int gl = a.seg >> 2;
  const int nseg = whole ? 1 : a.H / a.seg;
  const float invS = 1.f / static_cast<float>(whole ? a.H : a.seg);
  float mean[CH], rstd[CH];
  if (whole) {
    const float m = wsum(s) * invS;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      if (!on[c]) continue;
      const float dx = v[c].x - m, dy = v[c].y - m, dz = v[c].z - m, dw = v[c].w - m;
      q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
    }
    const float r = 1.f / sqrtf(wsum(q) * invS + a.eps);
#pragma unroll
    for (int c = 0; c < CH; ++c) { mean[c] = m; rstd[c] = r; }
    if (lane == 0) { mean_out[row] = m; rstd_out[row] = r; }
  } else {
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const float sc = (v[c].x + v[c].y) + (v[c].z + v[c].w);
      const float m = gsum(sc, gl) * invS;
      const float dx = v[c].x - m, dy = v[c].y - m, dz = v[c].z - m, dw = v[c].w - m;
      const float q = on[c] ? (dx * dx + dy * dy) + (dz * dz + dw * dw) : 0.f;
      const float r = 1.f / sqrtf(gsum(q, gl) * invS + a.eps);
      mean[c] = m;
      rstd[c] = r;
      const int f = c * 256 + lane * 4;
      if (on[c] && (f % a.seg) == 0) { mean_out[row * nseg + f / a.seg] = m; rstd_out[row * nseg + f / a.seg] = r; }
    }
  }
  const float scale = drop ? 1.f / (1.f - a.p) : 1.f;
  float4* yp = reinterpret_cast<float4*>(y + row * ld_y);
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    if (!on[c]) continue;
    const float4 g = reinterpret_cast<const float4*>(a.gamma)[c * 64 + lane], b = reinterpret_cast<const float4*>(a.beta)[c * 64 + lane];
    float4 o = make_float4((v[c].x - mean[c]) * rstd[c] * g.x + b.x, (v[c].y - mean[c]) * rstd[c] * g.y + b.y,
                           (v[c].z - mean[c]) * rstd[c] * g.z + b.z, (v[c].w - mean[c]) * rstd[c] * g.w + b.w);
    if (post) { o.x = agnn::relu_nan(o.x); o.y = agnn::relu_nan(o.y); o.z = agnn::relu_nan(o.z); o.w = agnn::relu_nan(o.w); }
    if (drop) {
      const float4 m = keep_mask(a, row, c * 64 + lane, scale);
      o.x *= m.x; o.y *= m.y; o.z *= m.z; o.w *= m.w;
    }
    yp[c * 64 + lane] = o;
  }
}

// grid-stride over rows; per-block partial dgamma / dbeta rows -> part[block][2][H]
template <int CH>
__global__ __launch_bounds__(256) void k_na_bwd(NaArgs a, const float* __restrict__ dy, int64_t ld_dy, const float* __restrict__ mean_in,
                                                const float* __restrict__ rstd_in, float* __restrict__ dx, int64_t ld_dx,
                                                float* __restrict__ part) {
  const int lane = threadIdx.x & 63;
  // blockIdx.y: a column block of CH * 256 floats (segmented statistics only: the segments of a row are independent, so a wide
  // row — the task heads' [N, 21 * 64] — is split over 6 light waves with 8 per SIMD instead of one 254-register wave with 2:
  // 68 -> 4x us at C2; a whole-row LayerNorm always has blockIdx.y = 0)
  const int cb = blockIdx.y * 256 * CH;
  const int wave_g = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int n_waves = gridDim.x * 4;
  const bool pre = a.flags & AGNN_NA_PRE_RELU, post = a.flags & AGNN_NA_POST_RELU, drop = a.p > 0.f;
  const float scale = drop ? 1.f / (1.f - a.p) : 1.f;
  bool on[CH];
  float4 gm[CH], bt[CH], dg[CH], db[CH];
  const bool whole = a.seg >= a.H;
  const int gl = a.seg >> 2;
  const int nseg = whole ? 1 : a.H / a.seg;
  const float invS = 1.f / static_cast<float>(whole ? a.H : a.seg);
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    on[c] = (cb + c * 256 + lane * 4) < a.H;
    gm[c] = on[c] ? reinterpret_cast<const float4*>(a.gamma + cb)[c * 64 + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
    bt[c] = on[c] ? reinterpret_cast<const float4*>(a.beta + cb)[c * 64 + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
    dg[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    db[c] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int64_t row = wave_g; row < a.n; row += n_waves) {
    const float4* xp = reinterpret_cast<const float4*>(a.x + row * a.ld_x + cb);
    const float4* gp = reinterpret_cast<const float4*>(dy + row * ld_dy + cb);
    float4 xr[CH], xh[CH], gx[CH];
    float rs[CH], p1[CH], p2[CH];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int sidx = whole ? 0 : (on[c] ? (cb + c * 256 + lane * 4) / a.seg : 0);
      const float mean = mean_in[row * nseg + sidx], rstd = rstd_in[row * nseg + sidx];
      rs[c] = rstd;
      xr[c] = on[c] ? xp[c * 64 + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
      float4 g = on[c] ? gp[c * 64 + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
      float4 v = xr[c];
      if (pre) { v.x = agnn::relu_nan(v.x); v.y = agnn::relu_nan(v.y); v.z = agnn::relu_nan(v.z); v.w = agnn::relu_nan(v.w); }
      xh[c] = make_float4((v.x - mean) * rstd, (v.y - mean) * rstd, (v.z - mean) * rstd, (v.w - mean) * rstd);
      if (!on[c]) xh[c] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (drop) {
        const float4 m = keep_mask(a, row, cb / 4 + c * 64 + lane, scale);
        g.x *= m.x; g.y *= m.y; g.z *= m.z; g.w *= m.w;
      }
      if (post) {
        if (xh[c].x * gm[c].x + bt[c].x <= 0.f) g.x = 0.f;
        if (xh[c].y * gm[c].y + bt[c].y <= 0.f) g.y = 0.f;
        if (xh[c].z * gm[c].z + bt[c].z <= 0.f) g.z = 0.f;
        if (xh[c].w * gm[c].w + bt[c].w <= 0.f) g.w = 0.f;
      }
      dg[c].x += g.x * xh[c].x; dg[c].y += g.y * xh[c].y; dg[c].z += g.z * xh[c].z; dg[c].w += g.w * xh[c].w;
      db[c].x += g.x; db[c].y += g.y; db[c].z += g.z; db[c].w += g.w;
      gx[c] = make_float4(g.x * gm[c].x, g.y * gm[c].y, g.z * gm[c].z, g.w * gm[c].w);           // d xhat
      p1[c] = (gx[c].x + gx[c].y) + (gx[c].z + gx[c].w);
      p2[c] = (gx[c].x * xh[c].x + gx[c].y * xh[c].y) + (gx[c].z * xh[c].z + gx[c].w * xh[c].w);
      s1 += p1[c];
      s2 += p2[c];
    }
    float m1[CH], m2[CH];
    if (whole) {
      const float t1 = wsum(s1) * invS, t2 = wsum(s2) * invS;
#pragma unroll
      for (int c = 0; c < CH; ++c) { m1[c] = t1; m2[c] = t2; }
    } else {
#pragma unroll
      for (int c = 0; c < CH; ++c) { m1[c] = gsum(p1[c], gl) * invS; m2[c] = gsum(p2[c], gl) * invS; }
    }
    float4* op = reinterpret_cast<float4*>(dx + row * ld_dx + cb);
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      if (!on[c]) continue;
      float4 o = make_float4(rs[c] * (gx[c].x - m1[c] - xh[c].x * m2[c]), rs[c] * (gx[c].y - m1[c] - xh[c].y * m2[c]),
                             rs[c] * (gx[c].z - m1[c] - xh[c].z * m2[c]), rs[c] * (gx[c].w - m1[c] - xh[c].w * m2[c]));
      if (pre) {
        if (xr[c].x <= 0.f) o.x = 0.f;
        if (xr[c].y <= 0.f) o.y = 0.f;
        if (xr[c].z <= 0.f) o.z = 0.f;
        if (xr[c].w <= 0.f) o.w = 0.f;
      }
      op[c * 64 + lane] = o;
    }
  }
  // the block's four waves add their partial rows in LDS in a fixed order, then one row per BLOCK goes to `part`
  __shared__ float4 sm[2 * 64 * CH];
  const int wave = threadIdx.x >> 6;
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        float4 g = dg[c], b = db[c];
        if (w > 0) {
          const float4 pg = sm[c * 64 + lane], pb = sm[(CH + c) * 64 + lane];
          g.x += pg.x; g.y += pg.y; g.z += pg.z; g.w += pg.w;
          b.x += pb.x; b.y += pb.y; b.z += pb.z; b.w += pb.w;
        }
        sm[c * 64 + lane] = g;
        sm[(CH + c) * 64 + lane] = b;
      }
    }
    __syncthreads();
  }
  if (wave == 0) {
    float4* pg = reinterpret_cast<float4*>(part + static_cast<int64_t>(blockIdx.x) * 2 * a.H + cb);
    float4* pb = reinterpret_cast<float4*>(part + static_cast<int64_t>(blockIdx.x) * 2 * a.H + a.H + cb);
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      if (!on[c]) continue;
      pg[c * 64 + lane] = sm[c * 64 + lane];
      pb[c * 64 + lane] = sm[(CH + c) * 64 + lane];
    }
  }
}

// dgamma[j] | dbeta[j] = sum over blocks of part[b][j | H + j].  Block = 32 columns x 32 partial sums (fixed order).
__device__ __forceinline__ void na_colsum_block(const float* __restrict__ part, int n_rows, int width, float* __restrict__ dgamma,
                                                float* __restrict__ dbeta, int H, const int block) {
  __shared__ float sm[32][33];
  const int col = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int j = block * 32 + col;
  float s = 0.f;
  if (j < width) {
    // all 32 loads of a pass are in flight together (the slab has just been written: L2 / MALL hits); a chain of
    // 8-at-a-time loads made this launch four memory round trips long
    for (int w0 = grp; w0 < n_rows; w0 += 32 * 32) {
      float v[32];
#pragma unroll
      for (int k = 0; k < 32; ++k) {
        const int w = w0 + 32 * k;
        v[k] = w < n_rows ? part[static_cast<int64_t>(w) * width + j] : 0.f;
      }
#pragma unroll
      for (int k = 0; k < 32; ++k) s += v[k];
    }
  }
  sm[grp][col] = s;
  __syncthreads();
  if (grp == 0 && j < width) {
    float t = sm[0][col];
#pragma unroll
    for (int k = 1; k < 32; ++k) t += sm[k][col];
    if (j < H) dgamma[j] = t; else dbeta[j - H] = t;
  }
}

// ---- HGT layer epilogue (round 3):  z = dropout_p( relu?( x + sigmoid(s) * (o - x) ) )  ------------------------------------------
// PyG HGTConv's learnable skip connection (`out = alpha * out_lin(gelu(agg)) + (1 - alpha) * x`, alpha = sigmoid(skip[type]))
// followed by the ReLU + dropout the encoders put between layers (via graphmuse HybridHGT, models/analysis.py:445-453): as torch
// ops that is sigmoid, lerp, relu, dropout forward (4 - 5 launches) and ~7 backward (incl. a full reduction for d skip) per node
// type and layer — ~90 launches per C3 step, most of them on the beat / measure types' few thousand rows.  Here one launch each
// way: a wave per row, masks recomputed in backward (ReLU from the recomputed pre-activation, dropout from the counter-based
// generator), d skip = sigmoid'(s) * sum g (o - x) through per-block partial sums that the LAST block to finish (integer ticket)
// adds in index order — deterministic, no float atomics.
struct MixArgs {
  const float* x;          // skip input [n, H] or NULL (no skip connection: z = act(o))
  int64_t ld_x;
  const float* o;
  int64_t ld_o;
  const float* skip;       // device scalar s (NULL with x == NULL)
  int64_t n;
  int32_t H;
  float p;
  uint32_t flags;          // bit 0: ReLU
  const int64_t* rng;
  uint32_t call_id;
};

__device__ __forceinline__ float mix_alpha(const MixArgs& a) { return a.x != nullptr ? 1.f / (1.f + __expf(-*a.skip)) : 1.f; }

template <int CH>
__global__ __launch_bounds__(256) void k_mix_fwd(MixArgs a, float* __restrict__ z, int64_t ld_z, int64_t* __restrict__ rng_used) {
  const int lane = threadIdx.x & 63;
  const int64_t row = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (rng_used != nullptr && a.rng != nullptr && blockIdx.x == 0 && threadIdx.x == 0) {
    rng_used[0] = a.rng[0];
    rng_used[1] = a.rng[1];
  }
  if (row >= a.n) return;
  const float al = mix_alpha(a);
  const bool relu = a.flags & 1u, drop = a.p > 0.f;
  const float scale = drop ? 1.f / (1.f - a.p) : 1.f;
  const float4* op = reinterpret_cast<const float4*>(a.o + row * a.ld_o);
  const float4* xp = a.x != nullptr ? reinterpret_cast<const float4*>(a.x + row * a.ld_x) : nullptr;
  float4* zp = reinterpret_cast<float4*>(z + row * ld_z);
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    if ((c * 256 + lane * 4) >= a.H) continue;
    float4 y = op[c * 64 + lane];
    if (xp != nullptr) {
      const float4 xv = xp[c * 64 + lane];
      y = make_float4(fmaf(al, y.x - xv.x, xv.x), fmaf(al, y.y - xv.y, xv.y), fmaf(al, y.z - xv.z, xv.z), fmaf(al, y.w - xv.w, xv.w));
    }
    if (relu) { y.x = agnn::relu_nan(y.x); y.y = agnn::relu_nan(y.y); y.z = agnn::relu_nan(y.z); y.w = agnn::relu_nan(y.w); }
    if (drop) {
      const float4 m = keep_mask(a, row, c * 64 + lane, scale);
      y.x *= m.x; y.y *= m.y; y.z *= m.z; y.w *= m.w;
    }
    zp[c * 64 + lane] = y;
  }
}

constexpr int kMixBlocks = 1024;     // partial sums of d skip: one per block

template <int CH>
__global__ __launch_bounds__(256) void k_mix_bwd(MixArgs a, const float* __restrict__ dz, int64_t ld_dz, float* __restrict__ dx,
                                                 int64_t ld_dx, float* __restrict__ dout, int64_t ld_do, float* __restrict__ part,
                                                 unsigned int* __restrict__ ticket, float* __restrict__ dskip) {
  const int lane = threadIdx.x & 63;
  const int wave_g = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int n_waves = gridDim.x * 4;
  const float al = mix_alpha(a);
  const bool relu = a.flags & 1u, drop = a.p > 0.f;
  const float scale = drop ? 1.f / (1.f - a.p) : 1.f;
  float acc = 0.f;
  for (int64_t row = wave_g; row < a.n; row += n_waves) {
    const float4* op = reinterpret_cast<const float4*>(a.o + row * a.ld_o);
    const float4* xp = a.x != nullptr ? reinterpret_cast<const float4*>(a.x + row * a.ld_x) : nullptr;
    const float4* gp = reinterpret_cast<const float4*>(dz + row * ld_dz);
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      if ((c * 256 + lane * 4) >= a.H) continue;
      const float4 ov = op[c * 64 + lane];
      const float4 xv = xp != nullptr ? xp[c * 64 + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
      float4 g = gp[c * 64 + lane];
      const float4 d = make_float4(ov.x - xv.x, ov.y - xv.y, ov.z - xv.z, ov.w - xv.w);
      if (drop) {
        const float4 m = keep_mask(a, row, c * 64 + lane, scale);
        g.x *= m.x; g.y *= m.y; g.z *= m.z; g.w *= m.w;
      }
      if (relu) {                          // the pre-activation, recomputed exactly as the forward computed it
        if ((xp != nullptr ? fmaf(al, d.x, xv.x) : ov.x) <= 0.f) g.x = 0.f;
        if ((xp != nullptr ? fmaf(al, d.y, xv.y) : ov.y) <= 0.f) g.y = 0.f;
        if ((xp != nullptr ? fmaf(al, d.z, xv.z) : ov.z) <= 0.f) g.z = 0.f;
        if ((xp != nullptr ? fmaf(al, d.w, xv.w) : ov.w) <= 0.f) g.w = 0.f;
      }
      reinterpret_cast<float4*>(dout + row * ld_do)[c * 64 + lane] = make_float4(al * g.x, al * g.y, al * g.z, al * g.w);
      if (dx != nullptr) {
        const float be = 1.f - al;
        reinterpret_cast<float4*>(dx + row * ld_dx)[c * 64 + lane] = make_float4(be * g.x, be * g.y, be * g.z, be * g.w);
      }
      acc += (g.x * d.x + g.y * d.y) + (g.z * d.z + g.w * d.w);
    }
  }
  if (dskip == nullptr) return;            // block-uniform
  acc = wsum(acc);
  __shared__ float sw[4];
  __shared__ bool last;
  if (lane == 0) sw[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float v = (sw[0] + sw[1]) + (sw[2] + sw[3]);
    const unsigned int before = atomicExch(reinterpret_cast<unsigned int*>(part) + blockIdx.x, __float_as_uint(v));   // device-scope publication
    asm volatile("" ::"v"(before));
    last = atomicAdd(ticket, 1u) == gridDim.x - 1;
  }
  __syncthreads();
  if (last) {
    __shared__ float sp[kMixBlocks];
    for (int i = threadIdx.x; i < static_cast<int>(gridDim.x); i += 256) sp[i] = __uint_as_float(atomicOr(reinterpret_cast<unsigned int*>(part) + i, 0u));
    __syncthreads();
    if (threadIdx.x == 0) {
      float t = 0.f;
      for (int i = 0; i < static_cast<int>(gridDim.x); ++i) t += sp[i];                    // index order: the same bits every run
      *dskip = t * al * (1.f - al);
      atomicExch(ticket, 0u);
    }
  }
}

__global__ __launch_bounds__(1024) void k_na_colsum(const float* __restrict__ part, int n_rows, int width, float* __restrict__ dgamma,
                                                    float* __restrict__ dbeta, int H) {
  na_colsum_block(part, n_rows, width, dgamma, dbeta, H, blockIdx.x);
}

// several pending column sums in one launch (they only feed the optimizer and are pending together at the flush points)
constexpr int kColsumBatchMax = 16;
struct ColsumBatch {
  const float* part[kColsumBatchMax];
  float* dgamma[kColsumBatchMax];
  float* dbeta[kColsumBatchMax];
  int32_t n_rows[kColsumBatchMax], H[kColsumBatchMax], first[kColsumBatchMax + 1];
  int32_t n;
};

__global__ __launch_bounds__(1024) void k_na_colsum_batch(ColsumBatch b) {
  int i = 0;
  const int blk = blockIdx.x;
  while (i + 1 < b.n && blk >= b.first[i + 1]) ++i;
  na_colsum_block(b.part[i], b.n_rows[i], 2 * b.H[i], b.dgamma[i], b.dbeta[i], b.H[i], blk - b.first[i]);
}

constexpr int kBwdBlocks = 1024;    // 4 waves per SIMD; partial slab = 1024 * 2H floats (one row per block)

int na_check(const char* who, const void* x, int64_t ld_x, const void* gamma, const void* beta, int64_t n, int32_t H, int32_t seg,
             float p, const void* rng) {
  using namespace agnn;
  if (n < 0 || H <= 0 || (H & 3) || H > 2048) return fail(AGNN_EINVAL, "%s: H=%d must be a multiple of 4 in [4,2048], n=%lld", who, H, (long long)n);
  if (seg != H) {
    const int gl = seg / 4;
    if (seg <= 0 || (seg & 3) || 256 % seg != 0 || H % seg != 0 || (gl & (gl - 1)))
      return fail(AGNN_EINVAL, "%s: segment %d must be 4*2^k, divide 256 and divide H=%d", who, seg, H);
  }
  if (p < 0.f || p >= 1.f) return fail(AGNN_EINVAL, "%s: dropout p=%f", who, p);
  if (n == 0) return 1;
  if (!x || !gamma || !beta) return fail(AGNN_EINVAL, "%s: null argument", who);
  if (p > 0.f && !rng) return fail(AGNN_EINVAL, "%s: dropout needs the device rng state", who);
  if (!aligned16(x) || !aligned16(gamma) || !aligned16(beta) || (ld_x & 3) || ld_x < H) return fail(AGNN_EALIGN, "%s: misaligned", who);
  return 0;
}

}  // namespace

extern "C" size_t agnn_norm_act_workspace_bytes(int32_t H) { return static_cast<size_t>(kBwdBlocks) * 2 * static_cast<size_t>(H > 0 ? H : 0) * sizeof(float); }

#define AGNN_NA_DISPATCH(KERN, ...)                                                          \
  do {                                                                                       \
    if (H <= 256) hipLaunchKernelGGL(KERN<1>, grid, block, 0, s, __VA_ARGS__);               \
    else if (H <= 512) hipLaunchKernelGGL(KERN<2>, grid, block, 0, s, __VA_ARGS__);          \
    else if (H <= 1024) hipLaunchKernelGGL(KERN<4>, grid, block, 0, s, __VA_ARGS__);         \
    else if (H <= 1536) hipLaunchKernelGGL(KERN<6>, grid, block, 0, s, __VA_ARGS__);         \
    else hipLaunchKernelGGL(KERN<8>, grid, block, 0, s, __VA_ARGS__);                        \
  } while (0)

extern "C" int agnn_norm_act_fwd_f32(const float* x, int64_t ld_x, const float* gamma, const float* beta, int32_t seg, int64_t n,
                                     int32_t H, float eps, float p, uint32_t flags, const int64_t* rng_state, uint32_t call_id,
                                     float* y, int64_t ld_y, float* mean, float* rstd, int64_t* rng_used, agnn_stream_t stream_) {
  using namespace agnn;
  if (seg <= 0 || seg > H) seg = H;
  if (int rc = na_check("norm_act_fwd", x, ld_x, gamma, beta, n, H, seg, p, rng_state)) return rc > 0 ? AGNN_OK : rc;
  if (!y || !mean || !rstd || !aligned16(y) || (ld_y & 3) || ld_y < H) return fail(AGNN_EALIGN, "norm_act_fwd: output misaligned");
  NaArgs a{x, ld_x, gamma, beta, n, H, eps, p, flags, rng_state, call_id, seg};
  const dim3 grid(static_cast<unsigned>((n + 3) / 4)), block(256);
  hipStream_t s = static_cast<hipStream_t>(stream_);
  AGNN_NA_DISPATCH(k_na_fwd, a, y, ld_y, mean, rstd, rng_used);
  return check_launch("norm_act_fwd");
}

extern "C" int agnn_norm_act_bwd_f32(const float* x, int64_t ld_x, const float* gamma, const float* beta, int32_t seg, int64_t n,
                                     int32_t H, float eps, float p, uint32_t flags, const int64_t* rng_state, uint32_t call_id,
                                     const float* dy, int64_t ld_dy, const float* mean, const float* rstd, float* dx, int64_t ld_dx,
                                     float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, agnn_stream_t stream_) {
  using namespace agnn;
  if (seg <= 0 || seg > H) seg = H;
  if (int rc = na_check("norm_act_bwd", x, ld_x, gamma, beta, n, H, seg, p, rng_state)) return rc > 0 ? AGNN_OK : rc;
  if (!dy || !mean || !rstd || !dx || !workspace || (dgamma == nullptr) != (dbeta == nullptr)) return fail(AGNN_EINVAL, "norm_act_bwd: null argument");
  if (!aligned16(dy) || !aligned16(dx) || !aligned16(workspace) || (ld_dy & 3) || (ld_dx & 3)) return fail(AGNN_EALIGN, "norm_act_bwd: misaligned");
  if (workspace_bytes < agnn_norm_act_workspace_bytes(H)) return fail(AGNN_ENOMEM, "norm_act_bwd: workspace too small");
  NaArgs a{x, ld_x, gamma, beta, n, H, eps, p, flags, rng_state, call_id, seg};
  float* part = static_cast<float*>(workspace);
  hipStream_t s = static_cast<hipStream_t>(stream_);
  int nb = static_cast<int>((n + 3) / 4);                 // at least one row per wave
  if (nb > kBwdBlocks) nb = kBwdBlocks;
  const dim3 block(256);
  if (seg < H && H > 256) {                               // segmented statistics: one light wave per 256-column block of a row
    const dim3 grid(nb, static_cast<unsigned>((H + 255) / 256));
    hipLaunchKernelGGL(k_na_bwd<1>, grid, block, 0, s, a, dy, ld_dy, mean, rstd, dx, ld_dx, part);
  } else {
    const dim3 grid(nb);
    AGNN_NA_DISPATCH(k_na_bwd, a, dy, ld_dy, mean, rstd, dx, ld_dx, part);
  }
  if (int rc = check_launch("norm_act_bwd")) return rc;
  if (dgamma == nullptr) return AGNN_OK;              // the column sums are launched later: agnn_norm_act_colsum_f32
  const int width = 2 * H;
  hipLaunchKernelGGL(k_na_colsum, dim3((width + 31) / 32), dim3(1024), 0, s, part, nb, width, dgamma, dbeta, H);
  return check_launch("norm_act_colsum");
}

extern "C" int agnn_norm_act_colsum_f32(const void* workspace, size_t workspace_bytes, int64_t n, int32_t H, float* dgamma, float* dbeta,
                                        agnn_stream_t stream_) {
  using namespace agnn;
  if (n < 0 || H <= 0 || (H & 3) || H > 2048) return fail(AGNN_EINVAL, "norm_act_colsum: H=%d n=%lld", H, (long long)n);
  if (n == 0) return AGNN_OK;
  if (!workspace || !dgamma || !dbeta) return fail(AGNN_EINVAL, "norm_act_colsum: null argument");
  if (workspace_bytes < agnn_norm_act_workspace_bytes(H)) return fail(AGNN_ENOMEM, "norm_act_colsum: workspace too small");
  int nb = static_cast<int>((n + 3) / 4);                 // the partial rows agnn_norm_act_bwd_f32 wrote for n rows
  if (nb > kBwdBlocks) nb = kBwdBlocks;
  const int width = 2 * H;
  hipLaunchKernelGGL(k_na_colsum, dim3((width + 31) / 32), dim3(1024), 0, static_cast<hipStream_t>(stream_),
                     static_cast<const float*>(workspace), nb, width, dgamma, dbeta, H);
  return check_launch("norm_act_colsum");
}

// ---- HGT layer epilogue -----------------------------------------------------------------------------------------------------
namespace {
int mix_check(const char* who, const float* x, int64_t ld_x, const float* o, int64_t ld_o, const float* skip, int64_t n, int32_t H, float p,
              const void* rng) {
  using namespace agnn;
  if (n < 0 || H <= 0 || (H & 3) || H > 2048) return fail(AGNN_EINVAL, "%s: H=%d must be a multiple of 4 in [4,2048], n=%lld", who, H, (long long)n);
  if (p < 0.f || p >= 1.f) return fail(AGNN_EINVAL, "%s: dropout p=%f", who, p);
  if (n == 0) return 1;
  if (!o || (x != nullptr) != (skip != nullptr)) return fail(AGNN_EINVAL, "%s: null output operand, or x and skip not given together", who);
  if (p > 0.f && !rng) return fail(AGNN_EINVAL, "%s: dropout needs the device rng state", who);
  if (!aligned16(o) || (ld_o & 3) || ld_o < H || (x && (!aligned16(x) || (ld_x & 3) || ld_x < H))) return fail(AGNN_EALIGN, "%s: misaligned", who);
  return 0;
}
}  // namespace

extern "C" size_t agnn_skip_act_workspace_bytes(void) { return 256 + kMixBlocks * sizeof(float); }

extern "C" int agnn_skip_act_fwd_f32(const float* x, int64_t ld_x, const float* o, int64_t ld_o, const float* skip, int64_t n, int32_t H,
                                     float p, uint32_t flags, const int64_t* rng_state, uint32_t call_id, float* z, int64_t ld_z,
                                     int64_t* rng_used, agnn_stream_t stream_) {
  using namespace agnn;
  if (int rc = mix_check("skip_act_fwd", x, ld_x, o, ld_o, skip, n, H, p, rng_state)) return rc > 0 ? AGNN_OK : rc;
  if (!z || !aligned16(z) || (ld_z & 3) || ld_z < H) return fail(AGNN_EALIGN, "skip_act_fwd: output misaligned");
  MixArgs a{x, ld_x, o, ld_o, skip, n, H, p, flags, rng_state, call_id};
  const dim3 grid(static_cast<unsigned>((n + 3) / 4)), block(256);
  hipStream_t s = static_cast<hipStream_t>(stream_);
  AGNN_NA_DISPATCH(k_mix_fwd, a, z, ld_z, rng_used);
  return check_launch("skip_act_fwd");
}

/* workspace: agnn_skip_act_workspace_bytes(), 256-byte aligned, ZERO-FILLED once by the caller; every call leaves its ticket at zero. */
extern "C" int agnn_skip_act_bwd_f32(const float* x, int64_t ld_x, const float* o, int64_t ld_o, const float* skip, int64_t n, int32_t H,
                                     float p, uint32_t flags, const int64_t* rng_state, uint32_t call_id, const float* dz, int64_t ld_dz,
                                     float* dx, int64_t ld_dx, float* dout, int64_t ld_do, float* dskip, void* workspace,
                                     size_t workspace_bytes, agnn_stream_t stream_) {
  using namespace agnn;
  if (int rc = mix_check("skip_act_bwd", x, ld_x, o, ld_o, skip, n, H, p, rng_state)) return rc > 0 ? AGNN_OK : rc;
  if (!dz || !dout || !aligned16(dz) || !aligned16(dout) || (ld_dz & 3) || (ld_do & 3) || ld_dz < H || ld_do < H) return fail(AGNN_EALIGN, "skip_act_bwd: gradients misaligned");
  if (dx && (!aligned16(dx) || (ld_dx & 3) || ld_dx < H)) return fail(AGNN_EALIGN, "skip_act_bwd: dx misaligned");
  if (dskip && (!workspace || workspace_bytes < agnn_skip_act_workspace_bytes() || (reinterpret_cast<uintptr_t>(workspace) & 255u)))
    return fail(AGNN_ENOMEM, "skip_act_bwd: d skip needs the zero-filled 256-byte aligned workspace");
  if (dskip && !x) return fail(AGNN_EINVAL, "skip_act_bwd: d skip without a skip connection");
  MixArgs a{x, ld_x, o, ld_o, skip, n, H, p, flags, rng_state, call_id};
  int64_t nb = (n + 3) / 4;
  if (nb > kMixBlocks) nb = kMixBlocks;
  const dim3 grid(static_cast<unsigned>(nb)), block(256);
  hipStream_t s = static_cast<hipStream_t>(stream_);
  unsigned int* ticket = static_cast<unsigned int*>(workspace);
  float* part = workspace ? reinterpret_cast<float*>(static_cast<char*>(workspace) + 256) : nullptr;
  AGNN_NA_DISPATCH(k_mix_bwd, a, dz, ld_dz, dx, ld_dx, dout, ld_do, part, ticket, dskip);
  return check_launch("skip_act_bwd");
}

extern "C" int agnn_norm_act_colsum_batch_f32(int32_t n_items, const agnn_colsum_item_t* items, agnn_stream_t stream_) {
  using namespace agnn;
  if (n_items <= 0 || n_items > kColsumBatchMax || !items) return fail(AGNN_EINVAL, "norm_act_colsum_batch: %d items (1 .. %d)", n_items, kColsumBatchMax);
  ColsumBatch b{};
  b.n = n_items;
  int blocks = 0;
  for (int i = 0; i < n_items; ++i) {
    const agnn_colsum_item_t& it = items[i];
    if (it.n <= 0 || it.H <= 0 || (it.H & 3) || it.H > 2048 || !it.workspace || !it.dgamma || !it.dbeta)
      return fail(AGNN_EINVAL, "norm_act_colsum_batch: item %d: bad sizes or null argument", i);
    if (it.workspace_bytes < agnn_norm_act_workspace_bytes(it.H)) return fail(AGNN_ENOMEM, "norm_act_colsum_batch: item %d: workspace too small", i);
    int nb = static_cast<int>((it.n + 3) / 4);
    if (nb > kBwdBlocks) nb = kBwdBlocks;
    b.part[i] = static_cast<const float*>(it.workspace);
    b.dgamma[i] = it.dgamma;
    b.dbeta[i] = it.dbeta;
    b.n_rows[i] = nb;
    b.H[i] = it.H;
    b.first[i] = blocks;
    blocks += (2 * it.H + 31) / 32;
  }
  b.first[n_items] = blocks;
  hipLaunchKernelGGL(k_na_colsum_batch, dim3(blocks), dim3(1024), 0, static_cast<hipStream_t>(stream_), b);
  return check_launch("norm_act_colsum_batch");
}
