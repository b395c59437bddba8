// Batched small 2-D gather/sum:  dst_i[r, c] = sum_{k < n_src_i} src_{i,k}[r, c]   for up to AGNN_PACK_MAX_ITEMS items per launch
// (n_src_i = 0: the piece is zero-filled — the HGT backward's column-block clears, one launch instead of one per block).
//
// The fused layers present per-relation / per-task parameters to the GEMMs as ONE operand (PyG HeteroConv's four
// SAGEConv.lin_l weights side by side, their lin_r weights and biases summed — ref: models/cadence.py:147-159,174) and
// hand the gradient of that operand back as per-parameter tensors.  Done with torch ops that is a cat, six adds and
// nine copies per layer, each a 5-7 us launch moving a few hundred KB; here it is one launch in each direction.
// Item descriptors travel as kernel arguments (no device table to keep in sync, capturable in a hipGraph).
#include "agnn_common.h"

namespace {

struct PackTable {
  agnn_pack_item_t it[AGNN_PACK_MAX_ITEMS];
  int32_t first_block[AGNN_PACK_MAX_ITEMS + 1];   // prefix of blocks per item (a block = 256 threads x 4 floats x 4 rows-steps)
  int32_t n;
};

constexpr int kElemsPerBlock = 4096;

__global__ __launch_bounds__(256) void k_pack(PackTable t) {
  int i = 0;
  while (i + 1 < t.n && static_cast<int>(blockIdx.x) >= t.first_block[i + 1]) ++i;       // <= 24 uniform steps
  const agnn_pack_item_t& it = t.it[i];
  const int64_t total = static_cast<int64_t>(it.rows) * it.cols;
  const int64_t base = static_cast<int64_t>(blockIdx.x - t.first_block[i]) * kElemsPerBlock;
  const bool vec = (it.cols & 3) == 0 && it.vec_ok;
  if (vec) {
    const int c4 = it.cols >> 2;
#pragma unroll
    for (int u = 0; u < kElemsPerBlock / 1024; ++u) {
      const int64_t e = (base >> 2) + u * 256 + threadIdx.x;          // float4 index
      if (e >= (total >> 2)) break;
      const int64_t r = static_cast<uint32_t>(e) / static_cast<uint32_t>(c4);       // 32-bit: an item has < 2^31 elements (host-checked)
      const int c = static_cast<int>(e - r * c4) * 4;
      float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
      if (it.n_src > 0) s = *reinterpret_cast<const float4*>(it.src[0] + r * it.ld_src + c);
      for (int k = 1; k < it.n_src; ++k) {
        const float4 v = *reinterpret_cast<const float4*>(it.src[k] + r * it.ld_src + c);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
      *reinterpret_cast<float4*>(it.dst + r * it.ld_dst + c) = s;
    }
  } else {
#pragma unroll 4
    for (int u = 0; u < kElemsPerBlock / 256; ++u) {
      const int64_t e = base + u * 256 + threadIdx.x;
      if (e >= total) break;
      const int64_t r = static_cast<uint32_t>(e) / static_cast<uint32_t>(it.cols);
      const int c = static_cast<int>(e - r * it.cols);
      float s = it.n_src > 0 ? it.src[0][r * it.ld_src + c] : 0.f;
      for (int k = 1; k < it.n_src; ++k) s += it.src[k][r * it.ld_src + c];
      it.dst[r * it.ld_dst + c] = s;
    }
  }
}

}  // namespace

extern "C" int agnn_pack_f32(int32_t n_items, const agnn_pack_item_t* items, agnn_stream_t stream_) {
  using namespace agnn;
  if (n_items < 0) return fail(AGNN_EINVAL, "pack: negative item count");
  if (n_items > 0 && !items) return fail(AGNN_EINVAL, "pack: null item table");
  hipStream_t s = static_cast<hipStream_t>(stream_);
  for (int32_t at = 0; at < n_items; at += AGNN_PACK_MAX_ITEMS) {
    PackTable t;
    t.n = 0;
    int32_t blocks = 0;
    for (int32_t i = at; i < n_items && t.n < AGNN_PACK_MAX_ITEMS; ++i) {
      agnn_pack_item_t it = items[i];
      if (it.rows < 0 || it.cols < 0 || it.n_src < 0 || it.n_src > AGNN_PACK_MAX_SRC)
        return fail(AGNN_EINVAL, "pack: item %d has rows=%d cols=%d n_src=%d", i, it.rows, it.cols, it.n_src);
      if (it.rows == 0 || it.cols == 0) continue;
      if (static_cast<int64_t>(it.rows) * it.cols >= (int64_t{1} << 31)) return fail(AGNN_EINVAL, "pack: item %d has 2^31 or more elements", i);
      if (!it.dst || it.ld_dst < it.cols || (it.n_src > 0 && it.ld_src < it.cols)) return fail(AGNN_EINVAL, "pack: item %d: null dst or ld < cols", i);
      if (it.n_src == 0) it.ld_src = it.ld_dst;
      bool v = aligned16(it.dst) && (it.ld_dst & 3) == 0 && (it.ld_src & 3) == 0;
      for (int k = 0; k < it.n_src; ++k) {
        if (!it.src[k]) return fail(AGNN_EINVAL, "pack: item %d: null source %d", i, k);
        v = v && aligned16(it.src[k]);
      }
      it.vec_ok = v ? 1 : 0;
      t.first_block[t.n] = blocks;
      const int64_t total = static_cast<int64_t>(it.rows) * it.cols;
      blocks += static_cast<int32_t>((total + kElemsPerBlock - 1) / kElemsPerBlock);
      t.it[t.n++] = it;
    }
    if (t.n == 0) continue;
    t.first_block[t.n] = blocks;
    hipLaunchKernelGGL(k_pack, dim3(blocks), dim3(256), 0, s, t);
    if (int rc = check_launch("pack")) return rc;
  }
  return AGNN_OK;
}

namespace {
__global__ void k_stamp(unsigned long long* slot) { *slot = wall_clock64(); }
}  // namespace

extern "C" int agnn_debug_stamp(uint64_t* slot, agnn_stream_t stream_) {
  using namespace agnn;
  if (!slot) return fail(AGNN_EINVAL, "debug_stamp: null slot");
  hipLaunchKernelGGL(k_stamp, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream_), reinterpret_cast<unsigned long long*>(slot));
  return check_launch("debug_stamp");
}
