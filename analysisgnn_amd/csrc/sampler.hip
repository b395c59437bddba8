// Device-side batch assembly: neighbour sampling of note windows out of score graphs that are RESIDENT in HBM.
//
// The reference assembles every training batch on the host: graphmuse's `MuseNeighborLoader` (reference
// analysisgnn/data/datamodules/analysis.py:270-293: subgraph_size = 500 target notes per window, num_neighbors =
// [5] * (num_layers - 1), batch_size windows per batch, worker processes, then a host-to-device copy) hands over a
// hop-ordered HeteroData whose per-hop counts drive PyG's trim_to_layer.  At a few milliseconds per training step that
// loader is the bottleneck, and its batches have data-dependent shapes, which a replayed hipGraph cannot take.
// Here the scores stay on the device (one CSR by destination per relation over ALL notes of ALL scores: 288 GB of HBM
// hold any corpus of this kind) and ONE launch samples a whole batch into buffers of STATIC shape:
//   * nodes:  [ n_sub * n_targets targets | n_sub * cap[0] hop-1 slots | n_sub * cap[1] hop-2 slots | ... ]; a subgraph's
//     new nodes of a hop fill its slots in ascending global id, the rest are padding (gid -1: no features, no edges);
//   * edges of relation r:  hop h owns n_sub * F_h * fan[h] slots (F_1 = n_targets, F_h = cap[h-2]), slot
//     ((s * F_h + i) * fan + k) belongs to the k-th sampled in-neighbour of frontier node i of subgraph s — no
//     compaction, no atomics on the edge list; unused slots hold (-1, -1), which agnn_csr_build drops.
// Hop-ordered as PyG's NeighborLoader lays a batch out, so `trim_to_layer` with the STATIC per-hop capacities
// (num_sampled_nodes = [n_sub*n_targets, n_sub*cap[0], ...], num_sampled_edges[r] = [n_sub*F_1*fan[0], ...]) trims
// exactly the padded hop blocks: the whole training step, sampler included, is one fixed launch sequence.
// Sampling: all in-neighbours when a node has at most `fan` of them (the usual case in a score graph), otherwise
// `fan` of them without replacement by selection sampling (Knuth's algorithm S) on Philox-4x32-10 keyed by
// (seed; step, destination, relation, hop): the batch is a pure function of (windows, seed, step) — bitwise
// reproducible, checked against oracle/sampler_ref.py.
// One workgroup per subgraph; new nodes are collected in an LDS hash set, sorted, numbered; integer work only.
#include "agnn_common.h"

namespace {

// LDS hash set of EVERY distinct out-of-window source a subgraph's hops propose — the kept ones (<= sum of caps) and the
// ones the capacity cut dropped, which stay known so that a later hop does not bring them back.  A window of consecutive
// notes proposes a few dozen such sources; the worst case (n_targets * n_rel * fan distinct ones) does not fit any LDS
// table, so a full table is an ERROR the kernel reports through `status` (the source is treated as dropped, every probe
// loop is bounded by kHash), never a hang.
constexpr int kHash = 4096;
constexpr int kNewMax = 512;      // list of one hop's candidates; more than that: the cap smallest are found by a search over the table
constexpr int kEmpty = -1;
constexpr int kValDropped = -1;   // hval: known, without a batch slot
constexpr int kValNew = -2;       // hval: proposed in the hop being processed

struct SamplerArgs {
  agnn_sampler_t c;
};

__device__ __forceinline__ uint32_t s_mulhi(uint32_t a, uint32_t b) { return __umulhi(a, b); }

__device__ __forceinline__ uint4 s_philox(uint4 c, uint2 k) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t h0 = s_mulhi(0xD2511F53u, c.x), l0 = 0xD2511F53u * c.x;
    const uint32_t h1 = s_mulhi(0xCD9E8D57u, c.z), l1 = 0xCD9E8D57u * c.z;
    c = make_uint4(h1 ^ c.y ^ k.x, l1, h0 ^ c.w ^ k.y, l0);
    k.x += 0x9E3779B9u;
    k.y += 0xBB67AE85u;
  }
  return c;
}

// Positions (into the in-neighbour list of length deg) of the sampled neighbours, ascending; returns how many.
__device__ __forceinline__ int select_positions(int deg, int fan, uint32_t dst, uint32_t tag, uint32_t step, uint2 key, int* pos) {
  if (deg <= fan) {
    for (int k = 0; k < deg; ++k) pos[k] = k;
    return deg;
  }
  int m = 0;
  uint4 rnd = make_uint4(0, 0, 0, 0);
  for (int t = 0; t < deg && m < fan; ++t) {
    if ((t & 3) == 0) rnd = s_philox(make_uint4(dst, tag, static_cast<uint32_t>(t >> 2), step), key);
    const uint32_t u = (t & 3) == 0 ? rnd.x : (t & 3) == 1 ? rnd.y : (t & 3) == 2 ? rnd.z : rnd.w;
    // select with probability (fan - m) / (deg - t)
    if (s_mulhi(u, static_cast<uint32_t>(deg - t)) < static_cast<uint32_t>(fan - m)) pos[m++] = t;
  }
  return m;
}

__device__ __forceinline__ uint32_t hslot(int gid) { return (static_cast<uint32_t>(gid) * 2654435761u) >> 20; }   // 12 bits

constexpr int kThreads = 1024;    // one workgroup per subgraph: 16 waves walk the (frontier node, relation) items side by side

__global__ __launch_bounds__(kThreads) void k_sample_hops(SamplerArgs A) {
  const agnn_sampler_t& c = A.c;
  __shared__ int hkey[kHash];
  __shared__ int hval[kHash];
  __shared__ int newl[kNewMax];
  __shared__ int fr_gid[AGNN_SAMPLER_MAX_CAP];
  __shared__ int n_new, n_drop, n_over, n_le;
  const int s = blockIdx.x, tid = threadIdx.x;
  const int T = c.n_targets;
  const int w = c.win_start[s];
  const uint2 key = make_uint2(static_cast<uint32_t>(c.rng[0]), static_cast<uint32_t>(static_cast<uint64_t>(c.rng[0]) >> 32));
  const uint32_t step = static_cast<uint32_t>(c.rng[1]);
  for (int i = tid; i < kHash; i += kThreads) { hkey[i] = kEmpty; hval[i] = kValDropped; }
  for (int i = tid; i < T; i += kThreads) c.node_gid[static_cast<int64_t>(s) * T + i] = w + i;
  if (tid == 0) { n_drop = 0; n_over = 0; }
  __syncthreads();

  int64_t nbase = static_cast<int64_t>(c.n_sub) * T;          // first batch node of this hop's block
  int64_t ebase = 0;                                            // first edge slot of this hop's block (same for every relation)
  int F = T, Fcap = T;                                          // frontier: actual size, slots
  int64_t fr_local0 = static_cast<int64_t>(s) * T;             // batch id of frontier node 0
  for (int h = 0; h < c.n_hops; ++h) {
    const int fan = c.fan[h], cap = c.cap[h];
    if (tid == 0) n_new = 0;
    for (int i = tid; i < kNewMax; i += kThreads) newl[i] = 0x7fffffff;
    __syncthreads();
    // ---- A: the sampled sources that are neither in the window nor known yet
    for (int it = tid; it < F * c.n_rel; it += kThreads) {
      const int i = it / c.n_rel, r = it - i * c.n_rel;
      const int dst = h == 0 ? w + i : fr_gid[i];
      const int st = c.rowptr[r][dst], deg = c.rowptr[r][dst + 1] - st;
      int pos[AGNN_SAMPLER_MAX_FAN];
      const int cnt = select_positions(deg, fan, static_cast<uint32_t>(dst), static_cast<uint32_t>(r + (h << 8)), step, key, pos);
      for (int k = 0; k < cnt; ++k) {
        const int src = c.col[r][st + pos[k]];
        if (src >= w && src < w + T) continue;
        uint32_t q = hslot(src);
        int probe = 0;
        for (; probe < kHash; ++probe, q = (q + 1) & (kHash - 1)) {
          const int old = atomicCAS(&hkey[q], kEmpty, src);
          if (old == kEmpty) {                              // first sight of this node
            hval[q] = kValNew;
            const int idx = atomicAdd(&n_new, 1);
            if (idx < kNewMax) newl[idx] = src;
            break;
          }
          if (old == src) break;
        }
        if (probe == kHash) atomicAdd(&n_over, 1);          // table full: the source is lost (phase C finds nothing), reported
      }
    }
    __syncthreads();
    if (n_new > kNewMax) {
      // More candidates than the list holds (which ones it holds depends on the order of the atomics): take the `cap`
      // smallest ids out of the table instead.  Binary search for the cap-th smallest id among this hop's entries (ids are
      // distinct), then list the entries up to it: a pure function of the candidate SET.
      int lo = 0, hi = 0x7fffffff;                          // invariant: count(id <= hi) >= cap (n_new > kNewMax >= cap)
      while (lo < hi) {
        const int mid = lo + ((hi - lo) >> 1);
        if (tid == 0) n_le = 0;
        __syncthreads();
        int mine = 0;
        for (int i = tid; i < kHash; i += kThreads) mine += (hval[i] == kValNew && hkey[i] <= mid) ? 1 : 0;
        if (mine) atomicAdd(&n_le, mine);
        __syncthreads();
        const int le = n_le;
        __syncthreads();
        if (le >= cap) hi = mid; else lo = mid + 1;
      }
      for (int i = tid; i < kNewMax; i += kThreads) newl[i] = 0x7fffffff;
      if (tid == 0) n_le = 0;
      __syncthreads();
      for (int i = tid; i < kHash; i += kThreads)
        if (hval[i] == kValNew && hkey[i] <= lo) newl[atomicAdd(&n_le, 1)] = hkey[i];    // exactly cap entries, sorted below
      __syncthreads();
    }
    // ---- B: ascending global id (bitonic sort of the padded list), the first `cap` get this hop's slots
    const int listed = n_new <= kNewMax ? n_new : cap;
    int P = 2;                                              // sort only as many slots as there are candidates (usually a handful)
    while (P < listed && P < kNewMax) P <<= 1;
    for (int k2 = 2; k2 <= P; k2 <<= 1) {
      for (int j = k2 >> 1; j > 0; j >>= 1) {
        for (int i = tid; i < P; i += kThreads) {
          const int ixj = i ^ j;
          if (ixj > i) {
            const int a = newl[i], b = newl[ixj];
            const bool up = (i & k2) == 0;
            if ((a > b) == up) { newl[i] = b; newl[ixj] = a; }
          }
        }
        __syncthreads();
      }
    }
    const int kept = listed < cap ? listed : cap;
    if (tid == 0 && n_new > kept) n_drop += n_new - kept;
    if (tid == 0 && c.kept != nullptr) c.kept[h * c.n_sub + s] = kept;                   // for agnn_sample_compact
    for (int i = tid; i < cap; i += kThreads) {
      const int g = i < kept ? newl[i] : -1;
      c.node_gid[nbase + static_cast<int64_t>(s) * cap + i] = g;
      if (i < kept) {
        uint32_t q = hslot(g);
        for (int probe = 0; probe < kHash && hkey[q] != g; ++probe) q = (q + 1) & (kHash - 1);   // g is in the table: it was listed from it
        hval[q] = static_cast<int>(nbase + static_cast<int64_t>(s) * cap + i);
      }
    }
    __syncthreads();
    if (n_new > kept) {                                     // the candidates the cut dropped stay known, without a slot
      for (int i = tid; i < kHash; i += kThreads)
        if (hval[i] == kValNew) hval[i] = kValDropped;
    }
    __syncthreads();
    // ---- C: the edges, every (frontier node, relation, k) in its own slot
    for (int it = tid; it < Fcap * c.n_rel; it += kThreads) {
      const int i = it / c.n_rel, r = it - i * c.n_rel;
      int64_t* e0 = c.edges[r] + ebase + (static_cast<int64_t>(s) * Fcap + i) * fan;
      int64_t* e1 = e0 + c.e_cap;
      int cnt = 0, st = 0;
      int pos[AGNN_SAMPLER_MAX_FAN];
      if (i < F) {
        const int dst = h == 0 ? w + i : fr_gid[i];
        st = c.rowptr[r][dst];
        const int deg = c.rowptr[r][dst + 1] - st;
        cnt = select_positions(deg, fan, static_cast<uint32_t>(dst), static_cast<uint32_t>(r + (h << 8)), step, key, pos);
      }
      for (int k = 0; k < fan; ++k) {
        int64_t src_l = -1;
        if (k < cnt) {
          const int src = c.col[r][st + pos[k]];
          if (src >= w && src < w + T) {
            src_l = static_cast<int64_t>(s) * T + (src - w);
          } else {
            uint32_t q = hslot(src);
            int probe = 0;
            for (; probe < kHash && hkey[q] != src && hkey[q] != kEmpty; ++probe) q = (q + 1) & (kHash - 1);
            src_l = (probe < kHash && hkey[q] == src) ? hval[q] : -1;   // -1: dropped by the capacity cut (or lost to a full table)
          }
        }
        e0[k] = src_l;
        e1[k] = src_l >= 0 ? fr_local0 + i : -1;
      }
    }
    __syncthreads();
    // ---- next frontier = this hop's new nodes, in slot order
    for (int i = tid; i < cap; i += kThreads) fr_gid[i] = i < kept ? newl[i] : -1;
    ebase += static_cast<int64_t>(c.n_sub) * Fcap * fan;
    fr_local0 = nbase + static_cast<int64_t>(s) * cap;
    nbase += static_cast<int64_t>(c.n_sub) * cap;
    F = kept;
    Fcap = cap;
    __syncthreads();
  }
  if (tid == 0 && n_drop > 0 && c.drops != nullptr) atomicAdd(c.drops, n_drop);        // a statistic: expected on crowded scores
  if (tid == 0 && n_over > 0 && c.status != nullptr) atomicAdd(c.status, n_over);      // an error: the batch is incomplete
}

// Membership of the batch's notes in the metrical nodes (beats, measures) of their subgraph — agnn_sample_members.
// Notes are sorted by onset, so the groups a window's target notes belong to form ONE contiguous id range
// [group_of[w], group_of[w + T - 1]]: no set, no sort.  Thread i: group slot i (i < n_sub * cap_g) and note slot i (i < n_nodes).
struct MemberArgs {
  const int64_t* batch;      // optional: subgraph id per note slot (pool layout); NULL = the padded hop-block layout
  const int32_t* node_gid;
  const int32_t* group_of;
  const int32_t* win_start;
  int32_t* group_gid;
  int64_t* edges;
  int32_t* drops;
  int64_t n_nodes;
  int32_t n_sub, n_targets, n_hops, cap_g;
  int32_t cap[AGNN_SAMPLER_MAX_HOPS];
};

__global__ __launch_bounds__(256) void k_sample_members(MemberArgs a) {
  const int64_t idx = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (idx < static_cast<int64_t>(a.n_sub) * a.cap_g) {
    const int s = static_cast<int>(idx / a.cap_g), j = static_cast<int>(idx - static_cast<int64_t>(s) * a.cap_g);
    const int w = a.win_start[s];
    const int gmin = a.group_of[w], gmax = a.group_of[w + a.n_targets - 1];
    a.group_gid[idx] = gmin + j <= gmax ? gmin + j : -1;
    if (j == 0 && gmax - gmin + 1 > a.cap_g && a.drops != nullptr) atomicAdd(a.drops, gmax - gmin + 1 - a.cap_g);
  }
  if (idx < a.n_nodes) {
    int s;
    int64_t base = static_cast<int64_t>(a.n_sub) * a.n_targets;
    if (a.batch != nullptr) {
      s = static_cast<int>(a.batch[idx]);
    } else if (idx < base) {
      s = static_cast<int>(idx / a.n_targets);
    } else {
      s = 0;
      for (int h = 0; h < a.n_hops; ++h) {
        const int64_t blk = static_cast<int64_t>(a.n_sub) * a.cap[h];
        if (idx < base + blk || h == a.n_hops - 1) { s = static_cast<int>((idx - base) / a.cap[h]); break; }
        base += blk;
      }
    }
    const int g = a.node_gid[idx];
    int64_t src = -1, dst = -1;
    if (g >= 0) {
      const int w = a.win_start[s];
      const int gmin = a.group_of[w], gmax = a.group_of[w + a.n_targets - 1];
      const int gg = a.group_of[g];
      if (gg >= gmin && gg <= gmax && gg - gmin < a.cap_g) {
        src = idx;
        dst = static_cast<int64_t>(s) * a.cap_g + (gg - gmin);
      }
    }
    a.edges[idx] = src;
    a.edges[a.n_nodes + idx] = dst;
  }
}

// ---- agnn_sample_compact: the padded hop blocks [n_sub x cap[h]] of a sampled batch squeezed into pools of pool[h] slots --------
// Subgraph s's kept nodes of hop h move to  pool_base[h] + (sum of kept[h][s'] over s' < s) + i  — subgraph order, rank order:
// still hop-ordered, a subgraph's nodes still contiguous.  Every workgroup scans the kept counts itself (n_sub x n_hops small
// integers); then a grid-stride pass moves node ids, writes the subgraph id of every pool slot and rewrites BOTH endpoints of
// every edge slot in place (an edge whose endpoint fell past the pool's end becomes (-1, -1)).
struct CompactArgs {
  const int32_t* kept;       // [n_hops][n_sub]
  const int32_t* gid_in;     // padded layout
  int32_t* gid_out;          // pool layout
  int64_t* batch_out;        // [n_out]
  int64_t* edges[AGNN_SAMPLER_MAX_REL];
  int32_t* drops;
  int64_t e_cap;
  int32_t n_rel, n_sub, n_targets, n_hops;
  int32_t cap[AGNN_SAMPLER_MAX_HOPS], pool[AGNN_SAMPLER_MAX_HOPS];
};

constexpr int kCompactMaxSub = 1024;   // n_sub * n_hops offsets in LDS

__global__ __launch_bounds__(256) void k_sample_compact(CompactArgs a) {
  __shared__ int s_off[kCompactMaxSub];      // exclusive prefix of min(kept, room) per hop
  __shared__ int s_keep[kCompactMaxSub];     // nodes of (h, s) that found room in the pool
  __shared__ int s_tot[AGNN_SAMPLER_MAX_HOPS];
  const int B = a.n_sub, T = a.n_targets;
  if (threadIdx.x < a.n_hops) {
    const int h = threadIdx.x;
    int run = 0, lost = 0;
    for (int s = 0; s < B; ++s) {
      const int k = a.kept[h * B + s];
      const int room = a.pool[h] - run;
      const int take = k < room ? k : (room > 0 ? room : 0);
      s_off[h * B + s] = run;
      s_keep[h * B + s] = take;
      run += take;
      lost += k - take;
    }
    s_tot[h] = run;
    if (lost > 0 && blockIdx.x == 0 && a.drops != nullptr) atomicAdd(a.drops, lost);
  }
  __syncthreads();
  const int64_t n_tgt = static_cast<int64_t>(B) * T;
  int64_t n_in = n_tgt, n_out = n_tgt;
  for (int h = 0; h < a.n_hops; ++h) { n_in += static_cast<int64_t>(B) * a.cap[h]; n_out += a.pool[h]; }
  auto remap = [&](int64_t id) -> int64_t {
    if (id < n_tgt) return id;                       // targets (and -1) stay
    int64_t base = n_tgt, pbase = n_tgt;
    for (int h = 0; h < a.n_hops; ++h) {
      const int64_t blk = static_cast<int64_t>(B) * a.cap[h];
      if (id < base + blk) {
        const int s = static_cast<int>((id - base) / a.cap[h]), i = static_cast<int>((id - base) - static_cast<int64_t>(s) * a.cap[h]);
        return i < s_keep[h * B + s] ? pbase + s_off[h * B + s] + i : -1;
      }
      base += blk;
      pbase += a.pool[h];
    }
    return -1;
  };
  const int64_t stride = static_cast<int64_t>(gridDim.x) * 256;
  const int64_t t0 = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  // nodes: targets copy through; pool slots default to padding, then the kept nodes land on theirs
  for (int64_t j = t0; j < n_out; j += stride) {
    if (j < n_tgt) {
      a.gid_out[j] = a.gid_in[j];
      a.batch_out[j] = j / T;
    } else {
      int64_t pbase = n_tgt;
      int h = 0;
      while (h + 1 < a.n_hops && j >= pbase + a.pool[h]) { pbase += a.pool[h]; ++h; }
      if (j - pbase >= s_tot[h]) { a.gid_out[j] = -1; a.batch_out[j] = B - 1; }     // unused pool slot: no features, no edges
    }
  }
  for (int64_t j = t0 + n_tgt; j < n_in; j += stride) {
    const int64_t nj = remap(j);
    if (nj >= 0) {
      int64_t base = n_tgt;
      int h = 0;
      while (j >= base + static_cast<int64_t>(B) * a.cap[h]) { base += static_cast<int64_t>(B) * a.cap[h]; ++h; }
      a.gid_out[nj] = a.gid_in[j];
      a.batch_out[nj] = (j - base) / a.cap[h];
    }
  }
  for (int r = 0; r < a.n_rel; ++r) {
    int64_t* e0 = a.edges[r];
    int64_t* e1 = e0 + a.e_cap;
    for (int64_t e = t0; e < a.e_cap; e += stride) {
      const int64_t s0 = e0[e], d0 = e1[e];
      if (s0 < n_tgt && d0 < n_tgt) continue;          // both ends targets or the slot is empty: nothing moves
      const int64_t s1 = remap(s0), d1 = remap(d0);
      const bool ok = s1 >= 0 && d1 >= 0;
      e0[e] = ok ? s1 : -1;
      e1[e] = ok ? d1 : -1;
    }
  }
}

// out[i, :] = gid[i] >= 0 ? src[gid[i], :] : 0   (float rows, H % 4 == 0; one 16-lane group per row)
__global__ __launch_bounds__(256) void k_gather_rows(const float* __restrict__ src, int64_t ld_src, const int32_t* __restrict__ gid,
                                                     int64_t n, int32_t H4, float* __restrict__ out, int64_t ld_out) {
  const int64_t total = n * H4;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; e < total; e += static_cast<int64_t>(gridDim.x) * 256) {
    const int64_t i = e / H4;
    const int c4 = static_cast<int>(e - i * H4);
    const int g = gid[i];
    const float4 v = g >= 0 ? *reinterpret_cast<const float4*>(src + static_cast<int64_t>(g) * ld_src + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
    *reinterpret_cast<float4*>(out + i * ld_out + 4 * c4) = v;
  }
}

// out[t, i] = gid[i] >= 0 ? src[t * ld_src + gid[i]] : fill      (T int64 attribute vectors: labels, spelling, key)
__global__ __launch_bounds__(256) void k_gather_i64(const int64_t* __restrict__ src, int64_t ld_src, const int32_t* __restrict__ gid,
                                                    int64_t n, int32_t T, int64_t fill, int64_t* __restrict__ out, int64_t ld_out) {
  const int64_t total = n * T;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; e < total; e += static_cast<int64_t>(gridDim.x) * 256) {
    const int64_t t = e / n, i = e - t * n;
    const int g = gid[i];
    out[t * ld_out + i] = g >= 0 ? src[t * ld_src + g] : fill;
  }
}

}  // namespace

extern "C" int64_t agnn_sampler_num_nodes(const agnn_sampler_t* c) {
  if (!c) return -1;
  int64_t n = static_cast<int64_t>(c->n_sub) * c->n_targets;
  for (int h = 0; h < c->n_hops; ++h) n += static_cast<int64_t>(c->n_sub) * c->cap[h];
  return n;
}

extern "C" int64_t agnn_sampler_edge_capacity(const agnn_sampler_t* c) {
  if (!c) return -1;
  int64_t e = 0, F = c->n_targets;
  for (int h = 0; h < c->n_hops; ++h) {
    e += static_cast<int64_t>(c->n_sub) * F * c->fan[h];
    F = c->cap[h];
  }
  return e;
}

extern "C" int agnn_sample_hops(const agnn_sampler_t* cfg, agnn_stream_t stream_) {
  using namespace agnn;
  if (!cfg) return fail(AGNN_EINVAL, "sample_hops: null config");
  const agnn_sampler_t& c = *cfg;
  if (c.n_rel <= 0 || c.n_rel > AGNN_SAMPLER_MAX_REL) return fail(AGNN_EINVAL, "sample_hops: n_rel=%d not in [1,%d]", c.n_rel, AGNN_SAMPLER_MAX_REL);
  if (c.n_hops <= 0 || c.n_hops > AGNN_SAMPLER_MAX_HOPS) return fail(AGNN_EINVAL, "sample_hops: n_hops=%d not in [1,%d]", c.n_hops, AGNN_SAMPLER_MAX_HOPS);
  if (c.n_sub <= 0 || c.n_targets <= 0) return fail(AGNN_EINVAL, "sample_hops: n_sub=%d n_targets=%d", c.n_sub, c.n_targets);
  int cap_sum = 0;
  for (int h = 0; h < c.n_hops; ++h) {
    if (c.fan[h] <= 0 || c.fan[h] > AGNN_SAMPLER_MAX_FAN) return fail(AGNN_EINVAL, "sample_hops: fan[%d]=%d not in [1,%d]", h, c.fan[h], AGNN_SAMPLER_MAX_FAN);
    if (c.cap[h] <= 0 || c.cap[h] > AGNN_SAMPLER_MAX_CAP) return fail(AGNN_EINVAL, "sample_hops: cap[%d]=%d not in [1,%d]", h, c.cap[h], AGNN_SAMPLER_MAX_CAP);
    cap_sum += c.cap[h];
  }
  if (cap_sum > kHash / 4) return fail(AGNN_EINVAL, "sample_hops: sum of capacities %d > %d", cap_sum, kHash / 4);
  if (!c.win_start || !c.rng || !c.node_gid) return fail(AGNN_EINVAL, "sample_hops: null argument");
  if (c.e_cap != agnn_sampler_edge_capacity(cfg)) return fail(AGNN_EINVAL, "sample_hops: e_cap=%lld, expected %lld", (long long)c.e_cap, (long long)agnn_sampler_edge_capacity(cfg));
  for (int r = 0; r < c.n_rel; ++r)
    if (!c.rowptr[r] || !c.col[r] || !c.edges[r]) return fail(AGNN_EINVAL, "sample_hops: relation %d incomplete", r);
  SamplerArgs A{c};
  hipLaunchKernelGGL(k_sample_hops, dim3(static_cast<unsigned>(c.n_sub)), dim3(kThreads), 0, static_cast<hipStream_t>(stream_), A);
  return check_launch("sample_hops");
}

extern "C" int64_t agnn_sample_compact_nodes(const agnn_sampler_t* c, const int32_t* pool) {
  if (!c || !pool) return -1;
  int64_t n = static_cast<int64_t>(c->n_sub) * c->n_targets;
  for (int h = 0; h < c->n_hops; ++h) n += pool[h];
  return n;
}

extern "C" int agnn_sample_compact(const agnn_sampler_t* cfg, const int32_t* pool, int32_t* node_gid_out, int64_t* batch_out,
                                   agnn_stream_t stream_) {
  using namespace agnn;
  if (!cfg || !pool || !node_gid_out || !batch_out) return fail(AGNN_EINVAL, "sample_compact: null argument");
  const agnn_sampler_t& c = *cfg;
  if (!c.kept || !c.node_gid) return fail(AGNN_EINVAL, "sample_compact: the sampler configuration must carry `kept` (agnn_sample_hops fills it)");
  if (c.n_hops <= 0 || c.n_hops > AGNN_SAMPLER_MAX_HOPS || c.n_rel <= 0 || c.n_rel > AGNN_SAMPLER_MAX_REL || c.n_sub <= 0 ||
      static_cast<int64_t>(c.n_sub) * c.n_hops > kCompactMaxSub)
    return fail(AGNN_EINVAL, "sample_compact: n_sub=%d x n_hops=%d (at most %d counters)", c.n_sub, c.n_hops, kCompactMaxSub);
  CompactArgs a{};
  a.kept = c.kept; a.gid_in = c.node_gid; a.gid_out = node_gid_out; a.batch_out = batch_out; a.drops = c.drops; a.e_cap = c.e_cap;
  a.n_rel = c.n_rel; a.n_sub = c.n_sub; a.n_targets = c.n_targets; a.n_hops = c.n_hops;
  for (int h = 0; h < c.n_hops; ++h) {
    if (pool[h] <= 0 || pool[h] > static_cast<int64_t>(c.n_sub) * c.cap[h]) return fail(AGNN_EINVAL, "sample_compact: pool[%d]=%d not in [1, n_sub * cap]", h, pool[h]);
    a.cap[h] = c.cap[h];
    a.pool[h] = pool[h];
  }
  for (int r = 0; r < c.n_rel; ++r) a.edges[r] = c.edges[r];
  const int64_t work = c.e_cap > agnn_sampler_num_nodes(cfg) ? c.e_cap : agnn_sampler_num_nodes(cfg);
  int64_t blocks = (work + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(k_sample_compact, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, static_cast<hipStream_t>(stream_), a);
  return check_launch("sample_compact");
}

extern "C" int agnn_sample_members(const int32_t* node_gid, int64_t n_nodes, const int64_t* batch, const int32_t* group_of,
                                   const int32_t* win_start, int32_t n_sub, int32_t n_targets, int32_t n_hops, const int32_t* cap,
                                   int32_t cap_g, int32_t* group_gid, int64_t* edges, int32_t* drops, agnn_stream_t stream_) {
  using namespace agnn;
  if (n_sub <= 0 || n_targets <= 0 || cap_g <= 0 || n_hops < 0 || n_hops > AGNN_SAMPLER_MAX_HOPS || (n_hops > 0 && !cap))
    return fail(AGNN_EINVAL, "sample_members: n_sub=%d n_targets=%d n_hops=%d cap_g=%d", n_sub, n_targets, n_hops, cap_g);
  int64_t expect = static_cast<int64_t>(n_sub) * n_targets;
  MemberArgs a{};
  for (int h = 0; h < n_hops; ++h) {
    if (cap[h] <= 0) return fail(AGNN_EINVAL, "sample_members: cap[%d]=%d", h, cap[h]);
    a.cap[h] = cap[h];
    expect += static_cast<int64_t>(n_sub) * cap[h];
  }
  if (batch == nullptr && n_nodes != expect) return fail(AGNN_EINVAL, "sample_members: n_nodes=%lld, the hop layout has %lld slots", (long long)n_nodes, (long long)expect);
  if (batch != nullptr && (n_nodes < static_cast<int64_t>(n_sub) * n_targets || n_nodes > expect)) return fail(AGNN_EINVAL, "sample_members: n_nodes=%lld outside the pool layout's range", (long long)n_nodes);
  if (!node_gid || !group_of || !win_start || !group_gid || !edges) return fail(AGNN_EINVAL, "sample_members: null argument");
  a.batch = batch; a.node_gid = node_gid; a.group_of = group_of; a.win_start = win_start; a.group_gid = group_gid; a.edges = edges; a.drops = drops;
  a.n_nodes = n_nodes; a.n_sub = n_sub; a.n_targets = n_targets; a.n_hops = n_hops; a.cap_g = cap_g;
  const int64_t work = n_nodes > static_cast<int64_t>(n_sub) * cap_g ? n_nodes : static_cast<int64_t>(n_sub) * cap_g;
  hipLaunchKernelGGL(k_sample_members, dim3(static_cast<unsigned>((work + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream_), a);
  return check_launch("sample_members");
}

extern "C" int agnn_gather_rows_f32(const float* src, int64_t ld_src, const int32_t* gid, int64_t n, int32_t H, float* out, int64_t ld_out,
                                    agnn_stream_t stream_) {
  using namespace agnn;
  if (n < 0 || H <= 0 || (H & 3)) return fail(AGNN_EINVAL, "gather_rows: n=%lld H=%d (H must be a multiple of 4)", (long long)n, H);
  if (n == 0) return AGNN_OK;
  if (!src || !gid || !out) return fail(AGNN_EINVAL, "gather_rows: null argument");
  if (!aligned16(src) || !aligned16(out) || (ld_src & 3) || (ld_out & 3) || ld_src < H || ld_out < H) return fail(AGNN_EALIGN, "gather_rows: rows must be 16-byte aligned");
  int64_t blocks = (n * (H / 4) + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(k_gather_rows, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, static_cast<hipStream_t>(stream_), src, ld_src, gid, n, H / 4,
                     out, ld_out);
  return check_launch("gather_rows");
}

extern "C" int agnn_gather_i64(const int64_t* src, int64_t ld_src, const int32_t* gid, int64_t n, int32_t n_vec, int64_t fill, int64_t* out,
                               int64_t ld_out, agnn_stream_t stream_) {
  using namespace agnn;
  if (n < 0 || n_vec <= 0) return fail(AGNN_EINVAL, "gather_i64: n=%lld n_vec=%d", (long long)n, n_vec);
  if (n == 0) return AGNN_OK;
  if (!src || !gid || !out || ld_out < n) return fail(AGNN_EINVAL, "gather_i64: null argument or ld_out < n");
  int64_t blocks = (n * n_vec + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(k_gather_i64, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, static_cast<hipStream_t>(stream_), src, ld_src, gid, n, n_vec, fill,
                     out, ld_out);
  return check_launch("gather_i64");
}
