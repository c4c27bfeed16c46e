// voxel_scatter.hip -- points -> voxels (InputLayer, modes 1..4) on gfx950: three kernels and one fill.
//
// Replaces Metadata<3>::inputLayer -> inputLayerRules (SCN/Metadata/Metadata.cpp:405-417,
// SCN/Metadata/IOLayersRules.h:18-125: serial host hash inserts, first-seen numbering `sg.mp[p] = nActive++`
// :86-91, V x (1+maxActive) rule table :112-124) and InputLayer_ForwardPass / _BackwardPass
// (SCN/CPU/IOLayers.cpp:11-47; CUDA twin SCN/CUDA/IOLayers.cu:14-70).
//
//   fill      keys | first | vals | meta  <- 0xFF                            (one memset; meta counts up from -1)
//   K1 k_voxel_insert : point i -> hash slot (one 64-bit CAS) ; first[slot] = min point index (one atomicMin)
//   K2 k_voxel_number : ONE pass over the points in chunks of 1024: flag "I am my voxel's first point",
//                       chunk scan in LDS, decoupled look-back across chunks (wave-parallel, 64 predecessors per
//                       probe) => first-seen site number, exactly the reference's order, with no serial pass and
//                       no second launch; first points write vals[slot], the site's coordinates and first_pt;
//                       the other points of a voxel (rare: N/V ~ 1.03-1.6) pick their site up from vals[slot]
//                       and push themselves on the site's chain (one atomicExch) and count (one atomicAdd)
//   K3 k_voxel_mean   : out[site] = mean / sum / first / last of its points' features, bit-identical to
//                       InputLayer_ForwardPass: multiply and add kept separate, points in ascending index order
//                       (single-point sites -- the bulk -- read one row; chains are walked and ordered in registers)
// Per point the HBM side sees: 32 B of coordinates (K1) + 4 B slot (K1 write, K2 read) + 32 B of coordinates again
// for first points (K2) + 4 B point_site + the feature row in and out (K3), and three random 4-8 B sectors in the
// hash / first / vals arrays.  No CSR build, no per-site sort pass, no slot-table scan.
//
// Why not a coarse-cell partition with per-bin LDS hashes: on this data the dedup ratio is ~1 (80k points ->
// 77k voxels at 2 cm), so an LDS pre-aggregation removes almost no global inserts, and the partition itself
// (histogram + scan + stable scatter of 44-byte records) costs more passes over the points than the whole of the
// above.  LDS is used where it pays: the chunk scan / look-back of K2.
#include "common.h"
#include <stdlib.h>

namespace aabr {

constexpr int kVsThreads = 256;
// points per K2 workgroup = 256 x 4.  All chunks are resident at once on this chip, so a chunk walks back to chunk 0
// (chunk/64 probes of relaxed polls); measured 80k .. 1.5 M points, 1024-point chunks beat 4096-point ones at every
// size (more workgroups for the dependent first[slot] reads; the look-back is not the cost).
static inline int vs_items(int64_t) { return 4; }
constexpr int kMetaTicket = 4;                  // meta word used as the chunk ticket counter

__global__ __launch_bounds__(256) void k_voxel_insert(const int64_t *__restrict__ coords, int64_t n, int ncols,
                                                      GridEnt *keys, uint64_t mask,
                                                      int32_t *__restrict__ slot, int32_t *__restrict__ cnt_extra,
                                                      int32_t *__restrict__ head,
                                                      unsigned long long *__restrict__ status, int64_t nchunks,
                                                      int32_t *meta) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nchunks) status[i] = 0ull;
  if (i >= n) return;
  cnt_extra[i] = 0;
  head[i] = -1;
  const int64_t *c = coords + i * ncols;
  int64_t x = c[0], y = c[1], z = c[2], b = ncols == 4 ? c[3] : 0;
  if (x == -1 && y == -1 && z == -1) { // dropped by aabr_quantize_points (outside FULL_SCALE): skip silently
    slot[i] = -1;
    return;
  }
  if (x < 0 || y < 0 || z < 0 || b < 0 || x > kMaxCoord || y > kMaxCoord || z > kMaxCoord || b > kMaxCoord) {
    slot[i] = -1;
    atomicAnd(&meta[2], 0); // meta starts at 0xFFFFFFFF (the one fill): 0 = a coordinate was out of range;
    return;                 // K2 rewrites the word as 0 / 1
  }
  uint32_t h = grid_insert(keys, mask, pack_key((int)b, (int)x, (int)y, (int)z));
  atomicMin(&keys[h].first, (uint32_t)i);   // same 64-byte sector as the CAS just done
  slot[i] = (int32_t)h;
}

// status word of a chunk: bits 63..62 = 0 nothing yet / 1 chunk aggregate / 2 inclusive prefix; low 32 bits = value
__device__ inline unsigned long long st_pack(unsigned flag, unsigned v) {
  return ((unsigned long long)flag << 62) | (unsigned long long)v;
}

template <int kVsItems>
__global__ __launch_bounds__(kVsThreads) void k_voxel_number(
    const int64_t *__restrict__ coords, int64_t n, int ncols, const int32_t *__restrict__ slot,
    GridEnt *grid, int32_t *__restrict__ site_coords,
    int32_t *__restrict__ first_pt, int32_t *__restrict__ point_site, int32_t *cnt_extra, int32_t *head,
    int32_t *__restrict__ nxt, unsigned long long *status, int32_t *meta) {
  constexpr int kVsChunk = kVsThreads * kVsItems;
  __shared__ int s_chunk;
  __shared__ int s_wsum[kVsThreads / 64];
  __shared__ unsigned s_excl;
  if (threadIdx.x == 0) s_chunk = atomicAdd(&meta[kMetaTicket], 1) + 1; // (counter starts at -1) chunks start in
  __syncthreads();                           // ticket order: a chunk only ever waits for chunks that already run
  const int chunk = s_chunk;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t base = (int64_t)chunk * kVsChunk + (int64_t)threadIdx.x * kVsItems;
  int s[kVsItems], f[kVsItems], a = 0;
#pragma unroll
  for (int j = 0; j < kVsItems; ++j) {
    const int64_t i = base + j;
    s[j] = i < n ? slot[i] : -1;
    f[j] = (s[j] >= 0 && grid[s[j]].first == (uint32_t)i) ? 1 : 0;
    a += f[j];
  }
  // chunk-exclusive scan of `a`: wave scan + 4 wave totals through LDS
  int incl = a;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    int o = __shfl_up(incl, d);
    if (lane >= d) incl += o;
  }
  if (lane == 63) s_wsum[wave] = incl;
  __syncthreads();
  int wpre = 0, total = 0;
#pragma unroll
  for (int w = 0; w < kVsThreads / 64; ++w) {
    if (w < wave) wpre += s_wsum[w];
    total += s_wsum[w];
  }
  const int ex_in_chunk = wpre + incl - a;
  // decoupled look-back (wave 0): publish the aggregate, then sum the predecessors' aggregates back to the nearest
  // inclusive prefix, 64 chunks per probe
  if (wave == 0) {
    if (lane == 0)
      __hip_atomic_store(&status[chunk], st_pack(chunk == 0 ? 2u : 1u, (unsigned)total), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    unsigned excl = 0;
    int j = chunk - 1;
    while (j >= 0) {
      const int idx = j - lane;
      unsigned long long st = st_pack(2u, 0u); // virtual chunks before 0: inclusive prefix 0
      if (idx >= 0) {
        do { // relaxed (cache-bypassing) polls; the values are self-contained, no other data is read through them
          st = __hip_atomic_load(&status[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if ((st >> 62) == 0ull) __builtin_amdgcn_s_sleep(1);
        } while ((st >> 62) == 0ull);
      }
      const unsigned long long m2 = __ballot((st >> 62) == 2ull);
      const int stop = m2 ? __ffsll((long long)m2) - 1 : 63; // nearest chunk with an inclusive prefix
      unsigned v = lane <= stop ? (unsigned)(st & 0xffffffffull) : 0u;
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
      excl += v;
      if (m2) break;
      j -= 64;
    }
    if (lane == 0) {
      if (chunk != 0)
        __hip_atomic_store(&status[chunk], st_pack(2u, excl + (unsigned)total), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      s_excl = excl;
      if ((int64_t)(chunk + 1) * kVsChunk >= n) {
        meta[0] = (int32_t)(excl + (unsigned)total);   // V
        meta[2] = meta[2] == 0 ? 1 : 0;                // error flag in the documented form (K1 has finished)
      }
    }
  }
  __syncthreads();
  int site_next = (int)s_excl + ex_in_chunk;
  int mysite[kVsItems];
#pragma unroll
  for (int j = 0; j < kVsItems; ++j) {
    mysite[j] = -1;
    if (f[j]) {
      const int64_t i = base + j;
      const int v = site_next++;
      mysite[j] = v;
      const int64_t *cp = coords + i * ncols;
      *reinterpret_cast<int4 *>(site_coords + 4 * (int64_t)v) =
          make_int4((int)cp[0], (int)cp[1], (int)cp[2], ncols == 4 ? (int)cp[3] : 0);
      first_pt[v] = (int32_t)i;
      __hip_atomic_store(&grid[s[j]].val, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // self-contained value
    }
  }
  // the other points of a voxel: their first point lies in this or an earlier (already running) chunk
#pragma unroll
  for (int j = 0; j < kVsItems; ++j) {
    const int64_t i = base + j;
    if (i >= n) continue;
    int v = mysite[j];
    if (!f[j] && s[j] >= 0) {
      do {
        v = __hip_atomic_load(&grid[s[j]].val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v < 0) __builtin_amdgcn_s_sleep(1);
      } while (v < 0);
      nxt[i] = atomicExch(&head[v], (int32_t)i);
      atomicAdd(&cnt_extra[v], 1);
    }
    point_site[i] = v;
  }
}

// points of site v in ascending index order: first_pt[v], then the chain (unordered) sorted in registers.
// Calls `fn(point)` for each, in order.  Chains longer than kChainRegs are walked by repeated selection.
constexpr int kChainRegs = 8;
template <typename F>
__device__ inline void for_points_ascending(int first, int extra, int hd, const int32_t *__restrict__ nxt, F fn) {
  fn(first);
  if (extra == 0) return;
  if (extra == 1) { fn(hd); return; }       // the common multi-point case: the chain point is later than `first`
  if (extra == 2) {
    const int p1 = nxt[hd];
    fn(hd < p1 ? hd : p1);
    fn(hd < p1 ? p1 : hd);
    return;
  }
  if (extra <= kChainRegs) {
    int idx[kChainRegs];
    int p = hd;
#pragma unroll
    for (int q = 0; q < kChainRegs; ++q) {
      idx[q] = 0x7fffffff;
      if (q < extra) { idx[q] = p; p = nxt[p]; }
    }
#pragma unroll
    for (int a = 1; a < kChainRegs; ++a) { // insertion sort, fully unrolled on registers
#pragma unroll
      for (int b = a; b > 0; --b) {
        const int lo = idx[b - 1] < idx[b] ? idx[b - 1] : idx[b];
        const int hi = idx[b - 1] < idx[b] ? idx[b] : idx[b - 1];
        idx[b - 1] = lo; idx[b] = hi;
      }
    }
#pragma unroll
    for (int q = 0; q < kChainRegs; ++q)
      if (q < extra) fn(idx[q]);
  } else {
    int last = first;
    for (int t = 0; t < extra; ++t) { // next larger index each round
      int best = 0x7fffffff;
      for (int p = hd; p >= 0; p = nxt[p])
        if (p > last && p < best) best = p;
      fn(best);
      last = best;
    }
  }
}

// out[v] = sum_j mult * in[pts[j]] in ascending point order, separate multiply and add (no FMA contraction) so
// the fp32 result is bit-identical to InputLayer_ForwardPass (CPU/IOLayers.cpp:18-27:
// `out_f[plane] += multiplier * in_f[plane]`).  Also records the site's last point and the largest point count.
__global__ __launch_bounds__(256) void k_voxel_mean(const float *__restrict__ in, float *__restrict__ out, int64_t V,
                                                    int planes, const int32_t *__restrict__ first_pt,
                                                    const int32_t *__restrict__ cnt_extra,
                                                    const int32_t *__restrict__ head,
                                                    const int32_t *__restrict__ nxt, int32_t *__restrict__ last_pt,
                                                    int mode, int32_t *meta) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int extra = 0;
  if (idx < V * planes) {
    const int64_t v = idx / planes;
    const int p = (int)(idx - v * planes);
    const int first = first_pt[v];
    extra = cnt_extra[v];
    float acc = 0.0f;
    if (extra == 0) {
      acc = __fadd_rn(0.0f, __fmul_rn(1.0f, in[(int64_t)first * planes + p]));
      if (p == 0 && last_pt) last_pt[v] = first;
    } else {
      const int hd = head[v];
      const float mult = mode == 4 ? __fdiv_rn(1.0f, (float)(extra + 1)) : 1.0f;
      int last = first;
      if (mode == 1) {
        acc = __fadd_rn(0.0f, in[(int64_t)first * planes + p]);
        for (int q = hd; q >= 0; q = nxt[q]) last = q > last ? q : last;
      } else {
        for_points_ascending(first, extra, hd, nxt, [&](int pt) {
          last = pt;
          if (mode != 2) acc = __fadd_rn(acc, __fmul_rn(mult, in[(int64_t)pt * planes + p]));
        });
        if (mode == 2) acc = __fadd_rn(0.0f, in[(int64_t)last * planes + p]);
      }
      if (p == 0 && last_pt) last_pt[v] = last;
    }
    out[idx] = acc;
  }
  // largest point count of a site (IOLayersRules.h:96-103 maxActive): one atomic per block, only when it grows
  int m = extra + (idx < V * planes ? 1 : 0);
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const int o = __shfl_xor(m, d);
    m = o > m ? o : m;
  }
  __shared__ int wm[4];
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = wm[0];
    for (int w = 1; w < 4; ++w) m = wm[w] > m ? wm[w] : m;
    if (m > __builtin_nontemporal_load(&meta[1])) atomicMax(&meta[1], m);
  }
}

// Narrow rows (planes <= 16, the detector's 9): ONE thread per site walks the chain once and carries the whole row.
template <int MAXP>
__global__ __launch_bounds__(256) void k_voxel_mean_row(const float *__restrict__ in, float *__restrict__ out,
                                                        int64_t V, int planes, const int32_t *__restrict__ first_pt,
                                                        const int32_t *__restrict__ cnt_extra,
                                                        const int32_t *__restrict__ head,
                                                        const int32_t *__restrict__ nxt,
                                                        int32_t *__restrict__ last_pt, int mode, int32_t *meta) {
  const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int extra = -1;
  if (v < V) {
    const int first = first_pt[v];
    extra = cnt_extra[v];
    float acc[MAXP];
#pragma unroll
    for (int p = 0; p < MAXP; ++p) acc[p] = 0.0f;
    auto add_row = [&](int pt, float mult) {
      const float *r = in + (int64_t)pt * planes;
#pragma unroll
      for (int p = 0; p < MAXP; ++p)
        if (p < planes) acc[p] = __fadd_rn(acc[p], __fmul_rn(mult, r[p]));
    };
    int last = first;
    if (extra == 0) {
      add_row(first, 1.0f);
    } else {
      const int hd = head[v];
      const float mult = mode == 4 ? __fdiv_rn(1.0f, (float)(extra + 1)) : 1.0f;
      if (mode == 1) {
        add_row(first, 1.0f);
        for (int q = hd; q >= 0; q = nxt[q]) last = q > last ? q : last;
      } else if (mode == 2) {
        for (int q = hd; q >= 0; q = nxt[q]) last = q > last ? q : last;
        add_row(last, 1.0f);
      } else {
        for_points_ascending(first, extra, hd, nxt, [&](int pt) { last = pt; add_row(pt, mult); });
      }
    }
    if (last_pt) last_pt[v] = last;
    float *o = out + v * planes;
#pragma unroll
    for (int p = 0; p < MAXP; ++p)
      if (p < planes) o[p] = acc[p];
  }
  int m = extra + 1;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const int o = __shfl_xor(m, d);
    m = o > m ? o : m;
  }
  __shared__ int wm[4];
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = wm[0];
    for (int w = 1; w < 4; ++w) m = wm[w] > m ? wm[w] : m;
    if (m > __builtin_nontemporal_load(&meta[1])) atomicMax(&meta[1], m);
  }
}

// d_in[i] = mult * d_out[site(i)] for the points that contributed (InputLayer_BackwardPass, CPU/IOLayers.cpp:30-47)
__global__ __launch_bounds__(256) void k_voxel_backward(float *__restrict__ d_in, const float *__restrict__ d_out,
                                                        int64_t n, int planes, const int32_t *__restrict__ point_site,
                                                        const int32_t *__restrict__ first_pt,
                                                        const int32_t *__restrict__ last_pt,
                                                        const int32_t *__restrict__ cnt_extra, int mode) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * planes) return;
  const int64_t i = idx / planes;
  const int p = (int)(idx - i * planes);
  const int v = point_site[i];
  float g = 0.0f;
  if (v >= 0) {
    bool take = true;
    if (mode == 1) take = (first_pt[v] == (int32_t)i);
    if (mode == 2) take = (last_pt[v] == (int32_t)i);
    if (take) {
      const float mult = mode == 4 ? __fdiv_rn(1.0f, (float)(cnt_extra[v] + 1)) : 1.0f;
      g = __fadd_rn(0.0f, __fmul_rn(mult, d_out[(int64_t)v * planes + p]));
    }
  }
  d_in[idx] = g;
}

// reference-format rule table rules[1] (IOLayersRules.h:112-124): [V, 1+maxActive] = (count, ascending points..)
__global__ __launch_bounds__(256) void k_voxel_rule_table(const int32_t *__restrict__ first_pt,
                                                          const int32_t *__restrict__ last_pt,
                                                          const int32_t *__restrict__ cnt_extra,
                                                          const int32_t *__restrict__ head,
                                                          const int32_t *__restrict__ nxt, int64_t V, int width,
                                                          int mode, int32_t *__restrict__ rules) {
  int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V) return;
  int32_t *r = rules + v * width;
  if (mode == 3 || mode == 4) {
    const int extra = cnt_extra[v];
    r[0] = extra + 1;
    int j = 1;
    for_points_ascending(first_pt[v], extra, head[v], nxt, [&](int pt) { if (j < width) r[j++] = pt; });
    for (; j < width; ++j) r[j] = 0;
  } else {
    r[0] = 1;
    r[1] = mode == 1 ? first_pt[v] : last_pt[v];
  }
}

static inline dim3 grid1(int64_t n, int bs) { return dim3((unsigned)ceil_div(n > 0 ? n : 1, bs)); }

} // namespace aabr
using namespace aabr;

extern "C" int64_t aabr_input_layer_status_words(int64_t n) {
  return 2 * ceil_div(n > 0 ? n : 1, (int64_t)kVsThreads * vs_items(n));
}

extern "C" int aabr_input_layer_sites(const int64_t *coords, int64_t n, int ncols, uint64_t *keys, uint32_t *first,
                                      int32_t *vals, int64_t cap, int32_t *slot, int32_t *point_site,
                                      int32_t *site_coords, int32_t *first_pt, int32_t *cnt_extra, int32_t *head,
                                      int32_t *nxt, int32_t *status, int32_t *meta, void *stream_) {
  hipStream_t st = (hipStream_t)stream_;
  AABR_CHECK_ARG(n >= 0 && n < (1ll << 31) - 65536 && (ncols == 3 || ncols == 4), "0 <= n < 2^31, ncols in {3,4}");
  AABR_CHECK_ARG(is_pow2(cap) && cap >= 2 * n && cap >= 64, "cap must be a power of two >= max(64, 2n)");
  AABR_CHECK_ARG(keys && slot && point_site && site_coords && first_pt && cnt_extra && head && nxt && status && meta,
                 "null pointer");
  AABR_CHECK_ARG(((uintptr_t)status & 7) == 0 && ((uintptr_t)site_coords & 15) == 0 && ((uintptr_t)keys & 15) == 0,
                 "status 8-, site_coords / grid entries 16-byte aligned");
  AABR_CHECK_ARG(n == 0 || coords, "null coords");
  (void)first; (void)vals;   // round-2 layout (separate arrays); the grid's 16-byte entries hold them now
  GridEnt *grid = reinterpret_cast<GridEnt *>(keys);
  const int items = vs_items(n);
  const int64_t nchunks = ceil_div(n > 0 ? n : 1, (int64_t)kVsThreads * items);
  if (n == 0) {
    hipMemsetAsync(grid, 0xFF, cap * sizeof(GridEnt), st);
    hipMemsetAsync(meta, 0, AABR_META_WORDS * sizeof(int32_t), st);
    AABR_CHECK_LAUNCH();
    return AABR_OK;
  }
  if ((void *)(grid + cap) == (void *)meta) // contiguous (the layout the Python side uses): ONE fill
    hipMemsetAsync(grid, 0xFF, cap * sizeof(GridEnt) + AABR_META_WORDS * sizeof(int32_t), st);
  else {
    hipMemsetAsync(grid, 0xFF, cap * sizeof(GridEnt), st);
    hipMemsetAsync(meta, 0xFF, AABR_META_WORDS * sizeof(int32_t), st);
  }
  // K1 also clears the chunk status words and the per-site chain heads / counts (consumed by K2 only)
  hipLaunchKernelGGL(k_voxel_insert, grid1(n > nchunks ? n : nchunks, 256), dim3(256), 0, st, coords, n, ncols, grid,
                     (uint64_t)(cap - 1), slot, cnt_extra, head, (unsigned long long *)status, nchunks, meta);
  hipLaunchKernelGGL(k_voxel_number<4>, dim3((unsigned)nchunks), dim3(kVsThreads), 0, st, coords, n, ncols, slot,
                     grid, site_coords, first_pt, point_site, cnt_extra, head, nxt,
                     (unsigned long long *)status, meta);
  AABR_CHECK_LAUNCH();
  return AABR_OK;
}

extern "C" int aabr_input_layer_forward(const float *in_feats, float *out_feats, int64_t V, int planes,
                                        const int32_t *first_pt, const int32_t *cnt_extra, const int32_t *head,
                                        const int32_t *nxt, int32_t *last_pt, int mode, int32_t *meta, void *stream_) {
  AABR_CHECK_ARG(V >= 0 && planes > 0 && mode >= 1 && mode <= 4, "bad V/planes/mode");
  if (V == 0) return AABR_OK;
  AABR_CHECK_ARG(in_feats && out_feats && first_pt && cnt_extra && head && nxt && meta, "null pointer");
  const int mean_knob = knob(K_VOXEL_MEAN);
  // one thread per (site, plane): consecutive lanes read consecutive floats of consecutive first points -- 73 us
  // instead of 100 us at 1.5 M points for the one-thread-per-site form (AABR_VOXEL_MEAN=2 selects that one)
  if (planes <= 16 && mean_knob == 2)
    hipLaunchKernelGGL(k_voxel_mean_row<16>, grid1(V, 256), dim3(256), 0, (hipStream_t)stream_, in_feats, out_feats, V,
                       planes, first_pt, cnt_extra, head, nxt, last_pt, mode, meta);
  else
    hipLaunchKernelGGL(k_voxel_mean, grid1(V * planes, 256), dim3(256), 0, (hipStream_t)stream_, in_feats, out_feats,
                       V, planes, first_pt, cnt_extra, head, nxt, last_pt, mode, meta);
  AABR_CHECK_LAUNCH();
  return AABR_OK;
}

extern "C" int aabr_input_layer_backward(float *d_in_feats, const float *d_out_feats, int64_t n, int planes,
                                         const int32_t *point_site, const int32_t *first_pt, const int32_t *last_pt,
                                         const int32_t *cnt_extra, int mode, void *stream_) {
  AABR_CHECK_ARG(n >= 0 && planes > 0 && mode >= 1 && mode <= 4, "bad n/planes/mode");
  if (n == 0) return AABR_OK;
  AABR_CHECK_ARG(d_in_feats && d_out_feats && point_site && first_pt && cnt_extra, "null pointer");
  AABR_CHECK_ARG(mode != 2 || last_pt, "mode 2 needs last_pt");
  hipLaunchKernelGGL(k_voxel_backward, grid1(n * planes, 256), dim3(256), 0, (hipStream_t)stream_, d_in_feats,
                     d_out_feats, n, planes, point_site, first_pt, last_pt, cnt_extra, mode);
  AABR_CHECK_LAUNCH();
  return AABR_OK;
}

extern "C" int aabr_input_layer_rule_table(const int32_t *first_pt, const int32_t *last_pt, const int32_t *cnt_extra,
                                           const int32_t *head, const int32_t *nxt, int64_t V, int max_active, int mode,
                                           int32_t *rules, void *stream_) {
  AABR_CHECK_ARG(V >= 0 && max_active >= 0 && mode >= 1 && mode <= 4, "bad V/max_active/mode");
  if (V == 0) return AABR_OK;
  AABR_CHECK_ARG(first_pt && cnt_extra && head && nxt && rules, "null pointer");
  AABR_CHECK_ARG(mode != 2 || last_pt, "mode 2 needs last_pt");
  const int width = ((mode == 3 || mode == 4) ? max_active : 1) + 1;
  hipLaunchKernelGGL(k_voxel_rule_table, grid1(V, 256), dim3(256), 0, (hipStream_t)stream_, first_pt, last_pt,
                     cnt_extra, head, nxt, V, width, mode, rules);
  AABR_CHECK_LAUNCH();
  return AABR_OK;
}
