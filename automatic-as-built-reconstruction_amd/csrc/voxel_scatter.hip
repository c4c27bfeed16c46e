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
// Round 4: three forms of the insert, one numbering kernel (K2) and one mean kernel behind them.
//   generic (any coordinates up to 65534, any batch index)  : K1 above, two device atomics per point;
//   packed  (aabr_input_layer_sites_packed, variant 1)      : ONE device atomic per point -- the 64-bit word
//       (batch | x | y | z | point index), field widths from the layer's spatial size and n, goes through a CAS into an
//       8-byte side table (same slot sequence as the grid), a point that finds its own voxel's word takes the minimum
//       with one atomicMin only when it is the smaller one; K2 reads the winner's index from the word and writes the
//       finished 16-byte grid entry;
//   binned  (variant 2, the default when the fields fit)    : LDS-staged hash binning with coalesced HBM writes, NO
//       device atomics on the table -- P1 k_voxel_bin partitions the points' packed words by hash BLOCK (4096 slots =
//       one 64 KiB window of the grid: LDS histogram per workgroup, one reservation per (workgroup, block), records
//       written in runs), P2 k_voxel_bin_build builds each block's table in LDS (ds_cmpst / ds_min on 64-bit words),
//       tells every point its slot and whether it is its voxel's first (one 4-byte store by point index into an array
//       that fits the L2s) and writes the block's 4096 finished grid entries, empty ones included, with 16-byte
//       coalesced stores -- the table fill disappears.  Blocks are self-contained because every grid probes inside
//       its home block (common.h grid_next).
// The packed / binned forms set meta[5] = 0 when a point does not fit the word (coordinate beyond the spatial size,
// batch index beyond the bits left) or a block overflows; the caller then runs the generic form.
#include "common.h"
#include <stdlib.h>

namespace aabr {

constexpr int kVsThreads = 256;
// points per K2 workgroup = 256 x 4.  All chunks are resident at once on this chip, so a chunk walks back to chunk 0
// (chunk/64 probes of relaxed polls); measured 80k .. 1.5 M points, 1024-point chunks beat 4096-point ones at every
// size (more workgroups for the dependent first[slot] reads; the look-back is not the cost).
static inline int vs_items(int64_t) { return 4; }
constexpr int kMetaTicket = 4;                  // meta word used as the chunk ticket counter
constexpr int kMetaExtent = 8;                  // meta[8..11]: largest x, y, z, batch index over the valid points

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

// MODE 0: generic (slot[i] + grid[slot].first); 1: packed side table (slot[i] + low bits of words[slot]; the first
// point writes the whole grid entry); 2: binned (slot[i] = slot | first << 31, written by k_voxel_bin_build)
constexpr int32_t kInfoNone = 0x7fffffff;
constexpr int32_t kInfoMulti = 0x40000000;      // info bit 30: the point's voxel holds more than one point
template <int kVsItems, int MODE>
__global__ __launch_bounds__(kVsThreads) void k_voxel_number(
    const int64_t *__restrict__ coords, int64_t n, int ncols, const int32_t *__restrict__ slot,
    GridEnt *grid, int32_t *__restrict__ site_coords,
    int32_t *__restrict__ first_pt, int32_t *__restrict__ point_site, int32_t *cnt_extra, int32_t *head,
    int32_t *__restrict__ nxt, unsigned long long *status, int32_t *meta,
    const unsigned long long *__restrict__ words, unsigned long long imask) {
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
    if (MODE == 2) {
      const int32_t v = i < n ? slot[i] : kInfoNone;
      s[j] = (v & kInfoNone) == kInfoNone ? -1 : (v & (kInfoMulti - 1));
      f[j] = (s[j] >= 0 && v < 0) ? ((v & kInfoMulti) ? 2 : 1) : 0;
    } else {
      s[j] = i < n ? slot[i] : -1;
      if (MODE == 1)
        f[j] = (s[j] >= 0 && (words[s[j]] & imask) == (unsigned long long)i) ? 1 : 0;
      else
        f[j] = (s[j] >= 0 && grid[s[j]].first == (uint32_t)i) ? 1 : 0;
    }
    a += f[j] ? 1 : 0;
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
  int ext[4] = {-1, -1, -1, -1};       // largest x, y, z, batch index of this thread's first points
#pragma unroll
  for (int j = 0; j < kVsItems; ++j) {
    mysite[j] = -1;
    if (f[j]) {
      const int64_t i = base + j;
      const int v = site_next++;
      mysite[j] = v;
      const int64_t *cp = coords + i * ncols;
      const int4 sc = make_int4((int)cp[0], (int)cp[1], (int)cp[2], ncols == 4 ? (int)cp[3] : 0);
      *reinterpret_cast<int4 *>(site_coords + 4 * (int64_t)v) = sc;
      ext[0] = sc.x > ext[0] ? sc.x : ext[0]; ext[1] = sc.y > ext[1] ? sc.y : ext[1];
      ext[2] = sc.z > ext[2] ? sc.z : ext[2]; ext[3] = sc.w > ext[3] ? sc.w : ext[3];
      first_pt[v] = (int32_t)i;
      if (MODE == 2 && f[j] == 2) {   // the voxel's chain was linked by k_voxel_bin_build: head and count by slot
        const unsigned long long e = words[s[j]];
        head[v] = (int32_t)(uint32_t)e;
        cnt_extra[v] = (int32_t)(e >> 32);
      }
      if (MODE == 1) {   // the grid entry is created here: key and first index first, the polled word last
        GridEnt *e = grid + s[j];
        e->key = pack_key(ncols == 4 ? (int)cp[3] : 0, (int)cp[0], (int)cp[1], (int)cp[2]);
        e->first = (uint32_t)i;
      }
      __hip_atomic_store(&grid[s[j]].val, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // self-contained value
    }
  }
  // the scene's extent (meta[8..11], counting up from the fill's -1): what a brick grid's directory is sized by.  One
  // atomic per CHUNK and axis at most, and none when the running maximum already covers it (one per wave was 20 k
  // same-address atomics at 320 k points: +35 us on this kernel)
  {
    __shared__ int s_ext[kVsThreads / 64][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      int m = ext[q];
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) {
        const int o = __shfl_xor(m, d);
        m = o > m ? o : m;
      }
      if (lane == 0) s_ext[wave][q] = m;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
      const int q = threadIdx.x;
      int m = s_ext[0][q];
#pragma unroll
      for (int w = 1; w < kVsThreads / 64; ++w) m = s_ext[w][q] > m ? s_ext[w][q] : m;
      if (m > __builtin_nontemporal_load(&meta[kMetaExtent + q])) atomicMax(&meta[kMetaExtent + q], m);
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
      if (MODE != 2) {
        nxt[i] = atomicExch(&head[v], (int32_t)i);
        atomicAdd(&cnt_extra[v], 1);
      }
    }
    point_site[i] = v;
  }
}

// ---- packed words --------------------------------------------------------------------------------------------------
// word = (((batch << xb | x) << yb | y) << zb | z) << ib | point index: for one voxel the smallest word is its first
// point.  All ones = empty (the batch field is never all ones).
struct PackSpec {
  int xb, yb, zb, bb, ib;
};
__device__ inline unsigned long long pk_word(const PackSpec &p, int b, int x, int y, int z, unsigned long long i) {
  unsigned long long k = (unsigned long long)b;
  k = (k << p.xb) | (unsigned long long)x;
  k = (k << p.yb) | (unsigned long long)y;
  k = (k << p.zb) | (unsigned long long)z;
  return (k << p.ib) | i;
}
__device__ inline uint64_t pk_canonical(const PackSpec &p, unsigned long long w) {
  unsigned long long k = w >> p.ib;
  const int z = (int)(k & ((1ull << p.zb) - 1)); k >>= p.zb;
  const int y = (int)(k & ((1ull << p.yb) - 1)); k >>= p.yb;
  const int x = (int)(k & ((1ull << p.xb) - 1)); k >>= p.xb;
  return pack_key((int)k, x, y, z);
}
// load + validate point i; returns 0 ok, 1 dropped silently, 2 out of the generic range, 3 does not fit the word
__device__ inline int pk_load(const int64_t *__restrict__ coords, int64_t i, int ncols, const PackSpec &p, int &b,
                              int &x, int &y, int &z) {
  const int64_t *c = coords + i * ncols;
  const int64_t X = c[0], Y = c[1], Z = c[2], B = ncols == 4 ? c[3] : 0;
  if (X == -1 && Y == -1 && Z == -1) return 1;
  if (X < 0 || Y < 0 || Z < 0 || B < 0 || X > kMaxCoord || Y > kMaxCoord || Z > kMaxCoord || B > kMaxCoord) return 2;
  if (X >= (1ll << p.xb) || Y >= (1ll << p.yb) || Z >= (1ll << p.zb) || B >= (1ll << p.bb) - 1) return 3;
  b = (int)B; x = (int)X; y = (int)Y; z = (int)Z;
  return 0;
}
constexpr int kMetaRedo = 5;                    // meta word: 0 = run the generic form instead

// variant 1: one device atomic per point
__global__ __launch_bounds__(256) void k_voxel_insert_packed(const int64_t *__restrict__ coords, int64_t n, int ncols,
                                                             PackSpec sp, unsigned long long *words, uint64_t mask,
                                                             int32_t *__restrict__ slot,
                                                             int32_t *__restrict__ cnt_extra, int32_t *__restrict__ head,
                                                             unsigned long long *__restrict__ status, int64_t nchunks,
                                                             int32_t *meta) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nchunks) status[i] = 0ull;
  if (i >= n) return;
  cnt_extra[i] = 0;
  head[i] = -1;
  int b, x, y, z;
  const int rc = pk_load(coords, i, ncols, sp, b, x, y, z);
  if (rc) {
    slot[i] = -1;
    if (rc == 2) atomicAnd(&meta[2], 0);
    if (rc == 3) atomicAnd(&meta[kMetaRedo], 0);
    return;
  }
  const unsigned long long w = pk_word(sp, b, x, y, z, (unsigned long long)i);
  uint64_t h = mix64(pack_key(b, x, y, z)) & mask;
  for (uint32_t t = 0;; ++t) {
    const unsigned long long prev = atomicCAS(&words[h], ~0ull, w);
    if (prev == ~0ull) break;                                  // the usual case: ONE atomic
    if ((prev >> sp.ib) == (w >> sp.ib)) {                     // my voxel: keep the smaller point index
      if (w < prev) atomicMin(&words[h], w);
      break;
    }
    h = grid_next(h, mask, t);
  }
  slot[i] = (int32_t)h;
}

// variant 2, P1: partition the points' words by hash block.  LDS: [nbins] counts, [nbins] bases.
template <int ITEMS>
__global__ __launch_bounds__(256) void k_voxel_bin(const int64_t *__restrict__ coords, int64_t n, int ncols, PackSpec sp,
                                                   uint64_t mask, int nbins, int capbin,
                                                   unsigned long long *__restrict__ rec, int32_t *cursor,
                                                   int32_t *__restrict__ info, int32_t *__restrict__ cnt_extra,
                                                   int32_t *__restrict__ head, unsigned long long *__restrict__ status,
                                                   int64_t nchunks, int32_t *meta) {
  extern __shared__ int32_t s_hist[];
  for (int q = threadIdx.x; q < nbins; q += 256) s_hist[q] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * (256 * ITEMS);
  unsigned long long w[ITEMS];
  int bin[ITEMS], rank[ITEMS];
#pragma unroll
  for (int j = 0; j < ITEMS; ++j) {
    const int64_t i = base + j * 256 + threadIdx.x;      // consecutive lanes read consecutive points
    bin[j] = -1;
    if (i < nchunks) status[i] = 0ull;
    if (i >= n) continue;
    cnt_extra[i] = 0;
    head[i] = -1;
    int b, x, y, z;
    const int rc = pk_load(coords, i, ncols, sp, b, x, y, z);
    if (rc) {
      info[i] = kInfoNone;
      if (rc == 2) atomicAnd(&meta[2], 0);
      if (rc == 3) atomicAnd(&meta[kMetaRedo], 0);
      continue;
    }
    w[j] = pk_word(sp, b, x, y, z, (unsigned long long)i);
    bin[j] = (int)((mix64(pack_key(b, x, y, z)) & mask) / kGridBlock);
    rank[j] = atomicAdd(&s_hist[bin[j]], 1);
  }
  __syncthreads();
  for (int q = threadIdx.x; q < nbins; q += 256) {
    const int c = s_hist[q];
    s_hist[nbins + q] = c ? atomicAdd(&cursor[q], c) : 0;      // one reservation per (workgroup, block)
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < ITEMS; ++j) {
    if (bin[j] < 0) continue;
    const int pos = s_hist[nbins + bin[j]] + rank[j];
    if (pos < capbin)
      rec[(int64_t)bin[j] * capbin + pos] = w[j];
    else {                                                     // a block with more points than its region holds
      atomicAnd(&meta[kMetaRedo], 0);
      info[base + j * 256 + threadIdx.x] = kInfoNone;
    }
  }
}

// variant 2, P2: one workgroup per hash block builds the block's table in LDS, links the further points of every
// voxel into the voxel's chain (LDS exchange: the block sees ALL points of its voxels, so the chain heads and counts
// need no device atomics either; they go out per SLOT, over the block's own record region, and K2's first points
// carry them to their site rows) and writes the finished grid entries.
constexpr int kBinRec = 16;                     // records per thread (256 x 16 = one region = kGridBlock)
__global__ __launch_bounds__(256) void k_voxel_bin_build(unsigned long long *__restrict__ rec,
                                                         const int32_t *__restrict__ cursor, int capbin, PackSpec sp,
                                                         GridEnt *__restrict__ grid, int32_t *__restrict__ info,
                                                         int32_t *__restrict__ nxt, int32_t *meta) {
  __shared__ unsigned long long tab[kGridBlock];
  __shared__ int32_t lhead[kGridBlock];
  __shared__ int32_t lcnt[kGridBlock];
  const int bin = blockIdx.x;
  int cnt = cursor[bin];
  cnt = cnt < capbin ? cnt : capbin;
  for (int q = threadIdx.x; q < (int)kGridBlock; q += 256) { tab[q] = ~0ull; lhead[q] = -1; lcnt[q] = 0; }
  __syncthreads();
  unsigned long long *r = rec + (int64_t)bin * capbin;
  const unsigned long long imask = (1ull << sp.ib) - 1;
  unsigned long long w[kBinRec];
  int sl[kBinRec];
#pragma unroll
  for (int q = 0; q < kBinRec; ++q) {
    const int k = q * 256 + threadIdx.x;
    sl[q] = -1;
    w[q] = k < cnt ? r[k] : ~0ull;
  }
#pragma unroll
  for (int q = 0; q < kBinRec; ++q) {
    if (w[q] == ~0ull) continue;
    uint32_t h = (uint32_t)(mix64(pk_canonical(sp, w[q])) & (kGridBlock - 1));
    for (uint32_t t = 0; t < kGridBlock; ++t) {
      const unsigned long long prev = atomicCAS(&tab[h], ~0ull, w[q]);
      if (prev == ~0ull) { sl[q] = (int)h; break; }
      if ((prev >> sp.ib) == (w[q] >> sp.ib)) {
        if (w[q] < prev) atomicMin(&tab[h], w[q]);
        sl[q] = (int)h;
        break;
      }
      h = (h + 1) & (kGridBlock - 1);
    }
    if (sl[q] < 0) {                                           // the block is full: the generic form copes
      atomicAnd(&meta[kMetaRedo], 0);
      info[(int64_t)(w[q] & imask)] = kInfoNone;
    }
  }
  __syncthreads();                                             // every word is in: the minima are final
  int first[kBinRec];
#pragma unroll
  for (int q = 0; q < kBinRec; ++q) {
    first[q] = 0;
    if (sl[q] < 0) continue;
    const unsigned long long idx = w[q] & imask;
    first[q] = (tab[sl[q]] & imask) == idx ? 1 : 0;
    if (!first[q]) {                                           // a further point of its voxel: push it on the chain
      nxt[(int64_t)idx] = atomicExch(&lhead[sl[q]], (int32_t)idx);
      atomicAdd(&lcnt[sl[q]], 1);
    }
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < kBinRec; ++q) {
    if (sl[q] < 0) continue;
    const uint32_t multi = lcnt[sl[q]] > 0 ? (uint32_t)kInfoMulti : 0u;
    info[(int64_t)(w[q] & imask)] =
        (int32_t)(((uint32_t)bin * kGridBlock + (uint32_t)sl[q]) | multi | ((uint32_t)first[q] << 31));
  }
  GridEnt *g = grid + (int64_t)bin * kGridBlock;
  for (int q = threadIdx.x; q < (int)kGridBlock; q += 256) {
    const unsigned long long ww = tab[q];
    uint4 e;
    if (ww == ~0ull) {
      e = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);
    } else {
      const uint64_t key = pk_canonical(sp, ww);
      e = make_uint4((uint32_t)key, (uint32_t)(key >> 32), (uint32_t)(ww & imask), 0xffffffffu);
    }
    *reinterpret_cast<uint4 *>(g + q) = e;
    // per slot {chain head, further points}: over the block's own record region (every record was read above)
    r[q] = ((unsigned long long)(uint32_t)lcnt[q] << 32) | (unsigned long long)(uint32_t)lhead[q];
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
// (round 4: a 32-bit-index form and an 8-sites-per-thread form of this kernel were measured SLOWER -- 92 and 124 us
// against 73 us at 1.5 M points -- profiles/r04_scatter_ab.txt; this is the round-3 kernel)
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
constexpr int kBwdU = 4;    // points per thread: the two-hop chain point_site[i] -> d_out[site] carries 4 bytes per lane,
                            // several in flight per thread: 49 -> 35-38 us at 1.5 M points (1.75 -> 2.3-2.7 TB/s)
__global__ __launch_bounds__(256) void k_voxel_backward(float *__restrict__ d_in, const float *__restrict__ d_out,
                                                        int64_t n, int planes, const int32_t *__restrict__ point_site,
                                                        const int32_t *__restrict__ first_pt,
                                                        const int32_t *__restrict__ last_pt,
                                                        const int32_t *__restrict__ cnt_extra, int mode) {
  const bool narrow = planes <= 256;                        // points per workgroup and group etc.: see k_voxel_mean
  const unsigned spb = narrow ? 256u / (unsigned)planes : 1u;
  const unsigned sl = narrow ? threadIdx.x / (unsigned)planes : 0u;
  const int p0 = (int)(threadIdx.x - sl * (unsigned)planes);
  const int pstep = narrow ? planes : 256;
  if (sl >= spb) return;
  int v[kBwdU];
  float mult[kBwdU];
#pragma unroll
  for (int j = 0; j < kBwdU; ++j) {
    const int64_t i = ((int64_t)blockIdx.x * kBwdU + j) * spb + sl;
    v[j] = i < n ? point_site[i] : -2;
  }
#pragma unroll
  for (int j = 0; j < kBwdU; ++j) {
    const int64_t i = ((int64_t)blockIdx.x * kBwdU + j) * spb + sl;
    mult[j] = 1.0f;
    if (v[j] >= 0) {
      if (mode == 1 && first_pt[v[j]] != (int32_t)i) v[j] = -1;
      if (mode == 2 && last_pt[v[j]] != (int32_t)i) v[j] = -1;
      if (mode == 4 && v[j] >= 0) mult[j] = __fdiv_rn(1.0f, (float)(cnt_extra[v[j]] + 1));
    }
  }
  for (int p = p0; p < planes; p += pstep) {
    float g[kBwdU];
#pragma unroll
    for (int j = 0; j < kBwdU; ++j) g[j] = v[j] >= 0 ? d_out[(int64_t)v[j] * planes + p] : 0.0f;
#pragma unroll
    for (int j = 0; j < kBwdU; ++j) {
      const int64_t i = ((int64_t)blockIdx.x * kBwdU + j) * spb + sl;
      if (v[j] != -2) d_in[i * planes + p] = v[j] >= 0 ? __fadd_rn(0.0f, __fmul_rn(mult[j], g[j])) : 0.0f;
    }
  }
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

extern "C" int aabr_input_layer_sites(const int64_t *coords, int64_t n, int ncols, uint64_t *keys, int64_t cap,
                                      int32_t *slot, int32_t *point_site, int32_t *site_coords, int32_t *first_pt,
                                      int32_t *cnt_extra, int32_t *head, int32_t *nxt, int32_t *status, int32_t *meta,
                                      void *stream_) {
  hipStream_t st = (hipStream_t)stream_;
  AABR_CHECK_ARG(n >= 0 && n < (1ll << 31) - 65536 && (ncols == 3 || ncols == 4), "0 <= n < 2^31, ncols in {3,4}");
  AABR_CHECK_ARG(is_pow2(cap) && cap >= 2 * n && cap >= 64, "cap must be a power of two >= max(64, 2n)");
  AABR_CHECK_ARG(keys && slot && point_site && site_coords && first_pt && cnt_extra && head && nxt && status && meta,
                 "null pointer");
  AABR_CHECK_ARG(((uintptr_t)status & 7) == 0 && ((uintptr_t)site_coords & 15) == 0 && ((uintptr_t)keys & 15) == 0,
                 "status 8-, site_coords / grid entries 16-byte aligned");
  AABR_CHECK_ARG(n == 0 || coords, "null coords");
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
  hipLaunchKernelGGL((k_voxel_number<4, 0>), dim3((unsigned)nchunks), dim3(kVsThreads), 0, st, coords, n, ncols, slot,
                     grid, site_coords, first_pt, point_site, cnt_extra, head, nxt,
                     (unsigned long long *)status, meta, (const unsigned long long *)nullptr, 0ull);
  AABR_CHECK_LAUNCH();
  return AABR_OK;
}

static int bits_for(int64_t v) {   // bits that hold 0 .. v-1
  int b = 1;
  while ((1ll << b) < v) ++b;
  return b;
}
static bool pack_spec(int64_t n, const int32_t *spatial, PackSpec &sp) {
  for (int d = 0; d < 3; ++d)
    if (spatial[d] < 1 || spatial[d] > kMaxCoord + 1) return false;
  sp.xb = bits_for(spatial[0]); sp.yb = bits_for(spatial[1]); sp.zb = bits_for(spatial[2]);
  sp.ib = bits_for(n > 1 ? n : 2);
  sp.bb = 64 - sp.xb - sp.yb - sp.zb - sp.ib;
  if (sp.bb > 16) sp.bb = 16;
  return sp.bb >= 2;
}

extern "C" int aabr_input_layer_pack_bits(int64_t n, const int32_t *spatial_host) {
  PackSpec sp;
  if (!spatial_host || n < 0 || !pack_spec(n, spatial_host, sp)) return 0;
  return sp.bb;
}

extern "C" int aabr_input_layer_sites_packed(const int64_t *coords, int64_t n, int ncols, const int32_t *spatial_host,
                                             int variant, uint64_t *keys, int64_t cap, uint64_t *words, int32_t *cursor,
                                             int32_t *slot, int32_t *point_site, int32_t *site_coords, int32_t *first_pt,
                                             int32_t *cnt_extra, int32_t *head, int32_t *nxt, int32_t *status,
                                             int32_t *meta, void *stream_) {
  hipStream_t st = (hipStream_t)stream_;
  AABR_CHECK_ARG(n > 0 && n < (1ll << 31) - 65536 && (ncols == 3 || ncols == 4), "0 < n < 2^31, ncols in {3,4}");
  AABR_CHECK_ARG(is_pow2(cap) && cap >= 2 * n && cap >= 64, "cap must be a power of two >= max(64, 2n)");
  AABR_CHECK_ARG(variant == 1 || variant == 2, "variant: 1 packed (one atomic per point), 2 LDS-binned");
  AABR_CHECK_ARG(coords && spatial_host && keys && words && slot && point_site && site_coords && first_pt && cnt_extra &&
                     head && nxt && status && meta, "null pointer");
  AABR_CHECK_ARG(((uintptr_t)status & 7) == 0 && ((uintptr_t)site_coords & 15) == 0 && ((uintptr_t)keys & 15) == 0 &&
                     ((uintptr_t)words & 7) == 0, "status / words 8-, site_coords / grid entries 16-byte aligned");
  PackSpec sp;
  AABR_CHECK_ARG(pack_spec(n, spatial_host, sp), "the fields (spatial size, n) do not fit a 64-bit word: use "
                                                 "aabr_input_layer_sites (aabr_input_layer_pack_bits tells)");
  GridEnt *grid = reinterpret_cast<GridEnt *>(keys);
  const int64_t nchunks = ceil_div(n, (int64_t)kVsThreads * vs_items(n));
  const unsigned long long imask = (1ull << sp.ib) - 1;
  hipMemsetAsync(meta, 0xFF, AABR_META_WORDS * sizeof(int32_t), st);
  if (variant == 1) {
    hipMemsetAsync(grid, 0xFF, cap * sizeof(GridEnt), st);
    hipMemsetAsync(words, 0xFF, cap * sizeof(uint64_t), st);
    hipLaunchKernelGGL(k_voxel_insert_packed, grid1(n, 256), dim3(256), 0, st, coords, n, ncols, sp,
                       (unsigned long long *)words, (uint64_t)(cap - 1), slot, cnt_extra, head,
                       (unsigned long long *)status, nchunks, meta);
    hipLaunchKernelGGL((k_voxel_number<4, 1>), dim3((unsigned)nchunks), dim3(kVsThreads), 0, st, coords, n, ncols, slot,
                       grid, site_coords, first_pt, point_site, cnt_extra, head, nxt, (unsigned long long *)status,
                       meta, (const unsigned long long *)words, imask);
  } else {
    const int64_t nbins = cap / kGridBlock;
    AABR_CHECK_ARG(cap >= (int64_t)kGridBlock && nbins <= 4096 && cursor,
                   "LDS-binned form: 4096 <= cap <= 2^24 slots and a cursor array of cap/4096 words");
    hipMemsetAsync(cursor, 0, nbins * sizeof(int32_t), st);
    // ~8 points per (workgroup, block): few reservations, records written in 64-byte runs
    int64_t per = 8 * nbins;
    per = per < 1024 ? 1024 : (per > 8192 ? 8192 : per);
    const size_t lds = 2 * (size_t)nbins * sizeof(int32_t);
#define AABR_BIN(ITEMS)                                                                                              \
  hipLaunchKernelGGL((k_voxel_bin<ITEMS>), grid1(n, 256 * ITEMS), dim3(256), lds, st, coords, n, ncols, sp,          \
                     (uint64_t)(cap - 1), (int)nbins, (int)kGridBlock, (unsigned long long *)words, cursor, slot,    \
                     cnt_extra, head, (unsigned long long *)status, nchunks, meta)
    if (per <= 1024) AABR_BIN(4);
    else if (per <= 2048) AABR_BIN(8);
    else if (per <= 4096) AABR_BIN(16);
    else AABR_BIN(32);
#undef AABR_BIN
    hipLaunchKernelGGL(k_voxel_bin_build, dim3((unsigned)nbins), dim3(256), 0, st, (unsigned long long *)words,
                       (const int32_t *)cursor, (int)kGridBlock, sp, grid, slot, nxt, meta);
    hipLaunchKernelGGL((k_voxel_number<4, 2>), dim3((unsigned)nchunks), dim3(kVsThreads), 0, st, coords, n, ncols, slot,
                       grid, site_coords, first_pt, point_site, cnt_extra, head, nxt, (unsigned long long *)status,
                       meta, (const unsigned long long *)words, imask);
  }
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
  hipLaunchKernelGGL(k_voxel_backward, grid1(n, (planes <= 256 ? 256 / planes : 1) * kBwdU), dim3(256), 0, (hipStream_t)stream_, d_in_feats,
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
