// brick.hip -- brick-major site order: grids and rule tables without hashing (gfx950, round 5).
//
// The reference keeps one google::dense_hash_map per sample and scale (SCN/Metadata/Metadata.h:24-34) and builds every
// rule book by probing it once per (site, filter offset) (SubmanifoldConvolutionRules.h:26-45, ConvolutionRules.h:11-34);
// geometry.hip does the same on the device, wider: 27 random 64-byte probes per site into a table that does not fit the
// L2s (profiles/r04_pmc_fetch_write_per_kernel.json: 15-30 x the algorithmic bytes).  Here a level is stored by WHERE its
// sites are (geom.h: BrickLevel): a dense directory with one bit per 4^3-voxel brick of the level's extent, the occupied
// bricks in directory order with a 64-bit cell mask each, and the sites numbered brick by brick, cell by cell.  Then
//   * a lookup is two dependent 16-byte loads from small arrays that spatial neighbours share (no hash, no probing);
//   * the rows of a workgroup of any table builder are spatial neighbours, so its lookups hit the same few lines;
//   * every level is built by two rounds of bit-setting and two prefix sums -- the same code for the input level
//     (items = the voxel scatter's sites) and for strided levels (items = the finer level's sites x their output
//     region, OutputRegionCalculator, RectangularRegions.h:109-119) -- and needs its site count on the host only when
//     somebody sizes a tensor by it: a whole pyramid costs ONE read.
// Row order inside a sample differs from the reference's first-seen order: parity is modulo the per-sample permutation
// SURVEY.md 7 allows; an opt-in of Metadata_3 (site_order="brick"), first-seen stays the default.
#include "geom.h"

namespace aabr {

constexpr int kBkMetaV = 0, kBkMetaNB = 1, kBkMetaErr = 2;   // meta words of a brick level

// OR `bits` into arr[idx] (64-bit words) for every lane with idx >= 0.  Lanes of a wave that address the same word --
// the common case: consecutive items are spatial neighbours -- are merged into one atomic; after kAggRounds distinct
// words the rest go out on their own.  Nobody reads the old value: the atomics are fire-and-forget (a returning atomic
// in the merge loop made every round wait for its round trip).  Returns true in the lanes that issued an atomic (one per
// distinct word and round).  Wave-uniform call.
constexpr int kAggRounds = 4;
__device__ inline bool wave_or64(unsigned long long *arr, int64_t idx, unsigned long long bits) {
  bool active = idx >= 0, issued = false;
  const int lane = threadIdx.x & 63;
#pragma unroll 1
  for (int it = 0; it < kAggRounds; ++it) {
    const unsigned long long am = __ballot(active);
    if (!am) return issued;
    const int leader = __ffsll((long long)am) - 1;
    const int lo = __shfl((int)(uint32_t)idx, leader), hi = __shfl((int)(uint32_t)((uint64_t)idx >> 32), leader);
    const int64_t k = (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint64_t)(uint32_t)lo);
    const bool same = active && idx == k;
    uint32_t vlo = same ? (uint32_t)bits : 0u, vhi = same ? (uint32_t)(bits >> 32) : 0u;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      vlo |= (uint32_t)__shfl_xor((int)vlo, d);
      vhi |= (uint32_t)__shfl_xor((int)vhi, d);
    }
    if (lane == leader) { atomicOr(&arr[k], ((unsigned long long)vhi << 32) | vlo); issued = true; }
    active = active && !same;
  }
  if (active) { atomicOr(&arr[idx], bits); issued = true; }
  return issued;
}

// Rounds 1 and 2 of a level build.  Items = (input site u, l-th cell of its output region).
//   ROUND 1: the item's output voxel marks its brick in the directory word of its super-brick;
//   ROUND 2 (the directory prefix is known): the voxel sets its cell bit in its brick's mask and leaves the brick's
//            coordinates (no old value is read back: every sender stores the same coordinates).
template <int ROUND>
__global__ __launch_bounds__(256) void k_brick_mark(const int32_t *__restrict__ in_coords, int64_t vin_bound,
                                                    const int32_t *__restrict__ vin_dev, ConvGeom g, BrickDims d,
                                                    uint4 *dir, uint4 *bricks, int64_t nb_cap, int4 *__restrict__ bcoord,
                                                    int32_t *meta) {
  const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t V = vin_bound;
  if (vin_dev) { const int64_t dv = vin_dev[0]; V = dv < V ? dv : V; }
  const bool live = u < V;
  int4 ic = make_int4(0, 0, 0, 0);
  if (live) ic = *reinterpret_cast<const int4 *>(in_coords + 4 * u);
  const int p[3] = {ic.x, ic.y, ic.z};
  bool err = false;
#pragma unroll 1
  for (int l = 0; l < g.maxout; ++l) {
    int j[3] = {0, 0, 0};
    bool ok = live && output_region_lth(g, p, l, j);
    int64_t w = -1;
    if (ok) {
      w = brick_dir_index(d, ic.w, j[0], j[1], j[2]);
      if (w < 0) { err = true; ok = false; }          // a site outside the extent the directory was sized for
    }
    if (ROUND == 1) {
      // (round 6 measured "test before set" -- a plain read of the word first, no atomic when the bit already shows; three
      // quarters of the items at 1.5 M points: 392 -> 444 us for the scatter, 96.5 -> 99.2 at 320 k.  The atomics are
      // fire-and-forget and overlap; the read in front of them is a dependent round trip.  Not kept.)
      wave_or64(reinterpret_cast<unsigned long long *>(dir), ok ? 2 * w : -1, 1ull << brick_bit(j[0], j[1], j[2]));
    } else {
      int64_t bi = -1;
      if (ok) {
        const uint4 e = dir[w];
        const unsigned long long word = lo64(e);
        const int jb = brick_bit(j[0], j[1], j[2]);
        bi = (int64_t)e.z + __popcll(word & ((1ull << jb) - 1ull));     // the bit itself was set in round 1
        if (bi >= nb_cap) { err = true; bi = -1; }
      }
      const bool wrote = wave_or64(reinterpret_cast<unsigned long long *>(bricks), bi >= 0 ? 2 * bi : -1,
                                   1ull << brick_cell(j[0], j[1], j[2]));
      if (wrote)        // whoever sends a brick's atomic also stores its coordinates (the same value from every sender)
        bcoord[bi] = make_int4(j[0] >> 2, j[1] >> 2, j[2] >> 2, ic.w);
    }
  }
  if (err) meta[kBkMetaErr] = 1;
}

// ---- one-pass prefix sum over the popcounts of 16-byte entries (chunk scan + decoupled look-back, as k_voxel_number) ----
// MODE 0: entries = directory words, prefix of bricks into .z, total -> meta[NB];
// MODE 1: entries = bricks[0 .. meta[NB]), prefix of sites into .z, total -> meta[V].
constexpr int kBsThreads = 256, kBsItems = 4, kBsChunk = kBsThreads * kBsItems;
__device__ inline unsigned long long bs_pack(unsigned flag, unsigned v) {
  return ((unsigned long long)flag << 62) | (unsigned long long)v;
}
// MODE 1's write-out also produces what follows from its result: the level's site list (brick by brick, cell by cell:
// the brick's coordinates + the cell) -- no separate pass.
template <int MODE>
__global__ __launch_bounds__(kBsThreads) void k_brick_scan(uint4 *arr, int64_t n_bound, unsigned long long *status,
                                                           int32_t *ticket, int32_t *meta, int64_t total_cap, BrickDims d,
                                                           int4 *__restrict__ bcoord, int32_t *__restrict__ site_coords,
                                                           int64_t v_cap) {
  __shared__ int s_chunk;
  __shared__ int s_wsum[kBsThreads / 64];
  __shared__ unsigned s_excl;
  if (threadIdx.x == 0) s_chunk = atomicAdd(ticket, 1);    // chunks start in ticket order: a chunk only waits for
  __syncthreads();                                          // chunks that already run
  const int chunk = s_chunk;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int64_t n = n_bound;
  if (MODE == 1) { const int64_t nb = meta[kBkMetaNB]; n = nb < n ? nb : n; }
  const int64_t base = (int64_t)chunk * kBsChunk + (int64_t)threadIdx.x * kBsItems;
  int v[kBsItems], a = 0;
  unsigned long long bits[kBsItems];
#pragma unroll
  for (int j = 0; j < kBsItems; ++j) {
    const int64_t i = base + j;
    v[j] = 0;
    bits[j] = 0ull;
    if (i < n) {
      const uint4 e = arr[i];
      bits[j] = lo64(e);
      v[j] = __popc(e.x) + __popc(e.y);
    }
    a += v[j];
  }
  int incl = a;
#pragma unroll
  for (int dd = 1; dd < 64; dd <<= 1) {
    const int o = __shfl_up(incl, dd);
    if (lane >= dd) incl += o;
  }
  if (lane == 63) s_wsum[wave] = incl;
  __syncthreads();
  int wpre = 0, total = 0;
#pragma unroll
  for (int w = 0; w < kBsThreads / 64; ++w) {
    if (w < wave) wpre += s_wsum[w];
    total += s_wsum[w];
  }
  const int ex_in_chunk = wpre + incl - a;
  if (wave == 0) {
    if (lane == 0)
      __hip_atomic_store(&status[chunk], bs_pack(chunk == 0 ? 2u : 1u, (unsigned)total), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    unsigned excl = 0;
    int j = chunk - 1;
    while (j >= 0) {
      const int idx = j - lane;
      unsigned long long st = bs_pack(2u, 0u);
      if (idx >= 0) {
        do {
          st = __hip_atomic_load(&status[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if ((st >> 62) == 0ull) __builtin_amdgcn_s_sleep(1);
        } while ((st >> 62) == 0ull);
      }
      const unsigned long long m2 = __ballot((st >> 62) == 2ull);
      const int stop = m2 ? __ffsll((long long)m2) - 1 : 63;
      unsigned vv = lane <= stop ? (unsigned)(st & 0xffffffffull) : 0u;
#pragma unroll
      for (int dd = 32; dd >= 1; dd >>= 1) vv += __shfl_xor(vv, dd);
      excl += vv;
      if (m2) break;
      j -= 64;
    }
    if (lane == 0) {
      if (chunk != 0)
        __hip_atomic_store(&status[chunk], bs_pack(2u, excl + (unsigned)total), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      s_excl = excl;
      if (chunk == (int)gridDim.x - 1) {
        const int64_t tot = (int64_t)excl + total;
        meta[MODE == 0 ? kBkMetaNB : kBkMetaV] = (int32_t)tot;
        if (tot > total_cap) meta[kBkMetaErr] = 1;
      }
    }
  }
  __syncthreads();
  int run = (int)s_excl + ex_in_chunk;
#pragma unroll
  for (int j = 0; j < kBsItems; ++j) {
    const int64_t i = base + j;
    if (i < n) {
      reinterpret_cast<int32_t *>(arr + i)[2] = run;
      unsigned long long m = bits[j];
      if (MODE == 1 && m) {
        const int4 bc = bcoord[i];
        int64_t row = run;
        while (m) {
          const int c = __ffsll((long long)m) - 1;
          m &= m - 1ull;
          if (row < v_cap)
            *reinterpret_cast<int4 *>(site_coords + 4 * row) =
                make_int4(4 * bc.x + (c >> 4), 4 * bc.y + ((c >> 2) & 3), 4 * bc.z + (c & 3), bc.w);
          ++row;
        }
      }
      run += v[j];
    }
  }
}

// Input level: the voxel scatter numbered its sites in first-seen order (IOLayersRules.h:86-91); `old_coords` are those
// sites.  Carry the per-site arrays of the input layer over to the brick-major rows: new_row[old r], old_row[new i] and
// first_pt / cnt_extra / head by row.
__global__ __launch_bounds__(256) void k_brick_renumber(const int32_t *__restrict__ old_coords, int64_t V, BrickLevel L,
                                                        int32_t *__restrict__ new_of_old, int32_t *__restrict__ old_of_new,
                                                        const int32_t *__restrict__ first_pt,
                                                        const int32_t *__restrict__ cnt_extra,
                                                        const int32_t *__restrict__ head, int32_t *__restrict__ first_pt2,
                                                        int32_t *__restrict__ cnt_extra2, int32_t *__restrict__ head2,
                                                        int32_t *meta) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= V) return;
  const int4 c = *reinterpret_cast<const int4 *>(old_coords + 4 * r);
  const int i = brick_find(L, c.w, c.x, c.y, c.z);
  if (i < 0 || i >= V) { meta[kBkMetaErr] = 1; new_of_old[r] = 0; return; }
  new_of_old[r] = i;
  old_of_new[i] = (int32_t)r;
  first_pt2[i] = first_pt[r];
  cnt_extra2[i] = cnt_extra[r];
  head2[i] = head[r];
}
__global__ __launch_bounds__(256) void k_brick_remap_points(const int32_t *__restrict__ point_site, int64_t n,
                                                            const int32_t *__restrict__ new_of_old,
                                                            int32_t *__restrict__ point_site2) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const int s = point_site[p];
  point_site2[p] = s >= 0 ? new_of_old[s] : s;
}

// ---- voxel scatter straight into a brick grid (InputLayer without a hash table) ----------------------------------------
// The reference's inputLayerRules (IOLayersRules.h:18-125) dedups the points of a sample in a hash map and numbers the
// voxels in first-seen order; voxel_scatter.hip does that on the device (LDS-binned hash build + a first-seen scan) and
// the brick path then renumbered those sites.  Brick-major rows need neither the hash nor the first-seen numbers: a
// voxel's row follows from where it lies.  So the POINTS are the items of the level build (aabr_brick_build with size =
// stride = 1: duplicates set the same bits), and per point
//   k_points_prepare : int64 (x, y, z[, b]) -> int32x4 (x = -1: skipped), validity, the scene's extent (meta[8..11]);
//   k_points_sites   : row = brick_find(point) -> point_site; first_pt[row] = min point index (ONE atomic per point);
//   k_points_chains  : every point that is not its voxel's first pushes itself on the voxel's chain and counts itself --
//                      the same first_pt / cnt_extra / head / nxt the mean and backward kernels of voxel_scatter.hip read.
// (a few hundred workgroups stride over the points: the extent costs one atomic per workgroup and axis at most -- one per
// WAVE of a point-per-thread launch was 20 k same-address atomics, 280 us at 320 k points)
constexpr int kPrepItems = 4;      // points per thread, their loads issued together (a grid-stride loop of dependent
                                   // iterations at 1.2 workgroups per CU took 23 us for 15 MB at 320 k points)
__global__ __launch_bounds__(256) void k_points_prepare(const int64_t *__restrict__ coords, int64_t n, int ncols,
                                                        int4 *__restrict__ pc, int32_t *meta, int32_t *__restrict__ first_pt,
                                                        int32_t *__restrict__ cnt_extra, int32_t *__restrict__ head,
                                                        int vec16) {
  int ext[4] = {-1, -1, -1, -1};
  bool bad = false;
  const int64_t base = (int64_t)blockIdx.x * (256 * kPrepItems) + threadIdx.x;
  int64_t cx[kPrepItems], cy[kPrepItems], cz[kPrepItems], cb[kPrepItems];
#pragma unroll
  for (int q = 0; q < kPrepItems; ++q) {
    const int64_t i = base + q * 256;
    cx[q] = cy[q] = cz[q] = -1; cb[q] = 0;
    if (i < n) {
      const int64_t *c = coords + i * ncols;
      if (vec16) {             // ncols == 4 and a 16-byte aligned list: 32 bytes per point in two 16-byte loads
        const longlong2 a = *reinterpret_cast<const longlong2 *>(c), b2 = *reinterpret_cast<const longlong2 *>(c + 2);
        cx[q] = a.x; cy[q] = a.y; cz[q] = b2.x; cb[q] = b2.y;
      } else {
        cx[q] = c[0]; cy[q] = c[1]; cz[q] = c[2]; cb[q] = ncols == 4 ? c[3] : 0;
      }
    }
  }
#pragma unroll
  for (int q = 0; q < kPrepItems; ++q) {
    const int64_t i = base + q * 256;
    if (i >= n) continue;
    const int64_t x = cx[q], y = cy[q], z = cz[q], b = cb[q];
    int4 o = make_int4(-1, -1, -1, 0);
    if (x == -1 && y == -1 && z == -1) {
      // dropped by aabr_quantize_points (outside FULL_SCALE): skipped silently
    } else if (x < 0 || y < 0 || z < 0 || b < 0 || x > kMaxCoord || y > kMaxCoord || z > kMaxCoord || b > kMaxCoord) {
      bad = true;
    } else {
      o = make_int4((int)x, (int)y, (int)z, (int)b);
      ext[0] = o.x > ext[0] ? o.x : ext[0]; ext[1] = o.y > ext[1] ? o.y : ext[1];
      ext[2] = o.z > ext[2] ? o.z : ext[2]; ext[3] = o.w > ext[3] ? o.w : ext[3];
    }
    pc[i] = o;
    if (first_pt) { first_pt[i] = -1; cnt_extra[i] = 0; head[i] = -1; }   // the starting values aabr_points_sites wants
  }
  __shared__ int s_ext[4][4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    int m = ext[q];
#pragma unroll
    for (int dd = 32; dd >= 1; dd >>= 1) {
      const int o = __shfl_xor(m, dd);
      m = o > m ? o : m;
    }
    if ((threadIdx.x & 63) == 0) s_ext[threadIdx.x >> 6][q] = m;
  }
  const bool anybad = __syncthreads_or(bad ? 1 : 0) != 0;
  if (threadIdx.x < 4) {
    const int q = threadIdx.x;
    int m = s_ext[0][q];
    for (int w = 1; w < 4; ++w) m = s_ext[w][q] > m ? s_ext[w][q] : m;
    if (m > __builtin_nontemporal_load(&meta[8 + q])) atomicMax(&meta[8 + q], m);
  }
  if (anybad && threadIdx.x == 0) atomicAnd(&meta[2], 0);      // meta starts at all ones: 0 = a coordinate out of range
}
__global__ __launch_bounds__(256) void k_points_sites(const int4 *__restrict__ pc, int64_t n, BrickLevel L,
                                                      int32_t *__restrict__ point_site, uint32_t *first_pt,
                                                      int32_t *meta) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int4 c = pc[i];
  int row = -1;
  if (c.x >= 0) {
    row = brick_find(L, c.w, c.x, c.y, c.z);
    if (row < 0) meta[kBkMetaErr] = 1;           // (cannot happen: the point itself set the bits)
    else atomicMin(&first_pt[row], (uint32_t)i);
  }
  point_site[i] = row;
}
__global__ __launch_bounds__(256) void k_points_chains(int64_t n, const int32_t *__restrict__ point_site,
                                                       const uint32_t *__restrict__ first_pt, int32_t *head,
                                                       int32_t *cnt_extra, int32_t *__restrict__ nxt) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int row = point_site[i];
  if (row >= 0 && first_pt[row] != (uint32_t)i) {          // a further point of its voxel (rare: N / V ~ 1.03 on scenes)
    nxt[i] = atomicExch(&head[row], (int32_t)i);
    atomicAdd(&cnt_extra[row], 1);
  }
}

static int make_dims(const int32_t *dims_host, BrickDims &d) {
  if (!dims_host) return -1;
  for (int i = 0; i < 3; ++i) {
    if (dims_host[i] < 1 || dims_host[i] > 4096) return -1;
    d.sb[i] = dims_host[i];
  }
  if (dims_host[3] < 1 || dims_host[3] > 65535) return -1;
  d.nb = dims_host[3];
  return 0;
}
static inline int64_t dir_words(const BrickDims &d) { return (int64_t)d.nb * d.sb[0] * d.sb[1] * d.sb[2]; }

} // namespace aabr
using namespace aabr;

extern "C" int64_t aabr_brick_scratch_words(int64_t dir_words_, int64_t nb_cap) {
  const int64_t c0 = ceil_div(dir_words_ > 0 ? dir_words_ : 1, (int64_t)kBsChunk);
  const int64_t c1 = ceil_div(nb_cap > 0 ? nb_cap : 1, (int64_t)kBsChunk);
  return 2 * (c0 + c1) + 4;       // int32 words: two status arrays of 8-byte words + two tickets (+ pad)
}

extern "C" int aabr_brick_build(const int32_t *in_coords, int64_t vin_bound, const int32_t *vin_count_dev,
                                const int32_t *size_host, const int32_t *stride_host, const int32_t *out_spatial_host,
                                const int32_t *dims_host, void *dir, void *bricks, int64_t nb_cap, int32_t *bcoord,
                                int32_t *out_coords, int64_t v_cap, int32_t *meta, int32_t *scratch, int flags,
                                void *stream_) {
  hipStream_t st = (hipStream_t)stream_;
  AABR_CHECK_ARG((flags & ~1) == 0, "flags: bit 0 = the caller has cleared dir / bricks / meta / scratch");
  AABR_CHECK_ARG(vin_bound >= 0 && vin_bound < (int64_t)0x7fffffff && size_host && stride_host && out_spatial_host,
                 "bad arguments");
  ConvGeom g;
  AABR_CHECK_ARG(make_geom(size_host, stride_host, out_spatial_host, g, true) == 0, "bad filter geometry");
  BrickDims d;
  AABR_CHECK_ARG(make_dims(dims_host, d) == 0, "dims: super-bricks per axis in [1, 4096], samples in [1, 65535]");
  const int64_t nw = dir_words(d);
  AABR_CHECK_ARG(nw < (int64_t)1 << 31 && nb_cap >= 1 && nb_cap < (int64_t)1 << 31 && v_cap >= 1, "level too large");
  AABR_CHECK_ARG(dir && bricks && bcoord && out_coords && meta && scratch && (vin_bound == 0 || in_coords),
                 "null pointer");
  AABR_CHECK_ARG(((uintptr_t)dir & 15) == 0 && ((uintptr_t)bricks & 15) == 0 && ((uintptr_t)bcoord & 15) == 0 &&
                     ((uintptr_t)out_coords & 15) == 0 && ((uintptr_t)scratch & 7) == 0,
                 "directory / bricks / coordinates 16-byte, scratch 8-byte aligned");
  const int64_t c0 = ceil_div(nw, (int64_t)kBsChunk), c1 = ceil_div(nb_cap, (int64_t)kBsChunk);
  unsigned long long *status0 = reinterpret_cast<unsigned long long *>(scratch);
  unsigned long long *status1 = status0 + c0;
  int32_t *tickets = reinterpret_cast<int32_t *>(status1 + c1);
  if (!(flags & 1)) {
    if ((char *)bricks == (char *)dir + (size_t)nw * 16)     // one allocation (the layout the Python side uses): ONE fill
      AABR_CHECK_HIP(hipMemsetAsync(dir, 0, (size_t)(nw + nb_cap) * 16, st));
    else {
      AABR_CHECK_HIP(hipMemsetAsync(dir, 0, (size_t)nw * 16, st));
      AABR_CHECK_HIP(hipMemsetAsync(bricks, 0, (size_t)nb_cap * 16, st));
    }
    AABR_CHECK_HIP(hipMemsetAsync(meta, 0, AABR_META_WORDS * sizeof(int32_t), st));
    AABR_CHECK_HIP(hipMemsetAsync(scratch, 0, (size_t)(2 * (c0 + c1) + 4) * sizeof(int32_t), st));
  }
  if (vin_bound > 0)
    hipLaunchKernelGGL(k_brick_mark<1>, grid1(vin_bound, 256), dim3(256), 0, st, in_coords, vin_bound, vin_count_dev, g,
                       d, (uint4 *)dir, (uint4 *)bricks, nb_cap, (int4 *)bcoord, meta);
  hipLaunchKernelGGL(k_brick_scan<0>, dim3((unsigned)c0), dim3(kBsThreads), 0, st, (uint4 *)dir, nw, status0, tickets,
                     meta, nb_cap, d, (int4 *)bcoord, (int32_t *)nullptr, (int64_t)0);
  if (vin_bound > 0)
    hipLaunchKernelGGL(k_brick_mark<2>, grid1(vin_bound, 256), dim3(256), 0, st, in_coords, vin_bound, vin_count_dev, g,
                       d, (uint4 *)dir, (uint4 *)bricks, nb_cap, (int4 *)bcoord, meta);
  hipLaunchKernelGGL(k_brick_scan<1>, dim3((unsigned)c1), dim3(kBsThreads), 0, st, (uint4 *)bricks, nb_cap, status1,
                     tickets + 1, meta, v_cap, d, (int4 *)bcoord, out_coords, v_cap);
  AABR_CHECK_LAUNCH();
  return AABR_OK;
}

extern "C" int aabr_brick_renumber(const int32_t *old_coords, int64_t V, const int32_t *dims_host, const void *dir,
                                   const void *bricks, int32_t *new_of_old, int32_t *old_of_new, const int32_t *first_pt,
                                   const int32_t *cnt_extra, const int32_t *head, int32_t *first_pt2, int32_t *cnt_extra2,
                                   int32_t *head2, const int32_t *point_site, int64_t n, int32_t *point_site2,
                                   int32_t *meta, void *stream_) {
  hipStream_t st = (hipStream_t)stream_;
  AABR_CHECK_ARG(V >= 0 && n >= 0, "bad sizes");
  BrickLevel L;
  AABR_CHECK_ARG(make_dims(dims_host, L.d) == 0, "bad dims");
  if (V == 0) return AABR_OK;
  AABR_CHECK_ARG(old_coords && dir && bricks && new_of_old && old_of_new && first_pt && cnt_extra && head && first_pt2 &&
                     cnt_extra2 && head2 && meta && (n == 0 || (point_site && point_site2)), "null pointer");
  L.dir = (const uint4 *)dir;
  L.bricks = (const uint4 *)bricks;
  hipLaunchKernelGGL(k_brick_renumber, grid1(V, 256), dim3(256), 0, st, old_coords, V, L, new_of_old, old_of_new,
                     first_pt, cnt_extra, head, first_pt2, cnt_extra2, head2, meta);
  if (n > 0)
    hipLaunchKernelGGL(k_brick_remap_points, grid1(n, 256), dim3(256), 0, st, point_site, n,
                       (const int32_t *)new_of_old, point_site2);
  AABR_CHECK_LAUNCH();
  return AABR_OK;
}

extern "C" int aabr_brick_submanifold_table(const int32_t *site_coords, int64_t V, const int32_t *dims_host,
                                            const void *dir, const void *bricks, const int32_t *fs_host, int32_t *table,
                                            int32_t *counts, void *stream_) {
  hipStream_t st = (hipStream_t)stream_;
  AABR_CHECK_ARG(V >= 0 && fs_host, "bad V/filter");
  Filter3 fs;
  int64_t vol = 1;
  for (int i = 0; i < 3; ++i) {
    AABR_CHECK_ARG(fs_host[i] >= 1 && fs_host[i] <= 64, "filter size out of range");
    fs.size[i] = fs_host[i];
    vol *= fs_host[i];
  }
  AABR_CHECK_ARG(vol <= 65535, "filter volume too large");
  BrickFinder f;
  AABR_CHECK_ARG(make_dims(dims_host, f.L.d) == 0, "bad dims");
  if (V == 0) return AABR_OK;
  AABR_CHECK_ARG(site_coords && dir && bricks && table, "null pointer");
  f.L.dir = (const uint4 *)dir;
  f.L.bricks = (const uint4 *)bricks;
  hipLaunchKernelGGL(k_submanifold_table_rows<BrickFinder>, dim3((unsigned)ceil_div(V, 256)), dim3(256), 0, st, site_coords, V,
                     f, fs, table, counts);
  AABR_CHECK_LAUNCH();
  return AABR_OK;
}

extern "C" int aabr_brick_convolution_tables(const int32_t *in_coords, int64_t V_in, const int32_t *in_dims_host,
                                             const void *in_dir, const void *in_bricks, const int32_t *out_coords,
                                             int64_t V_out, const int32_t *out_dims_host, const void *out_dir,
                                             const void *out_bricks, const int32_t *size_host, const int32_t *stride_host,
                                             const int32_t *out_spatial_host, int32_t *table_out, int32_t *table_in,
                                             int32_t *counts, int32_t *counts_in, void *stream_) {
  hipStream_t st = (hipStream_t)stream_;
  AABR_CHECK_ARG(V_in >= 0 && V_out >= 0 && size_host && stride_host && out_spatial_host, "bad arguments");
  ConvGeom g;
  AABR_CHECK_ARG(make_geom(size_host, stride_host, out_spatial_host, g) == 0, "bad filter geometry");
  const int vol = g.size[0] * g.size[1] * g.size[2];
  BrickFinder fi, fo;
  AABR_CHECK_ARG(make_dims(in_dims_host, fi.L.d) == 0 && make_dims(out_dims_host, fo.L.d) == 0, "bad dims");
  fi.L.dir = (const uint4 *)in_dir; fi.L.bricks = (const uint4 *)in_bricks;
  fo.L.dir = (const uint4 *)out_dir; fo.L.bricks = (const uint4 *)out_bricks;
  if (V_out > 0 && table_out) {
    AABR_CHECK_ARG(out_coords && in_dir && in_bricks, "null pointer");
    hipLaunchKernelGGL((k_conv_table_rows<BrickFinder, 0>), dim3((unsigned)ceil_div(V_out, 256)), dim3(256), 0, st, out_coords,
                       V_out, fi, g, table_out, counts);
  }
  if (V_in > 0 && table_in) {
    AABR_CHECK_ARG(in_coords && out_dir && out_bricks, "null pointer");
    hipLaunchKernelGGL((k_conv_table_rows<BrickFinder, 1>), dim3((unsigned)ceil_div(V_in, 256)), dim3(256), 0, st, in_coords,
                       V_in, fo, g, table_in, counts_in);
  }
  AABR_CHECK_LAUNCH();
  return AABR_OK;
}

extern "C" int aabr_points_prepare(const int64_t *coords, int64_t n, int ncols, int32_t *pc, int32_t *meta,
                                   int32_t *first_pt, int32_t *cnt_extra, int32_t *head, void *stream_) {
  hipStream_t st = (hipStream_t)stream_;
  AABR_CHECK_ARG(n >= 0 && n < (1ll << 31) - 65536 && (ncols == 3 || ncols == 4), "0 <= n < 2^31, ncols in {3,4}");
  AABR_CHECK_ARG(meta && (n == 0 || (coords && pc)) && ((uintptr_t)pc & 15) == 0, "null / misaligned pointer");
  AABR_CHECK_ARG((first_pt != nullptr) == (cnt_extra != nullptr) && (first_pt != nullptr) == (head != nullptr),
                 "first_pt / cnt_extra / head: all three or none");
  AABR_CHECK_HIP(hipMemsetAsync(meta, 0xFF, AABR_META_WORDS * sizeof(int32_t), st));
  if (n > 0)
    hipLaunchKernelGGL(k_points_prepare, dim3((unsigned)ceil_div(n, 256 * kPrepItems)), dim3(256), 0, st, coords, n, ncols,
                       (int4 *)pc, meta, first_pt, cnt_extra, head, (ncols == 4 && ((uintptr_t)coords & 15) == 0) ? 1 : 0);
  AABR_CHECK_LAUNCH();
  return AABR_OK;
}

extern "C" int aabr_points_sites(const int32_t *pc, int64_t n, const int32_t *dims_host, const void *dir, const void *bricks,
                                 int32_t *point_site, int32_t *first_pt, int32_t *cnt_extra, int32_t *head, int32_t *nxt,
                                 int32_t *meta, int flags, void *stream_) {
  hipStream_t st = (hipStream_t)stream_;
  AABR_CHECK_ARG(n >= 0 && (flags & ~1) == 0, "bad n / flags");
  BrickLevel L;
  AABR_CHECK_ARG(make_dims(dims_host, L.d) == 0, "bad dims");
  if (n == 0) return AABR_OK;
  AABR_CHECK_ARG(pc && dir && bricks && point_site && first_pt && cnt_extra && head && nxt && meta, "null pointer");
  L.dir = (const uint4 *)dir;
  L.bricks = (const uint4 *)bricks;
  // first_pt and head start at all ones (no first point yet / empty chain), cnt_extra at 0
  if (flags & 1) {
    // aabr_points_prepare wrote them (three fills fewer)
  } else if (head == first_pt + n)
    AABR_CHECK_HIP(hipMemsetAsync(first_pt, 0xFF, (size_t)2 * n * sizeof(int32_t), st));
  else {
    AABR_CHECK_HIP(hipMemsetAsync(first_pt, 0xFF, (size_t)n * sizeof(int32_t), st));
    AABR_CHECK_HIP(hipMemsetAsync(head, 0xFF, (size_t)n * sizeof(int32_t), st));
  }
  if (!(flags & 1)) AABR_CHECK_HIP(hipMemsetAsync(cnt_extra, 0, (size_t)n * sizeof(int32_t), st));
  hipLaunchKernelGGL(k_points_sites, grid1(n, 256), dim3(256), 0, st, (const int4 *)pc, n, L, point_site,
                     (uint32_t *)first_pt, meta);
  hipLaunchKernelGGL(k_points_chains, grid1(n, 256), dim3(256), 0, st, n, (const int32_t *)point_site,
                     (const uint32_t *)first_pt, head, cnt_extra, nxt);
  AABR_CHECK_LAUNCH();
  return AABR_OK;
}
