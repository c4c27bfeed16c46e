// geometry.hip -- on-device sparse-grid state for the SparseConvNet hot path (gfx950).
//
// Replaces the host-side hash grids and rule-book builders of the reference
// (SparseConvNet/sparseconvnet/SCN/Metadata/{Metadata.cpp,IOLayersRules.h,
//  SubmanifoldConvolutionRules.h,ConvolutionRules.h,RectangularRegions.h}) with device
// kernels: the hash grid, the site numbering and every rule table live in HBM and never
// visit the host.  Integer work, HBM/latency bound: coalesced key/coord streams, random
// probes into a table that sits in L2 / Infinity Cache.
//
// Site numbering contract (DESIGN.md §Row order):
//   input layer  : first-seen order of the point list  == IOLayersRules.h:86-91 (exact)
//   strided conv : first-seen order over input rows ascending, then output-region order
//                  (the reference's order is dense_hash_map iteration order, an artefact;
//                   parity there is modulo a per-sample permutation)
#include "geom.h"

namespace aabr {

// ------------------------------------------------------------------ block-wide 2-lane scan
template <int NT>
__device__ inline void block_exscan2(int a, int b, int &ea, int &eb, int &ta, int &tb) {
  __shared__ int sa[NT / 64], sb[NT / 64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int ia = a, ib = b;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    int ua = __shfl_up(ia, d), ub = __shfl_up(ib, d);
    if (lane >= d) { ia += ua; ib += ub; }
  }
  if (lane == 63) { sa[w] = ia; sb[w] = ib; }
  __syncthreads();
  int pa = 0, pb = 0, xa = 0, xb = 0;
#pragma unroll
  for (int j = 0; j < NT / 64; ++j) {
    int va = sa[j], vb = sb[j];
    if (j < w) { pa += va; pb += vb; }
    xa += va; xb += vb;
  }
  ea = pa + ia - a; eb = pb + ib - b; ta = xa; tb = xb;
  __syncthreads();
}

constexpr int kScanThreads = 256;
constexpr int kScanItems = 1; // one item per thread: the flag lookups are dependent random loads, so more,
                              // smaller blocks beat longer per-thread chains (80k points: 313 blocks, not 79)
constexpr int kScanTile = kScanThreads * kScanItems; // 256 items per block
constexpr int kInlinePrefixBlocks = 2048;            // up to 512k items: block prefix summed in k_assign_sites

// "first" flag of encounter e: it won the atomicMin on its slot.
__device__ inline void first_flag(const int32_t *slot, const GridEnt *minidx, const uint32_t *slotcnt,
                                  int64_t e, int64_t n, int &f, int &c) {
  f = 0; c = 0;
  if (e < n) {
    int s = slot[e];
    if (s >= 0 && minidx[s].first == (uint32_t)e) {
      f = 1;
      c = slotcnt ? (int)(slotcnt[s] + 1u) : 1; // slotcnt starts at 0xFFFFFFFF
    }
  }
}

__global__ __launch_bounds__(kScanThreads) void k_scan_blocksums(
    const int32_t *__restrict__ slot, const GridEnt *__restrict__ minidx,
    const uint32_t *__restrict__ slotcnt, int64_t n, int32_t *__restrict__ blocksums) {
  int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
  int a = 0, b = 0;
#pragma unroll
  for (int j = 0; j < kScanItems; ++j) {
    int f, c;
    first_flag(slot, minidx, slotcnt, base + j, n, f, c);
    a += f; b += c;
  }
  int ea, eb, ta, tb;
  block_exscan2<kScanThreads>(a, b, ea, eb, ta, tb);
  if (threadIdx.x == 0) { blocksums[2 * blockIdx.x] = ta; blocksums[2 * blockIdx.x + 1] = tb; }
}

// single block: exclusive scan of the per-block sums; totals -> meta[0], meta[3]
__global__ __launch_bounds__(1024) void k_scan_blockprefix(const int32_t *__restrict__ blocksums,
                                                          int64_t nblk, int32_t *__restrict__ prefix,
                                                          int32_t *__restrict__ meta,
                                                          int32_t *__restrict__ site_off) {
  int ca = 0, cb = 0;
  for (int64_t base = 0; base < nblk; base += 1024) {
    int64_t i = base + threadIdx.x;
    int a = 0, b = 0;
    if (i < nblk) { a = blocksums[2 * i]; b = blocksums[2 * i + 1]; }
    int ea, eb, ta, tb;
    block_exscan2<1024>(a, b, ea, eb, ta, tb);
    if (i < nblk) { prefix[2 * i] = ca + ea; prefix[2 * i + 1] = cb + eb; }
    ca += ta; cb += tb;
  }
  if (threadIdx.x == 0) {
    meta[0] = ca;
    meta[3] = cb;
    if (site_off) site_off[ca] = cb; // CSR terminator: offsets[V] = number of valid points
  }
}

// MODE 0: input layer (items = points); MODE 1: strided-conv output sites (items = encounters)
template <int MODE>
__global__ __launch_bounds__(kScanThreads) void k_assign_sites(
    const int32_t *__restrict__ slot, GridEnt *__restrict__ minidx,
    const uint32_t *__restrict__ slotcnt, int64_t n, const int32_t *__restrict__ prefix,
    const int32_t *__restrict__ blocksums, int32_t *__restrict__ site_coords, int32_t *__restrict__ site_off,
    int32_t *__restrict__ meta, const int64_t *__restrict__ coords64, int ncols,
    const int32_t *__restrict__ in_coords, ConvGeom g) {
  int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
  int f[kScanItems], c[kScanItems], a = 0, b = 0;
#pragma unroll
  for (int j = 0; j < kScanItems; ++j) {
    first_flag(slot, minidx, slotcnt, base + j, n, f[j], c[j]);
    a += f[j]; b += c[j];
  }
  int ea, eb, ta, tb;
  block_exscan2<kScanThreads>(a, b, ea, eb, ta, tb);
  int pa, pb;
  if (blocksums) {
    // few blocks (<= kInlinePrefixBlocks): every block sums the block totals in front of it itself -- a
    // couple of KB from L2 -- instead of waiting for a one-block prefix kernel in between
    __shared__ int wsa[kScanThreads / 64], wsb[kScanThreads / 64];
    int sa = 0, sb = 0;
    for (int i = threadIdx.x; i < (int)blockIdx.x; i += kScanThreads) { sa += blocksums[2 * i]; sb += blocksums[2 * i + 1]; }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { sa += __shfl_xor(sa, d); sb += __shfl_xor(sb, d); }
    if ((threadIdx.x & 63) == 0) { wsa[threadIdx.x >> 6] = sa; wsb[threadIdx.x >> 6] = sb; }
    __syncthreads();
    pa = 0; pb = 0;
#pragma unroll
    for (int w = 0; w < kScanThreads / 64; ++w) { pa += wsa[w]; pb += wsb[w]; }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) { // totals, as k_scan_blockprefix writes them
      meta[0] = pa + ta;
      meta[3] = pb + tb;
      if (MODE == 0 && site_off) site_off[pa + ta] = pb + tb; // CSR terminator
    }
  } else {
    pa = prefix[2 * blockIdx.x]; pb = prefix[2 * blockIdx.x + 1];
  }
  int ra = pa + ea, rb = pb + eb;
  int maxc = 0;
#pragma unroll
  for (int j = 0; j < kScanItems; ++j) {
    if (f[j]) {
      int64_t e = base + j;
      minidx[slot[e]].val = ra;
      if (MODE == 0) {
        const int64_t *cp = coords64 + e * ncols;
        int4 sc = make_int4((int)cp[0], (int)cp[1], (int)cp[2], ncols == 4 ? (int)cp[3] : 0);
        *reinterpret_cast<int4 *>(site_coords + 4 * (int64_t)ra) = sc;
        site_off[ra] = rb;
        maxc = c[j] > maxc ? c[j] : maxc;
      } else {
        int64_t u = e / g.maxout;
        int l = (int)(e % g.maxout);
        int4 ic = *reinterpret_cast<const int4 *>(in_coords + 4 * u);
        int p[3] = {ic.x, ic.y, ic.z}, jj[3];
        output_region_lth(g, p, l, jj);
        *reinterpret_cast<int4 *>(site_coords + 4 * (int64_t)ra) = make_int4(jj[0], jj[1], jj[2], ic.w);
      }
      ra += 1; rb += c[j];
    }
  }
  if (MODE == 0) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      int o = __shfl_xor(maxc, d);
      maxc = o > maxc ? o : maxc;
    }
    // one atomic per BLOCK, and none when the running maximum already covers it (meta[1] only grows, so a
    // stale read can only cause a redundant atomic): same-address atomics serialise at ~14 ns each
    __shared__ int wmax[kScanThreads / 64];
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = maxc;
    __syncthreads();
    if (threadIdx.x == 0) {
      int m = wmax[0];
      for (int w = 1; w < kScanThreads / 64; ++w) m = wmax[w] > m ? wmax[w] : m;
      if (m > 0 && m > __builtin_nontemporal_load(&meta[1])) atomicMax(&meta[1], m);
    }
  }
}

// ------------------------------------------------------------------ rule tables
__global__ __launch_bounds__(256) void k_conv_insert_sites(const int32_t *__restrict__ in_coords,
                                                           int64_t V_in, ConvGeom g, GridEnt *keys,
                                                           uint64_t mask, int32_t *__restrict__ slot) {
  int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= V_in) return;
  int4 ic = *reinterpret_cast<const int4 *>(in_coords + 4 * u);
  int p[3] = {ic.x, ic.y, ic.z}, j[3];
  for (int l = 0; l < g.maxout; ++l) {
    int64_t e = u * g.maxout + l;
    if (output_region_lth(g, p, l, j)) {
      uint32_t h = grid_insert(keys, mask, pack_key(ic.w, j[0], j[1], j[2]));
      atomicMin(&keys[h].first, (uint32_t)e);
      slot[e] = (int32_t)h;
    } else {
      slot[e] = -1;
    }
  }
}

// one block per filter offset: ordered compaction of the table row into (entry, row) pairs
__global__ __launch_bounds__(1024) void k_table_to_rulebook(const int32_t *__restrict__ table, int64_t V,
                                                            int32_t *__restrict__ rules,
                                                            int32_t *__restrict__ counts) {
  const int k = blockIdx.x;
  int run = 0;
  for (int64_t base = 0; base < V; base += 1024) {
    int64_t v = base + threadIdx.x;
    int t = (v < V) ? table[(int64_t)k * V + v] : -1;
    int f = t >= 0, ea, eb, ta, tb;
    block_exscan2<1024>(f, 0, ea, eb, ta, tb);
    if (f) {
      int32_t *r = rules + ((int64_t)k * V + run + ea) * 2;
      r[0] = t; r[1] = (int32_t)v;
    }
    run += ta;
  }
  if (threadIdx.x == 0) counts[k] = run;
}

// A1: the dataset's host quantisation on the device (data3d/suncg_utils/suncg_dataset.py:126-188
// without the train-time augmentations): a = xyz*scale - min; keep 0 <= a < full_scale;
// locs = trunc(a); feature xyz = a / scale.  Dropped points get the (-1,-1,-1) sentinel that the
// input layer skips, so no compaction (and no host read-back) is needed.  Arithmetic in double
// like numpy's.
template <typename T>
__global__ __launch_bounds__(256) void k_quantize_points(const T *__restrict__ xyz, int64_t n, double scale,
                                                         const T *__restrict__ amin, const int32_t *fs,
                                                         int64_t batch, int64_t *__restrict__ locs,
                                                         float *__restrict__ feats, int fstride) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double a[3];
  bool keep = true;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    a[d] = (double)xyz[3 * i + d] * scale - (double)amin[d] * scale;
    keep = keep && a[d] >= 0.0 && a[d] < (double)fs[d];
  }
#pragma unroll
  for (int d = 0; d < 3; ++d) locs[4 * i + d] = keep ? (int64_t)a[d] : -1;
  locs[4 * i + 3] = batch;
  if (feats)
#pragma unroll
    for (int d = 0; d < 3; ++d) feats[i * fstride + d] = (float)(a[d] / scale);
}

__global__ __launch_bounds__(256) void k_spatial_locations(const int32_t *__restrict__ sc, int64_t n4,
                                                           int64_t *__restrict__ loc) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n4) loc[i] = sc[i];
}

// Row offset of every sample in a batch-contiguous site list (what the reference keeps as SparseGrid::ctr,
// Metadata.h:24-33): out[0] = V, out[1 + b] = first row whose batch index is >= b (b = 0 .. max_samples), so
// sample b owns rows [out[1+b], out[2+b]).  V is read from device memory (meta[0] of the grid builder), so the
// launch can follow the builder without a host read in between.
struct SampleJobs { SampleJob j[kSampleJobsMax]; };
__global__ __launch_bounds__(64) void k_sample_offsets(const SampleJobs js) {      // block = level (common.h SampleJob)
  const int32_t *__restrict__ sc = js.j[blockIdx.x].sc;
  const int32_t *__restrict__ meta = js.j[blockIdx.x].meta;
  int32_t *__restrict__ out = js.j[blockIdx.x].out;
  const int64_t V_max = js.j[blockIdx.x].V_max;
  const int max_samples = js.j[blockIdx.x].max_samples;
  int64_t V = meta[0];
  if (V > V_max) V = V_max;
  if (V < 0) V = 0;
  for (int b = threadIdx.x; b <= max_samples; b += 64) {
    int64_t lo = 0, hi = V;
    while (lo < hi) {
      int64_t mid = (lo + hi) >> 1;
      if (sc[mid * 4 + 3] < b) lo = mid + 1; else hi = mid;
    }
    out[1 + b] = (int32_t)lo;
  }
  if (threadIdx.x == 0) out[0] = (int32_t)V;
}

} // namespace aabr

using namespace aabr;

extern "C" int aabr_submanifold_table(const int32_t *site_coords, int64_t V, const uint64_t *keys,
                                      int64_t cap, const int32_t *fs_host,
                                      int32_t *table, int32_t *counts, void *stream_) {
  hipStream_t st = (hipStream_t)stream_;
  AABR_CHECK_ARG(V >= 0 && fs_host && is_pow2(cap), "bad V/filter/cap");
  Filter3 fs;
  int64_t vol = 1;
  for (int i = 0; i < 3; ++i) {
    AABR_CHECK_ARG(fs_host[i] >= 1 && fs_host[i] <= 64, "filter size out of range");
    fs.size[i] = fs_host[i];
    vol *= fs_host[i];
  }
  AABR_CHECK_ARG(vol <= 65535, "filter volume too large");
  if (V == 0) return AABR_OK;
  AABR_CHECK_ARG(site_coords && keys && table && ((uintptr_t)keys & 15) == 0, "null / misaligned pointer");
  hipLaunchKernelGGL(k_submanifold_table<HashFinder>, dim3((unsigned)ceil_div(V, 256), (unsigned)vol), dim3(256), 0, st,
                     site_coords, V, HashFinder{reinterpret_cast<const GridEnt *>(keys), (uint64_t)(cap - 1)}, fs, table,
                     counts);
  AABR_CHECK_LAUNCH();
  return AABR_OK;
}

extern "C" int aabr_convolution_sites(const int32_t *in_coords, int64_t V_in, const int32_t *size_host,
                                      const int32_t *stride_host, const int32_t *out_spatial_host,
                                      uint64_t *out_keys, int64_t out_cap,
                                      int32_t *scratch, int32_t *out_site_coords, int32_t *meta,
                                      void *stream_) {
  hipStream_t st = (hipStream_t)stream_;
  AABR_CHECK_ARG(V_in >= 0 && size_host && stride_host && out_spatial_host, "bad arguments");
  ConvGeom g;
  AABR_CHECK_ARG(make_geom(size_host, stride_host, out_spatial_host, g, true) == 0, "bad filter geometry");
  int64_t E = V_in * g.maxout;
  AABR_CHECK_ARG(E < (int64_t)0x7fffffff, "too many encounters");
  // at most E distinct keys are inserted: a load factor of 2/3 in the worst case (every input site its own output
  // site), 0.15-0.35 on strided levels; round 3 asked for 2 E, which made every derived grid of a pass as large as
  // the grid it was derived from (8 x 16 MB filled per bench step for grids of 49 .. 40 k sites)
  AABR_CHECK_ARG(is_pow2(out_cap) && 2 * out_cap >= 3 * E && out_cap >= 64,
                 "out_cap must be a power of two >= max(64, 1.5*V_in*max_out_per_in)");
  AABR_CHECK_ARG(out_keys && scratch && out_site_coords && meta && ((uintptr_t)out_keys & 15) == 0,
                 "null / misaligned pointer");
  int64_t nblk = ceil_div(E > 0 ? E : 1, kScanTile);
  GridEnt *grid = reinterpret_cast<GridEnt *>(out_keys);
  int32_t *slot = scratch;
  int32_t *blocksums = slot + E;
  int32_t *prefix = blocksums + 2 * nblk;
  hipMemsetAsync(grid, 0xFF, out_cap * sizeof(GridEnt), st);   // key = empty, first = max, val = -1
  hipMemsetAsync(meta, 0, AABR_META_WORDS * sizeof(int32_t), st);
  if (V_in > 0)
    hipLaunchKernelGGL(k_conv_insert_sites, grid1(V_in, 256), dim3(256), 0, st, in_coords, V_in, g, grid,
                       (uint64_t)(out_cap - 1), slot);
  hipLaunchKernelGGL(k_scan_blocksums, dim3((unsigned)nblk), dim3(kScanThreads), 0, st, slot, grid,
                     (const uint32_t *)nullptr, E, blocksums);
  const bool inline_prefix = nblk <= kInlinePrefixBlocks;
  if (!inline_prefix)
    hipLaunchKernelGGL(k_scan_blockprefix, dim3(1), dim3(1024), 0, st, blocksums, nblk, prefix, meta,
                       (int32_t *)nullptr);
  hipLaunchKernelGGL(k_assign_sites<1>, dim3((unsigned)nblk), dim3(kScanThreads), 0, st, slot, grid,
                     (const uint32_t *)nullptr, E, prefix,
                     inline_prefix ? (const int32_t *)blocksums : (const int32_t *)nullptr, out_site_coords, (int32_t *)nullptr, meta,
                     (const int64_t *)nullptr, 0, in_coords, g);
  AABR_CHECK_LAUNCH();
  return AABR_OK;
}

extern "C" int aabr_convolution_tables2(const int32_t *in_coords, int64_t V_in, const uint64_t *in_keys,
                                        int64_t in_cap, const int32_t *out_coords,
                                        int64_t V_out, const uint64_t *out_keys,
                                        int64_t out_cap, const int32_t *size_host, const int32_t *stride_host,
                                        const int32_t *out_spatial_host, int32_t *table_out, int32_t *table_in,
                                        int32_t *counts, int32_t *counts_in, void *stream_);

extern "C" int aabr_convolution_tables(const int32_t *in_coords, int64_t V_in, const uint64_t *in_keys,
                                       int64_t in_cap, const int32_t *out_coords,
                                       int64_t V_out, const uint64_t *out_keys,
                                       int64_t out_cap, const int32_t *size_host, const int32_t *stride_host,
                                       const int32_t *out_spatial_host, int32_t *table_out, int32_t *table_in,
                                       int32_t *counts, void *stream_) {
  return aabr_convolution_tables2(in_coords, V_in, in_keys, in_cap, out_coords, V_out, out_keys,
                                  out_cap, size_host, stride_host, out_spatial_host, table_out, table_in, counts,
                                  nullptr, stream_);
}

extern "C" int aabr_convolution_tables2(const int32_t *in_coords, int64_t V_in, const uint64_t *in_keys,
                                        int64_t in_cap, const int32_t *out_coords,
                                        int64_t V_out, const uint64_t *out_keys,
                                        int64_t out_cap, const int32_t *size_host, const int32_t *stride_host,
                                        const int32_t *out_spatial_host, int32_t *table_out, int32_t *table_in,
                                        int32_t *counts, int32_t *counts_in, void *stream_) {
  hipStream_t st = (hipStream_t)stream_;
  AABR_CHECK_ARG(V_in >= 0 && V_out >= 0 && size_host && stride_host && out_spatial_host, "bad arguments");
  AABR_CHECK_ARG(is_pow2(in_cap) && is_pow2(out_cap), "capacities must be powers of two");
  ConvGeom g;
  AABR_CHECK_ARG(make_geom(size_host, stride_host, out_spatial_host, g) == 0, "bad filter geometry");
  int vol = g.size[0] * g.size[1] * g.size[2];
  if (V_out > 0 && table_out) {
    AABR_CHECK_ARG(out_coords && in_keys && ((uintptr_t)in_keys & 15) == 0, "null / misaligned pointer");
    hipLaunchKernelGGL(k_conv_table_out<HashFinder>, dim3((unsigned)ceil_div(V_out, 256), (unsigned)vol), dim3(256), 0,
                       st, out_coords, V_out,
                       HashFinder{reinterpret_cast<const GridEnt *>(in_keys), (uint64_t)(in_cap - 1)}, g, table_out,
                       counts);
  }
  if (V_in > 0 && table_in) {
    AABR_CHECK_ARG(in_coords && out_keys && ((uintptr_t)out_keys & 15) == 0, "null / misaligned pointer");
    hipLaunchKernelGGL(k_conv_table_in<HashFinder>, dim3((unsigned)ceil_div(V_in, 256), (unsigned)vol), dim3(256), 0, st,
                       in_coords, V_in, HashFinder{reinterpret_cast<const GridEnt *>(out_keys), (uint64_t)(out_cap - 1)},
                       g, table_in, counts_in);
  }
  AABR_CHECK_LAUNCH();
  return AABR_OK;
}

extern "C" int aabr_table_to_rulebook(const int32_t *table, int64_t V, int vol, int32_t *rules,
                                      int32_t *counts, void *stream_) {
  AABR_CHECK_ARG(V >= 0 && vol > 0 && counts, "bad arguments");
  if (V == 0) {
    hipMemsetAsync(counts, 0, vol * sizeof(int32_t), (hipStream_t)stream_);
    return AABR_OK;
  }
  AABR_CHECK_ARG(table && rules, "null pointer");
  hipLaunchKernelGGL(k_table_to_rulebook, dim3((unsigned)vol), dim3(1024), 0, (hipStream_t)stream_, table, V,
                     rules, counts);
  AABR_CHECK_LAUNCH();
  return AABR_OK;
}

extern "C" int aabr_quantize_points(const void *xyz, int is_double, int64_t n, double scale, const void *xyz_min,
                                    const int32_t *full_scale_dev, int64_t batch_index, int64_t *locs,
                                    float *feats_xyz, int feat_stride, void *stream_) {
  AABR_CHECK_ARG(n >= 0 && scale > 0 && batch_index >= 0 && batch_index <= kMaxCoord, "bad arguments");
  if (n == 0) return AABR_OK;
  AABR_CHECK_ARG(xyz && xyz_min && full_scale_dev && locs, "null pointer");
  AABR_CHECK_ARG(!feats_xyz || feat_stride >= 3, "feature stride must cover the 3 xyz planes");
  if (is_double)
    hipLaunchKernelGGL(k_quantize_points<double>, grid1(n, 256), dim3(256), 0, (hipStream_t)stream_,
                       (const double *)xyz, n, scale, (const double *)xyz_min, full_scale_dev, batch_index, locs,
                       feats_xyz, feat_stride);
  else
    hipLaunchKernelGGL(k_quantize_points<float>, grid1(n, 256), dim3(256), 0, (hipStream_t)stream_,
                       (const float *)xyz, n, scale, (const float *)xyz_min, full_scale_dev, batch_index, locs,
                       feats_xyz, feat_stride);
  AABR_CHECK_LAUNCH();
  return AABR_OK;
}

extern "C" int aabr_spatial_locations(const int32_t *site_coords, int64_t V, int64_t *locations,
                                      void *stream_) {
  AABR_CHECK_ARG(V >= 0, "bad V");
  if (V == 0) return AABR_OK;
  AABR_CHECK_ARG(site_coords && locations, "null pointer");
  hipLaunchKernelGGL(k_spatial_locations, grid1(4 * V, 256), dim3(256), 0, (hipStream_t)stream_, site_coords,
                     4 * V, locations);
  AABR_CHECK_LAUNCH();
  return AABR_OK;
}

extern "C" int aabr_sample_offsets(const int32_t *site_coords, const int32_t *meta, int64_t V_max, int max_samples,
                                   int32_t *out, void *stream_) {
  AABR_CHECK_ARG(site_coords && meta && out && V_max >= 0 && max_samples >= 1 && max_samples <= 4096, "bad arguments");
  const SampleJob job{site_coords, meta, out, V_max, max_samples};
  return launch_sample_offsets_jobs(&job, 1, (hipStream_t)stream_);
}

int aabr::launch_sample_offsets_jobs(const SampleJob *jobs, int n, hipStream_t st) {
  for (int j0 = 0; j0 < n; j0 += kSampleJobsMax) {
    SampleJobs js;
    const int m = n - j0 < kSampleJobsMax ? n - j0 : kSampleJobsMax;
    for (int j = 0; j < m; ++j) js.j[j] = jobs[j0 + j];
    hipLaunchKernelGGL(k_sample_offsets, dim3((unsigned)m), dim3(64), 0, st, js);
  }
  AABR_CHECK_LAUNCH();
  return AABR_OK;
}
