// common.h -- shared host/device helpers of libaabr_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <atomic>
#include "../../include/aabr_hip.h"

namespace aabr {

void set_error(const char *fmt, ...);

// Tuning knobs (experiments and tests only; the defaults are what ships).  Each is read from its environment
// variable AABR_<NAME> ONCE, at its first use in the process, and can be set explicitly through aabr_set_knob --
// no entry point calls getenv on its launch path.
// ONE list: the enumerators and the names (api.cpp) are generated from it, so they cannot drift apart.
#define AABR_KNOB_LIST(X)                                                                                            \
  X(CONV_WLDS) X(CONV_SMALL) X(CONV_NBW) X(CONV_WPB) X(WIDE_ROWS) X(CONV_WIDE) X(WIDE_NBUF) X(CONV_WIDE_BF16)         \
  X(VOXEL_MEAN) X(WIDE_NCB) X(BN_SMALL) X(WIDE_PRIO) X(PLAN_SIDE_BATCH) X(PLAN_SIDE_PRIO) X(SMALL_WPB) X(SMALL_MAX)   \
  X(WIDE_SPLIT) X(SPLIT_TARGET) X(SPLIT_NBUF) X(SPLIT_MIN_ITEMS) X(CONV_NARROW) X(DW_FULL) X(DW_FULL_MIN) X(DW_FULL_WGS) X(SPLIT_ROWS) X(GEOM_JOBS)
#define AABR_KNOB_ENUM(n) K_##n,
enum Knob { AABR_KNOB_LIST(AABR_KNOB_ENUM) K_COUNT };
#undef AABR_KNOB_ENUM
constexpr int kKnobUnset = -2147483647 - 1;
int knob(Knob k);            // kKnobUnset when neither the environment nor aabr_set_knob gave a value

#define AABR_CHECK_ARG(cond, msg)                                   \
  do {                                                              \
    if (!(cond)) { aabr::set_error("%s: %s", __func__, msg); return AABR_EINVAL; } \
  } while (0)

#define AABR_CHECK_LAUNCH()                                                   \
  do {                                                                        \
    hipError_t e__ = hipGetLastError();                                       \
    if (e__ != hipSuccess) {                                                  \
      aabr::set_error("%s: HIP error %s", __func__, hipGetErrorString(e__));  \
      return AABR_ELAUNCH;                                                    \
    }                                                                         \
  } while (0)

#define AABR_CHECK_HIP(call)                                                  \
  do {                                                                        \
    hipError_t e__ = (call);                                                  \
    if (e__ != hipSuccess) {                                                  \
      aabr::set_error("%s: HIP error %s", __func__, hipGetErrorString(e__));  \
      return AABR_ELAUNCH;                                                    \
    }                                                                         \
  } while (0)

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a PER-DEVICE setting: one flag per (call site, device), set only
// after the call succeeded, readable from the launcher thread and the caller's thread alike.
struct DynLdsOnce {
  std::atomic<uint64_t> done{0};   // bit d = set on device d (64 devices; beyond that the call is simply repeated)
};
inline hipError_t dyn_lds_once(DynLdsOnce &o, const void *kernel, int bytes) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const uint64_t bit = dev < 64 ? 1ull << dev : 0ull;
  if (o.done.load(std::memory_order_acquire) & bit) return hipSuccess;
  e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) o.done.fetch_or(bit, std::memory_order_release);
  return e;
}

constexpr uint64_t kEmptyKey = 0xFFFFFFFFFFFFFFFFull;
constexpr int kMaxCoord = 65534;

// (b,x,y,z) -> 64-bit key, 16 bits each, spatial coordinates biased by +1 so a probe one
// cell outside the grid (-1) is still a valid, never-inserted key.
__host__ __device__ inline uint64_t pack_key(int b, int x, int y, int z) {
  return ((uint64_t)(uint32_t)(b & 0xffff) << 48) | ((uint64_t)(uint32_t)((x + 1) & 0xffff) << 32) |
         ((uint64_t)(uint32_t)((y + 1) & 0xffff) << 16) | (uint64_t)(uint32_t)((z + 1) & 0xffff);
}
__host__ __device__ inline bool coord_in_range(int v) { return v >= -1 && v <= kMaxCoord; }

__host__ __device__ inline uint64_t mix64(uint64_t k) {
  k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33;
  k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
  return k;
}

// One slot of a hash grid: key, the smallest item index that hit the slot (first-seen numbering) and the site row,
// INTERLEAVED in 16 bytes -- the CAS on the key, the atomicMin on `first`, the later read of `first`, the store and
// every probe of `val` touch ONE 64-byte sector (round 2 kept three arrays: three sectors per point, 119 B of
// write-back per point counted for k_voxel_insert alone).  The library's `keys` parameters point at these entries.
struct __align__(16) GridEnt {
  unsigned long long key;
  uint32_t first;
  int32_t val;
};
static_assert(sizeof(GridEnt) == 16, "GridEnt is 16 bytes");

// Probe sequence of every grid: linear inside the key's home BLOCK of kGridBlock slots (wrapping at the block's end),
// and only when all kGridBlock slots of that block have been visited -- input grids are at most half full (a full block
// is a > 40 sigma event for mixed keys), derived grids at most 2/3 full (SCN.derived_cap / `2 cap >= 3 E` in geometry.hip:
// ~26 sigma; unsuccessful probe chains about twice as long as at load 1/2) -- on through the slots behind the block.  Find and insert walk the same
// sequence.  The block rule is what lets the LDS-binned voxel scatter (voxel_scatter.hip) build one block per
// workgroup entirely in LDS and write it out with coalesced stores; it also keeps a probe chain inside one 64 KiB
// window.  Tables smaller than a block are one block.
constexpr uint32_t kGridBlock = 4096;
__device__ inline uint64_t grid_next(uint64_t h, uint64_t mask, uint32_t t) {
  const uint64_t bm = mask < (uint64_t)(kGridBlock - 1) ? mask : (uint64_t)(kGridBlock - 1);
  return t < bm ? ((h & ~bm) | ((h + 1) & bm)) : (((t == bm ? (h | bm) : h) + 1) & mask);
}

// read-only probe of a finished table: one 16-byte load per slot visited
__device__ inline int grid_find(const GridEnt *__restrict__ g, uint64_t mask, uint64_t key) {
  uint64_t h = mix64(key) & mask;
  for (uint32_t t = 0;; ++t) {
    const uint4 e = *reinterpret_cast<const uint4 *>(g + h);
    const uint64_t k = (uint64_t)e.x | ((uint64_t)e.y << 32);
    if (k == key) return (int)e.w;
    if (k == kEmptyKey) return -1;
    h = grid_next(h, mask, t);
  }
}

// insert-or-find; returns the slot
__device__ inline uint32_t grid_insert(GridEnt *g, uint64_t mask, uint64_t key) {
  uint64_t h = mix64(key) & mask;
  for (uint32_t t = 0;; ++t) {
    unsigned long long prev = atomicCAS(&g[h].key, (unsigned long long)kEmptyKey, (unsigned long long)key);
    if (prev == kEmptyKey || prev == key) return (uint32_t)h;
    h = grid_next(h, mask, t);
  }
}

// ---- stream builders over MANY rule books in one launch (round 6) ----------------------------------------------------
// A pass compiles every gather table into its streaming forms: wide tile blocks (conv_wide.hip), 64-row tile blocks and
// the offset-major pair list of the weight gradient (conv.hip).  Round 5 issued them book by book -- 4 launches x ~45
// books of 2.5-14 us each, a third of a step's launches.  A job list hands up to kStreamJobsMax books to ONE launch:
// block b of the grid serves job j with first[j] <= b < first[j + 1].  Same per-book code, bit-identical words.
constexpr int kStreamJobsMax = 48;
struct StreamJob {
  const int32_t *table;    // gather table [vol][V]
  const int32_t *counts;   // per-256-row hit counts [vol][ceil(V / 256)] (pair lists only)
  int32_t *words;          // the stream being built
  int64_t V;
  int32_t vol, T;          // T: rows per wide tile (wide blocks only)
};
struct StreamJobs {
  StreamJob j[kStreamJobsMax];
  uint32_t first[kStreamJobsMax + 1];   // first block of job j; first[n] = grid size
  int n;
};
// job of block b (wave-uniform: a linear walk over <= 48 prefix entries held in SGPRs)
__device__ inline int stream_job_of(const StreamJobs &js, unsigned b) {
  int j = 0;
  while (j + 1 < js.n && b >= js.first[j + 1]) ++j;
  return j;
}
int launch_wide_blocks_jobs(const StreamJob *jobs, int n, hipStream_t st);     // conv_wide.hip
int launch_tile_blocks_jobs(const StreamJob *jobs, int n, hipStream_t st);     // conv.hip
int launch_offset_pairs_jobs(const StreamJob *jobs, int n, hipStream_t st);    // conv.hip
// per-sample row offsets of many levels in one launch (geometry.hip k_sample_offsets: one 64-thread block per level)
constexpr int kSampleJobsMax = 32;
struct SampleJob {
  const int32_t *sc, *meta;
  int32_t *out;
  int64_t V_max;
  int32_t max_samples;
};
int launch_sample_offsets_jobs(const SampleJob *jobs, int n, hipStream_t st);  // geometry.hip

inline bool is_pow2(int64_t v) { return v > 0 && (v & (v - 1)) == 0; }
inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

} // namespace aabr
