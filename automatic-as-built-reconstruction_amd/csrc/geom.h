// geom.h -- what the grid / rule-table builders share (geometry.hip: hash grids, brick.hip: brick grids).
#pragma once
#include "common.h"

namespace aabr {

struct ConvGeom {
  int size[3], stride[3], out_sp[3];
  int maxout;
};
struct Filter3 { int size[3]; };

// l-th point (region order, z fastest) of the output region of input point p
// (OutputRegionCalculator, RectangularRegions.h:109-119).  Returns false if l is past the end.
__device__ inline bool output_region_lth(const ConvGeom &g, const int p[3], int l, int j[3]) {
  int lb[3], n[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    int num = p[i] - g.size[i] + g.stride[i];
    int q = num / g.stride[i]; // C division (toward zero), then clamp at 0 like std::max(0L, ..)
    lb[i] = q > 0 ? q : 0;
    int ub = p[i] / g.stride[i];
    if (ub > g.out_sp[i] - 1) ub = g.out_sp[i] - 1;
    n[i] = ub - lb[i] + 1;
    if (n[i] <= 0) return false;
  }
  if (l >= n[0] * n[1] * n[2]) return false;
  j[2] = lb[2] + l % n[2]; l /= n[2];
  j[1] = lb[1] + l % n[1]; l /= n[1];
  j[0] = lb[0] + l;
  return true;
}

// sites_only: output-grid construction alone also accepts the composition of several non-overlapping levels
// (size == stride up to 65536: exactly one output site per input site), see aabr_convolution_sites
static inline int make_geom(const int32_t *size, const int32_t *stride, const int32_t *out_sp, ConvGeom &g,
                            bool sites_only = false) {
  g.maxout = 1;
  for (int i = 0; i < 3; ++i) {
    const bool composed = sites_only && size[i] == stride[i] && size[i] >= 1 && size[i] <= 65536;
    if (!composed && (size[i] < 1 || size[i] > 64 || stride[i] < 1 || stride[i] > 64)) return -1;
    if (out_sp[i] < 1) return -1;
    g.size[i] = size[i]; g.stride[i] = stride[i]; g.out_sp[i] = out_sp[i];
    g.maxout *= (size[i] + stride[i] - 1) / stride[i];
  }
  return 0;
}

// counts[k * gridDim.x + blockIdx.x] = number of hits in this block (plain store: 27 hot
// addresses hammered by one atomic per wave cost more than the whole table build)
__device__ inline void block_count_store(int hit, int32_t *__restrict__ counts, int k) {
  __shared__ int wc[4];
  unsigned long long m = __ballot(hit);
  if ((threadIdx.x & 63) == 0) wc[threadIdx.x >> 6] = (int)__popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) counts[(int64_t)k * gridDim.x + blockIdx.x] = wc[0] + wc[1] + wc[2] + wc[3];
}

// ---- site lookup: (batch, x, y, z) -> row or -1 ----------------------------------------------------------------------
// Hash grid (first-seen numbering, the reference's order): one random 16-byte probe per lookup.
struct HashFinder {
  const GridEnt *keys;
  uint64_t mask;
  __device__ inline int operator()(int b, int x, int y, int z) const {
    if (!(coord_in_range(x) && coord_in_range(y) && coord_in_range(z))) return -1;
    return grid_find(keys, mask, pack_key(b, x, y, z));
  }
};

// Brick grid (round 5; DESIGN "Brick-major site order"): a site's row follows from where it lies, no hashing.
//   brick       = 4 x 4 x 4 voxels; its 64-bit mask holds one bit per cell, cell = (x&3)<<4 | (y&3)<<2 | (z&3);
//   super-brick = 4 x 4 x 4 bricks = 16^3 voxels; the DENSE directory has one 16-byte entry per super-brick of the
//                 level's extent: {64-bit word: one bit per brick, uint32 prefix: bricks in front of this word};
//   bricks[]    = the occupied bricks in directory order (batch, super-brick x, y, z, brick-in-super-brick),
//                 16 bytes each: {64-bit cell mask, int32 base: sites in front of this brick};
//   row(site)   = base(brick) + popcount(mask below the site's cell)  -- "brick-major" order.
// Two dependent 16-byte loads per lookup, both from small arrays that the rows of a workgroup (spatial neighbours by
// construction) share: the 27 look-ups of a site fall into <= 8 bricks.
struct BrickDims {
  int sb[3];   // super-bricks per axis (extent / 16, rounded up)
  int nb;      // samples
};
struct BrickLevel {
  const uint4 *dir;
  const uint4 *bricks;
  BrickDims d;
};
// directory word of a voxel, or -1 when it lies outside the level's extent
__device__ inline int64_t brick_dir_index(const BrickDims &d, int b, int x, int y, int z) {
  if ((unsigned)b >= (unsigned)d.nb || x < 0 || y < 0 || z < 0) return -1;
  const int sx = x >> 4, sy = y >> 4, sz = z >> 4;
  if (sx >= d.sb[0] || sy >= d.sb[1] || sz >= d.sb[2]) return -1;
  return (((int64_t)b * d.sb[0] + sx) * d.sb[1] + sy) * d.sb[2] + sz;
}
__device__ inline int brick_bit(int x, int y, int z) { return (((x >> 2) & 3) << 4) | (((y >> 2) & 3) << 2) | ((z >> 2) & 3); }
__device__ inline int brick_cell(int x, int y, int z) { return ((x & 3) << 4) | ((y & 3) << 2) | (z & 3); }
__device__ inline unsigned long long lo64(const uint4 &e) { return (unsigned long long)e.x | ((unsigned long long)e.y << 32); }

// index of the voxel's brick in bricks[], or -1
__device__ inline int brick_index(const BrickLevel &L, int b, int x, int y, int z) {
  const int64_t w = brick_dir_index(L.d, b, x, y, z);
  if (w < 0) return -1;
  const uint4 e = L.dir[w];
  const unsigned long long word = lo64(e);
  const int j = brick_bit(x, y, z);
  if (!((word >> j) & 1ull)) return -1;
  return (int)e.z + (int)__popcll(word & ((1ull << j) - 1ull));
}
__device__ inline int brick_find(const BrickLevel &L, int b, int x, int y, int z) {
  const int bi = brick_index(L, b, x, y, z);
  if (bi < 0) return -1;
  const uint4 br = L.bricks[bi];
  const unsigned long long m = lo64(br);
  const int c = brick_cell(x, y, z);
  if (!((m >> c) & 1ull)) return -1;
  return (int)br.z + (int)__popcll(m & ((1ull << c) - 1ull));
}
struct BrickFinder {
  BrickLevel L;
  __device__ inline int operator()(int b, int x, int y, int z) const { return brick_find(L, b, x, y, z); }
};

// ---- the rule-table kernels, one body for both grids (SubmanifoldConvolutionRules.h:26-45, ConvolutionRules.h:11-34) --
template <class Finder>
__global__ __launch_bounds__(256) void k_submanifold_table(const int32_t *__restrict__ site_coords, int64_t V, Finder find,
                                                           Filter3 fs, int32_t *__restrict__ table,
                                                           int32_t *__restrict__ counts) {
  const int k = blockIdx.y;
  int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int hit = 0;
  if (v < V) {
    int dz = k % fs.size[2], t = k / fs.size[2];
    int dy = t % fs.size[1], dx = t / fs.size[1];
    int4 c = *reinterpret_cast<const int4 *>(site_coords + 4 * v);
    // InputRegionCalculator_Submanifold: lb = p - size/2
    int x = c.x + dx - fs.size[0] / 2, y = c.y + dy - fs.size[1] / 2, z = c.z + dz - fs.size[2] / 2;
    const int r = find(c.w, x, y, z);
    table[(int64_t)k * V + v] = r;
    hit = r >= 0;
  }
  if (counts) block_count_store(hit, counts, k);
}

// table_out[k][o]: input row at o*stride + koff (InputRegionCalculator, :95-105)
template <class Finder>
__global__ __launch_bounds__(256) void k_conv_table_out(const int32_t *__restrict__ out_coords, int64_t V_out, Finder find,
                                                        ConvGeom g, int32_t *__restrict__ table,
                                                        int32_t *__restrict__ counts) {
  const int k = blockIdx.y;
  int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int hit = 0;
  if (o < V_out) {
    int dz = k % g.size[2], t = k / g.size[2];
    int dy = t % g.size[1], dx = t / g.size[1];
    int4 c = *reinterpret_cast<const int4 *>(out_coords + 4 * o);
    int x = c.x * g.stride[0] + dx, y = c.y * g.stride[1] + dy, z = c.z * g.stride[2] + dz;
    const int r = find(c.w, x, y, z);
    table[(int64_t)k * V_out + o] = r;
    hit = r >= 0;
  }
  if (counts) block_count_store(hit, counts, k);
}

// table_in[k][u]: the output row whose window holds input u at offset k, if that output cell
// is inside [0, out_spatial) (OutputRegionCalculator clamps, :109-119)
template <class Finder>
__global__ __launch_bounds__(256) void k_conv_table_in(const int32_t *__restrict__ in_coords, int64_t V_in, Finder find,
                                                       ConvGeom g, int32_t *__restrict__ table,
                                                       int32_t *__restrict__ counts) {
  const int k = blockIdx.y;
  int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int hit = 0;
  if (u < V_in) {
    int d[3];
    d[2] = k % g.size[2];
    int t = k / g.size[2];
    d[1] = t % g.size[1];
    d[0] = t / g.size[1];
    int4 c = *reinterpret_cast<const int4 *>(in_coords + 4 * u);
    int p[3] = {c.x, c.y, c.z}, j[3];
    bool ok = true;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      int q = p[i] - d[i];
      if (q < 0 || q % g.stride[i] != 0) { ok = false; break; }
      j[i] = q / g.stride[i];
      if (j[i] > g.out_sp[i] - 1) { ok = false; break; }
    }
    int r = -1;
    if (ok) r = find(c.w, j[0], j[1], j[2]);
    table[(int64_t)k * V_in + u] = r;
    hit = r >= 0;
  }
  if (counts) block_count_store(hit, counts, k);   // per 256-row block and offset, like k_conv_table_out
}

// ---- the same tables, one thread per ROW over all offsets (brick grids) --------------------------------------------------
// With a hash grid every (row, offset) is one independent random probe and a thread per (row, offset) -- the grid's y
// dimension over the offsets -- is the widest form.  With a brick grid the 27 look-ups of a row fall into <= 8 bricks that
// the thread (and its neighbours: rows are brick-major) has just touched, while the y-dimension form re-reads the site
// list once per offset (27 x 16 V bytes: PMC counted 120 MB of fetches for a 38 MB table when the L2s did not hold it).
// Here a thread reads its site once, walks the offsets and stores table[k][row] -- for a fixed k consecutive lanes store
// consecutive rows (coalesced) -- and the per-block counts come from one ballot per offset.
template <class Finder>
__global__ __launch_bounds__(256) void k_submanifold_table_rows(const int32_t *__restrict__ site_coords, int64_t V,
                                                                Finder find, Filter3 fs, int32_t *__restrict__ table,
                                                                int32_t *__restrict__ counts) {
  __shared__ int s_cnt[4][64];
  const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int vol = fs.size[0] * fs.size[1] * fs.size[2];
  int4 c = make_int4(0, 0, 0, 0);
  if (v < V) c = *reinterpret_cast<const int4 *>(site_coords + 4 * v);
  for (int k0 = 0; k0 < vol; k0 += 64) {
    int mine = 0;                                            // lane q keeps the wave's count of offset k0 + q
    const int kend = vol - k0 < 64 ? vol - k0 : 64;
    for (int q = 0; q < kend; ++q) {
      const int k = k0 + q;
      const int dz = k % fs.size[2], t = k / fs.size[2];
      const int dy = t % fs.size[1], dx = t / fs.size[1];
      int r = -1;
      if (v < V) {
        r = find(c.w, c.x + dx - fs.size[0] / 2, c.y + dy - fs.size[1] / 2, c.z + dz - fs.size[2] / 2);
        table[(int64_t)k * V + v] = r;
      }
      const int n = (int)__popcll(__ballot(r >= 0));
      if (lane == q) mine = n;
    }
    if (counts) {
      s_cnt[wave][lane] = mine;
      __syncthreads();
      if (threadIdx.x < kend)
        counts[(int64_t)(k0 + threadIdx.x) * gridDim.x + blockIdx.x] =
            s_cnt[0][threadIdx.x] + s_cnt[1][threadIdx.x] + s_cnt[2][threadIdx.x] + s_cnt[3][threadIdx.x];
      __syncthreads();
    }
  }
}

// MODE 0: table_out[k][o] = input row at o * stride + k (k_conv_table_out); MODE 1: table_in[k][u] = the output row whose
// window holds input u at offset k (k_conv_table_in).  vol <= 64 per pass of the offset loop, any vol.
template <class Finder, int MODE>
__global__ __launch_bounds__(256) void k_conv_table_rows(const int32_t *__restrict__ coords, int64_t V, Finder find,
                                                         ConvGeom g, int32_t *__restrict__ table,
                                                         int32_t *__restrict__ counts) {
  __shared__ int s_cnt[4][64];
  const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int vol = g.size[0] * g.size[1] * g.size[2];
  int4 c = make_int4(0, 0, 0, 0);
  if (v < V) c = *reinterpret_cast<const int4 *>(coords + 4 * v);
  for (int k0 = 0; k0 < vol; k0 += 64) {
    int mine = 0;
    const int kend = vol - k0 < 64 ? vol - k0 : 64;
    for (int q = 0; q < kend; ++q) {
      const int k = k0 + q;
      int d[3];
      d[2] = k % g.size[2];
      const int t = k / g.size[2];
      d[1] = t % g.size[1];
      d[0] = t / g.size[1];
      int r = -1;
      if (v < V) {
        if (MODE == 0) {
          r = find(c.w, c.x * g.stride[0] + d[0], c.y * g.stride[1] + d[1], c.z * g.stride[2] + d[2]);
        } else {
          const int p[3] = {c.x, c.y, c.z};
          int j[3];
          bool ok = true;
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            const int qq = p[i] - d[i];
            if (qq < 0 || qq % g.stride[i] != 0) { ok = false; break; }
            j[i] = qq / g.stride[i];
            if (j[i] > g.out_sp[i] - 1) { ok = false; break; }
          }
          if (ok) r = find(c.w, j[0], j[1], j[2]);
        }
        table[(int64_t)k * V + v] = r;
      }
      const int n = (int)__popcll(__ballot(r >= 0));
      if (lane == q) mine = n;
    }
    if (counts) {
      s_cnt[wave][lane] = mine;
      __syncthreads();
      if (threadIdx.x < kend)
        counts[(int64_t)(k0 + threadIdx.x) * gridDim.x + blockIdx.x] =
            s_cnt[0][threadIdx.x] + s_cnt[1][threadIdx.x] + s_cnt[2][threadIdx.x] + s_cnt[3][threadIdx.x];
      __syncthreads();
    }
  }
}

static inline dim3 grid1(int64_t n, int bs) { return dim3((unsigned)ceil_div(n > 0 ? n : 1, bs)); }

} // namespace aabr
