// conv_narrow.hip -- the 32 -> 32 plane layers of the finest scales (gfx950).
//
// Reference: the per-offset gather / matmul / scatter-add loop of SCN/CPU/Convolution.cpp:117-185 (submanifold and
// strided rule books alike; input-gradient form :46-79), `out[o] += in[i] @ W[k]` over the rules (i, o) of offset k.
//
// The 64-row-tile kernels of conv.hip stream 2-4 KiB of packed weights per 16-pair block for 1-2 KiB of gathered rows,
// and the wide kernel (conv_wide.hip) wants 64-column slabs and 128-byte row chunks.  A 32 x 32 layer is small enough for
// the opposite arrangement, which fits nowhere else:
//   * ALL filter offsets' weights (27 x 32 x 32: 54 KiB as bf16, 108 KiB as fp32) sit in LDS for the whole launch, in
//     MFMA operand order (one conflict-free ds_read_b128 per lane and operand);
//   * a wave owns 16 consecutive output rows x 32 columns in 8 accumulator registers over all offsets: no LDS tile, no
//     scatter-add, no block stream -- the gather TABLE [vol][rows] the rule-book builders leave behind is the input
//     (a wave reads its 16 entries per offset as one 64-byte line);
//   * the gathered rows go straight from global memory into the MFMA operand layout: lane (r, g) loads channels
//     8g .. 8g+7 of the input row of output row r -- one 16-byte load per lane and offset (two for fp32 rows), absent
//     neighbours masked, an offset none of the 16 rows has skipped wave-uniformly;
//   * one workgroup of 16 waves per CU, persistent over the row groups.
// Each output element is formed as bias + sum over the offsets in ascending order of the 32-channel dot product, the
// MFMA chain running through the accumulator (fp32 throughout; bf16 storage rounds once at the store).
#include "common.h"

namespace aabr {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8n __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4n __attribute__((ext_vector_type(4)));
extern thread_local const char *g_last_variant; // conv.hip
// (by value: __builtin_bit_cast applied to a vector ELEMENT expression reads element 0 whatever the subscript)
__device__ inline float bcfn(unsigned int v) { return __builtin_bit_cast(float, v); }

constexpr int kNarrowC = 32;       // planes in and out
constexpr int kNarrowVol = 28;     // filter offsets at most (their entries live in registers)
constexpr int kNarrowThreads = 1024;

// LDS image of the weights: bf16 [vol][2 column blocks][64 lanes][8 values = 16 B]; fp32 [vol][2][2 halves][64 lanes][4 values]
// (every read is 64 consecutive 16-byte granules).
// Lane (c = lane % 16, g = lane / 16) of column block cb holds, for out column n = cb * 16 + c, the weights of the 8 input
// channels it multiplies: bf16 (v_mfma_f32_16x16x32_bf16, A operand M = column, K = channel): channels 8g .. 8g+7;
// fp32 (eight v_mfma_f32_16x16x4_f32, MFMA t takes K slots {channel 8g' + t : g' = 0..3}): the same 8 channels, value t
// used by MFMA t -- the gathered row is loaded in exactly that grouping, so any channel order inside a lane is a
// consistent relabelling of K.
// STATS (bf16 storage; the BatchNorm around the layer takes its statistics from this launch's write-out as it does from
// k_conv_cs'): 1 = forward sums (v, v^2) of the STORED values; 2 = backward sums of the BatchNorm whose d_out this launch
// writes: (d, (x - mean) d), d = stored value masked by the sign of that BatchNorm's stored output (the terms of
// k_bn_partials<1, ., bf16>).  One [2][32] fp64 pair per WORKGROUP (parts[blockIdx.x]): rows summed over a wave's 16 lanes,
// then over its groups, then over the 16 waves in wave order -- fixed, whatever ran when.
struct NarrowStats {
  double *parts;
  const __bf16 *x, *y;      // STATS == 2: the BatchNorm's input and stored output
  const float *mean;
  float leak;
};

template <bool BF, int STATS = 0>
__global__ __launch_bounds__(kNarrowThreads, 1) void k_conv_narrow(const void *__restrict__ in_, int64_t rows_in,
                                                                   void *__restrict__ out_, int64_t V_out,
                                                                   const int32_t *__restrict__ table, int vol,
                                                                   const float *__restrict__ W, const float *__restrict__ bias,
                                                                   int flags, NarrowStats ns) {
  static_assert(STATS == 0 || BF, "write-out statistics: bf16 storage");
  extern __shared__ __align__(16) unsigned char smem_raw[];
  constexpr int ES = BF ? 2 : 4;                 // bytes per stored feature
  constexpr int NB = BF ? 7 : 4;                 // offsets per gather batch (two batches in flight)
  constexpr int NK = 28;                         // entry slots (vol <= 28 here; 28 .. 32 take the tile kernels)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c16 = lane & 15, g = lane >> 4;
  // ---- weights -> LDS (once per workgroup): W read in its own order (coalesced), scattered into operand order ----
  {
    const int transpose = flags & 1, mirror = (flags >> 1) & 1;
    const int total = vol * kNarrowC * kNarrowC;              // = vol x 1024: thread t handles element t of every W[kW]
#pragma unroll 9
    for (int i = threadIdx.x; i < total; i += kNarrowThreads) {
      const int q = i & 31, p = (i >> 5) & 31, kW = i >> 10;      // W[kW][p][q]
      const int k = mirror ? vol - 1 - kW : kW;
      const int ch = transpose ? q : p, n = transpose ? p : q;     // out[n] += in[ch] * w
      const int cb = n >> 4, ln = (ch >> 3) * 16 + (n & 15), t = ch & 7;
      const float w = W[i];
      if (BF) reinterpret_cast<__bf16 *>(smem_raw)[((k * 2 + cb) * 64 + ln) * 8 + t] = (__bf16)w;
      else reinterpret_cast<float *>(smem_raw)[(((k * 2 + cb) * 2 + (t >> 2)) * 64 + ln) * 4 + (t & 3)] = w;  // [half][lane][4]
    }
  }
  // behind the weights: [16 waves][2][32] fp64 statistics accumulators
  double *sacc = reinterpret_cast<double *>(smem_raw + (size_t)vol * 2 * 64 * (BF ? 16 : 32));
  if (STATS)
    for (int i = threadIdx.x; i < 16 * 2 * kNarrowC; i += kNarrowThreads) sacc[i] = 0.0;
  __syncthreads();
  // gathers through a buffer descriptor (32-bit offsets, bounds-checked): an absent neighbour's lane gets an offset past the
  // end (the matrix is < 2 GiB) -- the hardware returns zeros without touching memory and the load stays unconditional
  const unsigned rowbytes = kNarrowC * ES;
  const __amdgpu_buffer_rsrc_t rin =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(in_), 0, (int)(rows_in * rowbytes), 0x00020000);
  const int64_t ngroups = (V_out + 15) / 16;
  const int64_t gstride = (int64_t)gridDim.x * 16;
  auto load_entries = [&](int (&e)[NK], int64_t grp) {
    const int64_t row = grp * 16 + c16;
    const int64_t rowc = row < V_out ? row : V_out - 1;      // (clamped: the load is unconditional, the value is not)
#pragma unroll
    for (int k = 0; k < NK; ++k) e[k] = table[(int64_t)(k < vol ? k : vol - 1) * V_out + rowc];
  };
  int64_t grp = (int64_t)blockIdx.x * 16 + wave;
  int ent[NK];
  for (; grp < ngroups; grp += gstride) {
    load_entries(ent, grp);
    const int64_t row = grp * 16 + c16;
    const bool live = row < V_out;
    // which offsets ANY of the 16 rows has (wave-uniform): only those are gathered and multiplied
    unsigned m = 0;
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      if (!live || k >= vol) ent[k] = -1;
      m |= (__ballot(ent[k] >= 0) != 0ull ? 1u : 0u) << k;
    }
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    u32x4 a0[2][NB], a1[2][NB];
    // batches of NB offsets, two in flight: batch b + 1 is requested before batch b is multiplied; an offset none of the
    // 16 rows has is neither gathered nor multiplied (wave-uniform)
    auto issue = [&](int b) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int k = b * NB + j;
        if (k < NK && ((m >> k) & 1u)) {
          const unsigned off = ent[k] >= 0 ? (unsigned)ent[k] * rowbytes + (unsigned)g * (8u * ES) : 0x80000000u;
          a0[b & 1][j] = __builtin_amdgcn_raw_buffer_load_b128(rin, off, 0, 0);
          if (!BF) a1[b & 1][j] = __builtin_amdgcn_raw_buffer_load_b128(rin, off, 16, 0);
        }
      }
    };
    auto consume = [&](int b) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int k = b * NB + j;
        if (k < NK && ((m >> k) & 1u)) {
#pragma unroll
          for (int cb = 0; cb < 2; ++cb) {
            if (BF) {
              const u32x4 w = *reinterpret_cast<const u32x4 *>(smem_raw + (size_t)((k * 2 + cb) * 64 + lane) * 16);
              acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8n, w),
                                                                __builtin_bit_cast(bf16x8n, a0[b & 1][j]), acc[cb], 0, 0, 0);
            } else {
              const unsigned char *wq = smem_raw + (size_t)(k * 2 + cb) * (2 * 64 * 16) + (size_t)lane * 16;
              const u32x4 w0 = *reinterpret_cast<const u32x4 *>(wq), w1 = *reinterpret_cast<const u32x4 *>(wq + 64 * 16);
#pragma unroll
              for (int t = 0; t < 4; ++t)
                acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(bcfn(w0[t]), bcfn(a0[b & 1][j][t]), acc[cb], 0, 0, 0);
#pragma unroll
              for (int t = 0; t < 4; ++t)
                acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(bcfn(w1[t]), bcfn(a1[b & 1][j][t]), acc[cb], 0, 0, 0);
            }
          }
        }
      }
    };
    constexpr int NBATCH = NK / NB;
    issue(0);
#pragma unroll
    for (int b = 0; b < NBATCH; ++b) {
      if (b + 1 < NBATCH) issue(b + 1);
      consume(b);
    }
    // lane (c16, g) holds, per column block, columns cb * 16 + 4 g .. + 3 of output row `row`
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      const int n0 = cb * 16 + g * 4;
      double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
      if (live) {
        f32x4 v = acc[cb];
        if (bias) { v[0] += bias[n0]; v[1] += bias[n0 + 1]; v[2] += bias[n0 + 2]; v[3] += bias[n0 + 3]; }
        if (BF) {
          const bf16x4n o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
          *reinterpret_cast<bf16x4n *>(reinterpret_cast<__bf16 *>(out_) + row * kNarrowC + n0) = o;
          if (STATS == 1) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              const float sv = (float)o[t];
              s1[t] = (double)sv;
              s2[t] = (double)sv * (double)sv;
            }
          } else if (STATS == 2) {
            const bf16x4n xv = *reinterpret_cast<const bf16x4n *>(ns.x + row * kNarrowC + n0);
            const bf16x4n yv = *reinterpret_cast<const bf16x4n *>(ns.y + row * kNarrowC + n0);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              const float sv = (float)o[t];
              const float d = ((float)yv[t] > 0.0f) ? sv : sv * ns.leak;
              s1[t] = (double)d;
              s2[t] = (double)((float)xv[t] - ns.mean[n0 + t]) * (double)d;
            }
          }
        } else {
          *reinterpret_cast<f32x4 *>(reinterpret_cast<float *>(out_) + row * kNarrowC + n0) = v;
        }
      }
      if (STATS) {   // the 16 rows of the group (lanes of equal g), then this wave's accumulators
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
          for (int sh = 1; sh < 16; sh <<= 1) {
            s1[t] += __shfl_xor(s1[t], sh);
            s2[t] += __shfl_xor(s2[t], sh);
          }
          if (c16 == 0) {
            sacc[(wave * 2 + 0) * kNarrowC + n0 + t] += s1[t];
            sacc[(wave * 2 + 1) * kNarrowC + n0 + t] += s2[t];
          }
        }
      }
    }
  }
  if (STATS) {
    __syncthreads();
    if (threadIdx.x < 2 * kNarrowC) {
      const int sidx = threadIdx.x >> 5, col = threadIdx.x & 31;
      double a = 0.0;
      for (int w = 0; w < 16; ++w) a += sacc[(w * 2 + sidx) * kNarrowC + col];
      ns.parts[((int64_t)blockIdx.x * 2 + sidx) * kNarrowC + col] = a;
    }
  }
}

} // namespace aabr
using namespace aabr;

// 1: aabr_conv_forward_narrow[_bf16] takes this launch.  Measured on the bench's rule books (round 4, profiles/
// r04_conv_narrow_ab.txt): bf16 storage 31 us at 310 k rows against 30 us for the 64-row-tile kernel -- and 107 against 156 us
// at 890 k rows, where that kernel's per-block weight stream falls out of the L2s; fp32 storage 64 us against 48 at 310 k
// rows (twice the gathered bytes through the same number of waves).  So: bf16 storage from 400,000 output rows on.
extern "C" int aabr_conv_narrow_ok(int n_in, int n_out, int64_t rows_in, int64_t V_out, int vol, int bf16) {
  if (n_in != kNarrowC || n_out != kNarrowC || vol <= 0 || vol > kNarrowVol || V_out <= 0) return 0;
  if (rows_in <= 0 || rows_in * kNarrowC * 4 >= (1ll << 31)) return 0;   // (32-bit buffer offsets)
  {                                                // tuning experiments / tests only: 0 = never, 1 = whenever supported
    const int v = knob(K_CONV_NARROW);
    if (v == 0) return 0;
    if (v == 1) return 1;
  }
  return (bf16 && V_out >= 400000) ? 1 : 0;
}

static unsigned narrow_grid(int64_t V_out) {
  const int64_t ngroups = (V_out + 15) / 16;
  const int64_t want = (ngroups + 15) / 16;
  return (unsigned)(want < 256 ? want : 256);
}

// workgroups (= statistics parts) of the launch over V_out rows
extern "C" int aabr_conv_narrow_parts(int64_t V_out) { return V_out > 0 ? (int)narrow_grid(V_out) : 0; }

template <bool BF, int STATS>
static int conv_narrow_launch(const void *in, int64_t rows_in, void *out, int64_t V_out, const int32_t *table, int vol,
                              const float *W, const float *bias, int flags, NarrowStats ns, void *stream_) {
  AABR_CHECK_ARG(vol > 0 && vol <= kNarrowVol && V_out >= 0 && rows_in >= 0, "bad sizes (vol <= 28)");
  if (V_out == 0) return AABR_OK;
  AABR_CHECK_ARG(in && out && table && W && rows_in > 0, "null pointer / empty input");
  AABR_CHECK_ARG(((uintptr_t)in & 15) == 0 && ((uintptr_t)out & 15) == 0, "feature pointers must be 16-byte aligned");
  AABR_CHECK_ARG((flags & ~3) == 0, "flags: bit 0 transposed weights, bit 1 mirrored offsets");
  AABR_CHECK_ARG(rows_in * kNarrowC * (BF ? 2 : 4) < (1ll << 31), "input matrix must be < 2 GiB");
  const size_t lds = (size_t)vol * 2 * 64 * (BF ? 16 : 32) + (STATS ? 16 * 2 * kNarrowC * sizeof(double) : 0);
  static DynLdsOnce attr;
  AABR_CHECK_HIP(dyn_lds_once(attr, (const void *)(k_conv_narrow<BF, STATS>),
                              kNarrowVol * 2 * 64 * 32 + 16 * 2 * kNarrowC * (int)sizeof(double)));
  g_last_variant = BF ? (STATS == 2 ? "k_conv_narrow<bf16,bwd_stats>" : STATS == 1 ? "k_conv_narrow<bf16,stats>" : "k_conv_narrow<bf16>")
                      : "k_conv_narrow<f32>";
  hipLaunchKernelGGL((k_conv_narrow<BF, STATS>), dim3(narrow_grid(V_out)), dim3(kNarrowThreads), lds, (hipStream_t)stream_, in,
                     rows_in, out, V_out, table, vol, W, bias, flags, ns);
  AABR_CHECK_LAUNCH();
  return AABR_OK;
}

extern "C" int aabr_conv_forward_narrow(const float *in_feats, int64_t rows_in, float *out_feats, int64_t V_out,
                                        const int32_t *table, int vol, const float *W, const float *bias, int flags,
                                        void *stream) {
  return conv_narrow_launch<false, 0>(in_feats, rows_in, out_feats, V_out, table, vol, W, bias, flags, NarrowStats{}, stream);
}

extern "C" int aabr_conv_forward_narrow_bf16(const uint16_t *in_feats, int64_t rows_in, uint16_t *out_feats,
                                             int64_t V_out, const int32_t *table, int vol, const float *W,
                                             const float *bias, int flags, void *stream) {
  return conv_narrow_launch<true, 0>(in_feats, rows_in, out_feats, V_out, table, vol, W, bias, flags, NarrowStats{}, stream);
}

// ... with the forward statistics of the stored values: parts [aabr_conv_narrow_parts(V_out)][2][32] fp64 (layout of
// aabr_conv_forward_wide_bf16_stats, consumer aabr_bn_forward_parts_bf16)
extern "C" int aabr_conv_forward_narrow_bf16_stats(const uint16_t *in_feats, int64_t rows_in, uint16_t *out_feats,
                                                   int64_t V_out, const int32_t *table, int vol, const float *W,
                                                   const float *bias, int flags, double *stats, void *stream) {
  AABR_CHECK_ARG(stats && ((uintptr_t)stats & 7) == 0, "statistics buffer");
  NarrowStats ns{stats, nullptr, nullptr, nullptr, 0.0f};
  return conv_narrow_launch<true, 1>(in_feats, rows_in, out_feats, V_out, table, vol, W, bias, flags, ns, stream);
}

// ... with the BACKWARD statistics of the BatchNorm whose d_out the launch writes (as aabr_conv_forward_wide_bf16_bwd_stats;
// consumer aabr_bn_backward_parts_bf16)
extern "C" int aabr_conv_forward_narrow_bf16_bwd_stats(const uint16_t *in_feats, int64_t rows_in, uint16_t *out_feats,
                                                       int64_t V_out, const int32_t *table, int vol, const float *W,
                                                       const float *bias, int flags, double *stats, const uint16_t *bn_in,
                                                       const uint16_t *bn_out, const float *save_mean, float leakiness,
                                                       void *stream) {
  AABR_CHECK_ARG(stats && ((uintptr_t)stats & 7) == 0 && bn_in && bn_out && save_mean, "null pointer");
  AABR_CHECK_ARG((((uintptr_t)bn_in | (uintptr_t)bn_out) & 7) == 0, "the BatchNorm's input / output must be 8-byte aligned");
  NarrowStats ns{stats, reinterpret_cast<const __bf16 *>(bn_in), reinterpret_cast<const __bf16 *>(bn_out), save_mean, leakiness};
  return conv_narrow_launch<true, 2>(in_feats, rows_in, out_feats, V_out, table, vol, W, bias, flags, ns, stream);
}
