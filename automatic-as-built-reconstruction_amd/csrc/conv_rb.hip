// conv_rb.hip -- wide layers in bf16 storage with REGISTER accumulators over 256-row tiles (gfx950, round 5).
//
// Reference: the per-offset gather / matmul / scatter-add loop of SCN/CPU/Convolution.cpp:117-185 (input-gradient form
// :46-79): out[o] += in[i] @ W[k] over the rules (i, o) of offset k.
//
// What bounds k_conv_cs in bf16 storage (conv_wide.hip; DESIGN "bf16 storage"): every 64-row tile streams all 27 offsets'
// weights -- 27 x 32 KiB at 128 -> 128 planes -- through its CU's vector-memory path (~70 GB/s per CU for L2-resident
// lines, MI355X_MICROARCH.md "Indexed rows"): 864 KiB per 64 output rows, 1.1 GB per launch at 84 k rows = ~60 us of a
// 98 us launch before a single row is gathered; and every 16-pair block pays a gather -> stage -> LDS read-add-write
// chain.  With brick-major rows (csrc/brick.hip) the rows of a tile are spatial neighbours, so the work can be cut the
// other way:
//   * a workgroup owns 256 CONSECUTIVE output rows x all output columns; each of its 8 waves keeps 32 rows x n_out columns
//     in accumulator registers over the whole sweep -- no LDS tile, no read-add-write, no block stream: the gather TABLE
//     [vol][rows] the rule-book builders leave behind is the input (a tile's entries sit in LDS);
//   * an offset's weights are fetched ONCE per 256 rows (a quarter of the bytes per output row), by LDS-DMA
//     (global_load_lds_dwordx4: no registers, lane-linear MFMA-operand image, double-buffered: offset s + 1 lands while
//     offset s is multiplied) and read from LDS by all 8 waves, each fragment feeding the wave's two 16-row blocks;
//   * the partner rows go straight from L2 into the MFMA operand layout (lane (r, g): channels 32 c + 8 g .. + 7 of the
//     partner of row r), requested one of the wave's own active offsets ahead; an absent partner is an out-of-range
//     buffer offset (zeros, no branch); an offset none of a wave's 32 rows has is skipped by that wave, an offset none
//     of the tile's 256 rows has is not even staged -- spatial neighbours share their empty offsets.
// Sum order per output element: offsets ascending, 32-channel chunks ascending inside an offset, fp32 accumulation, ONE
// rounding to bf16 at the store -- fixed, run-to-run identical.
#include "common.h"

namespace aabr {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8r __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4r __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void rb_lds_void;
typedef __attribute__((address_space(1))) const void rb_g_void;
extern thread_local const char *g_last_variant; // conv.hip

constexpr int kRbWaves = 8, kRbThreads = 64 * kRbWaves, kRbTile = 32 * kRbWaves, kRbMaxVol = 27;

// KC = n_in / 32 (32-channel chunks), NNB = n_out / 16 (column blocks).  wpack: the bf16 pack of aabr_conv_pack_weights2_bf16 /
// aabr_conv_pack_weights_jobs: [kW][chunk][column block][lane][8 bf16] -- one offset = KC * NNB KiB, contiguous.
template <int KC, int NNB>
__global__ __launch_bounds__(kRbThreads, 1) void k_conv_rb(const __bf16 *__restrict__ in, int64_t rows_in,
                                                          __bf16 *__restrict__ out, int64_t V_out,
                                                          const int32_t *__restrict__ table, int vol,
                                                          const unsigned char *__restrict__ wpack,
                                                          const float *__restrict__ bias, int mirror, int ntiles) {
  extern __shared__ __align__(16) unsigned char smem[];
  constexpr int WB = KC * NNB * 1024;                       // bytes of one offset's weights
  constexpr int ROWB = KC * 64;                             // bytes of a feature row
  constexpr int NOUT = NNB * 16;
  unsigned char *wbuf = smem;                               // [2][WB]
  int32_t *ent = reinterpret_cast<int32_t *>(smem + 2 * WB);  // [vol][256]
  __shared__ unsigned s_mask[kRbWaves];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c16 = lane & 15, g = lane >> 4;
  const __amdgpu_buffer_rsrc_t rin =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16 *>(in), 0, (int)(rows_in * ROWB), 0x00020000);
  // XCD-contiguous tiles: workgroup ids go round-robin over the 8 XCDs, so XCD x takes the x-th eighth of the tiles and
  // its workgroups stride through it -- the rows a tile gathers are its spatial neighbours, i.e. the next tiles' rows
  const int nper = (int)gridDim.x >> 3, xcd = (int)blockIdx.x & 7, jx = (int)blockIdx.x >> 3;
  const int tpx = (ntiles + 7) >> 3;
  struct ASet { u32x4 a[2][KC]; };
  for (int ti = jx; ti < tpx; ti += nper) {
    const int tile = xcd * tpx + ti;
    if (tile >= ntiles) break;
    const int64_t row0 = (int64_t)tile * kRbTile;
    __syncthreads();                                         // the previous tile's readers of ent / wbuf / s_mask are done
    {                                                        // the tile's table entries -> LDS: all loads out, then the stores
      constexpr int NE = (kRbMaxVol * kRbTile + kRbThreads - 1) / kRbThreads;
      int32_t ev[NE];
#pragma unroll
      for (int j = 0; j < NE; ++j) {
        const int i = j * kRbThreads + (int)threadIdx.x;
        const int k = i >> 8, r = i & (kRbTile - 1);
        const int64_t row = row0 + r;
        ev[j] = (k < vol && row < V_out) ? table[(int64_t)k * V_out + row] : -1;
      }
#pragma unroll
      for (int j = 0; j < NE; ++j) {
        const int i = j * kRbThreads + (int)threadIdx.x;
        if (i < vol * kRbTile) ent[i] = ev[j];
      }
    }
    __syncthreads();
    const int lrA = wave * 32 + c16, lrB = lrA + 16;
    unsigned m = 0;
    for (int k = 0; k < vol; ++k) {
      const bool has = ent[k * kRbTile + lrA] >= 0 || ent[k * kRbTile + lrB] >= 0;
      m |= (__ballot(has) != 0ull ? 1u : 0u) << k;
    }
    if (lane == 0) s_mask[wave] = m;
    __syncthreads();
    unsigned wg = 0;
#pragma unroll
    for (int w = 0; w < kRbWaves; ++w) wg |= s_mask[w];
    f32x4 acc[2][NNB];
#pragma unroll
    for (int cb = 0; cb < NNB; ++cb) { acc[0][cb] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[1][cb] = acc[0][cb]; }

    auto stage = [&](int k, int b) {                         // offset k's weights -> wbuf[b] by LDS-DMA (1 KiB per wave-instruction)
      const int kW = mirror ? vol - 1 - k : k;
      const unsigned char *src = wpack + (int64_t)kW * WB;
#pragma unroll
      for (int j = 0; j < WB / (kRbThreads * 16); ++j) {
        const int piece = j * kRbThreads + (int)threadIdx.x;
        __builtin_amdgcn_global_load_lds((rb_g_void *)(src + (int64_t)piece * 16),
                                         (rb_lds_void *)(wbuf + b * WB + (j * kRbThreads + wave * 64) * 16), 16, 0, 0);
      }
    };
    auto gather = [&](ASet &s, int k) {
      const int eA = ent[k * kRbTile + lrA], eB = ent[k * kRbTile + lrB];
      const unsigned oA = eA >= 0 ? (unsigned)eA * (unsigned)ROWB + (unsigned)g * 16u : 0x80000000u;
      const unsigned oB = eB >= 0 ? (unsigned)eB * (unsigned)ROWB + (unsigned)g * 16u : 0x80000000u;
#pragma unroll
      for (int c = 0; c < KC; ++c) {
        s.a[0][c] = __builtin_amdgcn_raw_buffer_load_b128(rin, oA, c * 64, 0);
        s.a[1][c] = __builtin_amdgcn_raw_buffer_load_b128(rin, oB, c * 64, 0);
      }
    };
    // the weight fragments of step u + 1 leave LDS while step u is multiplied (a step = one 32-channel chunk x HB column
    // blocks; two register sets in static alternation).  sched_barrier: without it the compiler sinks every fragment read
    // to just in front of its MFMAs and waits lgkmcnt(0) each time -- an LDS round trip per four MFMAs.
    auto compute = [&](const ASet &s, const unsigned char *wb) {
      constexpr int HB = NNB > 4 ? NNB / 2 : NNB, NH = NNB / HB, U = KC * NH;
      static_assert(U % 2 == 0, "steps come in pairs");
      u32x4 w0[HB], w1[HB];
      auto ldw = [&](u32x4 (&w)[HB], int u) {
        const int c = u / NH, h = u % NH;
#pragma unroll
        for (int q = 0; q < HB; ++q)
          w[q] = *reinterpret_cast<const u32x4 *>(wb + ((c * NNB + h * HB + q) * 64 + lane) * 16);
      };
      auto mm = [&](const u32x4 (&w)[HB], int u) {
        const int c = u / NH, h = u % NH;
#pragma unroll
        for (int q = 0; q < HB; ++q) {
          const int cb = h * HB + q;
          acc[0][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8r, w[q]),
                                                               __builtin_bit_cast(bf16x8r, s.a[0][c]), acc[0][cb], 0, 0, 0);
          acc[1][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8r, w[q]),
                                                               __builtin_bit_cast(bf16x8r, s.a[1][c]), acc[1][cb], 0, 0, 0);
        }
      };
      ldw(w0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < U; u += 2) {
        ldw(w1, u + 1);
        __builtin_amdgcn_sched_barrier(0);
        mm(w0, u);
        __builtin_amdgcn_sched_barrier(0);
        if (u + 2 < U) ldw(w0, u + 2);
        __builtin_amdgcn_sched_barrier(0);
        mm(w1, u + 1);
        __builtin_amdgcn_sched_barrier(0);
      }
    };

    if (wg) {
      unsigned rem = wg, mrem = m;
      int ks = __ffs(rem) - 1; rem &= rem - 1;
      int p = mrem ? __ffs(mrem) - 1 : -1; mrem &= mrem ? mrem - 1 : 0;
      ASet cur, nxt;
#pragma unroll
      for (int c = 0; c < KC; ++c) { cur.a[0][c] = cur.a[1][c] = nxt.a[0][c] = nxt.a[1][c] = (u32x4){0u, 0u, 0u, 0u}; }
      stage(ks, 0);
      if (p >= 0) gather(cur, p);
      __syncthreads();                                       // (vmcnt(0): the weights and the first rows have landed)
      int b = 0;
      for (;;) {
        const int kn = rem ? __ffs(rem) - 1 : -1;
        rem &= rem ? rem - 1 : 0;
        if (kn >= 0) stage(kn, b ^ 1);                       // lands while this offset is multiplied
        const bool mine = ks == p;                           // (wave-uniform)
        int pn = -1;
        if (mine) {
          pn = mrem ? __ffs(mrem) - 1 : -1;
          mrem &= mrem ? mrem - 1 : 0;
          if (pn >= 0) gather(nxt, pn);                      // this wave's next active offset, one step ahead
          compute(cur, wbuf + b * WB);
        }
        __syncthreads();
        if (mine) { cur = nxt; p = pn; }
        if (kn < 0) break;
        ks = kn;
        b ^= 1;
      }
    }
    // lane (c16, g) holds, per row block and column block, columns cb * 16 + 4 g .. + 3 of its row
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      const int64_t row = row0 + wave * 32 + rb * 16 + c16;
      if (row < V_out) {
#pragma unroll
        for (int cb = 0; cb < NNB; ++cb) {
          const int n0 = cb * 16 + g * 4;
          f32x4 v = acc[rb][cb];
          if (bias) { v[0] += bias[n0]; v[1] += bias[n0 + 1]; v[2] += bias[n0 + 2]; v[3] += bias[n0 + 3]; }
          const bf16x4r o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
          *reinterpret_cast<bf16x4r *>(out + row * NOUT + n0) = o;
        }
      }
    }
  }
}

} // namespace aabr
using namespace aabr;

// 1: aabr_conv_forward_rb_bf16 takes this launch (the caller reads the gather table, not a block stream)
extern "C" int aabr_conv_rb_ok(int n_in, int n_out, int64_t rows_in, int64_t V_out, int vol) {
  if (!((n_in == 64 || n_in == 128) && (n_out == 64 || n_out == 128))) return 0;
  if (vol <= 0 || vol > kRbMaxVol || V_out <= 0 || rows_in <= 0) return 0;
  if (rows_in * n_in * 2 >= (1ll << 31)) return 0;             // 32-bit buffer offsets
  const int v = knob(K_CONV_RB);
  if (v == 0) return 0;
  if (v == 1) return 1;
  return 0;                                                    // (not dispatched by default yet: see DESIGN)
}

template <int KC, int NNB>
static int conv_rb_launch(const uint16_t *in, int64_t rows_in, uint16_t *out, int64_t V_out, const int32_t *table, int vol,
                          const uint16_t *wpack, const float *bias, int mirror, hipStream_t st) {
  const size_t lds = (size_t)2 * KC * NNB * 1024 + (size_t)vol * kRbTile * sizeof(int32_t);
  static DynLdsOnce attr;
  AABR_CHECK_HIP(dyn_lds_once(attr, (const void *)(k_conv_rb<KC, NNB>), 2 * KC * NNB * 1024 + kRbMaxVol * kRbTile * (int)sizeof(int32_t)));
  const int64_t ntiles = (V_out + kRbTile - 1) / kRbTile;
  int64_t grid = ntiles < 256 ? (ntiles + 7) / 8 * 8 : 256;   // one workgroup per CU, a multiple of the 8 XCDs
  hipLaunchKernelGGL((k_conv_rb<KC, NNB>), dim3((unsigned)grid), dim3(kRbThreads), lds, st,
                     reinterpret_cast<const __bf16 *>(in), rows_in, reinterpret_cast<__bf16 *>(out), V_out, table, vol,
                     reinterpret_cast<const unsigned char *>(wpack), bias, mirror, (int)ntiles);
  return AABR_OK;
}

extern "C" int aabr_conv_forward_rb_bf16(const uint16_t *in_feats, int n_in, int64_t rows_in, uint16_t *out_feats, int n_out,
                                         int64_t V_out, const int32_t *table, int vol, const float *bias, int flags,
                                         const uint16_t *wpack, void *stream_) {
  hipStream_t st = (hipStream_t)stream_;
  AABR_CHECK_ARG((n_in == 64 || n_in == 128) && (n_out == 64 || n_out == 128), "plane counts: 64 or 128 each way");
  AABR_CHECK_ARG(vol > 0 && vol <= kRbMaxVol && V_out >= 0 && rows_in >= 0, "bad sizes (vol <= 27)");
  if (V_out == 0) return AABR_OK;
  AABR_CHECK_ARG(in_feats && out_feats && table && wpack && rows_in > 0, "null pointer / empty input");
  AABR_CHECK_ARG((((uintptr_t)in_feats | (uintptr_t)out_feats | (uintptr_t)wpack) & 15) == 0, "pointers must be 16-byte aligned");
  AABR_CHECK_ARG((flags & ~3) == 0, "flags: bit 0 transposed weights (the caller passes that pack), bit 1 mirrored offsets");
  AABR_CHECK_ARG(rows_in * n_in * 2 < (1ll << 31) && V_out < (1ll << 31) - 256, "input matrix must be < 2 GiB");
  const int mirror = (flags >> 1) & 1;
  int rc;
  if (n_in == 128 && n_out == 128) { g_last_variant = "k_conv_rb<4,8>"; rc = conv_rb_launch<4, 8>(in_feats, rows_in, out_feats, V_out, table, vol, wpack, bias, mirror, st); }
  else if (n_in == 64 && n_out == 64) { g_last_variant = "k_conv_rb<2,4>"; rc = conv_rb_launch<2, 4>(in_feats, rows_in, out_feats, V_out, table, vol, wpack, bias, mirror, st); }
  else if (n_in == 64) { g_last_variant = "k_conv_rb<2,8>"; rc = conv_rb_launch<2, 8>(in_feats, rows_in, out_feats, V_out, table, vol, wpack, bias, mirror, st); }
  else { g_last_variant = "k_conv_rb<4,4>"; rc = conv_rb_launch<4, 4>(in_feats, rows_in, out_feats, V_out, table, vol, wpack, bias, mirror, st); }
  if (rc != AABR_OK) return rc;
  AABR_CHECK_LAUNCH();
  return AABR_OK;
}
