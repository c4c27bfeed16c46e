// conv_wide.hip -- the wide-layer form of the sparse convolution (gfx950): 128-row output tiles, output columns
// split across the waves of a workgroup, gathered rows shared through LDS, weights held in registers per offset.
//
// Why: in the 64-row-tile kernels of conv.hip every 16-pair block streams its own 2 KiB x NBW of packed weights
// per 32-channel chunk through the vector-memory path; on the layers that dominate the FPN_Net step (64..256
// planes, 3x3x3) that is ~5x the bytes of the gathered rows, and the L2->CU feed, not the MFMA pipe, bounds the
// kernel (34 % of the fp32 MFMA peak at 128->128).  Two other forms were built and measured on the way
// (profiles/r02_conv_wide_ab.txt): weights staged through LDS per (offset, chunk) with the
// waves in lockstep (slower: a barrier per ~1.5 blocks of work and 75 % balance), and columns split across waves
// with every wave gathering every row itself (slower: 4x the gather traffic misses the vector L1).  What is kept:
//
//   k_conv_cs  -- a workgroup (4 waves) owns 128 consecutive output rows x 64 output columns in LDS (32 KiB, XOR-
//   swizzled 16-byte granules so the read-add-write of a block's 16 rows is bank-conflict free); wave w owns column
//   block w.  Per filter offset a wave loads its 16-column weight slice ONCE into registers (8 KiB per 128 input
//   channels = 32 VGPRs) and reuses it for every block of that offset in the tile; per pair of 16-pair blocks the
//   workgroup gathers the 32 rows once into a double-buffered LDS stage (each wave a quarter; loads in flight
//   during the previous pair's MFMAs) and every wave reads its MFMA B-operands from there.  Waves never touch each
//   other's tile columns, every wave issues the same MFMAs (perfect balance), accumulation order is fixed (offset,
//   block) => bit-reproducible, no atomics.  One barrier per block pair; 64 KiB LDS => two workgroups per CU.
//
// Same contraction as conv.hip (reference: SCN/CPU/Convolution.cpp:45-185, SCN/CPU/Deconvolution.cpp:7-77):
//     out[o] = bias + sum_k in[table[k][o]] @ Wl[k]
// Requires n_in % 32 == 0 (above 128: % 128), n_out % 64 == 0, vol <= 63, rows_in < 2^23, buffers < 2 GiB.
#include "common.h"
#include <stdlib.h>

namespace aabr {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

extern thread_local const char *g_last_variant; // conv.hip

constexpr int kWS = 64;    // tile row stride in floats = slab width
constexpr int kNB = 4;     // 16-column blocks per slab
constexpr int kMaxVol = 63;  // vol + 1 prefix entries live in the lanes of one VGPR

// ------------------------------------------------------------------ compiled rule book, big-tile form
//   words: [ntiles][vol+1] block prefix per offset | [ntiles][(T/16)*vol][16] entries   (T rows per tile)
//   entry = (partner_row << 8) | local_row; padding entries repeat the block's first pair with bit 31 set.
// Pairs of one offset are in ascending local-row order (deterministic).
// T = rows per tile (a multiple of 16, <= 128); the workgroup has ceil(T/64) waves
__global__ __launch_bounds__(256) void k_build_tileT(const int32_t *__restrict__ table, int64_t V, int vol, int T,
                                                      int32_t *__restrict__ words) {
  const int kT = T;
  const int NWV = (T + 63) / 64;
  __shared__ int s_cnt[4][kMaxVol];
  __shared__ int s_base[kMaxVol + 1];
  __shared__ int s_tot[kMaxVol];
  __shared__ int s_first[kMaxVol];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t ntiles = (V + T - 1) / T, tile = blockIdx.x;
  const int maxb = (T / 16) * vol;
  int32_t *pre = words + tile * (vol + 1);
  int32_t *ent = words + ntiles * (vol + 1) + tile * (int64_t)maxb * 16;
  const int64_t row = tile * kT + threadIdx.x;
  const bool valid = (int)threadIdx.x < T && row < V;
  for (int k = 0; k < vol; ++k) {
    const int t = valid ? table[(int64_t)k * V + row] : -1;
    const unsigned long long m = __ballot(t >= 0);
    if (lane == 0) s_cnt[wave][k] = __popcll(m);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int b = 0;
    for (int k = 0; k < vol; ++k) {
      int tot = 0;
      for (int w = 0; w < NWV; ++w) tot += s_cnt[w][k];
      s_tot[k] = tot;
      s_base[k] = b;
      b += (tot + 15) >> 4;
    }
    s_base[vol] = b;
  }
  __syncthreads();
  if ((int)threadIdx.x <= vol) pre[threadIdx.x] = s_base[threadIdx.x];
  for (int k = 0; k < vol; ++k) {
    const int t = valid ? table[(int64_t)k * V + row] : -1;
    const unsigned long long m = __ballot(t >= 0);
    if (t >= 0) {
      int rank = __popcll(m & ((1ull << lane) - 1ull));
      for (int w = 0; w < wave; ++w) rank += s_cnt[w][k];
      const int e = (t << 8) | (int)threadIdx.x;
      ent[s_base[k] * 16 + rank] = e;
      if (rank == 0) s_first[k] = e;
    }
  }
  __syncthreads();
  for (int k = 0; k < vol; ++k) {
    const int tot = s_tot[k];
    if (tot == 0) continue;
    const int pad = ((tot + 15) & ~15) - tot;
    if ((int)threadIdx.x < pad) ent[s_base[k] * 16 + tot + threadIdx.x] = s_first[k] | (int)0x80000000;
  }
}

__device__ inline float bcf_(unsigned int v) { return __builtin_bit_cast(float, v); }

// ------------------------------------------------------------------ the kernel
// Per pair of blocks the workgroup gathers the 32 rows once (each wave a quarter, 4 x 16-byte loads per lane, in
// flight during the previous pair's MFMAs), stores them in a double-buffered, granule-swizzled stage (2 x 16 KiB)
// and every wave reads its MFMA B-operands from there with conflict-free ds_read_b128.  So per 16-pair block the
// vector-memory path carries the 8 KiB of gathered rows ONCE (64-row-tile kernels: 8 KiB + 32 KiB of weights), the
// weights cost 8 KiB per wave per OFFSET (registers), and every wave issues the same MFMAs.
// One barrier per block pair (~2 x 1024 MFMA cycles per wave) with the double-buffered stage (NBUF = 2), two with a
// single stage buffer (NBUF = 1: 49 KiB of LDS at 128-channel groups => three workgroups per CU instead of two).
constexpr int kMaxTileRows = 240; // (240 + 1) rows x 256 B + 16 KiB stage = 76 KiB: two workgroups per CU

// NBUF = LDS stage buffers: 2 (one barrier per pair, 2 workgroups per CU) or 1 (two barriers per pair, 3 per CU)
// BF: bf16 feature storage (extension): rows, stage and weight packs hold bf16, one v_mfma_f32_16x16x32_bf16 per 32
// channels; KG then counts 128-BYTE row chunks (64 channels), so the gather / stage code is the same byte for byte
typedef __bf16 bf16x8w __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4w __attribute__((ext_vector_type(4)));

template <int KG, int DBG, int NBUF, bool BF = false>
__global__ __launch_bounds__(256, NBUF == 2 ? 2 : 3) void k_conv_cs(const float *__restrict__ in, int ci, int64_t in_bytes,
                                                    float *__restrict__ out, int co, int64_t V_out,
                                                    const int32_t *__restrict__ words, int64_t words_bytes, int vol,
                                                    int wflip, const float *__restrict__ Wp, int64_t wp_bytes,
                                                    const float *__restrict__ bias, int kT2,
                                                    const float *__restrict__ res) {
  constexpr int RG = KG * 8;               // 16-byte granules per staged row
  constexpr int RF = KG * 32;              // floats per staged row
  constexpr int SWZ = (RG >= 16 && (RG & 15) == 0) ? 15 : 7; // XOR must stay inside the row's granules
  constexpr int STAGE = 2 * 16 * RF;       // floats per stage buffer (two blocks)
  extern __shared__ __align__(16) float smem[];
  float *Ct = smem;                        // [kT2 + 1][64] floats, granule-swizzled; the last row swallows padding entries
  float *St = smem + (kT2 + 1) * kWS;      // [2][2][16][RF]
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int g = lane >> 4, c16 = lane & 15;
  const int pr = wave * 8 + (lane >> 3), seg = lane & 7; // gather role: pair row 0..31, 16-byte segment
  const int nkc = BF ? ci >> 6 : ci >> 5, nnb = co >> 4;   // 128-byte chunks per row
  // Workgroups are dealt round-robin over the 8 XCDs (linear id % 8), each with its own 4 MiB L2.  Give every XCD a
  // CONTIGUOUS range of (tile, slab) work items instead of every eighth one: the slabs of a tile and the tiles next to
  // it gather the same input rows (consecutive sites lie on the same surface), so they hit that XCD's L2 instead of
  // all eight L2s fetching the whole input (round 2: 580 MB of fetches per launch for 91 MB of operands).  Speed
  // only -- which workgroup computes a tile does not change a bit of it.
  int64_t tile;
  int nb0;
  {
    const unsigned ny = gridDim.y, total = gridDim.x * ny;
    const unsigned lin = blockIdx.y * gridDim.x + blockIdx.x;
    const unsigned per = total >> 3, rem = total & 7u, x = lin & 7u;
    const unsigned wi = x * per + (x < rem ? x : rem) + (lin >> 3);
    tile = wi / ny;
    nb0 = (int)(wi % ny) * kNB;
  }
  const int64_t row0 = tile * kT2;
  const int64_t ntiles = (V_out + kT2 - 1) / kT2;
  const int maxb = (kT2 / 16) * vol;
  const int vpre = lane <= vol ? words[tile * (vol + 1) + lane] : 0;
  {
    f32x4 z = {0.f, 0.f, 0.f, 0.f};
    f32x4 *c4 = reinterpret_cast<f32x4 *>(Ct);
    for (int i = threadIdx.x; i < (kT2 + 1) * kWS / 4; i += 256) c4[i] = z;
  }
  const __amdgpu_buffer_rsrc_t rin =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(in), 0, (int)in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rwords =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t *>(words), 0, (int)words_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(Wp), 0, (int)wp_bytes, 0x00020000);
  const unsigned ebase = (unsigned)((ntiles * (vol + 1) + tile * (int64_t)maxb * 16) * 4);
  const unsigned rowbytes = (unsigned)ci * (BF ? 2u : 4u);
  const unsigned lane32 = (unsigned)lane * 32u;
  auto pre_of = [&](int k) { return __builtin_amdgcn_readlane(vpre, k); };
  auto next_offset = [&](int k) {
    ++k;
    while (k < vol && pre_of(k + 1) == pre_of(k)) ++k;
    return k;
  };
  struct WReg { u32x4 w0[KG], w1[KG]; };
  struct GReg { u32x4 v[KG]; };
  struct Ent { int eg, ea, eb, hb; };      // gather-role entry of this lane's pair row; compute-role entries (A, B);
                                           // hb (wave-uniform): the pair has a second block
  const int ngroups = (nkc + KG - 1) / KG;
  const int nblk_all = pre_of(vol);
  auto load_w = [&](WReg &w, int k, int kg) {
    const int kW = wflip ? vol - 1 - k : k;
#pragma unroll
    for (int c = 0; c < KG; ++c) { // nkc % KG == 0 (dispatch): every load is unconditional, so the compiler's
      const int kc = kg * KG + c;  // vmcnt bookkeeping stays exact and nothing waits for a prefetch it does not use
      if (BF) { // two 32-channel MFMA chunks per 128-byte row chunk, 1 KiB of packed weights each
        const unsigned so = (unsigned)((((int64_t)kW * (2 * nkc) + 2 * kc) * nnb + nb0 + wave) * 1024);
        w.w0[c] = __builtin_amdgcn_raw_buffer_load_b128(rw, (unsigned)lane * 16u, so, 0);
        w.w1[c] = __builtin_amdgcn_raw_buffer_load_b128(rw, (unsigned)lane * 16u, so + (unsigned)nnb * 1024u, 0);
      } else {
        const unsigned so = (unsigned)((((int64_t)kW * nkc + kc) * nnb + nb0 + wave) * 2048);
        w.w0[c] = __builtin_amdgcn_raw_buffer_load_b128(rw, lane32, so, 0);
        w.w1[c] = __builtin_amdgcn_raw_buffer_load_b128(rw, lane32 + 16u, so, 0);
      }
    }
  };
  // entries of the pair (bb, bb+1) of offset kk; the second block only if it belongs to the same offset,
  // otherwise block A again with the discard bit
  auto load_ent = [&](int bb, int kk) {
    Ent e;
    const bool hasB = bb + 1 < pre_of(kk + 1);
    const unsigned bA = (unsigned)bb * 64u, bB = (unsigned)(hasB ? bb + 1 : bb) * 64u;
    e.ea = (int)__builtin_amdgcn_raw_buffer_load_b32(rwords, (unsigned)c16 * 4u, ebase + bA, 0);
    e.eb = (int)__builtin_amdgcn_raw_buffer_load_b32(rwords, (unsigned)c16 * 4u, ebase + bB, 0);
    e.eg = (int)__builtin_amdgcn_raw_buffer_load_b32(rwords, (unsigned)(pr & 15) * 4u, ebase + (pr < 16 ? bA : bB), 0);
    if (!hasB) e.eb |= (int)0x80000000;
    e.hb = hasB ? 1 : 0;
    return e;
  };
  auto gather = [&](GReg &q, int eg, int kg) {
    const unsigned va = (((unsigned)eg & 0x7fffffffu) >> 8) * rowbytes + (unsigned)seg * 16u;
#pragma unroll
    for (int i = 0; i < KG; ++i) { // granule seg + 8 i of the row's current channel group
      const unsigned so = (unsigned)(kg * KG + i) * 128u;
      if (DBG & 2) q.v[i] = (u32x4){(unsigned)eg, 0u, 0u, 0u}; // timing experiments: no global gathers
      else q.v[i] = __builtin_amdgcn_raw_buffer_load_b128(rin, va, so, 0);
    }
  };
  auto stage_store = [&](const GReg &q, int buf) {
    float *rowp = St + (NBUF == 2 ? buf : 0) * STAGE + pr * RF;
#pragma unroll
    for (int i = 0; i < KG; ++i)
      *reinterpret_cast<u32x4 *>(rowp + (((seg + 8 * i) ^ (pr & SWZ)) << 2)) = q.v[i];
  };
  // branch-free: a padding entry (bit 31) lands in the dummy row; the two blocks of a pair never share a real row,
  // so both reads go out before either write
  auto accumulate2 = [&](int ea, const f32x4 &accA, int eb, const f32x4 &accB) {
    const int ra = ea >= 0 ? (ea & 255) : kT2, rb = eb >= 0 ? (eb & 255) : kT2;
    f32x4 *da = reinterpret_cast<f32x4 *>(Ct + ra * kWS + (((wave * 4 + g) ^ (ra & 15)) << 2));
    f32x4 *db = reinterpret_cast<f32x4 *>(Ct + rb * kWS + (((wave * 4 + g) ^ (rb & 15)) << 2));
    const f32x4 va = *da, vb = *db;
    *da = va + accA;
    *db = vb + accB;
  };
  // Software pipeline over the tile's block pairs (pairs never straddle an offset):
  //   entries two pairs ahead (registers), gathered rows one pair ahead (registers -> LDS stage after this pair's
  //   MFMAs), weights one offset ahead.  A gather never waits for an entry load issued in the same iteration.
  struct Pos { int b, k; };                // k >= vol: past the end
  auto adv = [&](Pos q) {
    if (q.k >= vol) return q;
    const int kend = pre_of(q.k + 1);
    q.b += 2;
    if (q.b >= kend) { q.b = kend; q.k = next_offset(q.k); }
    return q;
  };
  auto wg_barrier = [&]() {                // LDS traffic of this wave retired, then the workgroup barrier; unlike
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); // __syncthreads() it does not drain the prefetches
  };
  int par = 0;
  long long dbg_t[5] = {0, 0, 0, 0, 0};
  for (int kg = 0; kg < ngroups; ++kg) {
    Pos p0;
    p0.k = next_offset(-1);
    if (p0.k >= vol) break;
    p0.b = pre_of(p0.k);
    Pos p1 = adv(p0), p2 = adv(p1), p3 = adv(p2);
    auto ent_of = [&](const Pos &q) { const bool v = q.k < vol; return load_ent(v ? q.b : p0.b, v ? q.k : p0.k); };
    Ent e0 = ent_of(p0), e1 = ent_of(p1), e2 = ent_of(p2), e3 = e0;
    // gathered rows travel TWO pairs ahead of the MFMAs (a random-row gather takes ~2 us, one pair's MFMAs ~1 us):
    // two register sets, used alternately by the two copies (PAR) of the step body
    GReg gqA, gqB;
    gather(gqA, e0.eg, kg);
    __syncthreads();                       // zero fill done / previous group's stage reads done
    par = 0;                               // (the register-set roles below are tied to the stage parity)
    stage_store(gqA, par);
    gather(gqB, e1.eg, kg);                // pair p1: stored at the end of the first step
    __syncthreads();
    // one pipeline step: pair p0 with the weight registers `w` (passed by reference: the two weight sets are
    // used from fixed registers by separate copies of this body -- no register shuffling at an offset change)
    auto step = [&](const WReg &w, GReg &g_issue, GReg &g_store) __attribute__((always_inline)) {
      long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
      if (DBG & 4) t0 = __builtin_amdgcn_s_memtime();
      // unconditional: past the tile's last pair the loads repeat a valid pair and their results are dropped
      gather(g_issue, e2.eg, kg);                            // pair p2; its entry was loaded an iteration ago
      e3 = ent_of(p3);
      __builtin_amdgcn_sched_barrier(0);                     // the prefetches are issued HERE, ahead of the MFMAs
      if (DBG & 4) { t1 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }

      {
        const float *sa = St + (NBUF == 2 ? par : 0) * STAGE + c16 * RF;
        const float *sb = sa + 16 * RF;
        // all of the pair's B operands leave LDS before the first MFMA (counted lgkmcnt waits follow)
        u32x4 a0[KG], a1[KG], b0[KG], b1[KG];
#pragma unroll
        for (int c = 0; c < KG; ++c) {
          // fp32: the lane's 8 consecutive channels of a 32-channel chunk = granules 2g, 2g+1; bf16: its 8 channels
          // of each of the chunk's two 32-channel halves = granules g and 4+g
          const int q0 = ((c * 8 + (BF ? g : g * 2)) ^ (c16 & SWZ)) << 2,
                    q1 = ((c * 8 + (BF ? 4 + g : g * 2 + 1)) ^ (c16 & SWZ)) << 2;
          a0[c] = *reinterpret_cast<const u32x4 *>(sa + q0);
          b0[c] = *reinterpret_cast<const u32x4 *>(sb + q0);
          a1[c] = *reinterpret_cast<const u32x4 *>(sa + q1);
          b1[c] = *reinterpret_cast<const u32x4 *>(sb + q1);
        }
        f32x4 accA = {0.f, 0.f, 0.f, 0.f}, accB = accA;
        if (DBG & 1) { // timing experiments: operands consumed, no MFMAs
#pragma unroll
          for (int c = 0; c < KG; ++c) {
            accA[0] += bcf_(a0[c][0]) + bcf_(a1[c][0]) + bcf_(w.w0[c][0]) + bcf_(w.w1[c][0]);
            accB[0] += bcf_(b0[c][0]) + bcf_(b1[c][0]);
          }
        } else if (BF) {
#pragma unroll
          for (int c = 0; c < KG; ++c) {
            accA = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8w, w.w0[c]),
                                                           __builtin_bit_cast(bf16x8w, a0[c]), accA, 0, 0, 0);
            accA = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8w, w.w1[c]),
                                                           __builtin_bit_cast(bf16x8w, a1[c]), accA, 0, 0, 0);
          }
          if (e0.hb) {
#pragma unroll
            for (int c = 0; c < KG; ++c) {
              accB = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8w, w.w0[c]),
                                                             __builtin_bit_cast(bf16x8w, b0[c]), accB, 0, 0, 0);
              accB = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8w, w.w1[c]),
                                                             __builtin_bit_cast(bf16x8w, b1[c]), accB, 0, 0, 0);
            }
          }
        } else {
          // block A, then block B only if the pair has one (sparse rule books: most offsets of a tile hold a single
          // block; the branch is wave-uniform and covers nothing but MFMAs, so no load waits move)
#pragma unroll
          for (int c = 0; c < KG; ++c) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
              accA = __builtin_amdgcn_mfma_f32_16x16x4f32(bcf_(w.w0[c][t]), bcf_(a0[c][t]), accA, 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 4; ++t)
              accA = __builtin_amdgcn_mfma_f32_16x16x4f32(bcf_(w.w1[c][t]), bcf_(a1[c][t]), accA, 0, 0, 0);
          }
          if (e0.hb) {
#pragma unroll
            for (int c = 0; c < KG; ++c) {
#pragma unroll
              for (int t = 0; t < 4; ++t)
                accB = __builtin_amdgcn_mfma_f32_16x16x4f32(bcf_(w.w0[c][t]), bcf_(b0[c][t]), accB, 0, 0, 0);
#pragma unroll
              for (int t = 0; t < 4; ++t)
                accB = __builtin_amdgcn_mfma_f32_16x16x4f32(bcf_(w.w1[c][t]), bcf_(b1[c][t]), accB, 0, 0, 0);
            }
          }
        }
        if (DBG & 4) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_nop 0" ::"v"(accA), "v"(accB)); t2 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
        accumulate2(e0.ea, accA, e0.eb, accB);
      }
      if (NBUF == 1) wg_barrier();                           // single stage: every wave is done reading it
      if (p1.k < vol) stage_store(g_store, par ^ 1);         // pair p1, gathered during the previous step
      if (DBG & 4) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); t3 = __builtin_amdgcn_s_memtime(); }
      wg_barrier();
      if (DBG & 4) {
        t4 = __builtin_amdgcn_s_memtime();
        dbg_t[0] += t1 - t0; dbg_t[1] += t2 - t1; dbg_t[2] += t3 - t2; dbg_t[3] += t4 - t3; dbg_t[4] += 1;
      }
      par ^= 1;
      p0 = p1; p1 = p2; p2 = p3; p3 = adv(p3);
      e0 = e1; e1 = e2; e2 = e3;
    };
    // stage parity 0: issue into gqA, store gqB; parity 1: the other way round
    auto step2 = [&](const WReg &w) __attribute__((always_inline)) {
      if (par == 0) step(w, gqA, gqB);
      else step(w, gqB, gqA);
    };
    WReg wA, wB;
    int k = p0.k;
    load_w(wA, k, kg);
    for (;;) {
      int kn = next_offset(k);
      if (kn < vol) load_w(wB, kn, kg);                      // next offset's weights in flight during this offset
      while (p0.k == k) step2(wA);
      if (kn >= vol) break;
      k = kn;
      kn = next_offset(k);
      if (kn < vol) load_w(wA, kn, kg);
      while (p0.k == k) step2(wB);
      if (kn >= vol) break;
      k = kn;
    }
  }
  if (DBG & 4) { // timing experiments: per-wave phase clocks -> the buffer passed as `bias` (which is then not added)
    if (lane == 0) {
      long long *d = reinterpret_cast<long long *>(const_cast<float *>(bias)) +
                     (((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) * 5;
      for (int q = 0; q < 5; ++q) d[q] = dbg_t[q];
    }
    bias = nullptr;
  }
  __syncthreads();
  const int nrows = (int)((V_out - row0) < kT2 ? (V_out - row0) : kT2);
#pragma unroll 4
  for (int i = threadIdx.x; i < nrows * 16; i += 256) {
    const int r = i >> 4, q = i & 15;
    f32x4 v = *reinterpret_cast<const f32x4 *>(Ct + r * kWS + ((q ^ (r & 15)) << 2));
    if (bias) {
      const float *bb = bias + nb0 * 16 + q * 4;
      v[0] += bb[0]; v[1] += bb[1]; v[2] += bb[2]; v[3] += bb[3];
    }
    if (res) { // out = conv + res: the residual / lateral add of the consumer folded into the write-out
      const f32x4 rr = *reinterpret_cast<const f32x4 *>(res + (row0 + r) * co + nb0 * 16 + q * 4);
      v[0] += rr[0]; v[1] += rr[1]; v[2] += rr[2]; v[3] += rr[3];
    }
    if (BF) {
      bf16x4w o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
      *reinterpret_cast<bf16x4w *>(reinterpret_cast<__bf16 *>(out) + (row0 + r) * co + nb0 * 16 + q * 4) = o;
    } else {
      *reinterpret_cast<f32x4 *>(out + (row0 + r) * co + nb0 * 16 + q * 4) = v;
    }
  }
}

} // namespace aabr
using namespace aabr;

static int64_t wide_words(int64_t V, int vol, int T) {
  const int64_t nt = (V + T - 1) / T;
  return nt * (vol + 1) + nt * (int64_t)(T / 16) * vol * 16;
}

extern "C" int64_t aabr_wide_blocks_words(int64_t V, int vol, int tile_rows) { return wide_words(V, vol, tile_rows); }

extern "C" int aabr_build_wide_blocks(const int32_t *table, int64_t V, int vol, int tile_rows, int32_t *blocks,
                                      void *stream_) {
  AABR_CHECK_ARG(V >= 0 && vol > 0 && vol <= kMaxVol, "bad sizes (vol <= 63)");
  AABR_CHECK_ARG(tile_rows >= 16 && tile_rows <= kMaxTileRows && (tile_rows & 15) == 0, "tile_rows: multiple of 16, <= 240");
  if (V == 0) return AABR_OK;
  AABR_CHECK_ARG(table && blocks, "null pointer");
  const unsigned nt = (unsigned)((V + tile_rows - 1) / tile_rows);
  hipLaunchKernelGGL(k_build_tileT, dim3(nt), dim3(64 * ((tile_rows + 63) / 64)), 0, (hipStream_t)stream_, table, V, vol,
                     tile_rows, blocks);
  AABR_CHECK_LAUNCH();
  return AABR_OK;
}

// 0: use the 64-row-tile kernels of conv.hip; 128 / 256: rows per tile of the block stream aabr_conv_forward_wide wants
extern "C" int aabr_conv_wide_tile_rows(int n_in, int n_out, int64_t rows_in, int64_t V_out, int vol) {
  if (n_in <= 0 || n_out <= 0 || (n_in & 31) || (n_out & 63) || vol <= 0 || vol > kMaxVol) return 0;
  if (rows_in >= (1ll << 23) || rows_in * n_in * 4 >= (1ll << 31)) return 0;
  // rows per tile: 128, except when the whole launch fits the chip in ONE round (512 resident workgroups, 2 per
  // CU): then its time is the longest workgroup, so take the smallest tile (>= 64 rows) that still fits one round
  // (measured, profiles/r02_conv_wide_ab.txt: 22k rows x 2 slabs, 128 -> 96 rows: 184 -> 133 us; with several rounds
  // smaller tiles only lower the block fill: 84k rows 385 -> 400 us)
  int T = 128;
  {
    const int64_t slabs = n_out / 64;
    if (((V_out + 127) / 128) * slabs <= 512)
      for (int t = 64; t < 128; t += 16)
        if (((V_out + t - 1) / t) * slabs <= 512) { T = t; break; }
  }
  {                                                // tuning experiments only
    const int v = knob(K_WIDE_ROWS);
    if (v >= 16 && v <= kMaxTileRows && (v & 15) == 0) T = v;
  }
  if (wide_words(V_out, vol, T) * 4 >= (1ll << 31)) return 0;
  if ((int64_t)vol * n_in * n_out * 4 >= (1ll << 31)) return 0;
  if (n_in > 128 && (n_in & 127)) return 0; // channel groups of 128: every load of the inner loop unconditional
  {                                                // tuning experiments / tests only: 0 = never, 1 = whenever supported
    const int v = knob(K_CONV_WIDE);
    if (v == 0) return 0;
    if (v == 1) return T;
  }
  // enough workgroups to fill the chip twice over (measured, profiles/r02_conv_wide_ab.txt: wins from ~340
  // workgroups up, loses below ~180)
  return (((V_out + T - 1) / T) * (n_out / 64) >= 320) ? T : 0;
}

extern "C" int aabr_conv_forward_wide_res(const float *in_feats, int n_in, int64_t rows_in, float *out_feats,
                                          int n_out, int64_t V_out, const int32_t *blocks, int tile_rows, int vol,
                                          const float *bias, int flags, const float *wpack, const float *residual,
                                          void *stream_);

extern "C" int aabr_conv_forward_wide(const float *in_feats, int n_in, int64_t rows_in, float *out_feats, int n_out,
                                      int64_t V_out, const int32_t *blocks, int tile_rows, int vol, const float *bias,
                                      int flags, const float *wpack, void *stream_) {
  return aabr_conv_forward_wide_res(in_feats, n_in, rows_in, out_feats, n_out, V_out, blocks, tile_rows, vol, bias,
                                    flags, wpack, nullptr, stream_);
}

extern "C" int aabr_conv_forward_wide_res(const float *in_feats, int n_in, int64_t rows_in, float *out_feats,
                                          int n_out, int64_t V_out, const int32_t *blocks, int tile_rows, int vol,
                                          const float *bias, int flags, const float *wpack, const float *residual,
                                          void *stream_) {
  hipStream_t st = (hipStream_t)stream_;
  AABR_CHECK_ARG(((uintptr_t)residual & 15) == 0, "residual must be 16-byte aligned");
  AABR_CHECK_ARG(n_in > 0 && n_out > 0 && (n_in & 31) == 0 && (n_out & 63) == 0, "plane counts: n_in % 32, n_out % 64");
  AABR_CHECK_ARG(vol > 0 && vol <= kMaxVol && V_out >= 0 && rows_in >= 0, "bad sizes");
  AABR_CHECK_ARG(tile_rows >= 16 && tile_rows <= kMaxTileRows && (tile_rows & 15) == 0, "tile_rows: multiple of 16, <= 240");
  if (V_out == 0) return AABR_OK;
  AABR_CHECK_ARG(in_feats && out_feats && blocks && wpack && rows_in > 0, "null pointer / empty input");
  AABR_CHECK_ARG(rows_in < (1ll << 23), "too many input rows for the wide block format");
  const int64_t in_bytes = rows_in * n_in * 4, words_bytes = wide_words(V_out, vol, tile_rows) * 4;
  AABR_CHECK_ARG(in_bytes < (1ll << 31) && words_bytes < (1ll << 31), "buffers must be < 2 GiB");
  AABR_CHECK_ARG(((uintptr_t)in_feats & 15) == 0 && ((uintptr_t)out_feats & 15) == 0 && ((uintptr_t)wpack & 15) == 0,
                 "feature / weight pointers must be 16-byte aligned");
  const int dbg = flags >> 8;
  const int nkc = n_in / 32;
  const int64_t wp_bytes = (int64_t)vol * nkc * (n_out / 16) * 2048;
  AABR_CHECK_ARG(wp_bytes < (1ll << 31), "packed weights must be < 2 GiB");
  AABR_CHECK_ARG(n_in <= 128 || (n_in & 127) == 0, "n_in above 128 must be a multiple of 128");
  dim3 grid((unsigned)((V_out + tile_rows - 1) / tile_rows), (unsigned)(n_out / 64));
  const int flip = (flags >> 1) & 1;
  const int kg = nkc >= 4 ? 4 : nkc;
  // LDS stage buffers: with 128-channel groups the double-buffered stage (32 KiB) allows two workgroups per CU, a
  // single buffer three (49 KiB each) at the price of a second barrier per pair: measured +4...+10 % (128->128 at 84k
  // rows 380 -> 367 us, 256->256 1366 -> 1272 us); narrower groups fit three workgroups with the double buffer
  int nbuf = kg == 4 ? 1 : 2;
  {                                                // tuning experiments only
    const int v = knob(K_WIDE_NBUF);
    if (v == 1 || v == 2) nbuf = v;
  }
#define AABR_LAUNCH_WIDE(KERNEL, NAME, LDS, ...)                                                          \
  do {                                                                                                    \
    static bool attr = false;                                                                             \
    if (!attr) {                                                                                          \
      AABR_CHECK_HIP(hipFuncSetAttribute((const void *)KERNEL, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024)); \
      attr = true;                                                                                        \
    }                                                                                                     \
    g_last_variant = NAME;                                                                                \
    hipLaunchKernelGGL(KERNEL, grid, dim3(256), (LDS), st, __VA_ARGS__);                                  \
  } while (0)
  {
#define AABR_WIDE_CS_N(KG, D, NB)                                                                         \
  AABR_LAUNCH_WIDE((k_conv_cs<KG, D, NB>), "k_conv_cs<" #KG "," #D "," #NB ">",                           \
                   (size_t)((tile_rows + 1) * kWS + NB * 2 * 16 * KG * 32) * sizeof(float), in_feats, n_in, in_bytes,   \
                   out_feats, n_out, V_out, blocks, words_bytes, vol, flip, wpack, wp_bytes, bias, tile_rows, residual)
#define AABR_WIDE_CS(KG, D)                                                                               \
  do {                                                                                                    \
    if (nbuf == 1) AABR_WIDE_CS_N(KG, D, 1); else AABR_WIDE_CS_N(KG, D, 2);                               \
  } while (0)
#ifdef AABR_DEV
    if (dbg & 7) { // timing experiments (tools/, `make DEV=1`): only the 128-channel-group instance carries the debug variants
      AABR_CHECK_ARG(kg == 4, "debug variants exist for n_in >= 128 only");
      nbuf = 2;
      if (dbg & 4) AABR_WIDE_CS(4, 4);
      else if ((dbg & 3) == 1) AABR_WIDE_CS(4, 1); else if ((dbg & 3) == 2) AABR_WIDE_CS(4, 2); else AABR_WIDE_CS(4, 3);
    } else
#else
    AABR_CHECK_ARG(!(dbg & 7), "the timing-experiment variants of k_conv_cs exist in a `make DEV=1` build only");
#endif
    {
      if (kg == 1) AABR_WIDE_CS(1, 0); else if (kg == 2) AABR_WIDE_CS(2, 0); else if (kg == 3) AABR_WIDE_CS(3, 0);
      else AABR_WIDE_CS(4, 0);
    }
#undef AABR_WIDE_CS
#undef AABR_WIDE_CS_N
  }
#undef AABR_LAUNCH_WIDE
  AABR_CHECK_LAUNCH();
  return AABR_OK;
}

// ---- bf16 feature storage (extension): the same kernel on 128-byte row chunks of 64 bf16 channels ------------------
// 0: use aabr_conv_forward_bf16 (64-row tiles); otherwise rows per tile for aabr_conv_forward_wide_bf16
extern "C" int aabr_conv_wide_tile_rows_bf16(int n_in, int n_out, int64_t rows_in, int64_t V_out, int vol) {
  if (n_in <= 0 || n_out <= 0 || (n_in & 63) || (n_out & 63) || vol <= 0 || vol > kMaxVol) return 0;
  if (rows_in >= (1ll << 23) || rows_in * n_in * 2 >= (1ll << 31)) return 0;
  if (n_in > 256 && (n_in & 255)) return 0; // channel groups of 256: every load of the inner loop unconditional
  // bf16: two MFMAs per block and 64-channel chunk -- the gather / stage / barrier skeleton sets the pace, so more
  // resident workgroups pay: 96-row tiles with a single stage buffer (measured on the bench's rule books: convolution
  // time of a bf16 step 5.25 -> 5.01 ms against 128 rows + two buffers; 64 rows: 5.24)
  int T = 96;
  {
    const int64_t slabs = n_out / 64;
    if (((V_out + 95) / 96) * slabs <= 512)
      for (int t = 64; t < 96; t += 16)
        if (((V_out + t - 1) / t) * slabs <= 512) { T = t; break; }
  }
  {                                                // tuning experiments only
    const int v = knob(K_WIDE_ROWS);
    if (v >= 16 && v <= kMaxTileRows && (v & 15) == 0) T = v;
  }
  if (wide_words(V_out, vol, T) * 4 >= (1ll << 31)) return 0;
  if ((int64_t)vol * n_in * n_out * 2 >= (1ll << 31)) return 0;
  {                                                // tuning experiments / tests only: 0 = never, 1 = whenever supported
    const int v = knob(K_CONV_WIDE_BF16);
    if (v == 0) return 0;
    if (v == 1) return T;
  }
  return (((V_out + T - 1) / T) * (n_out / 64) >= 320) ? T : 0;
}

extern "C" int aabr_conv_forward_wide_bf16(const uint16_t *in_feats, int n_in, int64_t rows_in, uint16_t *out_feats,
                                           int n_out, int64_t V_out, const int32_t *blocks, int tile_rows, int vol,
                                           const float *bias, int flags, const uint16_t *wpack, void *stream_) {
  hipStream_t st = (hipStream_t)stream_;
  AABR_CHECK_ARG(n_in > 0 && n_out > 0 && (n_in & 63) == 0 && (n_out & 63) == 0, "plane counts: n_in % 64, n_out % 64");
  AABR_CHECK_ARG(vol > 0 && vol <= kMaxVol && V_out >= 0 && rows_in >= 0, "bad sizes");
  AABR_CHECK_ARG(tile_rows >= 16 && tile_rows <= kMaxTileRows && (tile_rows & 15) == 0, "tile_rows: multiple of 16, <= 240");
  if (V_out == 0) return AABR_OK;
  AABR_CHECK_ARG(in_feats && out_feats && blocks && wpack && rows_in > 0, "null pointer / empty input");
  AABR_CHECK_ARG(rows_in < (1ll << 23), "too many input rows for the wide block format");
  const int64_t in_bytes = rows_in * n_in * 2, words_bytes = wide_words(V_out, vol, tile_rows) * 4;
  AABR_CHECK_ARG(in_bytes < (1ll << 31) && words_bytes < (1ll << 31), "buffers must be < 2 GiB");
  AABR_CHECK_ARG(((uintptr_t)in_feats & 15) == 0 && ((uintptr_t)out_feats & 15) == 0 && ((uintptr_t)wpack & 15) == 0,
                 "feature / weight pointers must be 16-byte aligned");
  const int nkc = n_in / 64; // 128-byte chunks per row
  const int64_t wp_bytes = (int64_t)vol * (n_in / 32) * (n_out / 16) * 1024;
  AABR_CHECK_ARG(wp_bytes < (1ll << 31), "packed weights must be < 2 GiB");
  AABR_CHECK_ARG(n_in <= 256 || (n_in & 255) == 0, "n_in above 256 must be a multiple of 256");
  dim3 grid((unsigned)((V_out + tile_rows - 1) / tile_rows), (unsigned)(n_out / 64));
  const int flip = (flags >> 1) & 1;
  const int kg = nkc >= 4 ? 4 : nkc;
  int nbuf = 1;
  {                                                // tuning experiments only
    const int v = knob(K_WIDE_NBUF);
    if (v == 1 || v == 2) nbuf = v;
  }
  const float *in_f = reinterpret_cast<const float *>(in_feats), *wp_f = reinterpret_cast<const float *>(wpack);
  float *out_f = reinterpret_cast<float *>(out_feats);
#define AABR_WIDE_BF(KG, NB)                                                                                       \
  do {                                                                                                             \
    static bool attr = false;                                                                                      \
    if (!attr) {                                                                                                   \
      AABR_CHECK_HIP(hipFuncSetAttribute((const void *)(k_conv_cs<KG, 0, NB, true>),                               \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));                  \
      attr = true;                                                                                                 \
    }                                                                                                              \
    g_last_variant = "k_conv_cs<" #KG ",0," #NB ",bf16>";                                                          \
    hipLaunchKernelGGL((k_conv_cs<KG, 0, NB, true>), grid, dim3(256),                                              \
                       (size_t)((tile_rows + 1) * kWS + NB * 2 * 16 * KG * 32) * sizeof(float), st, in_f, n_in,    \
                       in_bytes, out_f, n_out, V_out, blocks, words_bytes, vol, flip, wp_f, wp_bytes, bias,        \
                       tile_rows, (const float *)nullptr);                                                         \
  } while (0)
#define AABR_WIDE_BF_K(KG)                                                                                         \
  do {                                                                                                             \
    if (nbuf == 1) AABR_WIDE_BF(KG, 1); else AABR_WIDE_BF(KG, 2);                                                  \
  } while (0)
  if (kg == 1) AABR_WIDE_BF_K(1); else if (kg == 2) AABR_WIDE_BF_K(2); else if (kg == 3) AABR_WIDE_BF_K(3);
  else AABR_WIDE_BF_K(4);
#undef AABR_WIDE_BF_K
#undef AABR_WIDE_BF
  AABR_CHECK_LAUNCH();
  return AABR_OK;
}
