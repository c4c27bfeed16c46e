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
// (one launch serves the books of a job list that need the same number of waves per tile, common.h StreamJobs)
__global__ __launch_bounds__(256) void k_build_tileT(const StreamJobs js) {
  const int job = stream_job_of(js, blockIdx.x);
  const int32_t *__restrict__ table = js.j[job].table;
  int32_t *__restrict__ words = js.j[job].words;
  const int64_t V = js.j[job].V;
  const int vol = js.j[job].vol, T = js.j[job].T;
  const int kT = T;
  const int NWV = (T + 63) / 64;
  __shared__ int s_cnt[4][kMaxVol];
  __shared__ int s_base[kMaxVol + 1];
  __shared__ int s_tot[kMaxVol];
  __shared__ int s_first[kMaxVol];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t ntiles = (V + T - 1) / T, tile = blockIdx.x - js.first[job];
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
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// NCB = 16-column blocks per wave: 1 = 64-column slabs; 2 (bf16 storage) = 128-column slabs -- the kernel is bound by
// the rate at which a CU gathers random rows (~20 GB/s per CU measured in both this and the row-stationary kernel,
// profiles/r03_conv_rs_ab.txt), and a 128-column layer gathered every row once per 64-column slab, i.e. twice
// `stats` in the input-gradient form: the launch writes d_out of the BatchNorm(+leaky ReLU) whose OUTPUT the convolution
// consumed in the forward pass; with `x` set the write-out forms that BatchNorm's BACKWARD statistics instead of the
// forward ones -- per tile [2][co] fp64 sums of d and (x - mean) * d, d = d_out masked by the sign of the forward
// activation recomputed from x exactly as k_bn_partials<1> does (same floats, same operations)
// bf16 storage: the sign comes from the STORED (rounded) activation `out` as in k_bn_partials<1, .., bf16> (x, out bf16)
struct BnBwdStats {
  const float *x, *mean, *invstd, *weight, *bias;   // x == nullptr: forward statistics (or none)
  float leak;
  const float *out;                                 // bf16 storage only
};

// Forms that were built, measured slower and removed in round 5 (their A/B tables are the record): fp32 on the bf16 pipe by
// a three-term split (profiles/r03_conv_x3_ab.txt), a fourth resident workgroup through a two-slot operand ring
// (r04_conv_occ4_ab.txt), the tile rows as the MFMA chains' starting accumulators (r04_conv_bf16_defer_ab.txt), eight-wave
// workgroups on 128-column slabs (r04_conv_bf16_w8_ab.txt), the row-stationary kernel (r03_conv_rs_ab.txt).
template <int KG, int DBG, int NBUF, bool BF = false, int NCB = 1, int NSET = 2>
__global__ __launch_bounds__(256, NBUF == 2 ? 2 : 3) void k_conv_cs(const float *__restrict__ in, int ci, int64_t in_bytes,
                                                    float *__restrict__ out, int co, int64_t V_out,
                                                    const int32_t *__restrict__ words, int64_t words_bytes, int vol,
                                                    int wflip, const float *__restrict__ Wp, int64_t wp_bytes,
                                                    const float *__restrict__ bias, int kT2,
                                                    const float *__restrict__ res, double *__restrict__ stats,
                                                    BnBwdStats bn) {
  constexpr int NW = 4;                   // waves per workgroup
  constexpr int NT = 64 * NW;              // threads
  constexpr int LPR = NT / 32;             // lanes per gathered pair row (8 or 16)
  constexpr int RG = KG * 8;               // 16-byte granules per staged row
  constexpr int RF = KG * 32;              // floats per staged row
  constexpr int SWZ = (RG >= 16 && (RG & 15) == 0) ? 15 : 7; // XOR must stay inside the row's granules
  constexpr int STAGE = 2 * 16 * RF;       // floats per stage buffer (two blocks)
  constexpr bool BFM = BF;                 // bf16 operands in the stage and the weight pack
  constexpr int NP = 1;                    // operand planes
  constexpr int NGL = KG * 8 / LPR;        // 16-byte gather loads per lane and pair
  extern __shared__ __align__(16) float smem[];
  float *Ct = smem;                        // [kT2 + 1][64] floats, granule-swizzled; the last row swallows padding entries
  constexpr int WS = 16 * NW * NCB;        // tile row stride in floats = slab width
  float *St = smem + (kT2 + 1) * WS;       // [2][2][16][RF]
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int g = lane >> 4, c16 = lane & 15;
  const int pr = wave * (64 / LPR) + lane / LPR, seg = lane % LPR; // gather role: pair row 0..31, 16-byte segment
  const int nkc = BFM ? ci >> 6 : ci >> 5, nnb = co >> 4;  // chunks per row (64 channels with bf16 operands, else 32)
  // Workgroups are dealt round-robin over the 8 XCDs (linear id % 8), each with its own 4 MiB L2.  Give every XCD a
  // CONTIGUOUS range of (tile, slab) work items instead of every eighth one: the slabs of a tile and the tiles next to
  // it gather the same input rows (consecutive sites lie on the same surface), so they hit that XCD's L2 instead of
  // all eight L2s fetching the whole input (round 2: 580 MB of fetches per launch for 91 MB of operands).  Speed
  // only -- which workgroup computes a tile does not change a bit of it.
  // Offset split (coarse maps: too few (tile, slab) items to fill the chip): grid.y = slabs x P, part p of a (tile, slab)
  // sweeps the filter offsets [p vol / P, (p + 1) vol / P) only and writes its fp32 partial tile to out + p V_out co;
  // k_split_reduce adds the parts in part order (bias, residual and the storage rounding happen there).
  const int nparts = (wflip >> 8) & 0xff;
  int64_t tile;
  int nb0, part = 0;
  {
    const unsigned ny = gridDim.y, total = gridDim.x * ny;
    const unsigned lin = blockIdx.y * gridDim.x + blockIdx.x;
    const unsigned per = total >> 3, rem = total & 7u, x = lin & 7u;
    const unsigned wi = x * per + (x < rem ? x : rem) + (lin >> 3);
    tile = wi / ny;
    unsigned sp = wi % ny;
    if (nparts > 1) { part = (int)(sp % (unsigned)nparts); sp /= (unsigned)nparts; }
    nb0 = (int)sp * (NW * NCB);
  }
  const int k_lo = nparts > 1 ? part * vol / nparts : 0;
  const int kend = nparts > 1 ? (part + 1) * vol / nparts : vol;   // offsets >= kend are past the end for this part
  const int64_t row0 = tile * kT2;
  const int64_t ntiles = (V_out + kT2 - 1) / kT2;
  const int maxb = (kT2 / 16) * vol;
  const int vpre = lane <= vol ? words[tile * (vol + 1) + lane] : 0;
  {
    f32x4 z = {0.f, 0.f, 0.f, 0.f};
    f32x4 *c4 = reinterpret_cast<f32x4 *>(Ct);
    for (int i = threadIdx.x; i < (kT2 + 1) * WS / 4; i += NT) c4[i] = z;
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
    while (k < kend && pre_of(k + 1) == pre_of(k)) ++k;
    return k;
  };
  struct WReg { u32x4 w0[NP][NCB][KG], w1[NP][NCB][KG]; };
  struct GReg { u32x4 v[NGL]; };
  struct Ent { int eg, ea, eb, hb; };      // gather-role entry of this lane's pair row; compute-role entries (A, B);
                                           // hb (wave-uniform): the pair has a second block
  const int ngroups = (nkc + KG - 1) / KG;
  const int nblk_all = pre_of(vol);
  auto load_w = [&](WReg &w, int k, int kg) {
    const int kW = (wflip & 1) ? vol - 1 - k : k;
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int c = 0; c < KG; ++c) { // nkc % KG == 0 (dispatch): every load is unconditional, so the compiler's
        const int kc = kg * KG + c;  // vmcnt bookkeeping stays exact and nothing waits for a prefetch it does not use
        if (BFM) { // two 32-channel MFMA chunks per 64-channel row chunk, 1 KiB of packed weights each
          const unsigned so = (unsigned)((((int64_t)kW * (2 * nkc) + 2 * kc) * nnb + nb0 + wave * NCB + cb) * 1024);
#pragma unroll
          for (int pl = 0; pl < NP; ++pl) {
            const unsigned po = so + (unsigned)pl * (unsigned)(wp_bytes / NP);
            w.w0[pl][cb][c] = __builtin_amdgcn_raw_buffer_load_b128(rw, (unsigned)lane * 16u, po, 0);
            w.w1[pl][cb][c] = __builtin_amdgcn_raw_buffer_load_b128(rw, (unsigned)lane * 16u, po + (unsigned)nnb * 1024u, 0);
          }
        } else {
          const unsigned so = (unsigned)((((int64_t)kW * nkc + kc) * nnb + nb0 + wave * NCB + cb) * 2048);
          w.w0[0][cb][c] = __builtin_amdgcn_raw_buffer_load_b128(rw, lane32, so, 0);
          w.w1[0][cb][c] = __builtin_amdgcn_raw_buffer_load_b128(rw, lane32 + 16u, so, 0);
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
    for (int i = 0; i < NGL; ++i) { // granule seg + 8 i of the row's current channel group
      const unsigned so = (unsigned)(kg * NGL + i) * (unsigned)(LPR * 16);
      if (DBG & 2) q.v[i] = (u32x4){(unsigned)eg, 0u, 0u, 0u}; // timing experiments: no global gathers
      else q.v[i] = __builtin_amdgcn_raw_buffer_load_b128(rin, va, so, 0);
    }
  };
  auto stage_store = [&](const GReg &q, int buf) {
    float *rowp = St + (NBUF == 2 ? buf : 0) * (NP * STAGE) + pr * RF;
#pragma unroll
    for (int i = 0; i < NGL; ++i)
      *reinterpret_cast<u32x4 *>(rowp + (((seg + LPR * i) ^ (pr & SWZ)) << 2)) = q.v[i];
  };
  // branch-free: a padding entry (bit 31) lands in the dummy row; the two blocks of a pair never share a real row,
  // so both reads go out before either write
  auto accumulate2 = [&](int ea, const f32x4 (&accA)[NCB], int eb, const f32x4 (&accB)[NCB]) {
    const int ra = ea >= 0 ? (ea & 255) : kT2, rb = eb >= 0 ? (eb & 255) : kT2;
    // (LDS float adds without return, ds_add_f32, instead of this read-add-write: measured 10x slower -- 1147 us
    // instead of 100 us on the dominant bf16 instance, profiles/r03_conv_bf16_ab.txt)
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      f32x4 *da = reinterpret_cast<f32x4 *>(Ct + ra * WS + ((((wave * NCB + cb) * 4 + g) ^ (ra & 15)) << 2));
      f32x4 *db = reinterpret_cast<f32x4 *>(Ct + rb * WS + ((((wave * NCB + cb) * 4 + g) ^ (rb & 15)) << 2));
      const f32x4 va = *da, vb = *db;
      *da = va + accA[cb];
      *db = vb + accB[cb];
    }
  };
  // Software pipeline over the tile's block pairs (pairs never straddle an offset):
  //   entries two pairs ahead (registers), gathered rows one pair ahead (registers -> LDS stage after this pair's
  //   MFMAs), weights one offset ahead.  A gather never waits for an entry load issued in the same iteration.
  struct Pos { int b, k; };                // k >= kend: past the end
  auto adv = [&](Pos q) {
    if (q.k >= kend) return q;
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
    // NSET register sets of gathered rows: the rows of pair i + NSET are requested while pair i is multiplied (fp32:
    // NSET = 2 -- a pair's MFMAs take ~1 us, a random-row gather ~2 us; bf16 storage: the MFMAs of a pair take 0.1 us,
    // so the workgroup's pace is NSET pairs per gather latency and NSET = 4 nearly doubles it, measured).  Entries run
    // one pair further ahead.  Pair i lives in set i % NSET; the step body exists once per set (static registers).
    Pos pp[NSET + 2];
    pp[0].k = next_offset(k_lo - 1);
    if (pp[0].k >= kend) break;
    pp[0].b = pre_of(pp[0].k);
#pragma unroll
    for (int i = 1; i < NSET + 2; ++i) pp[i] = adv(pp[i - 1]);
    auto ent_of = [&](const Pos &q) { const bool v = q.k < kend; return load_ent(v ? q.b : pp[0].b, v ? q.k : pp[0].k); };
    Ent ee[NSET + 2];
#pragma unroll
    for (int i = 0; i <= NSET; ++i) ee[i] = ent_of(pp[i]);
    ee[NSET + 1] = ee[0];
    GReg gq[NSET];
    gather(gq[0], ee[0].eg, kg);
    __syncthreads();                       // zero fill done / previous group's stage reads done
    par = 0;
    int gs = 0;                            // register set of the current pair
    stage_store(gq[0], par);
#pragma unroll
    for (int i = 1; i < NSET; ++i) gather(gq[i], ee[i].eg, kg);   // pairs 1 .. NSET-1: stored one step before their turn
    __syncthreads();
    // one pipeline step: pair pp[0] with the weight registers `w` (passed by reference: the two weight sets are
    // used from fixed registers by separate copies of this body -- no register shuffling at an offset change)
    auto step = [&](const WReg &w, GReg &g_issue, GReg &g_store) __attribute__((always_inline)) {
      long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
      if (DBG & 4) t0 = __builtin_amdgcn_s_memtime();
      // unconditional: past the tile's last pair the loads repeat a valid pair and their results are dropped
      gather(g_issue, ee[NSET].eg, kg);                      // pair i + NSET; its entry was loaded an iteration ago
      ee[NSET + 1] = ent_of(pp[NSET + 1]);
      const Ent e0 = ee[0];
      __builtin_amdgcn_sched_barrier(0);                     // the prefetches are issued HERE, ahead of the MFMAs
      if (DBG & 4) { t1 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }

      {
        const float *sa = St + (NBUF == 2 ? par : 0) * STAGE + c16 * RF;
        const float *sb = sa + 16 * RF;
        f32x4 accA[NCB], accB[NCB];
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) { accA[cb] = (f32x4){0.f, 0.f, 0.f, 0.f}; accB[cb] = accA[cb]; }
        if (wflip & 2) __builtin_amdgcn_s_setprio(3);          // experiment (WIDE_PRIO knob): matrix phase at raised priority
        {
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
        if (DBG & 1) { // timing experiments: operands consumed, no MFMAs
#pragma unroll
          for (int c = 0; c < KG; ++c) {
            accA[0][0] += bcf_(a0[c][0]) + bcf_(a1[c][0]) + bcf_(w.w0[0][0][c][0]) + bcf_(w.w1[0][0][c][0]);
            accB[0][0] += bcf_(b0[c][0]) + bcf_(b1[c][0]);
          }
        } else if (BF) {
#pragma unroll
          for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int c = 0; c < KG; ++c) {
              accA[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8w, w.w0[0][cb][c]),
                                                                 __builtin_bit_cast(bf16x8w, a0[c]), accA[cb], 0, 0, 0);
              accA[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8w, w.w1[0][cb][c]),
                                                                 __builtin_bit_cast(bf16x8w, a1[c]), accA[cb], 0, 0, 0);
            }
          if (e0.hb) {
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
              for (int c = 0; c < KG; ++c) {
                accB[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8w, w.w0[0][cb][c]),
                                                                   __builtin_bit_cast(bf16x8w, b0[c]), accB[cb], 0, 0, 0);
                accB[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8w, w.w1[0][cb][c]),
                                                                   __builtin_bit_cast(bf16x8w, b1[c]), accB[cb], 0, 0, 0);
              }
          }
        } else {
          // block A, then block B only if the pair has one (sparse rule books: most offsets of a tile hold a single
          // block; the branch is wave-uniform and covers nothing but MFMAs, so no load waits move)
#pragma unroll
          for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int c = 0; c < KG; ++c) {
#pragma unroll
              for (int t = 0; t < 4; ++t)
                accA[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(bcf_(w.w0[0][cb][c][t]), bcf_(a0[c][t]), accA[cb], 0, 0, 0);
#pragma unroll
              for (int t = 0; t < 4; ++t)
                accA[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(bcf_(w.w1[0][cb][c][t]), bcf_(a1[c][t]), accA[cb], 0, 0, 0);
            }
          if (e0.hb) {
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
              for (int c = 0; c < KG; ++c) {
#pragma unroll
                for (int t = 0; t < 4; ++t)
                  accB[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(bcf_(w.w0[0][cb][c][t]), bcf_(b0[c][t]), accB[cb], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < 4; ++t)
                  accB[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(bcf_(w.w1[0][cb][c][t]), bcf_(b1[c][t]), accB[cb], 0, 0, 0);
              }
          }
        }
        }
        if (DBG & 4) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_nop 0" ::"v"(accA[0]), "v"(accB[0])); t2 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
        if (wflip & 2) __builtin_amdgcn_s_setprio(0);
        {
          accumulate2(e0.ea, accA, e0.eb, accB);
        }
      }
      {
        if (NBUF == 1) wg_barrier();                         // single stage: every wave is done reading it
        if (pp[1].k < kend) stage_store(g_store, par ^ 1);   // the next pair, gathered NSET - 1 steps ago
      }
      if (DBG & 4) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); t3 = __builtin_amdgcn_s_memtime(); }
      wg_barrier();
      if (DBG & 4) {
        t4 = __builtin_amdgcn_s_memtime();
        dbg_t[0] += t1 - t0; dbg_t[1] += t2 - t1; dbg_t[2] += t3 - t2; dbg_t[3] += t4 - t3; dbg_t[4] += 1;
      }
      par ^= 1;
#pragma unroll
      for (int i = 0; i < NSET + 1; ++i) { pp[i] = pp[i + 1]; ee[i] = ee[i + 1]; }
      pp[NSET + 1] = adv(pp[NSET + 1]);
    };
    // the current pair's set takes the new request, the next pair's set is stored
    auto step2 = [&](const WReg &w) __attribute__((always_inline)) {
      if (NSET == 2) {
        if (gs == 0) step(w, gq[0], gq[1]);
        else step(w, gq[1], gq[0]);
      } else {
        if (gs == 0) step(w, gq[0], gq[1 % NSET]);
        else if (gs == 1) step(w, gq[1 % NSET], gq[2 % NSET]);
        else if (gs == 2) step(w, gq[2 % NSET], gq[3 % NSET]);
        else step(w, gq[3 % NSET], gq[0]);
      }
      gs = gs + 1 == NSET ? 0 : gs + 1;
    };
    WReg wA;
    int k = pp[0].k;
    load_w(wA, k, kg);
    WReg wB;
    for (;;) {
      int kn = next_offset(k);
      if (kn < kend) load_w(wB, kn, kg);                     // next offset's weights in flight during this offset
      while (pp[0].k == k) step2(wA);
      if (kn >= kend) break;
      k = kn;
      kn = next_offset(k);
      if (kn < kend) load_w(wA, kn, kg);
      while (pp[0].k == k) step2(wB);
      if (kn >= kend) break;
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
  // `stats`: the following BatchNorm's statistics from the values this tile writes (exactly the stored values: after
  // bias / residual, after the bf16 rounding) -- per tile one [2][co] pair of fp64 column sums (x, x^2), combined by the
  // BatchNorm's finalize in tile order, so the result does not depend on which workgroup ran when
  constexpr int QW = 4 * NW * NCB;         // 16-byte columns of the slab; NT % QW == 0: a thread keeps its column
  if (nparts > 1) {                        // this part's fp32 partial tile (an empty offset range writes zeros)
    float *po = out + (int64_t)part * V_out * co;
    for (int i = threadIdx.x; i < nrows * QW; i += NT) {
      const int r = i / QW, q = i % QW;
      *reinterpret_cast<f32x4 *>(po + (row0 + r) * co + nb0 * 16 + q * 4) =
          *reinterpret_cast<const f32x4 *>(Ct + r * WS + ((q ^ (r & 15)) << 2));
    }
    return;
  }
  double sa[4] = {0.0, 0.0, 0.0, 0.0}, sb[4] = {0.0, 0.0, 0.0, 0.0};
  float bmu[4] = {0.f, 0.f, 0.f, 0.f}, bwc[4] = {0.f, 0.f, 0.f, 0.f}, bbc[4] = {0.f, 0.f, 0.f, 0.f};
  if (stats && bn.x) {                     // this thread's four columns of the BatchNorm's forward coefficients
    const int p0 = nb0 * 16 + (threadIdx.x % QW) * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bmu[j] = bn.mean[p0 + j];
      if (!BF) {
        bwc[j] = bn.invstd[p0 + j] * (bn.weight ? bn.weight[p0 + j] : 1.0f);
        bbc[j] = -bmu[j] * bwc[j] + (bn.bias ? bn.bias[p0 + j] : 0.0f);
      }
    }
  }
#pragma unroll 4
  for (int i = threadIdx.x; i < nrows * QW; i += NT) {
    const int r = i / QW, q = i % QW;
    f32x4 v = *reinterpret_cast<const f32x4 *>(Ct + r * WS + ((q ^ (r & 15)) << 2));
    if (bias) {
      const float *bb = bias + nb0 * 16 + q * 4;
      v[0] += bb[0]; v[1] += bb[1]; v[2] += bb[2]; v[3] += bb[3];
    }
    if (res) { // out = conv + res: the residual / lateral add of the consumer folded into the write-out
      if (BF) {
        // bf16 storage: the separate add would read the convolution's STORED value -- round it first, add the stored
        // residual in fp32, round once more below: bit for bit what k_add2<bf16> / torch's bf16 add give
        const bf16x4w rr = *reinterpret_cast<const bf16x4w *>(reinterpret_cast<const __bf16 *>(res) + (row0 + r) * co +
                                                              nb0 * 16 + q * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (float)(__bf16)v[j] + (float)rr[j];
      } else {
        const f32x4 rr = *reinterpret_cast<const f32x4 *>(res + (row0 + r) * co + nb0 * 16 + q * 4);
        v[0] += rr[0]; v[1] += rr[1]; v[2] += rr[2]; v[3] += rr[3];
      }
    }
    if (BF) {
      bf16x4w o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
      *reinterpret_cast<bf16x4w *>(reinterpret_cast<__bf16 *>(out) + (row0 + r) * co + nb0 * 16 + q * 4) = o;
      if (stats) { v[0] = (float)o[0]; v[1] = (float)o[1]; v[2] = (float)o[2]; v[3] = (float)o[3]; }
    } else {
      *reinterpret_cast<f32x4 *>(out + (row0 + r) * co + nb0 * 16 + q * 4) = v;
    }
    if (stats) {
      if (BF && bn.x) {                    // (v holds the rounded, stored d_out here)
        const int64_t at = (row0 + r) * co + nb0 * 16 + q * 4;
        const bf16x4w xv = *reinterpret_cast<const bf16x4w *>(reinterpret_cast<const __bf16 *>(bn.x) + at);
        const bf16x4w ov = *reinterpret_cast<const bf16x4w *>(reinterpret_cast<const __bf16 *>(bn.out) + at);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float d = ((float)ov[j] > 0.0f) ? v[j] : v[j] * bn.leak;
          sa[j] += (double)d;
          sb[j] += (double)((float)xv[j] - bmu[j]) * (double)d;
        }
      } else if (bn.x) {                   // backward statistics of the BatchNorm whose d_out this tile is
        const f32x4 xv = *reinterpret_cast<const f32x4 *>(bn.x + (row0 + r) * co + nb0 * 16 + q * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float o = xv[j] * bwc[j] + bbc[j];
          const float d = (o > 0.0f) ? v[j] : v[j] * bn.leak;
          sa[j] += (double)d;
          sb[j] += (double)(xv[j] - bmu[j]) * (double)d;
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) { sa[j] += (double)v[j]; sb[j] += (double)v[j] * (double)v[j]; }
      }
    }
  }
  if (stats) {
    __syncthreads();                       // every read of the tile is done: its LDS holds the partial sums now
    double *red = reinterpret_cast<double *>(smem); // [NT][8] = 16 / 32 KiB <= (kT2 + 1) * WS * 4 for kT2 >= 64
#pragma unroll
    for (int j = 0; j < 4; ++j) { red[threadIdx.x * 8 + j] = sa[j]; red[threadIdx.x * 8 + 4 + j] = sb[j]; }
    __syncthreads();
    if (threadIdx.x < 16 * NW * NCB) {
      const int q = threadIdx.x >> 2, j = threadIdx.x & 3;
      double a = 0.0, b = 0.0;
#pragma unroll
      for (int t = 0; t < NT / QW; ++t) { a += red[(t * QW + q) * 8 + j]; b += red[(t * QW + q) * 8 + 4 + j]; }
      stats[(tile * 2 + 0) * co + nb0 * 16 + threadIdx.x] = a;
      stats[(tile * 2 + 1) * co + nb0 * 16 + threadIdx.x] = b;
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
  const StreamJob job{table, nullptr, blocks, V, vol, tile_rows};
  return launch_wide_blocks_jobs(&job, 1, (hipStream_t)stream_);
}

int aabr::launch_wide_blocks_jobs(const StreamJob *jobs, int n, hipStream_t st) {
  // one launch per workgroup size: a tile of T rows needs ceil(T / 64) waves, and waves beyond that would walk the whole
  // offset loop for nothing (T = 64 and T = 96..128 are what the dispatch rules produce: two launches per list)
  for (int nwv = 1; nwv <= 4; ++nwv) {
    StreamJobs js;
    js.n = 0;
    uint64_t blocks = 0;
    auto flush = [&]() -> int {
      if (js.n == 0 || blocks == 0) { js.n = 0; blocks = 0; return AABR_OK; }
      js.first[js.n] = (uint32_t)blocks;
      hipLaunchKernelGGL(k_build_tileT, dim3((unsigned)blocks), dim3(64 * nwv), 0, st, js);
      js.n = 0;
      blocks = 0;
      return AABR_OK;
    };
    for (int j = 0; j < n; ++j) {
      if ((jobs[j].T + 63) / 64 != nwv) continue;
      const uint64_t b = (uint64_t)((jobs[j].V + jobs[j].T - 1) / jobs[j].T);
      if (js.n == kStreamJobsMax || blocks + b >= (1ull << 31)) flush();
      AABR_CHECK_ARG(b < (1ull << 31), "too many tiles in one rule book");
      js.j[js.n] = jobs[j];
      js.first[js.n] = (uint32_t)blocks;
      blocks += b;
      ++js.n;
    }
    flush();
  }
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
  // up to 64 input channels (channel groups of 32 / 64: 8 KiB of stage, ~100 registers) 112-row tiles with a single
  // stage buffer fit FOUR workgroups per CU (36.9 KiB each): 64->64 at 200k rows 204 -> 190 us, at 282k rows 180 -> 171
  // (round 4, measured with AABR_WIDE_NBUF / AABR_WIDE_ROWS); 128-channel groups stay at 128 rows / three per CU
  int T = n_in <= 64 ? 112 : 128;
  {
    const int64_t slabs = n_out / 64;
    if (((V_out + T - 1) / T) * slabs <= 512)
      for (int t = 64; t < T; t += 16)
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

extern "C" int64_t aabr_conv_wide_stats_doubles(int64_t V_out, int tile_rows, int n_out) {
  if (V_out <= 0 || tile_rows <= 0 || n_out <= 0) return 0;
  return ((V_out + tile_rows - 1) / tile_rows) * 2 * n_out;
}

extern "C" int aabr_conv_forward_wide_stats(const float *in_feats, int n_in, int64_t rows_in, float *out_feats,
                                            int n_out, int64_t V_out, const int32_t *blocks, int tile_rows, int vol,
                                            const float *bias, int flags, const float *wpack, const float *residual,
                                            double *stats, void *stream_);

extern "C" int aabr_conv_forward_wide_res(const float *in_feats, int n_in, int64_t rows_in, float *out_feats,
                                          int n_out, int64_t V_out, const int32_t *blocks, int tile_rows, int vol,
                                          const float *bias, int flags, const float *wpack, const float *residual,
                                          void *stream_) {
  return aabr_conv_forward_wide_stats(in_feats, n_in, rows_in, out_feats, n_out, V_out, blocks, tile_rows, vol, bias,
                                      flags, wpack, residual, nullptr, stream_);
}

static int wide_launch_f32(const float *in_feats, int n_in, int64_t rows_in, float *out_feats, int n_out, int64_t V_out,
                           const int32_t *blocks, int tile_rows, int vol, const float *bias, int flags,
                           const float *wpack, const float *residual, double *stats, BnBwdStats bn, void *stream_);

extern "C" int aabr_conv_forward_wide_stats(const float *in_feats, int n_in, int64_t rows_in, float *out_feats,
                                            int n_out, int64_t V_out, const int32_t *blocks, int tile_rows, int vol,
                                            const float *bias, int flags, const float *wpack, const float *residual,
                                            double *stats, void *stream_) {
  return wide_launch_f32(in_feats, n_in, rows_in, out_feats, n_out, V_out, blocks, tile_rows, vol, bias, flags, wpack,
                         residual, stats, BnBwdStats{}, stream_);
}

extern "C" int aabr_conv_forward_wide_bwd_stats(const float *in_feats, int n_in, int64_t rows_in, float *out_feats,
                                                int n_out, int64_t V_out, const int32_t *blocks, int tile_rows, int vol,
                                                const float *bias, int flags, const float *wpack, const float *residual,
                                                double *stats, const float *bn_in, const float *save_mean,
                                                const float *save_invstd, const float *bn_weight, const float *bn_bias,
                                                float leakiness, void *stream_) {
  AABR_CHECK_ARG(stats && bn_in && save_mean && save_invstd, "null pointer");
  AABR_CHECK_ARG(leakiness >= 0.0f, "the activation sign is recomputed from the BatchNorm input: leakiness >= 0");
  AABR_CHECK_ARG(((uintptr_t)bn_in & 15) == 0, "the BatchNorm input must be 16-byte aligned");
  BnBwdStats bn{bn_in, save_mean, save_invstd, bn_weight, bn_bias, leakiness};
  return wide_launch_f32(in_feats, n_in, rows_in, out_feats, n_out, V_out, blocks, tile_rows, vol, bias, flags, wpack,
                         residual, stats, bn, stream_);
}

static int wide_launch_f32(const float *in_feats, int n_in, int64_t rows_in, float *out_feats, int n_out, int64_t V_out,
                           const int32_t *blocks, int tile_rows, int vol, const float *bias, int flags,
                           const float *wpack, const float *residual, double *stats, BnBwdStats bn, void *stream_) {
  hipStream_t st = (hipStream_t)stream_;
  AABR_CHECK_ARG(((uintptr_t)residual & 15) == 0, "residual must be 16-byte aligned");
  AABR_CHECK_ARG(!stats || (tile_rows >= 64 && ((uintptr_t)stats & 7) == 0), "statistics need tiles of >= 64 rows");
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
  // bit 1: the waves raise their priority for the matrix phase of a step (s_setprio): the wave that holds its operands
  // gets the pipe, the others issue their gathers -- measured 331 -> 323.5 us on the dominant instance; WIDE_PRIO knob 0: off
  const int flip = ((flags >> 1) & 1) | (knob(K_WIDE_PRIO) == 0 ? 0 : 2);
  const int kg = nkc >= 4 ? 4 : nkc;
  // LDS stage buffers: with 128-channel groups the double-buffered stage (32 KiB) allows two workgroups per CU, a
  // single buffer three (49 KiB each) at the price of a second barrier per pair: measured +4...+10 % (128->128 at 84k
  // rows 380 -> 367 us, 256->256 1366 -> 1272 us); narrower groups fit three workgroups with the double buffer
  int nbuf = (kg == 4 || kg <= 2) ? 1 : 2;   // kg <= 2: single buffer + 112-row tiles = four workgroups per CU (above)
  {                                                // tuning experiments only
    const int v = knob(K_WIDE_NBUF);
    if (v == 1 || v == 2) nbuf = v;
  }
#define AABR_LAUNCH_WIDE(KERNEL, NAME, LDS, ...)                                                          \
  do {                                                                                                    \
    static DynLdsOnce attr;                                                                               \
    AABR_CHECK_HIP(dyn_lds_once(attr, (const void *)KERNEL, 80 * 1024));          \
    g_last_variant = NAME;                                                                                \
    hipLaunchKernelGGL(KERNEL, grid, dim3(256), (LDS), st, __VA_ARGS__);                                  \
  } while (0)
  {
#define AABR_WIDE_CS_N(KG, D, NB)                                                                         \
  AABR_LAUNCH_WIDE((k_conv_cs<KG, D, NB>), "k_conv_cs<" #KG "," #D "," #NB ">",                           \
                   (size_t)((tile_rows + 1) * kWS + NB * 2 * 16 * KG * 32) * sizeof(float), in_feats, n_in, in_bytes,   \
                   out_feats, n_out, V_out, blocks, words_bytes, vol, flip, wpack, wp_bytes, bias, tile_rows, residual, stats, bn)
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

// ---- offset split for coarse maps ---------------------------------------------------------------------------------------
// The coarse FPN scales (49 .. 5,565 rows x 128-256 planes) give the wide kernel 4 .. 180 (tile, slab) items -- a fraction
// of the chip -- and the 64-row-tile kernels that ran them instead re-fetch every block's weights (8-10 TB/s of L2->CU
// traffic for 28-35 TFLOP/s).  Split: P parts per (tile, slab), each sweeping vol / P filter offsets with the wide
// kernel's economy (weights per offset in registers, gathered rows shared through LDS) into its own fp32 partial tile;
// k_split_reduce sums the parts in part order -- fixed order, bit-reproducible -- and applies bias / residual / the
// bf16 rounding.  Scratch: P x V_out x n_out floats.  Measured on the bench's coarse rule books (profiles/r04_conv_split_ab.txt):
// 128->128 at 5,565 rows 62 -> 53 us, 256->256 at 1,382 rows 83 -> 62 us, at 332 rows 47 -> 28 us; single stage buffer (three
// workgroups per CU) over the double one: -10 %; below 8 (tile, slab) items the 16-column item kernel stays ahead.
namespace aabr {
template <bool BF>
__global__ __launch_bounds__(256) void k_split_reduce(const float *__restrict__ parts, int nparts, int64_t n4, int co4,
                                                      const float *__restrict__ bias, const float *__restrict__ res,
                                                      void *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  f32x4 v = reinterpret_cast<const f32x4 *>(parts)[i];
  for (int p = 1; p < nparts; ++p) {
    const f32x4 w = reinterpret_cast<const f32x4 *>(parts)[(int64_t)p * n4 + i];
    v[0] += w[0]; v[1] += w[1]; v[2] += w[2]; v[3] += w[3];
  }
  if (bias) {
    const float *bb = bias + (i % co4) * 4;
    v[0] += bb[0]; v[1] += bb[1]; v[2] += bb[2]; v[3] += bb[3];
  }
  if (res) {
    if (BF) {      // the sum of two stored bf16 values (see k_conv_cs' write-out)
      const bf16x4w rr = reinterpret_cast<const bf16x4w *>(res)[i];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = (float)(__bf16)v[j] + (float)rr[j];
    } else {
      const f32x4 rr = reinterpret_cast<const f32x4 *>(res)[i];
      v[0] += rr[0]; v[1] += rr[1]; v[2] += rr[2]; v[3] += rr[3];
    }
  }
  if (BF) {
    bf16x4w o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
    reinterpret_cast<bf16x4w *>(out)[i] = o;
  } else {
    reinterpret_cast<f32x4 *>(out)[i] = v;
  }
}
} // namespace aabr

// (P << 16) | tile_rows when this launch should go to aabr_conv_forward_wide_split, else 0.  Asked after
// aabr_conv_wide_tile_rows declined (fewer than 320 (tile, slab) items).
extern "C" int aabr_conv_wide_split(int n_in, int n_out, int64_t rows_in, int64_t V_out, int vol) {
  if (n_in <= 0 || n_out <= 0 || (n_in & 31) || (n_out & 63) || vol <= 1 || vol > kMaxVol || V_out <= 0) return 0;
  if (rows_in >= (1ll << 23) || rows_in * n_in * 4 >= (1ll << 31)) return 0;
  if (n_in < 64 || (n_in > 128 && (n_in & 127))) return 0;
  if (knob(K_WIDE_SPLIT) == 0) return 0;
  if (((V_out + 63) / 64) * (n_out / 64) >= 320) return 0;
  int T = V_out >= 1024 ? 96 : 64;     // fp32, round 5: 96-row tiles from ~1 k rows on (5,565 rows 54 -> 49 us, 1,382 rows 57 -> 53; 332 rows 27 vs 29)
  { const int v = knob(K_SPLIT_ROWS); if (v >= 64 && v <= 128 && (v & 15) == 0) T = v; }   // (A/B)
  const int64_t items = ((V_out + T - 1) / T) * (n_out / 64);
  const int min_items = knob(K_SPLIT_MIN_ITEMS) == kKnobUnset ? 8 : knob(K_SPLIT_MIN_ITEMS);   // below: the 16-column item kernel wins
  if (items < min_items) return 0;
  const int target = knob(K_SPLIT_TARGET) == kKnobUnset ? 768 : knob(K_SPLIT_TARGET);   // workgroups aimed at (round 5: 512 .. 1536 re-measured; 768 and 1280 best by ~3 %)
  int P = (int)((target + items - 1) / items);
  if (P > vol) P = vol;
  if (P > 32) P = 32;
  {                                                // tuning experiments only
    const int v = knob(K_WIDE_SPLIT);
    if (v >= 2 && v <= 32) P = v < vol ? v : vol;
  }
  if (P < 2) return 0;
  if (wide_words(V_out, vol, T) * 4 >= (1ll << 31) || (int64_t)vol * n_in * n_out * 4 >= (1ll << 31)) return 0;
  return (P << 16) | T;
}

extern "C" int64_t aabr_conv_wide_split_scratch_floats(int64_t V_out, int n_out, int parts) {
  return V_out > 0 && n_out > 0 && parts > 0 ? (int64_t)parts * V_out * n_out : 0;
}

extern "C" int aabr_conv_forward_wide_split(const float *in_feats, int n_in, int64_t rows_in, float *out_feats, int n_out,
                                            int64_t V_out, const int32_t *blocks, int tile_rows, int vol,
                                            const float *bias, int flags, const float *wpack, const float *residual,
                                            int parts, float *scratch, void *stream_) {
  hipStream_t st = (hipStream_t)stream_;
  AABR_CHECK_ARG(parts >= 2 && parts <= 32 && parts <= vol && scratch && ((uintptr_t)scratch & 15) == 0,
                 "2 <= parts <= min(32, vol) and a 16-byte aligned scratch of parts x V_out x n_out floats");
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
  const int nkc = n_in / 32;
  const int64_t wp_bytes = (int64_t)vol * nkc * (n_out / 16) * 2048;
  AABR_CHECK_ARG(wp_bytes < (1ll << 31), "packed weights must be < 2 GiB");
  AABR_CHECK_ARG(n_in <= 128 || (n_in & 127) == 0, "n_in above 128 must be a multiple of 128");
  dim3 grid((unsigned)((V_out + tile_rows - 1) / tile_rows), (unsigned)((n_out / 64) * parts));
  const int flip = ((flags >> 1) & 1) | (knob(K_WIDE_PRIO) == 0 ? 0 : 2) | (parts << 8);
  const int kg = nkc >= 4 ? 4 : nkc;
  const int nbuf = knob(K_SPLIT_NBUF) == 2 ? 2 : 1;   // single stage buffer: three workgroups per CU (latency-bound launches)
#define AABR_SPLIT_CS_N(KG, NB)                                                                                        \
  do {                                                                                                                 \
    static DynLdsOnce attr;                                                                               \
    AABR_CHECK_HIP(dyn_lds_once(attr, (const void *)(k_conv_cs<KG, 0, NB>), 80 * 1024));          \
    g_last_variant = "k_conv_cs<" #KG ",0," #NB ",split>";                                                             \
    hipLaunchKernelGGL((k_conv_cs<KG, 0, NB>), grid, dim3(256),                                                        \
                       (size_t)((tile_rows + 1) * kWS + NB * 2 * 16 * KG * 32) * sizeof(float), st, in_feats, n_in,     \
                       in_bytes, scratch, n_out, V_out, blocks, words_bytes, vol, flip, wpack, wp_bytes,                \
                       (const float *)nullptr, tile_rows, (const float *)nullptr, (double *)nullptr, BnBwdStats{});     \
  } while (0)
#define AABR_SPLIT_CS(KG) do { if (nbuf == 1) AABR_SPLIT_CS_N(KG, 1); else AABR_SPLIT_CS_N(KG, 2); } while (0)
  if (kg == 2) AABR_SPLIT_CS(2); else if (kg == 3) AABR_SPLIT_CS(3); else AABR_SPLIT_CS(4);
#undef AABR_SPLIT_CS
#undef AABR_SPLIT_CS_N
  const int64_t n4 = V_out * n_out / 4;
  hipLaunchKernelGGL((k_split_reduce<false>), dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, scratch, parts, n4,
                     n_out / 4, bias, residual, (void *)out_feats);
  AABR_CHECK_LAUNCH();
  return AABR_OK;
}

// ---- bf16 feature storage (extension): the same kernel on 128-byte row chunks of 64 bf16 channels ------------------
// 16-column blocks per wave of the bf16 launch: 2 (128-column slabs) for n_out % 128 == 0 up to 128 input channels
// (two weight register sets of 16 x NCB x KG registers), else 1; WIDE_NCB knob: 1 forces 64-column slabs
static int wide_bf16_ncb(int n_in, int n_out) {
  if (knob(K_WIDE_NCB) == 1) return 1;
  return ((n_out & 127) == 0 && n_in <= 128) ? 2 : 1;
}

// 0: use aabr_conv_forward_bf16 (64-row tiles); otherwise rows per tile for aabr_conv_forward_wide_bf16
extern "C" int aabr_conv_wide_tile_rows_bf16(int n_in, int n_out, int64_t rows_in, int64_t V_out, int vol) {
  if (n_in <= 0 || n_out <= 0 || (n_in & 63) || (n_out & 63) || vol <= 0 || vol > kMaxVol) return 0;
  if (rows_in >= (1ll << 23) || rows_in * n_in * 2 >= (1ll << 31)) return 0;
  if (n_in > 256 && (n_in & 255)) return 0; // channel groups of 256: every load of the inner loop unconditional
  // bf16: two MFMAs per block and 64-channel chunk -- the gather / stage / barrier skeleton sets the pace, so more
  // resident workgroups pay: 96-row tiles with a single stage buffer (measured on the bench's rule books: convolution
  // time of a bf16 step 5.25 -> 5.01 ms against 128 rows + two buffers; 64 rows: 5.24)
  // 128-column slabs when the layer has them: the kernel is bound by the CU's random-row gather rate and a 64-column
  // slab gathers every row once per slab.  Their fp32 tile is 512 B per row: 64 rows (33 KiB + 8 KiB of stage) keep
  // three workgroups per CU; 64-column slabs keep the 96-row tiles of round 2.
  const int ncb = wide_bf16_ncb(n_in, n_out);
  int T = ncb == 2 ? 64 : 96;
  {
    const int64_t slabs = n_out / (64 * ncb);
    if (((V_out + T - 1) / T) * slabs <= 512)
      for (int t = 64; t < T; t += 16)
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
  return (((V_out + T - 1) / T) * (n_out / (64 * ncb)) >= 320) ? T : 0;
}

extern "C" int aabr_conv_forward_wide_bf16_stats(const uint16_t *in_feats, int n_in, int64_t rows_in,
                                                 uint16_t *out_feats, int n_out, int64_t V_out, const int32_t *blocks,
                                                 int tile_rows, int vol, const float *bias, int flags,
                                                 const uint16_t *wpack, double *stats, void *stream_);

extern "C" int aabr_conv_forward_wide_bf16(const uint16_t *in_feats, int n_in, int64_t rows_in, uint16_t *out_feats,
                                           int n_out, int64_t V_out, const int32_t *blocks, int tile_rows, int vol,
                                           const float *bias, int flags, const uint16_t *wpack, void *stream_) {
  return aabr_conv_forward_wide_bf16_stats(in_feats, n_in, rows_in, out_feats, n_out, V_out, blocks, tile_rows, vol, bias,
                                           flags, wpack, nullptr, stream_);
}

static int wide_launch_bf16(const uint16_t *in_feats, int n_in, int64_t rows_in, uint16_t *out_feats, int n_out,
                            int64_t V_out, const int32_t *blocks, int tile_rows, int vol, const float *bias, int flags,
                            const uint16_t *wpack, const uint16_t *residual, double *stats, BnBwdStats bn, void *stream_);

extern "C" int aabr_conv_forward_wide_bf16_stats(const uint16_t *in_feats, int n_in, int64_t rows_in,
                                                 uint16_t *out_feats, int n_out, int64_t V_out, const int32_t *blocks,
                                                 int tile_rows, int vol, const float *bias, int flags,
                                                 const uint16_t *wpack, double *stats, void *stream_) {
  return wide_launch_bf16(in_feats, n_in, rows_in, out_feats, n_out, V_out, blocks, tile_rows, vol, bias, flags, wpack,
                          nullptr, stats, BnBwdStats{}, stream_);
}

// the general bf16-storage form behind the compiled pass: optional residual (out = conv + residual, both rounded as
// stored: see the write-out), optional statistics of the following BatchNorm (bn_in NULL) or backward statistics of the
// BatchNorm whose d_out this launch writes (bn_in, bn_out, save_mean, leakiness as aabr_conv_forward_wide_bf16_bwd_stats)
extern "C" int aabr_conv_forward_wide_bf16_res(const uint16_t *in_feats, int n_in, int64_t rows_in, uint16_t *out_feats,
                                               int n_out, int64_t V_out, const int32_t *blocks, int tile_rows, int vol,
                                               const float *bias, int flags, const uint16_t *wpack,
                                               const uint16_t *residual, double *stats, const uint16_t *bn_in,
                                               const uint16_t *bn_out, const float *save_mean, float leakiness,
                                               void *stream_) {
  AABR_CHECK_ARG(((uintptr_t)residual & 7) == 0, "residual must be 8-byte aligned");
  BnBwdStats bn{};
  if (bn_in) {
    AABR_CHECK_ARG(stats && bn_out && save_mean, "null pointer");
    AABR_CHECK_ARG((((uintptr_t)bn_in | (uintptr_t)bn_out) & 7) == 0, "the BatchNorm's input / output must be 8-byte aligned");
    bn = BnBwdStats{reinterpret_cast<const float *>(bn_in), save_mean, nullptr, nullptr, nullptr, leakiness,
                    reinterpret_cast<const float *>(bn_out)};
  }
  return wide_launch_bf16(in_feats, n_in, rows_in, out_feats, n_out, V_out, blocks, tile_rows, vol, bias, flags, wpack,
                          residual, stats, bn, stream_);
}

extern "C" int aabr_conv_forward_wide_bf16_bwd_stats(const uint16_t *in_feats, int n_in, int64_t rows_in,
                                                     uint16_t *out_feats, int n_out, int64_t V_out,
                                                     const int32_t *blocks, int tile_rows, int vol, const float *bias,
                                                     int flags, const uint16_t *wpack, double *stats,
                                                     const uint16_t *bn_in, const uint16_t *bn_out,
                                                     const float *save_mean, float leakiness, void *stream_) {
  AABR_CHECK_ARG(stats && bn_in && bn_out && save_mean, "null pointer");
  AABR_CHECK_ARG((((uintptr_t)bn_in | (uintptr_t)bn_out) & 7) == 0, "the BatchNorm's input / output must be 8-byte aligned");
  BnBwdStats bn{reinterpret_cast<const float *>(bn_in), save_mean, nullptr, nullptr, nullptr, leakiness,
                reinterpret_cast<const float *>(bn_out)};
  return wide_launch_bf16(in_feats, n_in, rows_in, out_feats, n_out, V_out, blocks, tile_rows, vol, bias, flags, wpack,
                          nullptr, stats, bn, stream_);
}

static int wide_launch_bf16(const uint16_t *in_feats, int n_in, int64_t rows_in, uint16_t *out_feats, int n_out,
                            int64_t V_out, const int32_t *blocks, int tile_rows, int vol, const float *bias, int flags,
                            const uint16_t *wpack, const uint16_t *residual, double *stats, BnBwdStats bn, void *stream_) {
  hipStream_t st = (hipStream_t)stream_;
  AABR_CHECK_ARG(!stats || (tile_rows >= 64 && ((uintptr_t)stats & 7) == 0), "statistics need tiles of >= 64 rows");
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
  const int ncb = wide_bf16_ncb(n_in, n_out);
  dim3 grid((unsigned)((V_out + tile_rows - 1) / tile_rows), (unsigned)(n_out / (64 * ncb)));
  const int flip = ((flags >> 1) & 1) | (knob(K_WIDE_PRIO) == 1 ? 2 : 0);   // bf16: priority only on request (measured below)
  const int kg = nkc >= 4 ? 4 : nkc;
  int nbuf = 1;
  {                                                // tuning experiments only
    const int v = knob(K_WIDE_NBUF);
    if (v == 1 || v == 2) nbuf = v;
  }
  const float *in_f = reinterpret_cast<const float *>(in_feats), *wp_f = reinterpret_cast<const float *>(wpack);
  float *out_f = reinterpret_cast<float *>(out_feats);
  const float *res_f = reinterpret_cast<const float *>(residual);
constexpr int kBfSets = 2;   // gather register sets of the bf16 launches (4: measured slower, 100 -> 112 us)
#define AABR_WIDE_BF(KG, NB, NCB) AABR_WIDE_BF_S(KG, NB, NCB, ((KG) <= 2 ? kBfSets : 2))
#define AABR_WIDE_BF_D(KG, NB, NCB, D)                                                                              \
  do {                                                                                                             \
    static DynLdsOnce attr;                                                                                        \
    AABR_CHECK_HIP(dyn_lds_once(attr, (const void *)(k_conv_cs<KG, D, NB, true, NCB, kBfSets>), 80 * 1024));       \
    hipLaunchKernelGGL((k_conv_cs<KG, D, NB, true, NCB, kBfSets>), grid, dim3(256),                                \
                       (size_t)((tile_rows + 1) * kWS * NCB + NB * 2 * 16 * KG * 32) * sizeof(float), st, in_f,    \
                       n_in, in_bytes, out_f, n_out, V_out, blocks, words_bytes, vol, flip, wp_f, wp_bytes, bias,  \
                       tile_rows, res_f, stats, bn);                                                      \
  } while (0)
#define AABR_WIDE_BF_S(KG, NB, NCB, NS) AABR_WIDE_BF_L(KG, NB, NCB, NS)
#define AABR_WIDE_BF_L(KG, NB, NCB, NS)                                                                            \
  do {                                                                                                             \
    static DynLdsOnce attr;                                                                                        \
    AABR_CHECK_HIP(dyn_lds_once(attr, (const void *)(k_conv_cs<KG, 0, NB, true, NCB, NS>), 80 * 1024));          \
    g_last_variant = NCB == 2 ? "k_conv_cs<" #KG ",0," #NB ",bf16,x128>" : "k_conv_cs<" #KG ",0," #NB ",bf16>";    \
    hipLaunchKernelGGL((k_conv_cs<KG, 0, NB, true, NCB, NS>), grid, dim3(256),                   \
                       (size_t)((tile_rows + 1) * kWS * NCB + NB * 2 * 16 * KG * 32) * sizeof(float), st, in_f,    \
                       n_in, in_bytes, out_f, n_out, V_out, blocks, words_bytes, vol, flip, wp_f, wp_bytes, bias,  \
                       tile_rows, res_f, stats, bn);                                              \
  } while (0)
#define AABR_WIDE_BF_K(KG)                                                                                         \
  do {                                                                                                             \
    if (ncb == 2) { if (nbuf == 1) AABR_WIDE_BF(KG, 1, 2); else AABR_WIDE_BF(KG, 2, 2); }                          \
    else { if (nbuf == 1) AABR_WIDE_BF(KG, 1, 1); else AABR_WIDE_BF(KG, 2, 1); }                                   \
  } while (0)
#define AABR_WIDE_BF_K1(KG) /* more than 128 input channels: 64-column slabs only (wide_bf16_ncb) */                 \
  do {                                                                                                             \
    if (nbuf == 1) AABR_WIDE_BF(KG, 1, 1); else AABR_WIDE_BF(KG, 2, 1);                                            \
  } while (0)
  AABR_CHECK_ARG(ncb == 1 || kg <= 2, "128-column slabs need n_in <= 128");
#ifdef AABR_DEV
  if ((flags >> 8) & 4) {   // timing experiments (tools/tools_cs_phases.py bf16): phase clocks of the 128-channel instance
    AABR_CHECK_ARG(kg == 2, "the bf16 phase-clock variant exists for n_in = 128 only");
    if (ncb == 2) AABR_WIDE_BF_D(2, 1, 2, 4);
    else AABR_WIDE_BF_D(2, 1, 1, 4);
    AABR_CHECK_LAUNCH();
    return AABR_OK;
  }
#endif
  if (kg == 1) AABR_WIDE_BF_K(1); else if (kg == 2) AABR_WIDE_BF_K(2); else if (kg == 3) AABR_WIDE_BF_K1(3);
  else AABR_WIDE_BF_K1(4);
#undef AABR_WIDE_BF_K1
#undef AABR_WIDE_BF_K
#undef AABR_WIDE_BF
#undef AABR_WIDE_BF_S
#undef AABR_WIDE_BF_L
#undef AABR_WIDE_BF_D
  AABR_CHECK_LAUNCH();
  return AABR_OK;
}


// ---- offset split, bf16 storage ---------------------------------------------------------------------------------------
// the coarse scales of a bf16-storage pass (round 4: 50 launches of k_conv_blocks_mfma_bf16 per step, 8 % of the bf16
// step's kernel time): the same cut over the filter offsets; the parts are fp32 (the kernel's LDS tile is), the reduce
// kernel rounds once to bf16 -- the same single rounding per stored value as the unsplit kernels.
extern "C" int aabr_conv_wide_split_bf16(int n_in, int n_out, int64_t rows_in, int64_t V_out, int vol) {
  if (n_in <= 0 || n_out <= 0 || (n_in & 63) || (n_out & 63) || vol <= 1 || vol > kMaxVol || V_out <= 0) return 0;
  if (rows_in >= (1ll << 23) || rows_in * n_in * 2 >= (1ll << 31)) return 0;
  if (n_in > 256 && (n_in & 255)) return 0;
  if (knob(K_WIDE_SPLIT) == 0 || knob(K_CONV_WIDE_BF16) == 0) return 0;
  int T = 64;
  if (((V_out + 63) / 64) * (n_out / 64) >= 320) return 0;
  { const int v = knob(K_SPLIT_ROWS); if (v >= 64 && v <= 128 && (v & 15) == 0) T = v; }   // (A/B)
  const int64_t items = ((V_out + T - 1) / T) * (n_out / 64);
  const int min_items = knob(K_SPLIT_MIN_ITEMS) == kKnobUnset ? 8 : knob(K_SPLIT_MIN_ITEMS);
  if (items < min_items) return 0;
  const int target = knob(K_SPLIT_TARGET) == kKnobUnset ? 768 : knob(K_SPLIT_TARGET);   // (bf16: 768 / 1024 / 1280 -> 1156 / 1192 / 1204 us of coarse-scale convolutions per step)
  int P = (int)((target + items - 1) / items);
  if (P > vol) P = vol;
  if (P > 32) P = 32;
  {
    const int v = knob(K_WIDE_SPLIT);
    if (v >= 2 && v <= 32) P = v < vol ? v : vol;
  }
  if (P < 2) return 0;
  if (wide_words(V_out, vol, T) * 4 >= (1ll << 31) || (int64_t)vol * n_in * n_out * 2 >= (1ll << 31)) return 0;
  return (P << 16) | T;
}

extern "C" int aabr_conv_forward_wide_split_bf16_res(const uint16_t *in_feats, int n_in, int64_t rows_in,
                                                     uint16_t *out_feats, int n_out, int64_t V_out, const int32_t *blocks,
                                                     int tile_rows, int vol, const float *bias, int flags,
                                                     const uint16_t *wpack, int parts, float *scratch,
                                                     const uint16_t *residual, void *stream_);
extern "C" int aabr_conv_forward_wide_split_bf16(const uint16_t *in_feats, int n_in, int64_t rows_in, uint16_t *out_feats,
                                                 int n_out, int64_t V_out, const int32_t *blocks, int tile_rows, int vol,
                                                 const float *bias, int flags, const uint16_t *wpack, int parts,
                                                 float *scratch, void *stream_) {
  return aabr_conv_forward_wide_split_bf16_res(in_feats, n_in, rows_in, out_feats, n_out, V_out, blocks, tile_rows, vol, bias,
                                               flags, wpack, parts, scratch, nullptr, stream_);
}

extern "C" int aabr_conv_forward_wide_split_bf16_res(const uint16_t *in_feats, int n_in, int64_t rows_in,
                                                     uint16_t *out_feats, int n_out, int64_t V_out, const int32_t *blocks,
                                                     int tile_rows, int vol, const float *bias, int flags,
                                                     const uint16_t *wpack, int parts, float *scratch,
                                                     const uint16_t *residual, void *stream_) {
  hipStream_t st = (hipStream_t)stream_;
  AABR_CHECK_ARG(((uintptr_t)residual & 7) == 0, "residual must be 8-byte aligned");
  AABR_CHECK_ARG(parts >= 2 && parts <= 32 && parts <= vol && scratch && ((uintptr_t)scratch & 15) == 0,
                 "2 <= parts <= min(32, vol) and a 16-byte aligned scratch of parts x V_out x n_out floats");
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
  const int nkc = n_in / 64;
  const int64_t wp_bytes = (int64_t)vol * (n_in / 32) * (n_out / 16) * 1024;
  AABR_CHECK_ARG(wp_bytes < (1ll << 31), "packed weights must be < 2 GiB");
  AABR_CHECK_ARG(n_in <= 256 || (n_in & 255) == 0, "n_in above 256 must be a multiple of 256");
  dim3 grid((unsigned)((V_out + tile_rows - 1) / tile_rows), (unsigned)((n_out / 64) * parts));
  const int flip = ((flags >> 1) & 1) | (parts << 8);
  const int kg = nkc >= 4 ? 4 : nkc;
  const float *in_f = reinterpret_cast<const float *>(in_feats), *wp_f = reinterpret_cast<const float *>(wpack);
#define AABR_SPLIT_BF(KG)                                                                                              \
  do {                                                                                                                 \
    static DynLdsOnce attr;                                                                                        \
    AABR_CHECK_HIP(dyn_lds_once(attr, (const void *)(k_conv_cs<KG, 0, 1, true, 1, 2>), 80 * 1024));          \
    g_last_variant = "k_conv_cs<" #KG ",0,1,bf16,split>";                                                              \
    hipLaunchKernelGGL((k_conv_cs<KG, 0, 1, true, 1, 2>), grid, dim3(256),                                             \
                       (size_t)((tile_rows + 1) * kWS + 2 * 16 * KG * 32) * sizeof(float), st, in_f, n_in, in_bytes,    \
                       scratch, n_out, V_out, blocks, words_bytes, vol, flip, wp_f, wp_bytes, (const float *)nullptr,   \
                       tile_rows, (const float *)nullptr, (double *)nullptr, BnBwdStats{});                            \
  } while (0)
  if (kg == 1) AABR_SPLIT_BF(1); else if (kg == 2) AABR_SPLIT_BF(2); else if (kg == 3) AABR_SPLIT_BF(3); else AABR_SPLIT_BF(4);
#undef AABR_SPLIT_BF
  const int64_t n4 = V_out * n_out / 4;
  hipLaunchKernelGGL((k_split_reduce<true>), dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, scratch, parts, n4,
                     n_out / 4, bias, reinterpret_cast<const float *>(residual), (void *)out_feats);
  AABR_CHECK_LAUNCH();
  return AABR_OK;
}

