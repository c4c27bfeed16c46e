// conv_rs.hip -- row-stationary form of the sparse convolution for bf16 feature storage (gfx950).
//
// Same contraction as conv.hip / conv_wide.hip (reference: SCN/CPU/Convolution.cpp:45-185,
// SCN/CPU/Deconvolution.cpp:7-77):      out[o] = bias + sum_k in[table[k][o]] @ Wl[k]
//
// Why another form: with bf16 operands a 16-pair block costs 16 MFMA cycles per 32 channels and 16 columns, and the
// LDS-tile kernels (conv_wide.hip) spend several times that on the fp32 read-add-write of the output tile and on
// streaming a weight slice that is reused by ~1.5 blocks (profiles/r02_conv_instances_bf16.txt: 247 TFLOP/s = 10 % of
// the bf16 MFMA peak on the dominant 128->128 layer; the L2->CU feed of weights + gathered rows bounds it).  Here:
//
//   * a workgroup owns a UNIT of U <= 256 consecutive output rows and ALL of up to 128 output columns; the
//     accumulators of the whole unit live in registers (MFMA C/D: lane = output row, registers = output columns) --
//     no output tile in LDS, no accumulate traffic, the output is written once;
//   * inside a unit the rows are regrouped (k_build_rs, once per rule book): sorted by their 27-bit "which offsets
//     have a partner" mask and cut into groups of 16.  Rows of one surface share their mask (a floor voxel has
//     partners in its own z-plane only), so a group is active for a third of the offsets and nearly full where it
//     is active: (group, offset) items with no partner at all are skipped, the rest run as MFMA N = 16 rows with
//     absent partners reading zeros (buffer range check, no memory traffic).  Measured on the bench's rule books
//     (tools/tools_rs_fill.py): 77-83 % of the issued MFMA columns are real pairs at 9 rules per row.  Each output row
//     is still the sum over offsets ascending of exact fp32 MFMA chains: bit-reproducible, independent of grouping;
//   * per offset a wave loads its 32-column weight slice ONCE into registers and uses it for every active group of
//     the unit (~5-7 groups = 80-110 rows instead of ~24 in the 96-row LDS-tile kernel);
//   * the gathered rows of two items (32 rows) are fetched once per workgroup into a double-buffered, granule-
//     swizzled LDS stage, two items ahead in registers, and every wave reads its MFMA B operands from there.
//
// bytes through L2->CU per padded flop: 1/P + 1/W with P = padded pairs per (unit, offset), W = columns per
// workgroup: 0.046 (96-row tile x 64 columns) -> 0.02 (192 rows x 128 columns).
#include "common.h"
#include <stdlib.h>

namespace aabr {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8r __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4r __attribute__((ext_vector_type(4)));

extern thread_local const char *g_last_variant; // conv.hip

constexpr int kRsMaxVol = 32;   // one mask bit per filter offset
constexpr int kRsMaxU = 256;    // rows per unit (16 groups of 16)

// ------------------------------------------------------------------ compiled rule book, row-stationary form
//   words: [nunits][32] header: [0] = number of active offsets n, [1..n] = (offset << 16) | 16-bit set of the
//          groups with at least one partner at that offset, offsets ascending
//        | [nunits][U] perm (output row of slot, -1 = padding)
//        | [nunits][vol][U] partner row of (offset, slot), -1 = none
constexpr int kRsHdr = 32;
static inline int64_t rs_words(int64_t V, int vol, int U) {
  const int64_t nu = (V + U - 1) / U;
  return nu * (kRsHdr + U + (int64_t)vol * U);
}

__global__ __launch_bounds__(256) void k_build_rs(const int32_t *__restrict__ table, int64_t V, int vol, int U,
                                                  int32_t *__restrict__ words) {
  __shared__ unsigned long long s_key[kRsMaxU];
  __shared__ unsigned int s_mask[kRsMaxU];
  __shared__ unsigned int s_gm[16];
  const int t = threadIdx.x;
  const int64_t unit = blockIdx.x, nunits = (V + U - 1) / U;
  const int64_t row = unit * U + t;
  const bool in_unit = t < U, valid = in_unit && row < V;
  unsigned int mask = 0;
  for (int k = 0; k < vol; ++k) {
    const int e = valid ? table[(int64_t)k * V + row] : -1;
    if (e >= 0) mask |= 1u << k;
  }
  // rows without any partner (possible in strided books) sort in front (mask 0), padding slots last
  const unsigned long long key = valid ? (((unsigned long long)mask << 9) | (unsigned)t)
                                       : ((1ull << 48) | (unsigned)t);
  if (in_unit) s_key[t] = key;
  __syncthreads();
  int rank = 0;
  if (in_unit)
    for (int s = 0; s < U; ++s) rank += s_key[s] < key;
  if (in_unit) s_mask[rank] = valid ? mask : 0u;
  int32_t *hdr = words + unit * kRsHdr;
  int32_t *perm = words + nunits * kRsHdr + unit * U;
  int32_t *tp = words + nunits * (kRsHdr + U) + unit * (int64_t)vol * U;
  if (in_unit) {
    perm[rank] = valid ? (int32_t)row : -1;
    for (int k = 0; k < vol; ++k) tp[(int64_t)k * U + rank] = valid ? table[(int64_t)k * V + row] : -1;
  }
  __syncthreads();
  if (t < 16) {
    unsigned int m = 0;
    if (t * 16 < U)
      for (int i = 0; i < 16; ++i) m |= s_mask[t * 16 + i];
    s_gm[t] = m;
  }
  __syncthreads();
  if (t == 0) {
    int n = 0;
    for (int k = 0; k < vol; ++k) {
      unsigned bits = 0;
      for (int g = 0; g < 16; ++g) bits |= ((s_gm[g] >> k) & 1u) << g;
      if (bits) hdr[++n] = (int32_t)(((unsigned)k << 16) | bits);
    }
    hdr[0] = n;
    for (int i = n + 1; i < kRsHdr; ++i) hdr[i] = 0;
  }
}

__device__ __align__(16) unsigned char g_rs_zero[1024];   // the row an absent partner reads (module-zero-initialised)

// ------------------------------------------------------------------ the kernel
// One workgroup = 4 consumer waves (wave w: the MFMAs of columns [32 w, 32 w + 32) of the slab, or 16 with NCB = 1)
// + 2 loader waves.  Everything the consumers need -- the weight slices of an offset, then the gathered rows of that
// offset's items, two items (32 rows) per slot -- travels through ONE ring of LDS slots in the order it is used,
// filled by the loader waves with LDS-DMA (global_load_lds_dwordx4: per-lane source address = a row gather; the
// destination is linear, so the XOR swizzle of the row image is applied to the SOURCE granule).  The loaders' vmcnt
// stream holds nothing but these DMAs (entries and headers come through the scalar cache), so a counted
// s_waitcnt keeps a dozen slots (~100 KiB) in flight per CU; the consumers issue no vector-memory instruction at all
// inside the loop.  One s_barrier per step hands the next slot(s) over.
// KC = 32-channel chunks per input row (n_in = 32 KC), NCB = 16-column blocks per consumer wave.
struct RsSeq {               // position in the workgroup's item sequence (wave-uniform)
  int u, uend, j, nact, k;
  unsigned bits;
  bool first, done;
};

template <int KC, int NCB, int NG>
__global__ __launch_bounds__(384, 2) void k_conv_rsq(
    const __bf16 *__restrict__ in, __bf16 *__restrict__ out, int co, int64_t V_out, const int32_t *__restrict__ words,
    int vol, int U, int upw, int wflip, const __bf16 *__restrict__ Wp, const float *__restrict__ bias, int NS) {
  constexpr int RB = KC * 64;                 // bytes per input row
  constexpr int SLOT = 32 * RB;               // ring slot: the rows of two items, or as many bytes of weights
  constexpr int NDMA = SLOT / 1024;           // DMA instructions (1 KiB each) per slot
  constexpr int IPW = NDMA / 2;               // ... per loader wave
  constexpr int LPR = RB / 16;                // lanes per row in a gather piece
  constexpr int RPP = 64 / LPR;               // rows per gather piece
  constexpr int SWZ = LPR - 1;
  constexpr int NWS = 2 * NCB;                // weight slots per offset: 4 waves x NCB x KC KiB
  static_assert(KC == 2 || KC == 4, "n_in = 64 or 128");
  extern __shared__ __align__(16) unsigned char ring[];
  typedef const __attribute__((address_space(4))) int32_t *cptr;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int nnb = co >> 4;
  const int nb_wg = blockIdx.y * (4 * NCB);
  const int64_t nunits = (V_out + U - 1) / U;
  const int u0 = blockIdx.x * upw;
  const int u1 = (int)((int64_t)(u0 + upw) < nunits ? (u0 + upw) : nunits);
  cptr hdr = (cptr)words;
  const int32_t *perm = words + nunits * kRsHdr;
  cptr tp = (cptr)(words + nunits * (kRsHdr + U));
  const int D = NS - (NWS + 1);               // slots in flight; a slot being filled is never one being read

  auto seq_offset = [&](RsSeq &q) {           // move to the next active offset (possibly of a later unit)
    ++q.j;
    while (q.j > q.nact) {
      ++q.u;
      if (q.u >= q.uend) { q.done = true; q.bits = 0; q.k = 0; return; }
      q.nact = hdr[(int64_t)q.u * kRsHdr];
      q.j = 1;
    }
    const unsigned w = (unsigned)hdr[(int64_t)q.u * kRsHdr + q.j];
    q.k = (int)(w >> 16);
    q.bits = w & 0xffffu;
    q.first = true;
  };
  auto seq_init = [&](RsSeq &q) {
    q.u = u0 - 1; q.uend = u1; q.j = 1; q.nact = 0; q.k = 0; q.bits = 0; q.first = false; q.done = false;
    seq_offset(q);
  };
  auto seq_next = [&](RsSeq &q) {             // past the current pair
    unsigned r = q.bits & (q.bits - 1);
    r &= r - 1;
    if (r) { q.bits = r; q.first = false; }
    else seq_offset(q);
  };

  if (wave >= 4) {
    // ------------------------------------------------------------------ loader waves
    const int pw = wave - 4;
    RsSeq head, tail;
    seq_init(head);
    seq_init(tail);
    int wpend = tail.done ? 0 : NWS;           // weight slots of tail's offset still to issue
    int tpos = 0, inflight = 0;
    const unsigned char *inb = reinterpret_cast<const unsigned char *>(in);
    const unsigned char *wpb = reinterpret_cast<const unsigned char *>(Wp);
    const int rl = lane / LPR, pl = lane % LPR;
    auto issue_slot = [&]() {                  // the next slot of the sequence into ring position tpos
      unsigned char *dst = ring + tpos * SLOT + pw * IPW * 1024;
      if (wpend > 0) {
        const int widx = NWS - wpend;
        const int kW = wflip ? vol - 1 - tail.k : tail.k;
#pragma unroll
        for (int i = 0; i < IPW; ++i) {
          const int gp = widx * NDMA + pw * IPW + i;
          const int w = gp / (NCB * KC), r = gp % (NCB * KC), cb = r / KC, kc = r % KC;
          const unsigned char *src = wpb + (((int64_t)kW * KC + kc) * nnb + nb_wg + w * NCB + cb) * 1024 + lane * 16;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                           (__attribute__((address_space(3))) void *)(dst + i * 1024), 16, 0, 0);
        }
        --wpend;
      } else {
        const unsigned b = tail.bits;
        const int g0 = __builtin_ctz(b);
        const unsigned r = b & (b - 1);
        const int g1 = r ? __builtin_ctz(r) : -1;
        cptr ent = tp + ((int64_t)tail.u * vol + tail.k) * U;
#pragma unroll
        for (int i = 0; i < IPW; ++i) {
          const int t = pw * IPW + i;          // piece: rows t RPP .. t RPP + RPP - 1 of the pair
          const int row0 = t * RPP;
          const int g = row0 < 16 ? g0 : g1;
          int e = -1;
          if (g >= 0) {
            cptr ep = ent + g * 16 + (row0 & 15);
#pragma unroll
            for (int q = 0; q < RPP; ++q) {
              const int v = ep[q];
              e = rl == q ? v : e;
            }
          }
          const int row = row0 + rl;
          const unsigned char *src = e >= 0 ? inb + (int64_t)e * RB + ((pl ^ (row & SWZ)) << 4) : g_rs_zero + (pl << 4);
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                           (__attribute__((address_space(3))) void *)(dst + i * 1024), 16, 0, 0);
        }
        seq_next(tail);
        if (!tail.done && tail.first) wpend = NWS;
      }
      tpos = tpos + 1 == NS ? 0 : tpos + 1;
      ++inflight;
    };
    // counted waits need immediates: two cases per step (the next step takes 1 slot or NWS + 1)
    auto wait_next = [&](int need) {
      if ((tail.done && wpend == 0) || inflight - need < 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); return; }
      const int allow = (inflight - need) * IPW;
      // allow is one of a few values in steady state; round DOWN to the nearest immediate we have
      if (allow >= 60) asm volatile("s_waitcnt vmcnt(60)" ::: "memory");
      else if (allow >= 56) asm volatile("s_waitcnt vmcnt(56)" ::: "memory");
      else if (allow >= 52) asm volatile("s_waitcnt vmcnt(52)" ::: "memory");
      else if (allow >= 48) asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
      else if (allow >= 44) asm volatile("s_waitcnt vmcnt(44)" ::: "memory");
      else if (allow >= 40) asm volatile("s_waitcnt vmcnt(40)" ::: "memory");
      else if (allow >= 36) asm volatile("s_waitcnt vmcnt(36)" ::: "memory");
      else if (allow >= 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
      else if (allow >= 28) asm volatile("s_waitcnt vmcnt(28)" ::: "memory");
      else if (allow >= 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
      else if (allow >= 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
      else if (allow >= 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else if (allow >= 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else if (allow >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (allow >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else if (allow >= 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    auto cost = [&](const RsSeq &q) { return q.done ? 0 : (q.first ? NWS + 1 : 1); };
    while (inflight < D && !(tail.done && wpend == 0)) issue_slot();
    wait_next(cost(head));
    asm volatile("s_barrier" ::: "memory");
    while (!head.done) {
      const int c = cost(head);
      seq_next(head);
      inflight -= c;                           // the consumers take these slots during this step
      // refill: the slots taken in the PREVIOUS steps are free; the ones being read now are not (NS >= D + NWS + 1)
      while (inflight < D && !(tail.done && wpend == 0)) issue_slot();
      wait_next(cost(head));
      asm volatile("s_barrier" ::: "memory");
    }
    return;
  }

  // -------------------------------------------------------------------- consumer waves
  const int g4 = lane >> 4, c16 = lane & 15;
  const int nb0 = nb_wg + wave * NCB;
  f32x4 acc[NG][NCB];
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) acc[g][cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
  auto write_out = [&](int u) {
    const int64_t rows_here = (V_out - (int64_t)u * U) < U ? (V_out - (int64_t)u * U) : U;
    const int ng = (int)((rows_here + 15) >> 4);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if (g < ng) {
        const int row = perm[(int64_t)u * U + g * 16 + c16];
        if (row >= 0) {
#pragma unroll
          for (int cb = 0; cb < NCB; ++cb) {
            const int col = (nb0 + cb) * 16 + g4 * 4;
            f32x4 v = acc[g][cb];
            if (bias) { v[0] += bias[col]; v[1] += bias[col + 1]; v[2] += bias[col + 2]; v[3] += bias[col + 3]; }
            bf16x4r o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
            *reinterpret_cast<bf16x4r *>(out + (int64_t)row * co + col) = o;
          }
        }
      }
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) acc[g][cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  };
  RsSeq cur;
  seq_init(cur);
  int cu = u0, hpos = 0;
  u32x4 w[NCB][KC];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int c = 0; c < KC; ++c) w[cb][c] = (u32x4){0u, 0u, 0u, 0u};
  asm volatile("s_barrier" ::: "memory");
  while (!cur.done) {
    while (cu < cur.u) { write_out(cu); ++cu; }
    if (cur.first) {                           // this offset's weight slices: NWS slots, wave w's pieces back to back
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int c = 0; c < KC; ++c) {
          const int off = (wave * NCB * KC + cb * KC + c) * 1024;
          int sp = hpos + off / SLOT;
          sp = sp >= NS ? sp - NS : sp;
          w[cb][c] = *reinterpret_cast<const u32x4 *>(ring + sp * SLOT + (off % SLOT) + lane * 16);
        }
      hpos += NWS;
      hpos = hpos >= NS ? hpos - NS : hpos;
    }
    {
      const unsigned b0 = cur.bits;
      const int g0 = __builtin_ctz(b0);
      const unsigned r0 = b0 & (b0 - 1);
      const int g1 = r0 ? __builtin_ctz(r0) : -1;
      const unsigned char *sa = ring + hpos * SLOT + c16 * RB;
      const unsigned char *sb = sa + 16 * RB;
      u32x4 fa[KC], fb[KC];
#pragma unroll
      for (int c = 0; c < KC; ++c) {           // the lane's 8 channels of chunk c: granule 4 c + g4 of row c16
        const int q = ((c * 4 + g4) ^ (c16 & SWZ)) << 4;
        fa[c] = *reinterpret_cast<const u32x4 *>(sa + q);
        fb[c] = *reinterpret_cast<const u32x4 *>(sb + q);
      }
#define AABR_RS_CASE(G, F)                                                                               \
  case G:                                                                                               \
    if (G < NG) {                                                                                        \
      _Pragma("unroll") for (int cb = 0; cb < NCB; ++cb) _Pragma("unroll") for (int c = 0; c < KC; ++c)   \
          acc[G < NG ? G : 0][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                             \
              __builtin_bit_cast(bf16x8r, w[cb][c]), __builtin_bit_cast(bf16x8r, F[c]),                  \
              acc[G < NG ? G : 0][cb], 0, 0, 0);                                                         \
    }                                                                                                    \
    break;
#define AABR_RS_SWITCH(GV, F)                                                                            \
  switch (GV) {                                                                                          \
    AABR_RS_CASE(0, F) AABR_RS_CASE(1, F) AABR_RS_CASE(2, F) AABR_RS_CASE(3, F) AABR_RS_CASE(4, F)        \
    AABR_RS_CASE(5, F) AABR_RS_CASE(6, F) AABR_RS_CASE(7, F) AABR_RS_CASE(8, F) AABR_RS_CASE(9, F)        \
    AABR_RS_CASE(10, F) AABR_RS_CASE(11, F) AABR_RS_CASE(12, F) AABR_RS_CASE(13, F) AABR_RS_CASE(14, F)   \
    AABR_RS_CASE(15, F)                                                                                  \
  default: break;                                                                                        \
  }
      AABR_RS_SWITCH(g0, fa)
      if (g1 >= 0) { AABR_RS_SWITCH(g1, fb) }
#undef AABR_RS_SWITCH
#undef AABR_RS_CASE
      hpos = hpos + 1 == NS ? 0 : hpos + 1;
    }
    seq_next(cur);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  while (cu < u1) { write_out(cu); ++cu; }
}

} // namespace aabr
using namespace aabr;

extern "C" int64_t aabr_rs_words(int64_t V, int vol, int unit_rows) { return rs_words(V, vol, unit_rows); }

extern "C" int aabr_build_rs(const int32_t *table, int64_t V, int vol, int unit_rows, int32_t *words, void *stream_) {
  AABR_CHECK_ARG(V >= 0 && vol > 0 && vol <= kRsMaxVol, "bad sizes (vol <= 32)");
  AABR_CHECK_ARG(unit_rows >= 16 && unit_rows <= kRsMaxU && (unit_rows & 15) == 0, "unit_rows: multiple of 16, <= 256");
  if (V == 0) return AABR_OK;
  AABR_CHECK_ARG(table && words, "null pointer");
  const unsigned nu = (unsigned)((V + unit_rows - 1) / unit_rows);
  hipLaunchKernelGGL(k_build_rs, dim3(nu), dim3(256), 0, (hipStream_t)stream_, table, V, vol, unit_rows, words);
  AABR_CHECK_LAUNCH();
  return AABR_OK;
}

namespace {
struct RsKnobs { int force; int unit; };
const RsKnobs &rs_knobs() {          // tuning knobs, read once per process
  static const RsKnobs k = [] {
    RsKnobs r{-1, 0};
    if (const char *v = getenv("AABR_CONV_RS")) r.force = v[0] == '0' ? 0 : (v[0] == '1' ? 1 : -1);
    if (const char *v = getenv("AABR_RS_UNIT")) r.unit = atoi(v);
    return r;
  }();
  return k;
}
} // namespace

// unit size for V_out rows: the number of units is a multiple of the workgroups the chip runs at once (one per
// CU and slab), so every workgroup gets the same number of units; a unit holds at most 256 rows (16 groups)
static int rs_unit(int64_t V_out, int slabs, int umax) {
  const int64_t wgs = 256 / slabs > 0 ? 256 / slabs : 1;
  int64_t n = (V_out + umax - 1) / umax;
  n = (n + wgs - 1) / wgs * wgs;
  int64_t u = (V_out + n - 1) / n;
  u = (u + 15) & ~15ll;
  return (int)(u < 32 ? 32 : u);
}

// 0: not for this launch; otherwise the unit size (rows) of the stream aabr_conv_forward_rs_bf16 wants
extern "C" int aabr_conv_rs_unit_rows(int n_in, int n_out, int64_t rows_in, int64_t V_out, int vol) {
  if ((n_in != 64 && n_in != 128) || n_out <= 0 || (n_out & 63) || vol <= 0 || vol >= kRsHdr) return 0;
  if (rows_in <= 0 || V_out <= 0 || rows_in * n_in * 2 >= (1ll << 40)) return 0;
  const RsKnobs &kn = rs_knobs();
  if (kn.force == 0) return 0;
  const int ncb = (n_out & 127) == 0 ? 2 : 1;
  const int umax = ncb == 2 ? 192 : 256;   // accumulators of 12 / 16 groups fit the register file
  int U = rs_unit(V_out, n_out / (64 * ncb), umax);
  if (kn.unit >= 16 && kn.unit <= umax && (kn.unit & 15) == 0) U = kn.unit;
  if (rs_words(V_out, vol, U) * 4 >= (1ll << 40)) return 0;
  if (kn.force == 1) return U;
  // enough work to fill the chip; the small coarse scales stay on the small-launch kernels
  return V_out >= 16384 ? U : 0;
}

extern "C" int aabr_conv_forward_rs_bf16(const uint16_t *in_feats, int n_in, int64_t rows_in, uint16_t *out_feats,
                                         int n_out, int64_t V_out, const int32_t *rs_stream, int unit_rows, int vol,
                                         const float *bias, int flags, const uint16_t *wpack, void *stream_) {
  hipStream_t st = (hipStream_t)stream_;
  AABR_CHECK_ARG((n_in == 64 || n_in == 128) && n_out > 0 && (n_out & 63) == 0, "plane counts: n_in 64|128, n_out % 64");
  AABR_CHECK_ARG(vol > 0 && vol < kRsHdr && V_out >= 0 && rows_in >= 0, "bad sizes (vol <= 31)");
  AABR_CHECK_ARG(unit_rows >= 16 && unit_rows <= ((n_out & 127) == 0 ? 192 : 256) && (unit_rows & 15) == 0,
                 "unit_rows: multiple of 16, <= 192 (n_out % 128 == 0) / 256");
  if (V_out == 0) return AABR_OK;
  AABR_CHECK_ARG(in_feats && out_feats && rs_stream && wpack && rows_in > 0, "null pointer / empty input");
  AABR_CHECK_ARG(((uintptr_t)in_feats & 15) == 0 && ((uintptr_t)out_feats & 7) == 0 && ((uintptr_t)wpack & 15) == 0 &&
                     ((uintptr_t)rs_stream & 15) == 0,
                 "feature / weight / stream pointers must be 16-byte aligned");
  const int ncb = (n_out & 127) == 0 ? 2 : 1;
  const int slabs = n_out / (64 * ncb);
  const int64_t nunits = (V_out + unit_rows - 1) / unit_rows;
  const int64_t wgs_max = 256 / slabs > 0 ? 256 / slabs : 1;
  const int upw = (int)((nunits + wgs_max - 1) / wgs_max);
  AABR_CHECK_ARG(nunits < (1ll << 30), "too many units");
  dim3 grid((unsigned)((nunits + upw - 1) / upw), (unsigned)slabs);
  const int flip = (flags >> 1) & 1;
  const __bf16 *in_b = reinterpret_cast<const __bf16 *>(in_feats), *wp_b = reinterpret_cast<const __bf16 *>(wpack);
  __bf16 *out_b = reinterpret_cast<__bf16 *>(out_feats);
  static int ns_knob = -1;
  if (ns_knob < 0) { const char *v = getenv("AABR_RS_SLOTS"); ns_knob = v ? atoi(v) : 0; }
#define AABR_RS_LAUNCH(KC, NCB, NG)                                                                              \
  do {                                                                                                         \
    constexpr int SLOT = 32 * KC * 64;                                                                         \
    int NS = (144 * 1024) / SLOT;                                                                              \
    if (NS > 36) NS = 36;                        /* vmcnt counts at most 63 outstanding DMAs per loader wave */ \
    if (ns_knob >= 4 * NCB + 2 && ns_knob <= NS) NS = ns_knob;  /* D = NS - NWS - 1 >= the slots of one step */                                                 \
    static bool attr = false;                                                                                  \
    if (!attr) {                                                                                               \
      AABR_CHECK_HIP(hipFuncSetAttribute((const void *)(k_conv_rsq<KC, NCB, NG>),                                  \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));             \
      attr = true;                                                                                             \
    }                                                                                                          \
    g_last_variant = "k_conv_rsq<" #KC "," #NCB "," #NG ",bf16>";                                                  \
    hipLaunchKernelGGL((k_conv_rsq<KC, NCB, NG>), grid, dim3(384), (size_t)NS * SLOT, st, in_b, out_b, n_out, V_out, \
                       rs_stream, vol, unit_rows, upw, flip, wp_b, bias, NS);                                  \
  } while (0)
  if (n_in == 128) { if (ncb == 2) AABR_RS_LAUNCH(4, 2, 12); else AABR_RS_LAUNCH(4, 1, 16); }
  else { if (ncb == 2) AABR_RS_LAUNCH(2, 2, 12); else AABR_RS_LAUNCH(2, 1, 16); }
#undef AABR_RS_LAUNCH
  AABR_CHECK_LAUNCH();
  return AABR_OK;
}
