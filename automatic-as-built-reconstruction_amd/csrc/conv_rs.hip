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
constexpr int kRsMaxUpw = 32;   // units per workgroup (their headers sit in LDS)

// ------------------------------------------------------------------ compiled rule book, row-stationary form
//   words: [nunits][32] header: [0] = number of active offsets n, [1..n] = (offset << 16) | 16-bit set of the
//          QUADS (4 consecutive groups = 64 slots) with at least one partner at that offset (offsets ascending),
//          [31] = steps of the unit
//        | [nunits][U] perm (output row of slot, -1 = padding)
//        | [nunits][vol * 4][4][16] step descriptors in the order they are consumed: a step = the 4 groups of one
//          active quad at one offset, an item = the 16 partner rows of a group (-1 = none)
constexpr int kRsHdr = 32;
constexpr int kRsSI = 4;     // items per step
static inline int64_t rs_words(int64_t V, int vol, int U) {
  const int64_t nu = (V + U - 1) / U;
  return nu * (kRsHdr + U + (int64_t)vol * 4 * kRsSI * 16);
}

#ifdef AABR_DEV   // the row-stationary kernels are an A/B experiment (measured slower): `make DEV=1` builds only
__global__ __launch_bounds__(256) void k_build_rs(const int32_t *__restrict__ table, int64_t V, int vol, int U,
                                                  int32_t *__restrict__ words) {
  __shared__ unsigned long long s_key[kRsMaxU];
  __shared__ unsigned int s_mask[kRsMaxU];
  __shared__ int s_row[kRsMaxU];
  __shared__ unsigned int s_gm[16];
  __shared__ int s_hdr[kRsHdr], s_sb[kRsHdr];
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
  for (int i = t; i < kRsMaxU; i += 256) { s_mask[i] = 0u; s_row[i] = -1; }
  __syncthreads();
  if (in_unit) { s_mask[rank] = valid ? mask : 0u; s_row[rank] = valid ? (int)t : -1; }
  int32_t *hdr = words + unit * kRsHdr;
  int32_t *perm = words + nunits * kRsHdr + unit * U;
  int32_t *desc = words + nunits * (kRsHdr + U) + unit * (int64_t)vol * 4 * kRsSI * 16;
  if (in_unit) perm[rank] = valid ? (int32_t)row : -1;
  __syncthreads();
  if (t < 16) {
    unsigned int m = 0;
    if (t * 16 < U)
      for (int i = 0; i < 16; ++i) m |= s_mask[t * 16 + i];
    s_gm[t] = m;
  }
  __syncthreads();
  if (t == 0) {
    int n = 0, steps = 0;
    for (int k = 0; k < vol; ++k) {
      unsigned bits = 0;      // quads (4 consecutive groups = 64 slots) with at least one partner at this offset
      for (int g = 0; g < 16; ++g) bits |= ((s_gm[g] >> k) & 1u) << (g >> 2);
      if (bits) {
        s_hdr[++n] = (int)(((unsigned)k << 16) | bits);
        s_sb[n] = steps;
        steps += __popc(bits);
      }
    }
    s_hdr[0] = n;
    for (int i = n + 1; i < kRsHdr; ++i) s_hdr[i] = 0;
    s_hdr[kRsHdr - 1] = steps;
  }
  __syncthreads();
  if (t < kRsHdr) hdr[t] = s_hdr[t];
  const int n = s_hdr[0];
  for (int j = 1; j <= n; ++j) {
    const unsigned wd = (unsigned)s_hdr[j], bits = wd & 0xfu;
    const int k = (int)(wd >> 16), nst = __popc(bits);
    if (t < nst * 64) {
      unsigned b = bits;                       // the (t / 64)-th active quad of this offset
      for (int i = 0; i < (t >> 6); ++i) b &= b - 1;
      const int q = __ffs(b) - 1, lr = s_row[q * 64 + (t & 63)];
      desc[((int64_t)s_sb[j] * kRsSI) * 16 + t] = lr >= 0 ? table[(int64_t)k * V + unit * U + lr] : -1;
    }
  }
}

// ------------------------------------------------------------------ the kernel
// One workgroup (512 threads, one per CU) = 4 consumer waves + 4 loader waves.
//   consumer wave w: the MFMAs of columns [16 NCB w, 16 NCB (w + 1)) of the slab for every item; per offset it loads
//     its weight slice straight into registers (next offset's slice in flight meanwhile); accumulators in registers.
//   loader wave p: gathers item p of every step (16 rows) into registers, kRsD steps ahead, and stores it -- granule-
//     swizzled -- into the step's LDS buffer one step before the consumers read it.  All its memory operations are plain
//     loads in consumption order (descriptor of step s + kRsD, then the rows of that step), so the compiler's own vmcnt
//     bookkeeping keeps kRsD steps (4 KiB each per loader, 7 x 16 KiB per CU) in flight with static register sets.
// One s_barrier per step.  KC = 32-channel chunks per input row (n_in = 32 KC), NCB = 16-column blocks per consumer.
constexpr int kRsD = 7;      // steps a loader runs ahead (register sets)

template <int KC, int NCB, int NG>
__global__ __launch_bounds__(512, 2) void k_conv_rsq(
    const __bf16 *__restrict__ in, int64_t in_bytes, __bf16 *__restrict__ out, int co, int64_t V_out,
    const int32_t *__restrict__ words, int64_t words_bytes, int vol, int U, int upw, int wflip,
    const __bf16 *__restrict__ Wp, int64_t wp_bytes, const float *__restrict__ bias, int dbg) {
  constexpr int RB = KC * 64;                 // bytes per input row
  constexpr int ITEM = 16 * RB;               // bytes of an item's rows
  constexpr int STEP = kRsSI * ITEM;          // LDS buffer of a step
  constexpr int LPR = RB / 16;                // lanes per row of a gather instruction (16 B per lane)
  constexpr int RPP = 64 / LPR;               // rows per gather instruction
  constexpr int IPW = 16 / RPP;               // gather instructions per item
  constexpr int SWZ = LPR - 1;
  static_assert(KC == 2 || KC == 4, "n_in = 64 or 128");
  __shared__ __align__(16) unsigned char sbuf[2][STEP];
  __shared__ int32_t s_hdr[kRsMaxUpw * kRsHdr];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int nnb = co >> 4;
  const int64_t nunits = (V_out + U - 1) / U;
  const int u0 = blockIdx.x * upw;
  const int u1 = (int)((int64_t)(u0 + upw) < nunits ? (u0 + upw) : nunits);
  for (int i = threadIdx.x; i < (u1 - u0) * kRsHdr; i += 512) s_hdr[i] = words[(int64_t)u0 * kRsHdr + i];
  __syncthreads();
  int nsteps = 0;
  for (int u = u0; u < u1; ++u) nsteps += s_hdr[(u - u0) * kRsHdr + kRsHdr - 1];
  nsteps = __builtin_amdgcn_readfirstlane(nsteps);

  if (wave >= 4) {
    // ------------------------------------------------------------------ loader waves
    const int pw = wave - 4;
    const __amdgpu_buffer_rsrc_t rin =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16 *>(in), 0, (int)(in_bytes > 0x7fffffffll ? 0x7fffffffll : in_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t rwords =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t *>(words), 0, (int)words_bytes, 0x00020000);
    const int rl = lane / LPR, pl = lane % LPR;
    const unsigned dbase = (unsigned)((nunits * (kRsHdr + U)) * 4);       // byte offset of the descriptors
    const unsigned ustride = (unsigned)(vol * 4 * kRsSI * 16 * 4);        // ... per unit
    // position of the step whose descriptor is fetched next (runs 2 kRsD steps ahead of the consumers)
    int du = u0, ds = 0, dn = u0 < u1 ? s_hdr[kRsHdr - 1] : 0;
    struct Ent { int e[IPW]; };
    struct Rows { u32x4 v[IPW]; };
    auto fetch_desc = [&](Ent &d) {            // the entries of item pw of the next step; past the end: none
      while (du < u1 && ds >= dn) { ++du; ds = 0; dn = du < u1 ? s_hdr[(du - u0) * kRsHdr + kRsHdr - 1] : 0; }
      // past the end: an offset behind the buffer -> zeros, i.e. row 0 is gathered into a buffer nobody reads (the
      // loaded value must not be touched here: a select on it would wait for the load, ~1 us per step measured)
      const unsigned so = du < u1 ? dbase + (unsigned)du * ustride + (unsigned)((ds * kRsSI + pw) * 64) : 0x7ffffff0u;
#pragma unroll
      for (int i = 0; i < IPW; ++i)
        d.e[i] = (int)__builtin_amdgcn_raw_buffer_load_b32(rwords, (unsigned)((i * RPP + rl) * 4), so, 0);
      ++ds;
    };
    auto gather = [&](Rows &r, const Ent &d) {
#pragma unroll
      for (int i = 0; i < IPW; ++i) {
        // absent partner: an offset behind the buffer -> the range check returns zeros, no memory traffic
        const unsigned va = d.e[i] >= 0 ? (unsigned)d.e[i] * (unsigned)RB + (unsigned)pl * 16u : 0x7ffffff0u;
        if (dbg & 1) r.v[i] = (u32x4){(unsigned)d.e[i], 0u, 0u, 0u};
        else r.v[i] = __builtin_amdgcn_raw_buffer_load_b128(rin, va, 0, 0);
      }
    };
    auto store = [&](const Rows &r, int buf) { // row i RPP + rl of the item, granule pl -> position pl ^ (row & SWZ)
      unsigned char *dst = &sbuf[buf][pw * ITEM];
#pragma unroll
      for (int i = 0; i < IPW; ++i) {
        const int row = i * RPP + rl;
        *reinterpret_cast<u32x4 *>(dst + row * RB + ((pl ^ (row & SWZ)) << 4)) = r.v[i];
      }
    };
    // software pipeline with static register sets: descriptors 2 kRsD steps ahead, rows kRsD steps ahead
    Ent de[kRsD];
    Rows rw[kRsD];
#pragma unroll
    for (int i = 0; i < kRsD; ++i) fetch_desc(de[i]);
#pragma unroll
    for (int i = 0; i < kRsD; ++i) { gather(rw[i], de[i]); fetch_desc(de[i]); }
    store(rw[0], 0);                           // step 0 is in LDS before the first barrier
    gather(rw[0], de[0]);
    fetch_desc(de[0]);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // iteration s (the consumers read buffer s & 1): store step s + 1, then fetch step s + 1 + kRsD
    int s = 0;
    long long stamp[5] = {0, 0, 0, 0, 0};
    while (s < nsteps) {
#pragma unroll
      for (int i = 1; i <= kRsD; ++i) {
        const int j = i == kRsD ? 0 : i;       // register set of step s + 1 (sets rotate: step t lives in set t % kRsD)
        if (s < nsteps) {
          long long t0 = 0, t1 = 0, t2 = 0, t3 = 0;
          if (dbg & 64) t0 = __builtin_amdgcn_s_memtime();
          if (!(dbg & 8)) {
            store(rw[j], (s + 1) & 1);
            if (dbg & 64) { __builtin_amdgcn_sched_barrier(0); t1 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
            gather(rw[j], de[j]);
            fetch_desc(de[j]);
          }
          if (dbg & 64) { __builtin_amdgcn_sched_barrier(0); t2 = __builtin_amdgcn_s_memtime(); }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          if (dbg & 64) t3 = __builtin_amdgcn_s_memtime();
          asm volatile("s_barrier" ::: "memory");
          if (dbg & 64) {
            const long long t4 = __builtin_amdgcn_s_memtime();
            stamp[0] += t1 - t0; stamp[1] += t2 - t1; stamp[2] += t3 - t2; stamp[3] += t4 - t3; stamp[4] += 1;
          }
          ++s;
        }
      }
    }
    if ((dbg & 64) && lane == 0) {   // timing experiments: per-wave phase clocks -> the buffer passed as `bias`
      long long *d = reinterpret_cast<long long *>(const_cast<float *>(bias)) + ((int64_t)blockIdx.x * 8 + wave) * 8;
      for (int q = 0; q < 5; ++q) d[q] = stamp[q];
    }
    return;
  }

  // -------------------------------------------------------------------- consumer waves
  const int g4 = lane >> 4, c16 = lane & 15;
  const float *bias_dbg = bias;
  if (dbg & 64) bias = nullptr;
  const int nb0 = blockIdx.y * (4 * NCB) + wave * NCB;
  const int32_t *perm = words + nunits * kRsHdr;
  const __amdgpu_buffer_rsrc_t rwp =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16 *>(Wp), 0, (int)wp_bytes, 0x00020000);
  f32x4 acc[NG][NCB];
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) acc[g][cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
  auto write_out = [&](int u) {
    const int64_t rows_here = (V_out - (int64_t)u * U) < U ? (V_out - (int64_t)u * U) : U;
    const int ng = (int)((rows_here + 15) >> 4);
    // all row numbers first (independent loads), then the stores: a load -> branch -> store chain per group costs a
    // memory latency each (16 x ~1.5 us per unit, measured as half of the kernel's time)
    int rowv[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) rowv[g] = perm[(int64_t)u * U + (g < ng ? g * 16 + c16 : c16)];
    float bv[NCB][4];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int i = 0; i < 4; ++i) bv[cb][i] = bias ? bias[(nb0 + cb) * 16 + g4 * 4 + i] : 0.f;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if (g < ng && rowv[g] >= 0) {
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
          const int col = (nb0 + cb) * 16 + g4 * 4;
          const f32x4 v = acc[g][cb];
          bf16x4r o = {(__bf16)(v[0] + bv[cb][0]), (__bf16)(v[1] + bv[cb][1]), (__bf16)(v[2] + bv[cb][2]),
                       (__bf16)(v[3] + bv[cb][3])};
          *reinterpret_cast<bf16x4r *>(out + (int64_t)rowv[g] * co + col) = o;
        }
      }
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) acc[g][cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  };
  struct Off { int u, j, k; unsigned bits; };  // u >= u1: past the end
  auto off_next = [&](Off q) {                 // the next active offset (possibly of a later unit)
    ++q.j;
    while (q.u < u1 && q.j > s_hdr[(q.u - u0) * kRsHdr]) { ++q.u; q.j = 1; }
    if (q.u < u1) {
      const unsigned wd = (unsigned)s_hdr[(q.u - u0) * kRsHdr + q.j];
      q.k = (int)(wd >> 16);
      q.bits = wd & 0xfu;
    } else { q.k = 0; q.bits = 0; }
    return q;
  };
  struct WReg { u32x4 w[NCB][KC]; };
  auto load_w = [&](WReg &w, int k) {
    const int kW = wflip ? vol - 1 - k : k;
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int c = 0; c < KC; ++c) {
        const unsigned so = (unsigned)((((int64_t)kW * KC + c) * nnb + nb0 + cb) * 1024);
        w.w[cb][c] = __builtin_amdgcn_raw_buffer_load_b128(rwp, (unsigned)lane * 16u, so, 0);
      }
  };
  int cu = u0, step = 0;
  long long cstamp[6] = {0, 0, 0, 0, 0, 0};
  // The groups of an offset are visited in a STATIC order (unrolled loop, a wave-uniform branch per group) so that
  // every accumulator is addressed by a constant and never leaves its registers -- a `switch` on the group number made
  // the compiler merge all accumulators with moves at every flow node (313 v_mov per step, 2 us per step measured).
  // What is dynamic is only where the item's rows sit in the step buffer.
  auto run_offset = [&](const WReg &w, unsigned b) __attribute__((always_inline)) {
    u32x4 f[2][KC];
    auto read_item = [&](int set, int item) {  // the lane's 8 channels of chunk c: granule 4 c + g4 of row c16
      const unsigned char *sa = &sbuf[step & 1][item * ITEM + c16 * RB];
#pragma unroll
      for (int c = 0; c < KC; ++c)
        f[set][c] = *reinterpret_cast<const u32x4 *>(sa + (((c * 4 + g4) ^ (c16 & SWZ)) << 4));
    };
#pragma unroll
    for (int q = 0; q < NG / 4; ++q) {
      if (b & (1u << q)) {                     // a step: the four groups of quad q, accumulators addressed statically
        long long t0 = 0, t1 = 0, t2 = 0;
        if (dbg & 64) t0 = __builtin_amdgcn_s_memtime();
        read_item(0, 0);
        read_item(1, 1);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (!(dbg & 2))
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
              for (int c = 0; c < KC; ++c)
                acc[4 * q + i][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    __builtin_bit_cast(bf16x8r, w.w[cb][c]), __builtin_bit_cast(bf16x8r, f[i & 1][c]), acc[4 * q + i][cb],
                    0, 0, 0);
          if (i + 2 < 4) read_item(i & 1, i + 2);   // in flight during the next item's MFMAs
        }
        ++step;
        if (dbg & 64) { __builtin_amdgcn_sched_barrier(0); t1 = __builtin_amdgcn_s_memtime(); }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (dbg & 64) t2 = __builtin_amdgcn_s_memtime();
        asm volatile("s_barrier" ::: "memory");
        if (dbg & 64) {
          const long long t3 = __builtin_amdgcn_s_memtime();
          cstamp[0] += t1 - t0; cstamp[1] += t2 - t1; cstamp[2] += t3 - t2; cstamp[3] += 1;
          if (cstamp[5]) cstamp[4] += t0 - cstamp[5];   // from the previous step's barrier to this step's start
          cstamp[5] = t3;
        }
      }
    }
  };
  Off cur;
  cur.u = u0; cur.j = 0; cur.k = 0; cur.bits = 0;
  cur = off_next(cur);
  WReg wA, wB;
  if (cur.u < u1) load_w(wA, cur.k);
  asm volatile("s_barrier" ::: "memory");
  while (cur.u < u1) {
    Off nxt = off_next(cur);
    if (nxt.u < u1 && !(dbg & 32)) load_w(wB, nxt.k);   // next offset's weights in flight during this offset
    while (cu < cur.u) { write_out(cu); ++cu; }
    run_offset(wA, cur.bits);
    cur = nxt;
    if (cur.u >= u1) break;
    nxt = off_next(cur);
    if (nxt.u < u1 && !(dbg & 32)) load_w(wA, nxt.k);
    while (cu < cur.u) { write_out(cu); ++cu; }
    run_offset(wB, cur.bits);
    cur = nxt;
  }
  if (dbg & 64) bias = nullptr;
  while (cu < u1) { write_out(cu); ++cu; }
  if ((dbg & 64) && lane == 0) {
    long long *d = reinterpret_cast<long long *>(const_cast<float *>(bias_dbg)) + ((int64_t)blockIdx.x * 8 + wave) * 8;
    for (int q = 0; q < 5; ++q) d[q] = cstamp[q];
  }
}

#endif  // AABR_DEV
} // namespace aabr
using namespace aabr;

extern "C" int64_t aabr_rs_words(int64_t V, int vol, int unit_rows) { return rs_words(V, vol, unit_rows); }

extern "C" int aabr_build_rs(const int32_t *table, int64_t V, int vol, int unit_rows, int32_t *words, void *stream_) {
#ifndef AABR_DEV
  (void)table; (void)V; (void)vol; (void)unit_rows; (void)words; (void)stream_;
  aabr::set_error("aabr_build_rs: the row-stationary kernels exist in a `make DEV=1` build only (aabr_build_flags)");
  return AABR_EINVAL;
#else
  AABR_CHECK_ARG(V >= 0 && vol > 0 && vol <= kRsMaxVol, "bad sizes (vol <= 32)");
  AABR_CHECK_ARG(unit_rows >= 16 && unit_rows <= kRsMaxU && (unit_rows & 15) == 0, "unit_rows: multiple of 16, <= 256");
  if (V == 0) return AABR_OK;
  AABR_CHECK_ARG(table && words, "null pointer");
  const unsigned nu = (unsigned)((V + unit_rows - 1) / unit_rows);
  hipLaunchKernelGGL(k_build_rs, dim3(nu), dim3(256), 0, (hipStream_t)stream_, table, V, vol, unit_rows, words);
  AABR_CHECK_LAUNCH();
  return AABR_OK;
#endif
}

// unit size for V_out rows: the number of units is a multiple of the workgroups the chip runs at once (one per
// CU and slab), so every workgroup gets the same number of units; a unit holds at most 256 rows (16 groups)
#ifdef AABR_DEV
static int rs_unit(int64_t V_out, int slabs, int umax) {
  const int64_t wgs = 256 / slabs > 0 ? 256 / slabs : 1;
  int64_t n = (V_out + umax - 1) / umax;
  n = (n + wgs - 1) / wgs * wgs;
  int64_t u = (V_out + n - 1) / n;
  u = (u + 15) & ~15ll;
  return (int)(u < 32 ? 32 : u);
}
#endif

// 0: not for this launch; otherwise the unit size (rows) of the stream aabr_conv_forward_rs_bf16 wants
extern "C" int aabr_conv_rs_unit_rows(int n_in, int n_out, int64_t rows_in, int64_t V_out, int vol) {
#ifndef AABR_DEV
  (void)n_in; (void)n_out; (void)rows_in; (void)V_out; (void)vol;
  return 0;                                  // never dispatched: the kernels are not in this build
#else
  if ((n_in != 64 && n_in != 128) || n_out <= 0 || (n_out & 63) || vol <= 0 || vol >= kRsHdr) return 0;
  if (rows_in <= 0 || V_out <= 0 || rows_in * n_in * 2 >= (1ll << 31) - 4096) return 0;
  if ((int64_t)vol * n_in * n_out * 2 >= (1ll << 31)) return 0;
  const int force = knob(K_CONV_RS), unit = knob(K_RS_UNIT);
  if (force != 1) return 0;
  const int ncb = (n_out & 127) == 0 ? 2 : 1;
  const int umax = kRsMaxU;                 // accumulators of 16 groups x 32 columns: 128 registers
  int U = rs_unit(V_out, n_out / (64 * ncb), umax);
  if (unit >= 16 && unit <= umax && (unit & 15) == 0) U = unit;
  if (rs_words(V_out, vol, U) * 4 >= (1ll << 31)) return 0;
  {
    const int slabs = n_out / (64 * ncb);
    const int64_t nunits = (V_out + U - 1) / U, wgs = 256 / slabs > 0 ? 256 / slabs : 1;
    if ((nunits + wgs - 1) / wgs > kRsMaxUpw) return 0;
  }
  // Measured (profiles/r03_conv_rs_ab.txt): correct, but at 138 us on the dominant 128->128 instance against 107 us
  // for the LDS-tile kernel -- one wave per SIMD and role with a barrier per step serialises gather latency, LDS
  // reads, weight loads and MFMAs instead of overlapping them.  Kept behind AABR_CONV_RS=1 for the A/B and the tests.
  return U;
#endif
}

extern "C" int aabr_conv_forward_rs_bf16(const uint16_t *in_feats, int n_in, int64_t rows_in, uint16_t *out_feats,
                                         int n_out, int64_t V_out, const int32_t *rs_stream, int unit_rows, int vol,
                                         const float *bias, int flags, const uint16_t *wpack, void *stream_) {
#ifndef AABR_DEV
  (void)in_feats; (void)n_in; (void)rows_in; (void)out_feats; (void)n_out; (void)V_out; (void)rs_stream; (void)unit_rows;
  (void)vol; (void)bias; (void)flags; (void)wpack; (void)stream_;
  aabr::set_error("aabr_conv_forward_rs_bf16: the row-stationary kernels exist in a `make DEV=1` build only");
  return AABR_EINVAL;
#else
  hipStream_t st = (hipStream_t)stream_;
  AABR_CHECK_ARG((n_in == 64 || n_in == 128) && n_out > 0 && (n_out & 63) == 0, "plane counts: n_in 64|128, n_out % 64");
  AABR_CHECK_ARG(vol > 0 && vol < kRsHdr && V_out >= 0 && rows_in >= 0, "bad sizes (vol <= 31)");
  AABR_CHECK_ARG(unit_rows >= 16 && unit_rows <= kRsMaxU && (unit_rows & 15) == 0, "unit_rows: multiple of 16, <= 256");
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
  AABR_CHECK_ARG(nunits < (1ll << 30) && upw <= kRsMaxUpw, "too many units per workgroup");
  const int64_t in_bytes = rows_in * n_in * 2, words_bytes = rs_words(V_out, vol, unit_rows) * 4;
  const int64_t wp_bytes = (int64_t)vol * (n_in / 32) * (n_out / 16) * 1024;
  AABR_CHECK_ARG(in_bytes < (1ll << 31) - 4096 && words_bytes < (1ll << 31) && wp_bytes < (1ll << 31),
                 "buffers must be < 2 GiB");
  dim3 grid((unsigned)((nunits + upw - 1) / upw), (unsigned)slabs);
  const int flip = (flags >> 1) & 1;
  const __bf16 *in_b = reinterpret_cast<const __bf16 *>(in_feats), *wp_b = reinterpret_cast<const __bf16 *>(wpack);
  __bf16 *out_b = reinterpret_cast<__bf16 *>(out_feats);
#ifdef AABR_DEV
#define AABR_RS_DBG(f) ((f) >> 8)   /* timing experiments (tools/tools_rs_probe.py, `make DEV=1`) */
#else
#define AABR_RS_DBG(f) 0
#endif
#define AABR_RS_LAUNCH(KC, NCB, NG)                                                                            \
  do {                                                                                                         \
    g_last_variant = "k_conv_rsq<" #KC "," #NCB "," #NG ",bf16>";                                              \
    hipLaunchKernelGGL((k_conv_rsq<KC, NCB, NG>), grid, dim3(512), 0, st, in_b, in_bytes, out_b, n_out, V_out, \
                       rs_stream, words_bytes, vol, unit_rows, upw, flip, wp_b, wp_bytes, bias, AABR_RS_DBG(flags));   \
  } while (0)
  if (n_in == 128) { if (ncb == 2) AABR_RS_LAUNCH(4, 2, 16); else AABR_RS_LAUNCH(4, 1, 16); }
  else { if (ncb == 2) AABR_RS_LAUNCH(2, 2, 16); else AABR_RS_LAUNCH(2, 1, 16); }
#undef AABR_RS_LAUNCH
  AABR_CHECK_LAUNCH();
  return AABR_OK;
#endif
}
