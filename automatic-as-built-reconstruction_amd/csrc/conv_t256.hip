// conv_t256.hip -- the wide-layer form of the sparse convolution (gfx950): 256-row output tiles,
// filter weights staged through LDS once per (workgroup, offset, 32-channel chunk).
//
// Why: in the 64-row-tile kernels of conv.hip every 16-pair block streams its own 2 KiB x NBW of packed
// weights per 32-channel chunk through the vector-memory path; on the layers that dominate the FPN_Net step
// (64..256 planes, 3x3x3) that is ~5x the bytes of the gathered rows and the L2->CU feed, not the MFMA pipe,
// bounds the kernel (34 % of the fp32 MFMA peak at 128->128; profiles/r02_*).  Here
//   * a workgroup (4 waves) owns 256 consecutive output rows x 64 output columns, kept in LDS (64 KiB, XOR-
//     swizzled 16-byte granules: the read-add-write of a block's 16 rows is bank-conflict free) for the whole
//     sweep and written once;
//   * the sweep is over STAGES (offset k, chunk kc): the stage's 8 KiB weight slab is copied to LDS once
//     (double-buffered: global -> registers during the previous stage's MFMAs -> LDS) and every wave reads its
//     MFMA A-operands from there with conflict-free ds_read_b128 -- weight traffic per block drops by the
//     number of blocks that share the stage (all of the tile's blocks of that offset);
//   * 256-row tiles fill their 16-pair blocks to ~90 % (64-row tiles: 64-69 %), i.e. fewer padded MFMAs;
//   * within one offset every output row occurs at most once (the property that makes the reference's
//     rule_index_add_ race-free, SCN/CPU/Convolution.cpp:28-43), so the waves of a workgroup update disjoint
//     rows of the shared tile inside a stage; stages are separated by a barrier => fixed accumulation order,
//     bit-reproducible, no atomics.
// Two workgroups per CU (80 KiB LDS each) overlap one's barrier / imbalance with the other's MFMAs.
//
// Same contraction as conv.hip (reference: SCN/CPU/Convolution.cpp:45-185, SCN/CPU/Deconvolution.cpp:7-77):
//     out[o] = bias + sum_k in[table[k][o]] @ Wl[k]
// Requires n_in % 32 == 0, n_out % 64 == 0, vol <= 63, rows_in < 2^23, buffers < 2 GiB.
#include "common.h"
#include <stdlib.h>

namespace aabr {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

extern thread_local const char *g_last_variant; // conv.hip

constexpr int kT = 256;    // output rows per tile
constexpr int kWS = 64;    // tile row stride in floats = slab width
constexpr int kNB = 4;     // 16-column blocks per slab
constexpr int kNW = 4;     // waves per workgroup
constexpr int kMaxVol = 63;  // vol + 1 prefix entries live in the lanes of one VGPR

__host__ __device__ inline int64_t t256_ntiles(int64_t V) { return (V + kT - 1) / kT; }
__host__ __device__ inline int t256_maxb(int vol) { return 16 * vol; }

// ------------------------------------------------------------------ compiled rule book, 256-row form
//   words: [ntiles][vol+1] block prefix per offset | [ntiles][16*vol][16] entries
//   entry = (partner_row << 8) | local_row; padding entries repeat the block's first pair with bit 31 set.
// Pairs of one offset are in ascending local-row order (deterministic).
template <int T> // rows per tile = threads per workgroup (128 or 256)
__global__ __launch_bounds__(T) void k_build_tileT(const int32_t *__restrict__ table, int64_t V, int vol,
                                                    int32_t *__restrict__ words) {
  constexpr int kT = T;
  constexpr int NWV = T / 64;
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
  const bool valid = row < V;
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

// one work item of a wave: block b of offset k, channel chunk kc
struct Item {
  int k, kc, b; // k == vol: past the end
};

template <int DBG>
__global__ __launch_bounds__(256, 2) void k_conv_t256(const float *__restrict__ in, int ci, int64_t in_bytes,
                                                      float *__restrict__ out, int co, int64_t V_out,
                                                      const int32_t *__restrict__ words, int64_t words_bytes,
                                                      int vol, int wflip, const float *__restrict__ Wp,
                                                      const float *__restrict__ bias) {
  extern __shared__ __align__(16) float smem[];
  float *Ct = smem;                 // [256][64] floats, granule-swizzled
  float *Wb = smem + kT * kWS;      // [2][kNB*512] floats
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int g = lane >> 4, c16 = lane & 15;
  const int nkc = ci >> 5, nnb = co >> 4;
  const int nb0 = blockIdx.y * kNB;
  const int64_t tile = blockIdx.x, row0 = tile * kT;
  const int64_t ntiles = t256_ntiles(V_out);
  const int maxb = t256_maxb(vol);
  // the tile's per-offset block prefix: lane l of every wave holds pre[l] (vol + 1 <= 64 entries), read back
  // with v_readlane -- no LDS beyond the 80 KiB of tile + weight buffers, so two workgroups fit a CU
  const int vpre = lane <= vol ? words[tile * (vol + 1) + lane] : 0;
  {
    f32x4 z = {0.f, 0.f, 0.f, 0.f};
    f32x4 *c4 = reinterpret_cast<f32x4 *>(Ct);
#pragma unroll
    for (int i = 0; i < (kT * kWS / 4) / 256; ++i) c4[i * 256 + threadIdx.x] = z;
  }
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rin =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(in), 0, (int)in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rwords =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t *>(words), 0, (int)words_bytes, 0x00020000);
  const unsigned ebase = (unsigned)((ntiles * (vol + 1) + tile * (int64_t)maxb * 16) * 4);
  const unsigned rowbytes = (unsigned)ci * 4u, g32 = (unsigned)g * 32u, c16x4 = (unsigned)c16 * 4u;
  const int64_t slab_floats = (int64_t)kNB * 512;

  auto pre_of = [&](int k) { return __builtin_amdgcn_readlane(vpre, k); };
  // ---- workgroup-level stage iteration (uniform): offsets that have blocks x channel chunks
  auto next_offset = [&](int k) {
    ++k;
    while (k < vol && pre_of(k + 1) == pre_of(k)) ++k;
    return k;
  };
  // ---- this wave's item iteration
  auto first_item = [&]() {
    Item it;
    it.k = next_offset(-1);
    it.kc = 0;
    while (it.k < vol) {
      it.b = pre_of(it.k) + wave;
      if (it.b < pre_of(it.k + 1)) break;
      it.k = next_offset(it.k);
    }
    return it;
  };
  auto advance = [&](Item it) {
    if (it.k >= vol) return it;
    it.b += kNW;
    if (it.b < pre_of(it.k + 1)) return it;
    for (;;) {
      if (++it.kc == nkc) { it.kc = 0; it.k = next_offset(it.k); }
      if (it.k >= vol) return it;
      it.b = pre_of(it.k) + wave;
      if (it.b < pre_of(it.k + 1)) return it;
      it.kc = nkc - 1; // this wave has no block at this offset: skip its remaining chunks
    }
  };
  auto load_entry = [&](const Item &it) -> int {
    if (it.k >= vol) return (int)0x80000000;
    return (int)__builtin_amdgcn_raw_buffer_load_b32(rwords, c16x4, ebase + (unsigned)it.b * 64u, 0);
  };
  struct G { u32x4 a0, a1; int e; };
  auto gather = [&](G &q, const Item &it, int e) {
    q.e = e;
    const unsigned va = (((unsigned)e & 0x7fffffffu) >> 8) * rowbytes + g32;
    const unsigned so = (unsigned)it.kc * 128u;
    q.a0 = __builtin_amdgcn_raw_buffer_load_b128(rin, va, so, 0);
    q.a1 = __builtin_amdgcn_raw_buffer_load_b128(rin, va + 16u, so, 0);
  };
  auto stage_weights_src = [&](int k, int kc) {
    const int kW = wflip ? vol - 1 - k : k;
    return Wp + (((int64_t)kW * nkc + kc) * nnb + nb0) * 512;
  };
  auto lds_slot = [&](int i) { // float4 index i of the slab -> float offset in the LDS image
    const int j = i >> 7, r = i & 127, ln = r >> 1, h = r & 1;
    return ((j * 2 + h) * 64 + ln) * 4;
  };

  int K = next_offset(-1), KC = 0; // current stage
  if (K < vol) {
    const f32x4 *src = reinterpret_cast<const f32x4 *>(stage_weights_src(K, 0));
    *reinterpret_cast<f32x4 *>(Wb + lds_slot(threadIdx.x)) = src[threadIdx.x];
    *reinterpret_cast<f32x4 *>(Wb + lds_slot(threadIdx.x + 256)) = src[threadIdx.x + 256];
  }
  Item cur = first_item();
  Item nxt = advance(cur);
  int e_cur = load_entry(cur), e_nxt = load_entry(nxt);
  G gc, gn;
  if (cur.k < vol) gather(gc, cur, e_cur);
  int s = 0;
  while (K < vol) {
    __syncthreads(); // wbuf[s&1] complete; everybody done with wbuf[(s+1)&1] and with the previous stage's rows
    // next stage's weights: global -> registers now, -> LDS after this stage's MFMAs
    int K2 = K, KC2 = KC + 1;
    if (KC2 == nkc) { KC2 = 0; K2 = next_offset(K); }
    f32x4 wq0 = {0.f, 0.f, 0.f, 0.f}, wq1 = wq0;
    if (K2 < vol) {
      const f32x4 *src = reinterpret_cast<const f32x4 *>(stage_weights_src(K2, KC2));
      wq0 = src[threadIdx.x];
      wq1 = src[threadIdx.x + 256];
    }
    const float *Wl = Wb + (s & 1) * slab_floats;
    while (cur.k == K && cur.kc == KC) {
      // the following item's gather goes in flight before this item's MFMAs; its entry was loaded an item ago
      if (nxt.k < vol) gather(gn, nxt, e_nxt);
      const Item nn = advance(nxt);
      const int e_nn = load_entry(nn);
      if (!(DBG & 1)) {
        f32x4 acc[kNB];
        u32x4 w0[kNB], w1[kNB];
#pragma unroll
        for (int j = 0; j < kNB; ++j) {
          w0[j] = *reinterpret_cast<const u32x4 *>(Wl + j * 512 + lane * 4);
          w1[j] = *reinterpret_cast<const u32x4 *>(Wl + j * 512 + 256 + lane * 4);
          acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
          for (int j = 0; j < kNB; ++j)
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bcf_(w0[j][t]), bcf_(gc.a0[t]), acc[j], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
          for (int j = 0; j < kNB; ++j)
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bcf_(w1[j][t]), bcf_(gc.a1[t]), acc[j], 0, 0, 0);
        }
        if (gc.e >= 0) {
          const int orow = gc.e & 255;
          float *rowp = Ct + orow * kWS;
          const int sw = orow & 15;
#pragma unroll
          for (int j = 0; j < kNB; ++j) {
            f32x4 *dst = reinterpret_cast<f32x4 *>(rowp + (((j * 4 + g) ^ sw) << 2));
            f32x4 v = *dst;
            v += acc[j];
            *dst = v;
          }
        }
      }
      cur = nxt; nxt = nn; gc = gn; e_cur = e_nxt; e_nxt = e_nn;
    }
    if (K2 < vol) {
      float *dstb = Wb + ((s + 1) & 1) * slab_floats;
      *reinterpret_cast<f32x4 *>(dstb + lds_slot(threadIdx.x)) = wq0;
      *reinterpret_cast<f32x4 *>(dstb + lds_slot(threadIdx.x + 256)) = wq1;
    }
    K = K2; KC = KC2; ++s;
  }
  __syncthreads();
  // write the tile once (+ bias, CPU/Convolution.cpp:59-62): 16 float4 granules per row, un-swizzled
  const int nrows = (int)((V_out - row0) < kT ? (V_out - row0) : kT);
#pragma unroll 4
  for (int i = threadIdx.x; i < nrows * 16; i += 256) {
    const int r = i >> 4, q = i & 15;
    f32x4 v = *reinterpret_cast<const f32x4 *>(Ct + r * kWS + ((q ^ (r & 15)) << 2));
    if (bias) {
      const float *bb = bias + nb0 * 16 + q * 4;
      v[0] += bb[0]; v[1] += bb[1]; v[2] += bb[2]; v[3] += bb[3];
    }
    *reinterpret_cast<f32x4 *>(out + (row0 + r) * co + nb0 * 16 + q * 4) = v;
  }
}


// ------------------------------------------------------------------ column-split form
// Same 256-row x 64-column output tile in LDS, but the 4 waves split the COLUMNS: wave w owns column block w
// (16 columns) of the tile for all rows.  Consequences:
//   * a wave's MFMA A-operand is ONE 16-column slice of the offset's weights: 8 KiB per 128 input channels =
//     32 VGPRs, loaded straight from L2 into registers once per (offset, 128-channel group) and reused by every
//     block of that offset in the tile -- no weight traffic per block, no LDS staging, no barrier;
//   * waves never touch each other's tile columns: no cross-wave hazard on the tile, fixed accumulation order
//     (offset, block) per element => bit-reproducible;
//   * every wave does the same number of MFMAs (all blocks of the tile): perfectly balanced whatever the number
//     of blocks per offset.
// The price: each wave gathers every block's rows itself (the four waves read the same 8 KiB per block; they
// walk the blocks in the same order, so the repeats are served by the CU's vector L1).
template <int KG, int DBG> // KG = 32-channel chunks handled per pass over the tile's blocks (<= 4)
__global__ __launch_bounds__(256, 2) void k_conv_t256c(const float *__restrict__ in, int ci, int64_t in_bytes,
                                                       float *__restrict__ out, int co, int64_t V_out,
                                                       const int32_t *__restrict__ words, int64_t words_bytes,
                                                       int vol, int wflip, const float *__restrict__ Wp,
                                                       int64_t wp_bytes, const float *__restrict__ bias) {
  extern __shared__ __align__(16) float smem[];
  float *Ct = smem; // [256][64] floats, granule-swizzled
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int g = lane >> 4, c16 = lane & 15;
  const int nkc = ci >> 5, nnb = co >> 4;
  const int nb0 = blockIdx.y * kNB;
  const int64_t tile = blockIdx.x, row0 = tile * kT;
  const int64_t ntiles = t256_ntiles(V_out);
  const int maxb = t256_maxb(vol);
  const int vpre = lane <= vol ? words[tile * (vol + 1) + lane] : 0;
  {
    f32x4 z = {0.f, 0.f, 0.f, 0.f};
    f32x4 *c4 = reinterpret_cast<f32x4 *>(Ct);
#pragma unroll
    for (int i = 0; i < (kT * kWS / 4) / 256; ++i) c4[i * 256 + threadIdx.x] = z;
  }
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rin =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(in), 0, (int)in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rwords =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t *>(words), 0, (int)words_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(Wp), 0, (int)wp_bytes, 0x00020000);
  const unsigned ebase = (unsigned)((ntiles * (vol + 1) + tile * (int64_t)maxb * 16) * 4);
  const unsigned rowbytes = (unsigned)ci * 4u, g32 = (unsigned)g * 32u, c16x4 = (unsigned)c16 * 4u;
  const unsigned lane32 = (unsigned)lane * 32u;
  auto pre_of = [&](int k) { return __builtin_amdgcn_readlane(vpre, k); };
  auto next_offset = [&](int k) {
    ++k;
    while (k < vol && pre_of(k + 1) == pre_of(k)) ++k;
    return k;
  };
  struct WReg { u32x4 w0[KG], w1[KG]; };
  struct GReg { u32x4 a0[KG], a1[KG]; };
  const int ngroups = (nkc + KG - 1) / KG;
  auto load_w = [&](WReg &w, int k, int kg) { // this wave's 16-column slice, chunks kg*KG .. +KG-1
    const int kW = wflip ? vol - 1 - k : k;
#pragma unroll
    for (int c = 0; c < KG; ++c) {
      const int kc = kg * KG + c;
      if (kc < nkc) {
        const unsigned so = (unsigned)((((int64_t)kW * nkc + kc) * nnb + nb0 + wave) * 2048);
        w.w0[c] = __builtin_amdgcn_raw_buffer_load_b128(rw, lane32, so, 0);
        w.w1[c] = __builtin_amdgcn_raw_buffer_load_b128(rw, lane32 + 16u, so, 0);
      }
    }
  };
  auto load_entry = [&](int b) -> int {
    return (int)__builtin_amdgcn_raw_buffer_load_b32(rwords, c16x4, ebase + (unsigned)b * 64u, 0);
  };
  auto gather = [&](GReg &q, int e, int kg) {
    const unsigned va = (((unsigned)e & 0x7fffffffu) >> 8) * rowbytes + g32;
#pragma unroll
    for (int c = 0; c < KG; ++c) {
      const int kc = kg * KG + c;
      if (kc < nkc) {
        const unsigned so = (unsigned)kc * 128u;
        q.a0[c] = __builtin_amdgcn_raw_buffer_load_b128(rin, va, so, 0);
        q.a1[c] = __builtin_amdgcn_raw_buffer_load_b128(rin, va + 16u, so, 0);
      }
    }
  };
  auto accumulate = [&](int e, const f32x4 &acc) {
    if (e >= 0) {
      const int orow = e & 255;
      f32x4 *dst = reinterpret_cast<f32x4 *>(Ct + orow * kWS + (((wave * 4 + g) ^ (orow & 15)) << 2));
      f32x4 v = *dst;
      v += acc;
      *dst = v;
    }
  };
  const int nblk_all = pre_of(vol);
  for (int kg = 0; kg < ngroups; ++kg) {
    const int nc = (nkc - kg * KG) < KG ? (nkc - kg * KG) : KG;
    int k = next_offset(-1);
    WReg wc, wn;
    if (k < vol) load_w(wc, k, kg);
    // block iteration in pairs (two independent MFMA chains per wave); entries / rows prefetched one pair ahead
    int b = k < vol ? pre_of(k) : nblk_all; // blocks are contiguous in offset order: b runs over all of them
    int eA = 0, eB = 0;
    GReg ga, gb, na, nb_;
    auto pair_entries = [&](int bb, int kk, int &ea, int &eb) {
      // pair = blocks bb, bb+1 of offset kk (the second only if it belongs to the same offset)
      ea = bb < nblk_all ? load_entry(bb) : (int)0x80000000;
      eb = (bb + 1 < nblk_all && bb + 1 < pre_of(kk + 1)) ? load_entry(bb + 1) : (ea | (int)0x80000000);
    };
    if (b < nblk_all) {
      pair_entries(b, k, eA, eB);
      gather(ga, eA, kg);
      gather(gb, eB, kg);
    }
    while (k < vol) {
      const int kend = pre_of(k + 1);
      const int k2 = next_offset(k);
      if (k2 < vol) load_w(wn, k2, kg); // next offset's weights in flight during this offset's blocks
      while (b < kend) {
        // next pair (may belong to the next offset)
        int b2 = b + 2, kk2 = k;
        if (b2 >= kend) { b2 = kend; kk2 = k2; }
        int nA = (int)0x80000000, nB = (int)0x80000000;
        if (b2 < nblk_all && kk2 < vol) {
          pair_entries(b2, kk2, nA, nB);
          gather(na, nA, kg);
          gather(nb_, nB, kg);
        }
        if (!(DBG & 1)) {
          f32x4 accA = {0.f, 0.f, 0.f, 0.f}, accB = accA;
#pragma unroll
          for (int c = 0; c < KG; ++c) {
            if (c < nc) {
#pragma unroll
              for (int t = 0; t < 4; ++t) {
                accA = __builtin_amdgcn_mfma_f32_16x16x4f32(bcf_(wc.w0[c][t]), bcf_(ga.a0[c][t]), accA, 0, 0, 0);
                accB = __builtin_amdgcn_mfma_f32_16x16x4f32(bcf_(wc.w0[c][t]), bcf_(gb.a0[c][t]), accB, 0, 0, 0);
              }
#pragma unroll
              for (int t = 0; t < 4; ++t) {
                accA = __builtin_amdgcn_mfma_f32_16x16x4f32(bcf_(wc.w1[c][t]), bcf_(ga.a1[c][t]), accA, 0, 0, 0);
                accB = __builtin_amdgcn_mfma_f32_16x16x4f32(bcf_(wc.w1[c][t]), bcf_(gb.a1[c][t]), accB, 0, 0, 0);
              }
            }
          }
          accumulate(eA, accA);
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // A and B never share a row inside an offset, but
          accumulate(eB, accB);                                   // keep the two read-add-writes ordered anyway
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
        b = b2;
        eA = nA; eB = nB; ga = na; gb = nb_;
      }
      k = k2;
      wc = wn;
    }
  }
  __syncthreads();
  const int nrows = (int)((V_out - row0) < kT ? (V_out - row0) : kT);
#pragma unroll 4
  for (int i = threadIdx.x; i < nrows * 16; i += 256) {
    const int r = i >> 4, q = i & 15;
    f32x4 v = *reinterpret_cast<const f32x4 *>(Ct + r * kWS + ((q ^ (r & 15)) << 2));
    if (bias) {
      const float *bb = bias + nb0 * 16 + q * 4;
      v[0] += bb[0]; v[1] += bb[1]; v[2] += bb[2]; v[3] += bb[3];
    }
    *reinterpret_cast<f32x4 *>(out + (row0 + r) * co + nb0 * 16 + q * 4) = v;
  }
}


// ------------------------------------------------------------------ column-split form with shared gathers
// 128-row x 64-column output tile in LDS (32 KiB); the 4 waves split the COLUMNS (wave w = column block w) as
// in k_conv_t256c, and the gathered rows are SHARED through LDS: per pair of blocks the workgroup gathers the
// 32 rows once (each wave a quarter, 4 x 16-byte loads per lane, in flight during the previous pair's MFMAs),
// stores them in a double-buffered, granule-swizzled stage (2 x 16 KiB) and every wave reads its MFMA
// B-operands from there with conflict-free ds_read_b128.  So per 16-pair block the vector-memory path carries
// the 8 KiB of gathered rows ONCE (64-row-tile kernels: 8 KiB + 32 KiB of weights; k_conv_t256c: 4 x 8 KiB),
// the weights cost 8 KiB per wave per OFFSET (registers), and every wave issues the same MFMAs.
// One barrier per block pair (~2 x 1024 MFMA cycles per wave); 64 KiB LDS => two workgroups per CU.
constexpr int kT2 = 128;

template <int KG, int DBG>
__global__ __launch_bounds__(256, 2) void k_conv_cs(const float *__restrict__ in, int ci, int64_t in_bytes,
                                                    float *__restrict__ out, int co, int64_t V_out,
                                                    const int32_t *__restrict__ words, int64_t words_bytes, int vol,
                                                    int wflip, const float *__restrict__ Wp, int64_t wp_bytes,
                                                    const float *__restrict__ bias) {
  constexpr int RG = KG * 8;               // 16-byte granules per staged row
  constexpr int RF = KG * 32;              // floats per staged row
  constexpr int SWZ = (RG >= 16 && (RG & 15) == 0) ? 15 : 7; // XOR must stay inside the row's granules
  constexpr int STAGE = 2 * 16 * RF;       // floats per stage buffer (two blocks)
  extern __shared__ __align__(16) float smem[];
  float *Ct = smem;                        // [128][64] floats, granule-swizzled
  float *St = smem + kT2 * kWS;            // [2][2][16][RF]
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int g = lane >> 4, c16 = lane & 15;
  const int pr = wave * 8 + (lane >> 3), seg = lane & 7; // gather role: pair row 0..31, 16-byte segment
  const int nkc = ci >> 5, nnb = co >> 4;
  const int nb0 = blockIdx.y * kNB;
  const int64_t tile = blockIdx.x, row0 = tile * kT2;
  const int64_t ntiles = (V_out + kT2 - 1) / kT2;
  const int maxb = (kT2 / 16) * vol;
  const int vpre = lane <= vol ? words[tile * (vol + 1) + lane] : 0;
  {
    f32x4 z = {0.f, 0.f, 0.f, 0.f};
    f32x4 *c4 = reinterpret_cast<f32x4 *>(Ct);
#pragma unroll
    for (int i = 0; i < (kT2 * kWS / 4) / 256; ++i) c4[i * 256 + threadIdx.x] = z;
  }
  const __amdgpu_buffer_rsrc_t rin =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(in), 0, (int)in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rwords =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t *>(words), 0, (int)words_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(Wp), 0, (int)wp_bytes, 0x00020000);
  const unsigned ebase = (unsigned)((ntiles * (vol + 1) + tile * (int64_t)maxb * 16) * 4);
  const unsigned rowbytes = (unsigned)ci * 4u;
  const unsigned lane32 = (unsigned)lane * 32u;
  auto pre_of = [&](int k) { return __builtin_amdgcn_readlane(vpre, k); };
  auto next_offset = [&](int k) {
    ++k;
    while (k < vol && pre_of(k + 1) == pre_of(k)) ++k;
    return k;
  };
  struct WReg { u32x4 w0[KG], w1[KG]; };
  struct GReg { u32x4 v[KG]; };
  struct Ent { int eg, ea, eb; };          // gather-role entry of this lane's pair row; compute-role entries (A, B)
  const int ngroups = (nkc + KG - 1) / KG;
  const int nblk_all = pre_of(vol);
  auto load_w = [&](WReg &w, int k, int kg) {
    const int kW = wflip ? vol - 1 - k : k;
#pragma unroll
    for (int c = 0; c < KG; ++c) { // nkc % KG == 0 (dispatch): every load is unconditional, so the compiler's
      const int kc = kg * KG + c;  // vmcnt bookkeeping stays exact and nothing waits for a prefetch it does not use
      const unsigned so = (unsigned)((((int64_t)kW * nkc + kc) * nnb + nb0 + wave) * 2048);
      w.w0[c] = __builtin_amdgcn_raw_buffer_load_b128(rw, lane32, so, 0);
      w.w1[c] = __builtin_amdgcn_raw_buffer_load_b128(rw, lane32 + 16u, so, 0);
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
    float *rowp = St + buf * STAGE + pr * RF;
#pragma unroll
    for (int i = 0; i < KG; ++i)
      *reinterpret_cast<u32x4 *>(rowp + (((seg + 8 * i) ^ (pr & SWZ)) << 2)) = q.v[i];
  };
  auto accumulate = [&](int e, const f32x4 &acc) {
    if (e >= 0) {
      const int orow = e & 255;
      f32x4 *dst = reinterpret_cast<f32x4 *>(Ct + orow * kWS + (((wave * 4 + g) ^ (orow & 15)) << 2));
      f32x4 v = *dst;
      v += acc;
      *dst = v;
    }
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
  for (int kg = 0; kg < ngroups; ++kg) {
    Pos p0;
    p0.k = next_offset(-1);
    if (p0.k >= vol) break;
    p0.b = pre_of(p0.k);
    Pos p1 = adv(p0), p2 = adv(p1);
    Ent e0 = load_ent(p0.b, p0.k), e2 = e0;
    Ent e1 = load_ent(p1.k < vol ? p1.b : p0.b, p1.k < vol ? p1.k : p0.k);
    GReg gq;
    gather(gq, e0.eg, kg);
    __syncthreads();                       // zero fill done / previous group's stage reads done
    stage_store(gq, par);
    __syncthreads();
    // one pipeline step: pair p0 with the weight registers `w` (passed by reference: the two weight sets are
    // used from fixed registers by two copies of this body -- no register shuffling at an offset change)
    auto step = [&](const WReg &w) __attribute__((always_inline)) {
      // unconditional: past the tile's last pair the loads repeat a valid pair and their results are dropped
      gather(gq, e1.eg, kg);                                 // entry loaded an iteration ago: no wait
      {
        const bool v2 = p2.k < vol;
        e2 = load_ent(v2 ? p2.b : p0.b, v2 ? p2.k : p0.k);
      }
      __builtin_amdgcn_sched_barrier(0);                     // the prefetches are issued HERE, ahead of the MFMAs

      {
        const float *sa = St + par * STAGE + c16 * RF;
        const float *sb = sa + 16 * RF;
        // all of the pair's B operands leave LDS before the first MFMA (counted lgkmcnt waits follow)
        u32x4 a0[KG], a1[KG], b0[KG], b1[KG];
#pragma unroll
        for (int c = 0; c < KG; ++c) {
          const int q0 = ((c * 8 + g * 2) ^ (c16 & SWZ)) << 2, q1 = ((c * 8 + g * 2 + 1) ^ (c16 & SWZ)) << 2;
          a0[c] = *reinterpret_cast<const u32x4 *>(sa + q0);
          b0[c] = *reinterpret_cast<const u32x4 *>(sb + q0);
          a1[c] = *reinterpret_cast<const u32x4 *>(sa + q1);
          b1[c] = *reinterpret_cast<const u32x4 *>(sb + q1);
        }
        f32x4 accA = {0.f, 0.f, 0.f, 0.f}, accB = accA;
#pragma unroll
        for (int c = 0; c < KG; ++c) {
          {
            if (DBG & 1) { // timing experiments: operands consumed, no MFMAs
              accA[0] += bcf_(a0[c][0]) + bcf_(a1[c][0]) + bcf_(w.w0[c][0]) + bcf_(w.w1[c][0]);
              accB[0] += bcf_(b0[c][0]) + bcf_(b1[c][0]);
            } else {
#pragma unroll
              for (int t = 0; t < 4; ++t) {
                accA = __builtin_amdgcn_mfma_f32_16x16x4f32(bcf_(w.w0[c][t]), bcf_(a0[c][t]), accA, 0, 0, 0);
                accB = __builtin_amdgcn_mfma_f32_16x16x4f32(bcf_(w.w0[c][t]), bcf_(b0[c][t]), accB, 0, 0, 0);
              }
#pragma unroll
              for (int t = 0; t < 4; ++t) {
                accA = __builtin_amdgcn_mfma_f32_16x16x4f32(bcf_(w.w1[c][t]), bcf_(a1[c][t]), accA, 0, 0, 0);
                accB = __builtin_amdgcn_mfma_f32_16x16x4f32(bcf_(w.w1[c][t]), bcf_(b1[c][t]), accB, 0, 0, 0);
              }
            }
          }
        }
        accumulate(e0.ea, accA);
        accumulate(e0.eb, accB);
      }
      if (p1.k < vol) stage_store(gq, par ^ 1);
      wg_barrier();
      par ^= 1;
      p0 = p1; p1 = p2; p2 = adv(p2);
      e0 = e1; e1 = e2;
    };
    WReg wA, wB;
    int k = p0.k;
    load_w(wA, k, kg);
    for (;;) {
      int kn = next_offset(k);
      if (kn < vol) load_w(wB, kn, kg);                      // next offset's weights in flight during this offset
      while (p0.k == k) step(wA);
      if (kn >= vol) break;
      k = kn;
      kn = next_offset(k);
      if (kn < vol) load_w(wA, kn, kg);
      while (p0.k == k) step(wB);
      if (kn >= vol) break;
      k = kn;
    }
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
    *reinterpret_cast<f32x4 *>(out + (row0 + r) * co + nb0 * 16 + q * 4) = v;
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
  AABR_CHECK_ARG(tile_rows == 128 || tile_rows == 256, "tile_rows must be 128 or 256");
  if (V == 0) return AABR_OK;
  AABR_CHECK_ARG(table && blocks, "null pointer");
  const unsigned nt = (unsigned)((V + tile_rows - 1) / tile_rows);
  if (tile_rows == 128)
    hipLaunchKernelGGL(k_build_tileT<128>, dim3(nt), dim3(128), 0, (hipStream_t)stream_, table, V, vol, blocks);
  else
    hipLaunchKernelGGL(k_build_tileT<256>, dim3(nt), dim3(256), 0, (hipStream_t)stream_, table, V, vol, blocks);
  AABR_CHECK_LAUNCH();
  return AABR_OK;
}

static char wide_form() { // tuning experiments only: AABR_WIDE_FORM = s | c | cs
  const char *f = getenv("AABR_WIDE_FORM");
  if (f && f[0] == 's') return 's';
  if (f && f[0] == 'c' && f[1] != 's') return 'c';
  return 'x'; // cs
}

// 0: use the 64-row-tile kernels of conv.hip; 128 / 256: rows per tile of the block stream aabr_conv_forward_wide wants
extern "C" int aabr_conv_wide_tile_rows(int n_in, int n_out, int64_t rows_in, int64_t V_out, int vol) {
  if (n_in <= 0 || n_out <= 0 || (n_in & 31) || (n_out & 63) || vol <= 0 || vol > kMaxVol) return 0;
  if (rows_in >= (1ll << 23) || rows_in * n_in * 4 >= (1ll << 31)) return 0;
  const int T = wide_form() == 'x' ? 128 : 256;
  if (wide_words(V_out, vol, T) * 4 >= (1ll << 31)) return 0;
  if ((int64_t)vol * n_in * n_out * 4 >= (1ll << 31)) return 0;
  if (n_in > 128 && (n_in & 127)) return 0; // channel groups of 128: every load of the inner loop unconditional
  if (const char *ov = getenv("AABR_CONV_WIDE")) { // tuning experiments only: 0 = never, 1 = whenever supported
    if (ov[0] == '0') return 0;
    if (ov[0] == '1') return T;
  }
  // enough workgroups to fill the chip twice over (measured, profiles/r02_conv_wide_ab.txt: wins from ~340
  // workgroups up, loses below ~180)
  return (((V_out + T - 1) / T) * (n_out / 64) >= 320) ? T : 0;
}

extern "C" int aabr_conv_forward_wide(const float *in_feats, int n_in, int64_t rows_in, float *out_feats, int n_out,
                                      int64_t V_out, const int32_t *blocks, int tile_rows, int vol, const float *bias,
                                      int flags, const float *wpack, void *stream_) {
  hipStream_t st = (hipStream_t)stream_;
  AABR_CHECK_ARG(n_in > 0 && n_out > 0 && (n_in & 31) == 0 && (n_out & 63) == 0, "plane counts: n_in % 32, n_out % 64");
  AABR_CHECK_ARG(vol > 0 && vol <= kMaxVol && V_out >= 0 && rows_in >= 0, "bad sizes");
  AABR_CHECK_ARG(tile_rows == 128 || tile_rows == 256, "tile_rows must be 128 or 256");
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
#define AABR_LAUNCH_WIDE(KERNEL, NAME, LDS, ...)                                                          \
  do {                                                                                                    \
    static bool attr = false;                                                                             \
    if (!attr) {                                                                                          \
      AABR_CHECK_HIP(hipFuncSetAttribute((const void *)KERNEL, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LDS))); \
      attr = true;                                                                                        \
    }                                                                                                     \
    g_last_variant = NAME;                                                                                \
    hipLaunchKernelGGL(KERNEL, grid, dim3(256), (LDS), st, __VA_ARGS__);                                  \
  } while (0)
  if (tile_rows == 256) {
    if (wide_form() == 's') {
      const size_t lds = (size_t)(kT * kWS + 2 * kNB * 512) * sizeof(float);
      if (dbg & 1)
        AABR_LAUNCH_WIDE((k_conv_t256<1>), "k_conv_t256<1>", lds, in_feats, n_in, in_bytes, out_feats, n_out, V_out, blocks,
                         words_bytes, vol, flip, wpack, bias);
      else
        AABR_LAUNCH_WIDE((k_conv_t256<0>), "k_conv_t256<0>", lds, in_feats, n_in, in_bytes, out_feats, n_out, V_out, blocks,
                         words_bytes, vol, flip, wpack, bias);
    } else {
      const size_t lds = (size_t)(kT * kWS) * sizeof(float);
#define AABR_WIDE_C(KG)                                                                                   \
  AABR_LAUNCH_WIDE((k_conv_t256c<KG, 0>), "k_conv_t256c<" #KG ",0>", lds, in_feats, n_in, in_bytes, out_feats, n_out, V_out, \
                   blocks, words_bytes, vol, flip, wpack, wp_bytes, bias)
      if (kg == 1) AABR_WIDE_C(1); else if (kg == 2) AABR_WIDE_C(2); else if (kg == 3) AABR_WIDE_C(3); else AABR_WIDE_C(4);
#undef AABR_WIDE_C
    }
  } else {
#define AABR_WIDE_CS(KG, D)                                                                               \
  AABR_LAUNCH_WIDE((k_conv_cs<KG, D>), "k_conv_cs<" #KG "," #D ">",                                       \
                   (size_t)(kT2 * kWS + 2 * 2 * 16 * KG * 32) * sizeof(float), in_feats, n_in, in_bytes, out_feats, \
                   n_out, V_out, blocks, words_bytes, vol, flip, wpack, wp_bytes, bias)
    if (dbg & 3) { // timing experiments (tools/): only the 128-channel-group instance carries the debug variants
      AABR_CHECK_ARG(kg == 4, "debug variants exist for n_in >= 128 only");
      if ((dbg & 3) == 1) AABR_WIDE_CS(4, 1); else if ((dbg & 3) == 2) AABR_WIDE_CS(4, 2); else AABR_WIDE_CS(4, 3);
    } else {
      if (kg == 1) AABR_WIDE_CS(1, 0); else if (kg == 2) AABR_WIDE_CS(2, 0); else if (kg == 3) AABR_WIDE_CS(3, 0);
      else AABR_WIDE_CS(4, 0);
    }
#undef AABR_WIDE_CS
  }
#undef AABR_LAUNCH_WIDE
  AABR_CHECK_LAUNCH();
  return AABR_OK;
}
