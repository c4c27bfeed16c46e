// conv.hip -- sparse 3-D convolution on gather tables, fp32 MFMA (gfx950).
//
// One formulation serves SubmanifoldConvolution, Convolution and Deconvolution, forward and
// input-gradient (reference: SCN/CPU/Convolution.cpp:45-185, SCN/CPU/Deconvolution.cpp:7-77,
// SCN/CUDA/Convolution.cu:57-667):
//
//     out[o] = bias + sum_k  in[ table[k][o] ] @ Wl[k]          (table entry -1 -> no term)
//
// It is OUTPUT-STATIONARY: a workgroup owns 64 consecutive output rows and a slab of output
// columns, keeps that tile in LDS for the whole sweep over the filter offsets and writes it to
// HBM exactly once -- no read-modify-write of `out` per offset and no atomics (the reference's
// GPU path re-reads and re-writes `out` 27 times and its rule book crosses PCIe per offset).
// The gather table is compiled once per rule book into per-tile blocks of 16 (partner row,
// local row) pairs sharing one offset; each block is one v_mfma_f32_16x16x4_f32 chain:
//     D^T[out col][pair] += Wl^T[out col][c] * in[partner(pair)][c]
// so each lane ends up with 4 consecutive output columns of one pair (one 16-byte LDS
// read-add-write).  Accumulation order is fixed (offset order, then channel order inside the
// MFMA chain) => results are bit-reproducible run to run.
//
// fp32 MFMA == k-ordered fmaf chain (exact fp32); peak 157 TFLOP/s.
#include "common.h"
#include <stdlib.h>

namespace aabr {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kKC = 32; // channels per K-chunk: 4 lane groups x 8 consecutive channels

__host__ __device__ inline int nkc_of(int ci) { return (ci + kKC - 1) / kKC; }
__host__ __device__ inline int nnb_of(int co) { return (co + 15) / 16; }

// Wp[k][kc][nb][lane][s] = Wl[k][kc*32 + (lane>>4)*8 + s][nb*16 + (lane&15)]   (0 outside)
// Wl[k][c][n] = W[wk][c][n] (plain) or W[wk][n][c] (transpose), wk = flip ? vol-1-k : k.
__global__ __launch_bounds__(256) void k_pack_weights(const float *__restrict__ W, int vol, int ci, int co,
                                                      int transpose, int flip, float *__restrict__ Wp) {
  const int nkc = nkc_of(ci), nnb = nnb_of(co);
  int64_t total = (int64_t)vol * nkc * nnb * 512;
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  int s = idx & 7;
  int lane = (idx >> 3) & 63;
  int64_t r = idx >> 9;
  int nb = (int)(r % nnb); r /= nnb;
  int kc = (int)(r % nkc); r /= nkc;
  int k = (int)r;
  int c = kc * kKC + (lane >> 4) * 8 + s;
  int n = nb * 16 + (lane & 15);
  float v = 0.0f;
  if (c < ci && n < co) {
    int wk = flip ? vol - 1 - k : k;
    v = transpose ? W[((int64_t)wk * co + n) * ci + c] : W[((int64_t)wk * ci + c) * co + n];
  }
  Wp[idx] = v;
}

// both orientations in one launch: the forward layout (Wl = W[k]) into Wf and the input-gradient
// layout (Wl = W[k]^T, plane counts swapped) into Wt -- a training step needs each exactly once per
// weight version, so the backward pass can run with the "prepacked" flag and no pack launch of its own
template <typename TO>
__device__ inline TO pack_cvt(float v);
template <> __device__ inline float pack_cvt<float>(float v) { return v; }
template <> __device__ inline __bf16 pack_cvt<__bf16>(float v) { return (__bf16)v; }

template <typename TO>
__global__ __launch_bounds__(256) void k_pack_weights2(const float *__restrict__ W, int vol, int n_in, int n_out,
                                                       TO *__restrict__ Wf, TO *__restrict__ Wt) {
  const int64_t total_f = (int64_t)vol * nkc_of(n_in) * nnb_of(n_out) * 512;
  const int64_t total_t = (int64_t)vol * nkc_of(n_out) * nnb_of(n_in) * 512;
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool tr = idx >= total_f;
  if (tr) idx -= total_f;
  if (idx >= (tr ? total_t : total_f)) return;
  const int ci = tr ? n_out : n_in, co = tr ? n_in : n_out;
  const int nkc = nkc_of(ci), nnb = nnb_of(co);
  int s = idx & 7;
  int lane = (idx >> 3) & 63;
  int64_t r = idx >> 9;
  int nb = (int)(r % nnb); r /= nnb;
  int kc = (int)(r % nkc); r /= nkc;
  int k = (int)r;
  int c = kc * kKC + (lane >> 4) * 8 + s;
  int n = nb * 16 + (lane & 15);
  float v = 0.0f;
  if (c < ci && n < co) v = tr ? W[((int64_t)k * co + n) * ci + c] : W[((int64_t)k * ci + c) * co + n];
  (tr ? Wt : Wf)[idx] = pack_cvt<TO>(v);
}

// every convolution of a network in ONE launch (a training step packs each weight once per version; 40-odd
// launches of a few microseconds each otherwise).  jobs (device, 48 bytes each):
//   { W, Wf, Wt : pointers; vol, n_in, n_out, bf16 : int32; first_block : int64 } -- job j owns the
// thread blocks [first_block[j], first_block[j+1]) of the grid; a block never straddles two jobs.
struct PackJob {
  const float *W;
  void *Wf, *Wt;
  int32_t vol, n_in, n_out, bf16;
  int64_t first_block;
};
static_assert(sizeof(PackJob) == 48, "PackJob layout is part of the C ABI");

// Round 4: a workgroup owns one 64 x 64 tile of one W[k]: the tile is read ONCE with coalesced 256-byte rows into LDS
// and both orientations' fragments (8 of 2 KiB each way) are written from there with coalesced stores.  Round 3's kernel
// read W element-wise in fragment order (32-byte segments) once per orientation: 167 us for the 21 M parameters of
// FPN_Net at the head of every training step, on the critical path.
constexpr int kPT = 64;                          // tile edge: 2 channel chunks x 4 column blocks
__device__ inline void pack_store(void *dst, int64_t idx, float v, int mode, int64_t plane) {
  (void)plane;
  if (mode) reinterpret_cast<__bf16 *>(dst)[idx] = (__bf16)v;
  else reinterpret_cast<float *>(dst)[idx] = v;
}

__global__ __launch_bounds__(256) void k_pack_weights_jobs(const PackJob *__restrict__ jobs, int n_jobs) {
  __shared__ int sj;
  __shared__ float tile[kPT][kPT + 1];
  if (threadIdx.x == 0) {
    int lo = 0, hi = n_jobs - 1; // last job whose first_block <= blockIdx.x
    while (lo < hi) {
      int mid = (lo + hi + 1) >> 1;
      if (jobs[mid].first_block <= (int64_t)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    sj = lo;
  }
  __syncthreads();
  const PackJob jb = jobs[sj];
  const int vol = jb.vol, n_in = jb.n_in, n_out = jb.n_out;
  const int tc = (n_in + kPT - 1) / kPT, tn = (n_out + kPT - 1) / kPT;
  int64_t b = (int64_t)blockIdx.x - jb.first_block;
  const int tj = (int)(b % tn); b /= tn;
  const int ti = (int)(b % tc); b /= tc;
  const int k = (int)b;
  if (k >= vol) return;
  const int c0 = ti * kPT, n0 = tj * kPT;
  // W[k][c0 + r][n0 + q]: 16 threads per row (4 floats each), 16 rows per sweep
  {
    const int q4 = (threadIdx.x & 15) * 4, r0 = threadIdx.x >> 4;
    const float *Wk = jb.W + (int64_t)k * n_in * n_out;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = r0 + 16 * j, c = c0 + r;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int n = n0 + q4 + e;
        tile[r][q4 + e] = (c < n_in && n < n_out) ? Wk[(int64_t)c * n_out + n] : 0.0f;
      }
    }
  }
  __syncthreads();
  // forward layout: Wl[c][n] = W[k][c][n], fragments (kc, nb) of 32 channels x 16 columns
  {
    const int nkc = nkc_of(n_in), nnb = nnb_of(n_out);
    const int64_t plane = (int64_t)vol * nkc * nnb * 512;
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
      const int e = i * 256 + threadIdx.x, fr = e >> 9, r = e & 511;
      const int lane = r >> 3, sidx = r & 7;
      const int kc = (c0 >> 5) + (fr >> 2), nb = (n0 >> 4) + (fr & 3);
      if (kc >= nkc || nb >= nnb) continue;
      const float v = tile[(fr >> 2) * 32 + (lane >> 4) * 8 + sidx][(fr & 3) * 16 + (lane & 15)];
      pack_store(jb.Wf, (((int64_t)k * nkc + kc) * nnb + nb) * 512 + r, v, jb.bf16, plane);
    }
  }
  // input-gradient layout: Wl[c'][n'] = W[k][n'][c'] (plane counts swapped): c' runs over this tile's columns
  {
    const int nkc = nkc_of(n_out), nnb = nnb_of(n_in);
    const int64_t plane = (int64_t)vol * nkc * nnb * 512;
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
      const int e = i * 256 + threadIdx.x, fr = e >> 9, r = e & 511;
      const int lane = r >> 3, sidx = r & 7;
      const int kc = (n0 >> 5) + (fr >> 2), nb = (c0 >> 4) + (fr & 3);
      if (kc >= nkc || nb >= nnb) continue;
      const float v = tile[(fr & 3) * 16 + (lane & 15)][(fr >> 2) * 32 + (lane >> 4) * 8 + sidx];
      pack_store(jb.Wt, (((int64_t)k * nkc + kc) * nnb + nb) * 512 + r, v, jb.bf16, plane);
    }
  }
}

// ------------------------------------------------------------------ compiled rule book, part 1
// Tile-major MFMA block lists.  For every tile of 64 consecutive output rows the gather table
// is compiled ONCE per rule book into blocks of 16 (partner row, local row) pairs that share a
// filter offset; every convolution that uses the rule book (forward of each layer at that scale,
// and the input-gradient passes) then streams these blocks with no ballots and no table reads.
//   words:  [ntiles] nblk | [ntiles][MAXB] offset k of each block | [ntiles][MAXB][16] entries
//   entry = (partner_row << 6) | local_row; padding entries repeat the block's first pair with
//   bit 31 set (gather is valid, result is discarded);  MAXB = 4 * vol.
__host__ __device__ inline int64_t tb_ntiles(int64_t V) { return (V + 63) / 64; }
__host__ __device__ inline int tb_maxb(int vol) { return 4 * vol; }

__global__ __launch_bounds__(256) void k_build_tile_blocks(const StreamJobs js) {      // common.h: one launch, many books
  const int job = stream_job_of(js, blockIdx.x);
  const int32_t *__restrict__ table = js.j[job].table;
  int32_t *__restrict__ words = js.j[job].words;
  const int64_t V = js.j[job].V;
  const int vol = js.j[job].vol;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t ntiles = tb_ntiles(V), tile = (int64_t)(blockIdx.x - js.first[job]) * 4 + wave;
  if (tile >= ntiles) return;
  const int maxb = tb_maxb(vol);
  int32_t *nblk = words;
  int32_t *blk_k = words + ntiles + tile * maxb;
  int32_t *ent = words + ntiles + ntiles * maxb + tile * maxb * 16;
  const int64_t row = tile * 64 + lane;
  const bool valid = row < V;
  int b0 = 0;
  constexpr int B = 9; // table rows fetched per batch: nine independent loads in flight, then nine ballots
  for (int k0 = 0; k0 < vol; k0 += B) {
    int tv[B];
#pragma unroll
    for (int j = 0; j < B; ++j) tv[j] = (valid && k0 + j < vol) ? table[(int64_t)(k0 + j) * V + row] : -1;
#pragma unroll
    for (int j = 0; j < B; ++j) {
    const int k = k0 + j;
    const int t = tv[j];
    const unsigned long long m = __ballot(t >= 0);
    if (m == 0) continue;
    const int cnt = __popcll(m);
    const int pos = __popcll(m & ((1ull << lane) - 1ull));
    const int nmb = (cnt + 15) >> 4;
    const int e = (t << 6) | lane;
    if (t >= 0) ent[b0 * 16 + pos] = e;
    // padding: a copy of the first pair with the discard bit (31) set, so the MFMA kernel can gather
    // unconditionally (no zero-fill selects) and simply drops the result
    const int e_first = __shfl(e, __ffsll((long long)m) - 1);
    if (lane < nmb * 16 - cnt) ent[b0 * 16 + cnt + lane] = e_first | (int)0x80000000;
    if (lane < nmb) blk_k[b0 + lane] = k;
    b0 += nmb;
    }
  }
  if (lane == 0) nblk[tile] = b0;
}

// gather 8 consecutive channels of the block entry's partner row (zeros for padding entries)
template <bool ALIGNED>
__device__ inline void conv_block_load(const float *__restrict__ in, int ci, int e, int g, int kc,
                                       float (&b8)[8]) {
  const int inrow = e >> 6;
  if (ALIGNED) {
    if (e >= 0) {
      const float4 *src = reinterpret_cast<const float4 *>(in + (int64_t)inrow * ci + kc * kKC + g * 8);
      float4 v0 = src[0], v1 = src[1];
      b8[0] = v0.x; b8[1] = v0.y; b8[2] = v0.z; b8[3] = v0.w;
      b8[4] = v1.x; b8[5] = v1.y; b8[6] = v1.z; b8[7] = v1.w;
    } else {
#pragma unroll
      for (int s = 0; s < 8; ++s) b8[s] = 0.0f;
    }
  } else {
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      int c = kc * kKC + g * 8 + s;
      b8[s] = (e >= 0 && c < ci) ? in[(int64_t)inrow * ci + c] : 0.0f;
    }
  }
}

// gathered operands of one pipeline step: the same 32-channel chunk of two blocks (A, B)
struct ConvStep {
  float a8[8], b8[8];
  int eA, eB;
};

// weights stream from L2 (packed, lane-linear, 2 KiB per block and chunk) right before their MFMAs;
// the two blocks' chains are interleaved: two independent accumulators keep the 40-cycle
// dependent latency of v_mfma_f32_16x16x4_f32 off the critical path
template <int NBW>
__device__ inline void conv_step_mfma(const ConvStep &st, const float *__restrict__ WkA,
                                      const float *__restrict__ WkB, int kc, int nnb, int nb0, int lane,
                                      f32x4 (&accA)[NBW], f32x4 (&accB)[NBW]) {
#pragma unroll
  for (int j = 0; j < NBW; ++j) {
    if (nb0 + j < nnb) {
      const int64_t off = (((int64_t)kc * nnb + nb0 + j) * 64 + lane) * 8;
      const float4 *wa = reinterpret_cast<const float4 *>(WkA + off);
      const float4 *wb = reinterpret_cast<const float4 *>(WkB + off);
      const float4 w0 = wa[0], w1 = wa[1], u0 = wb[0], u1 = wb[1];
      accA[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.x, st.a8[0], accA[j], 0, 0, 0);
      accB[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(u0.x, st.b8[0], accB[j], 0, 0, 0);
      accA[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.y, st.a8[1], accA[j], 0, 0, 0);
      accB[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(u0.y, st.b8[1], accB[j], 0, 0, 0);
      accA[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.z, st.a8[2], accA[j], 0, 0, 0);
      accB[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(u0.z, st.b8[2], accB[j], 0, 0, 0);
      accA[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.w, st.a8[3], accA[j], 0, 0, 0);
      accB[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(u0.w, st.b8[3], accB[j], 0, 0, 0);
      accA[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.x, st.a8[4], accA[j], 0, 0, 0);
      accB[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(u1.x, st.b8[4], accB[j], 0, 0, 0);
      accA[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.y, st.a8[5], accA[j], 0, 0, 0);
      accB[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(u1.y, st.b8[5], accB[j], 0, 0, 0);
      accA[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.z, st.a8[6], accA[j], 0, 0, 0);
      accB[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(u1.z, st.b8[6], accB[j], 0, 0, 0);
      accA[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.w, st.a8[7], accA[j], 0, 0, 0);
      accB[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(u1.w, st.b8[7], accB[j], 0, 0, 0);
    }
  }
}

// lane holds D^T[out col = j*16 + g*4 + r][pair c16]: 4 consecutive columns of the pair's row
template <int NBW, int WS>
__device__ inline void conv_block_accumulate(float *Ct, int e, int g, const f32x4 (&acc)[NBW]) {
  if (e >= 0) {
    const int orow = e & 63;
#pragma unroll
    for (int j = 0; j < NBW; ++j) {
      float4 *dst = reinterpret_cast<float4 *>(Ct + orow * WS + j * 16 + g * 4);
      float4 cur = *dst;
      cur.x += acc[j][0]; cur.y += acc[j][1]; cur.z += acc[j][2]; cur.w += acc[j][3];
      *dst = cur;
    }
  }
}

// One workgroup owns a tile of 64 output rows x (NBW*16) output columns.  Its WPB waves take
// the tile's blocks round-robin, two blocks per step, and software-pipeline three stages in
// registers: block entries two pairs ahead, weights + gathered rows one step ahead, MFMAs now.
// Each wave accumulates into a private LDS tile; the tiles are summed in wave order at the end,
// so the result is bit-reproducible.  ALIGNED: ci % 32 == 0 (float4 gathers, no bounds checks).
struct PairEnt { int eA, eB, kA, kB; };

template <int NBW, int WPB, bool ALIGNED>
__global__ __launch_bounds__(WPB * 64, (NBW <= 2 ? 3 : 2)) void k_conv_blocks_mfma(const float *__restrict__ in, int ci,
                                                              float *__restrict__ out, int co,
                                                              int64_t V_out, const int32_t *__restrict__ words,
                                                              int vol, int wflip, const float *__restrict__ Wp,
                                                              const float *__restrict__ bias) {
  constexpr int WS = NBW * 16;              // C-tile row stride (floats)
  constexpr int TILE = 64 * WS;             // floats per wave
  extern __shared__ __align__(16) float smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int g = lane >> 4, c16 = lane & 15;
  float *Ct = smem + (size_t)wave * TILE;
  const int maxb = tb_maxb(vol);

  const int nkc = nkc_of(ci), nnb = nnb_of(co);
  const int nb0 = blockIdx.y * NBW;                       // first global n-block of the slab
  const int64_t tile = blockIdx.x, row0 = tile * 64;
  const int64_t ntiles = tb_ntiles(V_out);
  const int nblk = words[tile];
  const int32_t *blk_k = words + ntiles + tile * maxb;
  const int32_t *ent = words + ntiles + ntiles * maxb + tile * maxb * 16;
  const int64_t wk_stride = (int64_t)nkc * nnb * 512;

  for (int i = lane; i < TILE; i += 64) Ct[i] = 0.0f;
  const int dbg = wflip >> 8; // timing experiments only (tools_conv_bench.py): 1 = no MFMAs, 2 = no gathers
  wflip &= 1;

  // this wave's blocks: wave, wave + WPB, ...; consumed two per pair
  const int nmine = (wave < nblk && !(dbg & 4)) ? (nblk - wave + WPB - 1) / WPB : 0; // dbg 4: skip the main loop
  const int npairs = (nmine + 1) >> 1;
  auto load_pair = [&](int pr) {
    PairEnt p;
    const int bA = wave + (2 * pr) * WPB, bB = bA + WPB;
    if (pr < npairs) {
      const bool hasB = bB < nblk;
      p.eA = ent[bA * 16 + c16];
      p.eB = hasB ? ent[bB * 16 + c16] : -1;
      p.kA = blk_k[bA];
      p.kB = hasB ? blk_k[bB] : p.kA;
    } else {
      p.eA = p.eB = -1; p.kA = p.kB = 0;
    }
    return p;
  };
  auto fetch = [&](ConvStep &st, const PairEnt &p, int kc) {
    st.eA = p.eA; st.eB = p.eB;
    conv_block_load<ALIGNED>(in, ci, st.eA, g, kc, st.a8);
    conv_block_load<ALIGNED>(in, ci, st.eB, g, kc, st.b8);
  };
  PairEnt p0 = load_pair(0), p1 = load_pair(1), p2 = load_pair(2);
  ConvStep cur, nxt;
  if (npairs > 0) fetch(cur, p0, 0);
  f32x4 accA[NBW], accB[NBW];
  for (int pr = 0; pr < npairs; ++pr) {
#pragma unroll
    for (int j = 0; j < NBW; ++j) {
      accA[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      accB[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    int kA = __builtin_amdgcn_readfirstlane(p0.kA), kB = __builtin_amdgcn_readfirstlane(p0.kB);
    if (wflip) { kA = vol - 1 - kA; kB = vol - 1 - kB; }
    const float *WkA = Wp + kA * wk_stride, *WkB = Wp + kB * wk_stride;
    for (int kc = 0; kc < nkc; ++kc) {
      // the gathers of the following step go in flight before this step's MFMAs issue
      if (!(dbg & 2)) {
        if (kc + 1 < nkc) fetch(nxt, p0, kc + 1);
        else if (pr + 1 < npairs) fetch(nxt, p1, 0);
      }
      if (!(dbg & 1)) conv_step_mfma<NBW>(cur, WkA, WkB, kc, nnb, nb0, lane, accA, accB);
      cur = nxt;
    }
    conv_block_accumulate<NBW, WS>(Ct, p0.eA, g, accA);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // A and B may hit the same row
    conv_block_accumulate<NBW, WS>(Ct, p0.eB, g, accB);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    p0 = p1; p1 = p2; p2 = load_pair(pr + 3);
  }
  __syncthreads();
  // combine the per-wave tiles in wave order (+ bias, CPU/Convolution.cpp:59-62) and write once
  const int nrows = (int)((V_out - row0) < 64 ? (V_out - row0) : 64);
  const int wcols = (co - nb0 * 16) < NBW * 16 ? (co - nb0 * 16) : NBW * 16;
  if ((co & 3) == 0) {
    const int q = wcols >> 2; // float4 per row
    for (int i = threadIdx.x; i < nrows * q; i += WPB * 64) {
      int r = i / q, cq = i % q;
      float4 v = *reinterpret_cast<const float4 *>(smem + r * WS + cq * 4);
#pragma unroll
      for (int w = 1; w < WPB; ++w) {
        float4 u = *reinterpret_cast<const float4 *>(smem + (size_t)w * TILE + r * WS + cq * 4);
        v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
      }
      if (bias) {
        const float *bb = bias + nb0 * 16 + cq * 4;
        v.x += bb[0]; v.y += bb[1]; v.z += bb[2]; v.w += bb[3];
      }
      *reinterpret_cast<float4 *>(out + (row0 + r) * co + nb0 * 16 + cq * 4) = v;
    }
  } else {
    for (int i = threadIdx.x; i < nrows * wcols; i += WPB * 64) {
      int r = i / wcols, cc = i % wcols;
      float v = smem[r * WS + cc];
#pragma unroll
      for (int w = 1; w < WPB; ++w) v += smem[(size_t)w * TILE + r * WS + cc];
      if (bias) v += bias[nb0 * 16 + cc];
      out[(row0 + r) * co + nb0 * 16 + cc] = v;
    }
  }
}

// Lean variant for ci % 32 == 0 and tensors < 2 GiB: every load goes through a buffer descriptor
// (32-bit per-lane offsets, wave-uniform parts in SGPRs, hardware range check), padding entries are
// gathered like real ones, the step registers ping-pong (no copies) and the first MFMA of a chain
// takes a literal zero accumulator.  fp32 MFMA shares the vector datapath with the VALU on gfx950
// (ablation: kernel time = MFMA time + everything-else time), so every VALU instruction removed
// from the loop is time given back to the matrix pipe.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct GStep { u32x4 a0, a1, b0, b1; };

__device__ inline float bcf(unsigned int v) { return __builtin_bit_cast(float, v); }

template <int NBW, bool ADJ>
__device__ inline void conv_step_mfma_buf(const GStep &q, __amdgpu_buffer_rsrc_t rw, unsigned lane32,
                                          unsigned soA, unsigned soB, int nvalid, f32x4 (&accA)[NBW],
                                          f32x4 (&accB)[NBW]) {
  const bool same = ADJ && soA == soB; // adjacent blocks of one offset: one set of weight fragments (wave-uniform)
  // all of the step's weight loads go out before its first MFMA: the L2 latency is paid once per
  // step, not once per column block (the last column slab may hold fewer than NBW blocks)
  u32x4 w0[NBW], w1[NBW], u0[NBW], u1[NBW];
#pragma unroll
  for (int j = 0; j < NBW; ++j) {
    if (j < nvalid) {
      w0[j] = __builtin_amdgcn_raw_buffer_load_b128(rw, lane32, soA + j * 2048, 0);
      w1[j] = __builtin_amdgcn_raw_buffer_load_b128(rw, lane32 + 16, soA + j * 2048, 0);
      if (!same) {
        u0[j] = __builtin_amdgcn_raw_buffer_load_b128(rw, lane32, soB + j * 2048, 0);
        u1[j] = __builtin_amdgcn_raw_buffer_load_b128(rw, lane32 + 16, soB + j * 2048, 0);
      }
    }
  }
  if (same) {
#pragma unroll
    for (int j = 0; j < NBW; ++j)
      if (j < nvalid) { u0[j] = w0[j]; u1[j] = w1[j]; }
  }
#pragma unroll
  for (int j = 0; j < NBW; ++j) {
    if (j < nvalid) {
      f32x4 ca = accA[j], cb = accB[j];
      ca = __builtin_amdgcn_mfma_f32_16x16x4f32(bcf(w0[j][0]), bcf(q.a0[0]), ca, 0, 0, 0);
      cb = __builtin_amdgcn_mfma_f32_16x16x4f32(bcf(u0[j][0]), bcf(q.b0[0]), cb, 0, 0, 0);
      ca = __builtin_amdgcn_mfma_f32_16x16x4f32(bcf(w0[j][1]), bcf(q.a0[1]), ca, 0, 0, 0);
      cb = __builtin_amdgcn_mfma_f32_16x16x4f32(bcf(u0[j][1]), bcf(q.b0[1]), cb, 0, 0, 0);
      ca = __builtin_amdgcn_mfma_f32_16x16x4f32(bcf(w0[j][2]), bcf(q.a0[2]), ca, 0, 0, 0);
      cb = __builtin_amdgcn_mfma_f32_16x16x4f32(bcf(u0[j][2]), bcf(q.b0[2]), cb, 0, 0, 0);
      ca = __builtin_amdgcn_mfma_f32_16x16x4f32(bcf(w0[j][3]), bcf(q.a0[3]), ca, 0, 0, 0);
      cb = __builtin_amdgcn_mfma_f32_16x16x4f32(bcf(u0[j][3]), bcf(q.b0[3]), cb, 0, 0, 0);
      ca = __builtin_amdgcn_mfma_f32_16x16x4f32(bcf(w1[j][0]), bcf(q.a1[0]), ca, 0, 0, 0);
      cb = __builtin_amdgcn_mfma_f32_16x16x4f32(bcf(u1[j][0]), bcf(q.b1[0]), cb, 0, 0, 0);
      ca = __builtin_amdgcn_mfma_f32_16x16x4f32(bcf(w1[j][1]), bcf(q.a1[1]), ca, 0, 0, 0);
      cb = __builtin_amdgcn_mfma_f32_16x16x4f32(bcf(u1[j][1]), bcf(q.b1[1]), cb, 0, 0, 0);
      ca = __builtin_amdgcn_mfma_f32_16x16x4f32(bcf(w1[j][2]), bcf(q.a1[2]), ca, 0, 0, 0);
      cb = __builtin_amdgcn_mfma_f32_16x16x4f32(bcf(u1[j][2]), bcf(q.b1[2]), cb, 0, 0, 0);
      ca = __builtin_amdgcn_mfma_f32_16x16x4f32(bcf(w1[j][3]), bcf(q.a1[3]), ca, 0, 0, 0);
      cb = __builtin_amdgcn_mfma_f32_16x16x4f32(bcf(u1[j][3]), bcf(q.b1[3]), cb, 0, 0, 0);
      accA[j] = ca; accB[j] = cb;
    }
  }
}

template <int NBW, int WPB, bool ADJ, bool ALIGNED>
__global__ __launch_bounds__(WPB * 64, (NBW <= 2 ? 3 : 2)) void k_conv_blocks_mfma_buf(
    const float *__restrict__ in, int ci, int64_t in_bytes, float *__restrict__ out, int co, int64_t V_out,
    const int32_t *__restrict__ words, int64_t words_bytes, int vol, int wflip, const float *__restrict__ Wp,
    int64_t wp_bytes, const float *__restrict__ bias) {
  constexpr int WS = NBW * 16;              // C-tile row stride (floats)
  constexpr int TILE = 64 * WS;             // floats per wave
  extern __shared__ __align__(16) float smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int g = lane >> 4, c16 = lane & 15;
  float *Ct = smem + (size_t)wave * TILE;
  const int maxb = tb_maxb(vol);
  const int nkc = ALIGNED ? (ci >> 5) : nkc_of(ci), nnb = nnb_of(co);
  const int nb0 = blockIdx.y * NBW;
  const int64_t tile = blockIdx.x, row0 = tile * 64;
  const int64_t ntiles = tb_ntiles(V_out);
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(in), 0, (int)in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(Wp), 0, (int)wp_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rwords =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t *>(words), 0, (int)words_bytes, 0x00020000);
  const int nblk = words[tile];
  const unsigned kbase = (unsigned)((ntiles + tile * maxb) * 4);                       // byte offset of blk_k
  const unsigned ebase = (unsigned)((ntiles + ntiles * maxb + tile * maxb * 16) * 4);  // byte offset of ent
  const unsigned rowbytes = (unsigned)ci * 4u, g32 = (unsigned)g * 32u, lane32 = (unsigned)lane * 32u;
  const unsigned wk_bytes = (unsigned)nkc * (unsigned)nnb * 2048u;                     // packed bytes per offset
  const unsigned c16x4 = (unsigned)c16 * 4u;
  const int nvalid = (nnb - nb0) < NBW ? (nnb - nb0) : NBW;

  for (int i = lane; i < TILE; i += 64) Ct[i] = 0.0f;

  // ADJ (throughput-bound launches): pairs of adjacent blocks dealt round-robin, same-offset pairs share
  // their weight fragments; else blocks dealt round-robin and a wave pairs its own consecutive ones
  const int nmine = wave < nblk ? (nblk - wave + WPB - 1) / WPB : 0;
  const int tpairs = (nblk + 1) >> 1;
  const int npairs = ADJ ? (wave < tpairs ? (tpairs - wave + WPB - 1) / WPB : 0) : ((nmine + 1) >> 1);
  auto load_pair = [&](int pr) {
    PairEnt p;
    int bA = ADJ ? 2 * (wave + pr * WPB) : wave + (2 * pr) * WPB;
    if (bA >= nblk) bA = nblk > 0 ? (ADJ ? ((nblk - 1) & ~1) : nblk - 1) : 0; // past the end: harmless re-read
    int bB = bA + (ADJ ? 1 : WPB);
    const bool hasB = bB < nblk;
    if (!hasB) bB = bA;
    p.eA = (int)__builtin_amdgcn_raw_buffer_load_b32(rwords, c16x4, ebase + (unsigned)bA * 64u, 0);
    p.eB = (int)__builtin_amdgcn_raw_buffer_load_b32(rwords, c16x4, ebase + (unsigned)bB * 64u, 0);
    if (!hasB) p.eB |= (int)0x80000000;
    p.kA = (int)__builtin_amdgcn_raw_buffer_load_b32(rwords, 0u, kbase + (unsigned)bA * 4u, 0);
    p.kB = (int)__builtin_amdgcn_raw_buffer_load_b32(rwords, 0u, kbase + (unsigned)bB * 4u, 0);
    return p;
  };
  auto gather = [&](GStep &q, const PairEnt &p, int kc) {
    const unsigned va = (((unsigned)p.eA & 0x7fffffffu) >> 6) * rowbytes + g32;
    const unsigned vb = (((unsigned)p.eB & 0x7fffffffu) >> 6) * rowbytes + g32;
    const unsigned so = (unsigned)kc * 128u;
    if (ALIGNED) {
      q.a0 = __builtin_amdgcn_raw_buffer_load_b128(rin, va, so, 0);
      q.a1 = __builtin_amdgcn_raw_buffer_load_b128(rin, va + 16u, so, 0);
      q.b0 = __builtin_amdgcn_raw_buffer_load_b128(rin, vb, so, 0);
      q.b1 = __builtin_amdgcn_raw_buffer_load_b128(rin, vb + 16u, so, 0);
    } else {
      // any plane count (the 9-plane first layer): element loads, planes past ci read as zero (their packed
      // weights are zero too); rows are ci*4 bytes apart, not 16-byte aligned
      const int c0 = kc * kKC + g * 8;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const bool ok0 = c0 + t < ci, ok1 = c0 + 4 + t < ci;
        q.a0[t] = ok0 ? __builtin_amdgcn_raw_buffer_load_b32(rin, va + 4u * t, so, 0) : 0u;
        q.a1[t] = ok1 ? __builtin_amdgcn_raw_buffer_load_b32(rin, va + 16u + 4u * t, so, 0) : 0u;
        q.b0[t] = ok0 ? __builtin_amdgcn_raw_buffer_load_b32(rin, vb + 4u * t, so, 0) : 0u;
        q.b1[t] = ok1 ? __builtin_amdgcn_raw_buffer_load_b32(rin, vb + 16u + 4u * t, so, 0) : 0u;
      }
    }
  };
  PairEnt p0 = load_pair(0), p1 = load_pair(1), p2 = load_pair(2);
  GStep s0, s1;
  if (npairs > 0) gather(s0, p0, 0);
  f32x4 accA[NBW], accB[NBW];
  int step = 0; // parity selects the ping-pong register set
  for (int pr = 0; pr < npairs; ++pr) {
    int kA = __builtin_amdgcn_readfirstlane(p0.kA), kB = __builtin_amdgcn_readfirstlane(p0.kB);
    if (wflip) { kA = vol - 1 - kA; kB = vol - 1 - kB; }
    const unsigned soA0 = (unsigned)kA * wk_bytes + (unsigned)nb0 * 2048u;
    const unsigned soB0 = (unsigned)kB * wk_bytes + (unsigned)nb0 * 2048u;
#pragma unroll
    for (int j = 0; j < NBW; ++j) {
      accA[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      accB[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    for (int kc = 0; kc < nkc; ++kc, ++step) {
      const bool more = (kc + 1 < nkc) || (pr + 1 < npairs);
      const PairEnt &pn = (kc + 1 < nkc) ? p0 : p1;
      const int kcn = (kc + 1 < nkc) ? kc + 1 : 0;
      const unsigned so = (unsigned)kc * (unsigned)nnb * 2048u;
      if (step & 1) {
        if (more) gather(s0, pn, kcn);
        conv_step_mfma_buf<NBW, ADJ>(s1, rw, lane32, soA0 + so, soB0 + so, nvalid, accA, accB);
      } else {
        if (more) gather(s1, pn, kcn);
        conv_step_mfma_buf<NBW, ADJ>(s0, rw, lane32, soA0 + so, soB0 + so, nvalid, accA, accB);
      }
    }
    conv_block_accumulate<NBW, WS>(Ct, p0.eA, g, accA);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // A and B may hit the same row
    conv_block_accumulate<NBW, WS>(Ct, p0.eB, g, accB);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    p0 = p1; p1 = p2; p2 = load_pair(pr + 3);
  }
  __syncthreads();
  const int nrows = (int)((V_out - row0) < 64 ? (V_out - row0) : 64);
  const int wcols = (co - nb0 * 16) < NBW * 16 ? (co - nb0 * 16) : NBW * 16;
  if ((co & 3) == 0) {
    const int q = wcols >> 2;
    for (int i = threadIdx.x; i < nrows * q; i += WPB * 64) {
      int r = i / q, cq = i % q;
      float4 v = *reinterpret_cast<const float4 *>(smem + r * WS + cq * 4);
#pragma unroll
      for (int w = 1; w < WPB; ++w) {
        float4 u = *reinterpret_cast<const float4 *>(smem + (size_t)w * TILE + r * WS + cq * 4);
        v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
      }
      if (bias) {
        const float *bb = bias + nb0 * 16 + cq * 4;
        v.x += bb[0]; v.y += bb[1]; v.z += bb[2]; v.w += bb[3];
      }
      *reinterpret_cast<float4 *>(out + (row0 + r) * co + nb0 * 16 + cq * 4) = v;
    }
  } else {
    for (int i = threadIdx.x; i < nrows * wcols; i += WPB * 64) {
      int r = i / wcols, cc = i % wcols;
      float v = smem[r * WS + cc];
#pragma unroll
      for (int w = 1; w < WPB; ++w) v += smem[(size_t)w * TILE + r * WS + cc];
      if (bias) v += bias[nb0 * 16 + cc];
      out[(row0 + r) * co + nb0 * 16 + cc] = v;
    }
  }
}

// packed weights of JG column blocks for the two blocks (A, B) of a step: JG x 4 x 16 B per lane
template <int JG> struct WSet { u32x4 a0[JG], a1[JG], b0[JG], b1[JG]; };

template <int JG>
__device__ inline void conv_load_w(WSet<JG> &w, __amdgpu_buffer_rsrc_t rw, unsigned lane32, unsigned soA,
                                   unsigned soB, int nvalid) {
#pragma unroll
  for (int j = 0; j < JG; ++j) {
    if (j < nvalid) { // last column slab may hold fewer blocks (wave-uniform)
      w.a0[j] = __builtin_amdgcn_raw_buffer_load_b128(rw, lane32, soA + j * 2048, 0);
      w.a1[j] = __builtin_amdgcn_raw_buffer_load_b128(rw, lane32 + 16, soA + j * 2048, 0);
      w.b0[j] = __builtin_amdgcn_raw_buffer_load_b128(rw, lane32, soB + j * 2048, 0);
      w.b1[j] = __builtin_amdgcn_raw_buffer_load_b128(rw, lane32 + 16, soB + j * 2048, 0);
    }
  }
}

template <int JG>
__device__ inline void conv_step_mfma_w(const GStep &q, const WSet<JG> &w, int nvalid, f32x4 *accA, f32x4 *accB) {
#pragma unroll
  for (int j = 0; j < JG; ++j) {
    if (j < nvalid) {
      f32x4 ca = accA[j], cb = accB[j];
      ca = __builtin_amdgcn_mfma_f32_16x16x4f32(bcf(w.a0[j][0]), bcf(q.a0[0]), ca, 0, 0, 0);
      cb = __builtin_amdgcn_mfma_f32_16x16x4f32(bcf(w.b0[j][0]), bcf(q.b0[0]), cb, 0, 0, 0);
      ca = __builtin_amdgcn_mfma_f32_16x16x4f32(bcf(w.a0[j][1]), bcf(q.a0[1]), ca, 0, 0, 0);
      cb = __builtin_amdgcn_mfma_f32_16x16x4f32(bcf(w.b0[j][1]), bcf(q.b0[1]), cb, 0, 0, 0);
      ca = __builtin_amdgcn_mfma_f32_16x16x4f32(bcf(w.a0[j][2]), bcf(q.a0[2]), ca, 0, 0, 0);
      cb = __builtin_amdgcn_mfma_f32_16x16x4f32(bcf(w.b0[j][2]), bcf(q.b0[2]), cb, 0, 0, 0);
      ca = __builtin_amdgcn_mfma_f32_16x16x4f32(bcf(w.a0[j][3]), bcf(q.a0[3]), ca, 0, 0, 0);
      cb = __builtin_amdgcn_mfma_f32_16x16x4f32(bcf(w.b0[j][3]), bcf(q.b0[3]), cb, 0, 0, 0);
      ca = __builtin_amdgcn_mfma_f32_16x16x4f32(bcf(w.a1[j][0]), bcf(q.a1[0]), ca, 0, 0, 0);
      cb = __builtin_amdgcn_mfma_f32_16x16x4f32(bcf(w.b1[j][0]), bcf(q.b1[0]), cb, 0, 0, 0);
      ca = __builtin_amdgcn_mfma_f32_16x16x4f32(bcf(w.a1[j][1]), bcf(q.a1[1]), ca, 0, 0, 0);
      cb = __builtin_amdgcn_mfma_f32_16x16x4f32(bcf(w.b1[j][1]), bcf(q.b1[1]), cb, 0, 0, 0);
      ca = __builtin_amdgcn_mfma_f32_16x16x4f32(bcf(w.a1[j][2]), bcf(q.a1[2]), ca, 0, 0, 0);
      cb = __builtin_amdgcn_mfma_f32_16x16x4f32(bcf(w.b1[j][2]), bcf(q.b1[2]), cb, 0, 0, 0);
      ca = __builtin_amdgcn_mfma_f32_16x16x4f32(bcf(w.a1[j][3]), bcf(q.a1[3]), ca, 0, 0, 0);
      cb = __builtin_amdgcn_mfma_f32_16x16x4f32(bcf(w.b1[j][3]), bcf(q.b1[3]), cb, 0, 0, 0);
      accA[j] = ca; accB[j] = cb;
    }
  }
}

// the two blocks of a pair are ADJACENT in the tile's (offset-sorted) list in the weight-prefetching
// kernel, so they often share the filter offset (the centre offset alone fills four blocks of a 64-row
// tile): then one set of weight fragments feeds both MFMA chains and the second fetch is skipped
template <int JG>
__device__ inline void conv_load_w_shared(WSet<JG> &w, __amdgpu_buffer_rsrc_t rw, unsigned lane32, unsigned soA,
                                          unsigned soB, int nvalid) {
#pragma unroll
  for (int j = 0; j < JG; ++j) {
    if (j < nvalid) {
      w.a0[j] = __builtin_amdgcn_raw_buffer_load_b128(rw, lane32, soA + j * 2048, 0);
      w.a1[j] = __builtin_amdgcn_raw_buffer_load_b128(rw, lane32 + 16, soA + j * 2048, 0);
      if (soA != soB) { // wave-uniform
        w.b0[j] = __builtin_amdgcn_raw_buffer_load_b128(rw, lane32, soB + j * 2048, 0);
        w.b1[j] = __builtin_amdgcn_raw_buffer_load_b128(rw, lane32 + 16, soB + j * 2048, 0);
      }
    }
  }
}

template <int JG, bool ADJ>
__device__ inline void conv_load_w_sel(WSet<JG> &w, __amdgpu_buffer_rsrc_t rw, unsigned lane32, unsigned soA,
                                       unsigned soB, int nvalid) {
  if (ADJ) conv_load_w_shared<JG>(w, rw, lane32, soA, soB, nvalid);
  else conv_load_w<JG>(w, rw, lane32, soA, soB, nvalid);
}

#define AABR_MFMA8(ACC, W0, W1, X0, X1)                                            \
  ACC = __builtin_amdgcn_mfma_f32_16x16x4f32(bcf(W0[0]), bcf(X0[0]), ACC, 0, 0, 0); \
  ACC = __builtin_amdgcn_mfma_f32_16x16x4f32(bcf(W0[1]), bcf(X0[1]), ACC, 0, 0, 0); \
  ACC = __builtin_amdgcn_mfma_f32_16x16x4f32(bcf(W0[2]), bcf(X0[2]), ACC, 0, 0, 0); \
  ACC = __builtin_amdgcn_mfma_f32_16x16x4f32(bcf(W0[3]), bcf(X0[3]), ACC, 0, 0, 0); \
  ACC = __builtin_amdgcn_mfma_f32_16x16x4f32(bcf(W1[0]), bcf(X1[0]), ACC, 0, 0, 0); \
  ACC = __builtin_amdgcn_mfma_f32_16x16x4f32(bcf(W1[1]), bcf(X1[1]), ACC, 0, 0, 0); \
  ACC = __builtin_amdgcn_mfma_f32_16x16x4f32(bcf(W1[2]), bcf(X1[2]), ACC, 0, 0, 0); \
  ACC = __builtin_amdgcn_mfma_f32_16x16x4f32(bcf(W1[3]), bcf(X1[3]), ACC, 0, 0, 0)

template <int JG>
__device__ inline void conv_step_mfma_w_shared(const GStep &q, const WSet<JG> &w, bool same, int nvalid, f32x4 *accA,
                                               f32x4 *accB) {
  if (!same) {
    conv_step_mfma_w<JG>(q, w, nvalid, accA, accB);
    return;
  }
#pragma unroll
  for (int j = 0; j < JG; ++j) {
    if (j < nvalid) {
      f32x4 ca = accA[j], cb = accB[j];
      // same weights for both blocks: the two chains stay interleaved through the scheduler
      AABR_MFMA8(ca, w.a0[j], w.a1[j], q.a0, q.a1);
      AABR_MFMA8(cb, w.a0[j], w.a1[j], q.b0, q.b1);
      accA[j] = ca; accB[j] = cb;
    }
  }
}

// weights streamed right before their MFMAs, one column block at a time (16 VGPRs live)
template <int NBW>
__device__ inline void conv_step_mfma_stream(const GStep &q, __amdgpu_buffer_rsrc_t rw, unsigned lane32,
                                             unsigned soA, unsigned soB, int nvalid, f32x4 *accA, f32x4 *accB) {
#pragma unroll
  for (int j = 0; j < NBW; ++j) {
    if (j >= nvalid) break; // last column slab may hold fewer than NBW blocks (wave-uniform)
    WSet<1> w;
    conv_load_w<1>(w, rw, lane32, soA + j * 2048, soB + j * 2048, 1);
    conv_step_mfma_w<1>(q, w, 1, accA + j, accB + j);
  }
}

// Three register-resident prefetch streams per wave: block entries two pairs ahead, gathered rows one
// step ahead, packed weights one micro-step (JG column blocks) ahead -- no load is waited on in the
// step that issues it.
// WPIPE: prefetch the packed weights one micro-step ahead (pays for wide slabs, NBW = 4; with narrow
// slabs the extra 32-64 VGPRs cost a wave of occupancy and the step is too short to need it).
// ADJ: pairs are ADJACENT blocks of the tile's list and share their weight fragments when they share the
// offset (less L2->CU traffic: pays when the launch is throughput-bound, i.e. thousands of workgroups;
// on small launches the longer loop body costs more than it saves)
template <int NBW, int WPB, bool WPIPE, bool ADJ>
__global__ __launch_bounds__(WPB * 64, 2) void k_conv_blocks_mfma_wpipe(
    const float *__restrict__ in, int ci, int64_t in_bytes, float *__restrict__ out, int co, int64_t V_out,
    const int32_t *__restrict__ words, int64_t words_bytes, int vol, int wflip, const float *__restrict__ Wp,
    int64_t wp_bytes, const float *__restrict__ bias) {
  constexpr int WS = NBW * 16;              // C-tile row stride (floats)
  constexpr int TILE = 64 * WS;             // floats per wave
  constexpr int JG = NBW >= 2 ? 2 : 1;      // column blocks per weight set
  constexpr int NJG = NBW / JG;             // weight sets (micro-steps) per step
  extern __shared__ __align__(16) float smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int g = lane >> 4, c16 = lane & 15;
  float *Ct = smem + (size_t)wave * TILE;
  const int maxb = tb_maxb(vol);
  const int nkc = ci >> 5, nnb = nnb_of(co);
  const int nb0 = blockIdx.y * NBW;
  const int64_t tile = blockIdx.x, row0 = tile * 64;
  const int64_t ntiles = tb_ntiles(V_out);
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(in), 0, (int)in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(Wp), 0, (int)wp_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rwords =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t *>(words), 0, (int)words_bytes, 0x00020000);
  const int nblk = words[tile];
  const unsigned kbase = (unsigned)((ntiles + tile * maxb) * 4);                       // byte offset of blk_k
  const unsigned ebase = (unsigned)((ntiles + ntiles * maxb + tile * maxb * 16) * 4);  // byte offset of ent
  const unsigned rowbytes = (unsigned)ci * 4u, g32 = (unsigned)g * 32u, lane32 = (unsigned)lane * 32u;
  const unsigned wk_bytes = (unsigned)nkc * (unsigned)nnb * 2048u;                     // packed bytes per offset
  const unsigned kc_bytes = (unsigned)nnb * 2048u;                                     // packed bytes per chunk
  const unsigned c16x4 = (unsigned)c16 * 4u;
  const int nvalid = (nnb - nb0) < NBW ? (nnb - nb0) : NBW;

  for (int i = lane; i < TILE; i += 64) Ct[i] = 0.0f;

  // ADJ: pairs of adjacent blocks (2j, 2j+1) dealt to the waves round-robin; else blocks dealt round-robin
  // and a wave pairs its own consecutive ones
  const int tpairs = (nblk + 1) >> 1;
  const int nmine = wave < nblk ? (nblk - wave + WPB - 1) / WPB : 0;
  const int npairs = ADJ ? (wave < tpairs ? (tpairs - wave + WPB - 1) / WPB : 0) : ((nmine + 1) >> 1);
  auto load_pair = [&](int pr) {
    PairEnt p;
    int bA = ADJ ? 2 * (wave + pr * WPB) : wave + (2 * pr) * WPB;
    if (bA >= nblk) bA = nblk > 0 ? (ADJ ? ((nblk - 1) & ~1) : nblk - 1) : 0; // past the end: harmless re-read
    int bB = bA + (ADJ ? 1 : WPB);
    const bool hasB = bB < nblk;
    if (!hasB) bB = bA;
    p.eA = (int)__builtin_amdgcn_raw_buffer_load_b32(rwords, c16x4, ebase + (unsigned)bA * 64u, 0);
    p.eB = (int)__builtin_amdgcn_raw_buffer_load_b32(rwords, c16x4, ebase + (unsigned)bB * 64u, 0);
    if (!hasB) p.eB |= (int)0x80000000;
    p.kA = (int)__builtin_amdgcn_raw_buffer_load_b32(rwords, 0u, kbase + (unsigned)bA * 4u, 0);
    p.kB = (int)__builtin_amdgcn_raw_buffer_load_b32(rwords, 0u, kbase + (unsigned)bB * 4u, 0);
    return p;
  };
  auto gather = [&](GStep &q, const PairEnt &p, int kc) {
    const unsigned va = (((unsigned)p.eA & 0x7fffffffu) >> 6) * rowbytes + g32;
    const unsigned vb = (((unsigned)p.eB & 0x7fffffffu) >> 6) * rowbytes + g32;
    const unsigned so = (unsigned)kc * 128u;
    q.a0 = __builtin_amdgcn_raw_buffer_load_b128(rin, va, so, 0);
    q.a1 = __builtin_amdgcn_raw_buffer_load_b128(rin, va + 16u, so, 0);
    q.b0 = __builtin_amdgcn_raw_buffer_load_b128(rin, vb, so, 0);
    q.b1 = __builtin_amdgcn_raw_buffer_load_b128(rin, vb + 16u, so, 0);
  };
  // byte offsets (wave-uniform) of a pair's packed weights for this column slab
  auto w_base = [&](int k) {
    if (wflip) k = vol - 1 - k;
    return (unsigned)k * wk_bytes + (unsigned)nb0 * 2048u;
  };
  // block entries run three pairs ahead: a pair's offsets (wave-uniform, needed one step early for the
  // weight prefetch) have been in flight for two pairs when they are first read, so waiting for them
  // does not drain the younger gathers behind them in the in-order VMEM queue
  PairEnt p0 = load_pair(0), p1 = load_pair(1), p2 = load_pair(2), p3 = load_pair(3);
  GStep s0, s1;
  WSet<JG> w0, w1;
  f32x4 accA[NBW], accB[NBW];
  unsigned soA = w_base(__builtin_amdgcn_readfirstlane(p0.kA)), soB = w_base(__builtin_amdgcn_readfirstlane(p0.kB));
  unsigned soA1 = 0, soB1 = 0;
  if (WPIPE) {
    soA1 = w_base(__builtin_amdgcn_readfirstlane(p1.kA));
    soB1 = w_base(__builtin_amdgcn_readfirstlane(p1.kB));
  }
  if (npairs > 0) {
    gather(s0, p0, 0);
    if (WPIPE) conv_load_w_sel<JG, ADJ>(w0, rw, lane32, soA, soB, nvalid);
  }
#pragma unroll
  for (int j = 0; j < NBW; ++j) {
    accA[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    accB[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  int pr = 0, kc = 0;
  // one step = one 32-channel chunk of one pair of blocks; the register sets a step reads and the
  // ones it prefetches into are fixed by the call site (the loop below is unrolled by two), so the
  // ping-pong costs no register moves
  auto do_step = [&](const GStep &xc, GStep &xn, WSet<JG> &wc, WSet<JG> &wo) {
    // loop state is wave-uniform; say so (keeps the buffer soffsets in SGPRs, no waterfall loops)
    kc = __builtin_amdgcn_readfirstlane(kc); pr = __builtin_amdgcn_readfirstlane(pr);
    soA = __builtin_amdgcn_readfirstlane(soA); soB = __builtin_amdgcn_readfirstlane(soB);
    const bool last_kc = kc + 1 == nkc;
    const bool more = !last_kc || (pr + 1 < npairs);
    const PairEnt &pn = last_kc ? p1 : p0;
    const int kcn = last_kc ? 0 : kc + 1;
    if (more) gather(xn, pn, kcn);
    const unsigned cA = soA + (unsigned)kc * kc_bytes, cB = soB + (unsigned)kc * kc_bytes;
    if (!WPIPE) {
      // weights stream from L2 right before their MFMAs (other waves cover the latency)
      conv_step_mfma_stream<NBW>(xc, rw, lane32, cA, cB, nvalid, accA, accB);
    } else {
      soA1 = __builtin_amdgcn_readfirstlane(soA1); soB1 = __builtin_amdgcn_readfirstlane(soB1);
      const unsigned nA = last_kc ? soA1 : cA + kc_bytes; // first weight set of the next step
      const unsigned nB = last_kc ? soB1 : cB + kc_bytes;
      const bool same = ADJ && soA == soB; // both blocks of the pair use the same filter offset (wave-uniform)
      if (NJG == 1) {
        if (more) conv_load_w_sel<JG, ADJ>(wo, rw, lane32, nA, nB, nvalid);
        conv_step_mfma_w_shared<JG>(xc, wc, same, nvalid, accA, accB);
      } else {
        conv_load_w_sel<JG, ADJ>(wo, rw, lane32, cA + JG * 2048u, cB + JG * 2048u, nvalid - JG);
        conv_step_mfma_w_shared<JG>(xc, wc, same, nvalid, accA, accB);
        if (more) conv_load_w_sel<JG, ADJ>(wc, rw, lane32, nA, nB, nvalid);
        conv_step_mfma_w_shared<JG>(xc, wo, same, nvalid - JG, accA + JG, accB + JG);
      }
    }
    if (last_kc) {
      conv_block_accumulate<NBW, WS>(Ct, p0.eA, g, accA);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // A and B may hit the same row
      conv_block_accumulate<NBW, WS>(Ct, p0.eB, g, accB);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
      for (int j = 0; j < NBW; ++j) {
        accA[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        accB[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
      p0 = p1; p1 = p2; p2 = p3; p3 = load_pair(pr + 4);
      if (WPIPE) {
        soA = soA1; soB = soB1;
        soA1 = w_base(__builtin_amdgcn_readfirstlane(p1.kA));
        soB1 = w_base(__builtin_amdgcn_readfirstlane(p1.kB));
      } else {
        soA = w_base(__builtin_amdgcn_readfirstlane(p0.kA));
        soB = w_base(__builtin_amdgcn_readfirstlane(p0.kB));
      }
      kc = 0; ++pr;
    } else {
      ++kc;
    }
  };
  while (pr < npairs) {
    if (NJG == 1) {
      do_step(s0, s1, w0, w1);
      if (pr >= npairs) break;
      do_step(s1, s0, w1, w0);
    } else {
      do_step(s0, s1, w0, w1);
      if (pr >= npairs) break;
      do_step(s1, s0, w0, w1);
    }
  }
  __syncthreads();
  const int nrows = (int)((V_out - row0) < 64 ? (V_out - row0) : 64);
  const int wcols = (co - nb0 * 16) < NBW * 16 ? (co - nb0 * 16) : NBW * 16;
  if ((co & 3) == 0) {
    const int q = wcols >> 2;
    for (int i = threadIdx.x; i < nrows * q; i += WPB * 64) {
      int r = i / q, cq = i % q;
      float4 v = *reinterpret_cast<const float4 *>(smem + r * WS + cq * 4);
#pragma unroll
      for (int w = 1; w < WPB; ++w) {
        float4 u = *reinterpret_cast<const float4 *>(smem + (size_t)w * TILE + r * WS + cq * 4);
        v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
      }
      if (bias) {
        const float *bb = bias + nb0 * 16 + cq * 4;
        v.x += bb[0]; v.y += bb[1]; v.z += bb[2]; v.w += bb[3];
      }
      *reinterpret_cast<float4 *>(out + (row0 + r) * co + nb0 * 16 + cq * 4) = v;
    }
  } else {
    for (int i = threadIdx.x; i < nrows * wcols; i += WPB * 64) {
      int r = i / wcols, cc = i % wcols;
      float v = smem[r * WS + cc];
#pragma unroll
      for (int w = 1; w < WPB; ++w) v += smem[(size_t)w * TILE + r * WS + cc];
      if (bias) v += bias[nb0 * 16 + cc];
      out[(row0 + r) * co + nb0 * 16 + cc] = v;
    }
  }
}

// ------------------------------------------------------------------ tiny rule books
// The coarse FPN scales have 1-50 tiles and 128-256 planes: a launch of the kernels above is a handful of
// workgroups walking 30-40 dependent (pair, channel-chunk) steps each (60 us for ~50 M MACs).  This variant
// cuts the work finer: 16 output columns per workgroup, 8 waves, and the unit dealt to a wave is one
// 32-channel chunk of one block pair, so a tile's ~15 pairs x 4-8 chunks spread over 8 waves x (co/16)
// workgroups.  Every item ends with its own LDS accumulate (private tile per wave, summed in wave order:
// still deterministic).  Gathers and weights are prefetched one item ahead (fixed register sets).
struct ItemEnt { int eA, eB, kA, kB; };

template <int WPB>
__global__ __launch_bounds__(WPB * 64, 1) void k_conv_blocks_mfma_small(
    const float *__restrict__ in, int ci, int64_t in_bytes, float *__restrict__ out, int co, int64_t V_out,
    const int32_t *__restrict__ words, int64_t words_bytes, int vol, int wflip, const float *__restrict__ Wp,
    int64_t wp_bytes, const float *__restrict__ bias) {
  constexpr int WS = 16, TILE = 64 * WS;
  extern __shared__ __align__(16) float smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int g = lane >> 4, c16 = lane & 15;
  float *Ct = smem + (size_t)wave * TILE;
  const int maxb = tb_maxb(vol);
  const int nkc = ci >> 5, nnb = nnb_of(co);
  const int nb0 = blockIdx.y;
  const int64_t tile = blockIdx.x, row0 = tile * 64;
  const int64_t ntiles = tb_ntiles(V_out);
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(in), 0, (int)in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(Wp), 0, (int)wp_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rwords =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t *>(words), 0, (int)words_bytes, 0x00020000);
  const int nblk = words[tile];
  const unsigned kbase = (unsigned)((ntiles + tile * maxb) * 4);
  const unsigned ebase = (unsigned)((ntiles + ntiles * maxb + tile * maxb * 16) * 4);
  const unsigned rowbytes = (unsigned)ci * 4u, g32 = (unsigned)g * 32u, lane32 = (unsigned)lane * 32u;
  const unsigned wk_bytes = (unsigned)nkc * (unsigned)nnb * 2048u, kc_bytes = (unsigned)nnb * 2048u;
  const unsigned c16x4 = (unsigned)c16 * 4u;

  for (int i = lane; i < TILE; i += 64) Ct[i] = 0.0f;

  const int nitems = ((nblk + 1) >> 1) * nkc;               // (block pair, channel chunk)
  const int nmine = wave < nitems ? (nitems - wave + WPB - 1) / WPB : 0;
  auto load_item = [&](int j) {                             // this wave's j-th item
    ItemEnt p;
    int it = wave + j * WPB;
    if (it >= nitems) it = nitems > 0 ? nitems - 1 : 0;     // past the end: harmless re-read, never consumed
    int bA = 2 * (it / nkc);
    if (bA >= nblk) bA = 0;
    int bB = bA + 1;
    const bool hasB = bB < nblk;
    if (!hasB) bB = bA;
    p.eA = (int)__builtin_amdgcn_raw_buffer_load_b32(rwords, c16x4, ebase + (unsigned)bA * 64u, 0);
    p.eB = (int)__builtin_amdgcn_raw_buffer_load_b32(rwords, c16x4, ebase + (unsigned)bB * 64u, 0);
    if (!hasB) p.eB |= (int)0x80000000;
    p.kA = (int)__builtin_amdgcn_raw_buffer_load_b32(rwords, 0u, kbase + (unsigned)bA * 4u, 0);
    p.kB = (int)__builtin_amdgcn_raw_buffer_load_b32(rwords, 0u, kbase + (unsigned)bB * 4u, 0);
    return p;
  };
  auto kc_of = [&](int j) {
    int it = wave + j * WPB;
    if (it >= nitems) it = nitems > 0 ? nitems - 1 : 0;
    return it % nkc;
  };
  auto fetch = [&](GStep &q, WSet<1> &w, const ItemEnt &p, int kc) {
    const unsigned va = (((unsigned)p.eA & 0x7fffffffu) >> 6) * rowbytes + g32;
    const unsigned vb = (((unsigned)p.eB & 0x7fffffffu) >> 6) * rowbytes + g32;
    const unsigned so = (unsigned)kc * 128u;
    q.a0 = __builtin_amdgcn_raw_buffer_load_b128(rin, va, so, 0);
    q.a1 = __builtin_amdgcn_raw_buffer_load_b128(rin, va + 16u, so, 0);
    q.b0 = __builtin_amdgcn_raw_buffer_load_b128(rin, vb, so, 0);
    q.b1 = __builtin_amdgcn_raw_buffer_load_b128(rin, vb + 16u, so, 0);
    int kA = __builtin_amdgcn_readfirstlane(p.kA), kB = __builtin_amdgcn_readfirstlane(p.kB);
    if (wflip) { kA = vol - 1 - kA; kB = vol - 1 - kB; }
    const unsigned off = (unsigned)kc * kc_bytes + (unsigned)nb0 * 2048u;
    conv_load_w<1>(w, rw, lane32, (unsigned)kA * wk_bytes + off, (unsigned)kB * wk_bytes + off, 1);
  };
  auto compute = [&](const GStep &q, const WSet<1> &w, const ItemEnt &p) {
    f32x4 accA[1] = {(f32x4){0.f, 0.f, 0.f, 0.f}}, accB[1] = {(f32x4){0.f, 0.f, 0.f, 0.f}};
    conv_step_mfma_w<1>(q, w, 1, accA, accB);
    conv_block_accumulate<1, WS>(Ct, p.eA, g, accA);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // A and B may hit the same row
    conv_block_accumulate<1, WS>(Ct, p.eB, g, accB);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  };
  ItemEnt e0 = load_item(0), e1 = load_item(1), e2 = load_item(2);
  GStep s0, s1;
  WSet<1> w0, w1;
  if (nmine > 0) fetch(s0, w0, e0, kc_of(0));
  for (int j = 0; j < nmine; j += 2) {
    if (j + 1 < nmine) fetch(s1, w1, e1, kc_of(j + 1));
    compute(s0, w0, e0);
    ItemEnt e3 = load_item(j + 3);
    if (j + 1 < nmine) {
      if (j + 2 < nmine) fetch(s0, w0, e2, kc_of(j + 2));
      compute(s1, w1, e1);
    }
    ItemEnt e4 = load_item(j + 4);
    e0 = e2; e1 = e3; e2 = e4;
  }
  __syncthreads();
  const int nrows = (int)((V_out - row0) < 64 ? (V_out - row0) : 64);
  const int wcols = (co - nb0 * 16) < 16 ? (co - nb0 * 16) : 16;
  for (int i = threadIdx.x; i < nrows * wcols; i += WPB * 64) {
    const int r = i / wcols, cc = i - r * wcols;
    float v = smem[r * WS + cc];
#pragma unroll
    for (int w = 1; w < WPB; ++w) v += smem[(size_t)w * TILE + r * WS + cc];
    if (bias) v += bias[nb0 * 16 + cc];
    out[(row0 + r) * co + nb0 * 16 + cc] = v;
  }
}

// ------------------------------------------------------------------ LDS-resident weights
// For layers whose whole packed filter bank fits in LDS next to the output tiles
// (vol * ceil(ci/32) * ceil(co/16) * 2 KiB <= ~110 KiB: 3x3x3 at <=32->32 planes, the 2x2x2 strided
// layers at 32<->64) the weights are copied to LDS once per workgroup and every MFMA A-operand is a
// conflict-free ds_read_b128.  The lean kernel above streams 2 KiB of weights through the vector
// memory path per 16-pair block and chunk -- twice the bytes of the gathered rows -- and the L2->CU
// path (~70 GB/s per CU measured, MI355X_MICROARCH.md "Indexed rows") is what bounds it; here that
// path carries only the gathered rows.  One wave owns one 64-row tile end to end (all its blocks,
// all output columns): no cross-wave summation, no barrier after the weight copy, and the grid is
// persistent (waves loop over tiles).  Same block order per tile => same bits as the other kernels.
//   LDS image:  Wl[k][kc][nb][half][lane][4]   (half h holds k-slots 4h..4h+3 of the 8 per lane)
template <int NBW>
__device__ inline void conv_step_mfma_lds(const GStep &q, const float *__restrict__ WA,
                                          const float *__restrict__ WB, int lane, f32x4 (&accA)[NBW],
                                          f32x4 (&accB)[NBW]) {
#pragma unroll
  for (int j = 0; j < NBW; ++j) {
    const f32x4 w0 = *reinterpret_cast<const f32x4 *>(WA + j * 512 + lane * 4);
    const f32x4 w1 = *reinterpret_cast<const f32x4 *>(WA + j * 512 + 256 + lane * 4);
    const f32x4 u0 = *reinterpret_cast<const f32x4 *>(WB + j * 512 + lane * 4);
    const f32x4 u1 = *reinterpret_cast<const f32x4 *>(WB + j * 512 + 256 + lane * 4);
    f32x4 ca = accA[j], cb = accB[j];
    ca = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[0], bcf(q.a0[0]), ca, 0, 0, 0);
    cb = __builtin_amdgcn_mfma_f32_16x16x4f32(u0[0], bcf(q.b0[0]), cb, 0, 0, 0);
    ca = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[1], bcf(q.a0[1]), ca, 0, 0, 0);
    cb = __builtin_amdgcn_mfma_f32_16x16x4f32(u0[1], bcf(q.b0[1]), cb, 0, 0, 0);
    ca = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[2], bcf(q.a0[2]), ca, 0, 0, 0);
    cb = __builtin_amdgcn_mfma_f32_16x16x4f32(u0[2], bcf(q.b0[2]), cb, 0, 0, 0);
    ca = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[3], bcf(q.a0[3]), ca, 0, 0, 0);
    cb = __builtin_amdgcn_mfma_f32_16x16x4f32(u0[3], bcf(q.b0[3]), cb, 0, 0, 0);
    ca = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[0], bcf(q.a1[0]), ca, 0, 0, 0);
    cb = __builtin_amdgcn_mfma_f32_16x16x4f32(u1[0], bcf(q.b1[0]), cb, 0, 0, 0);
    ca = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[1], bcf(q.a1[1]), ca, 0, 0, 0);
    cb = __builtin_amdgcn_mfma_f32_16x16x4f32(u1[1], bcf(q.b1[1]), cb, 0, 0, 0);
    ca = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[2], bcf(q.a1[2]), ca, 0, 0, 0);
    cb = __builtin_amdgcn_mfma_f32_16x16x4f32(u1[2], bcf(q.b1[2]), cb, 0, 0, 0);
    ca = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[3], bcf(q.a1[3]), ca, 0, 0, 0);
    cb = __builtin_amdgcn_mfma_f32_16x16x4f32(u1[3], bcf(q.b1[3]), cb, 0, 0, 0);
    accA[j] = ca; accB[j] = cb;
  }
}

// NBW = column blocks per workgroup slab (blockIdx.y selects the slab), NKC = ceil(ci/32); the
// number of waves per workgroup is the launch's block size / 64.  Gathers run RING-1 pairs ahead of
// the MFMAs and block entries 2*RING-1 pairs ahead (registers, static ring indices through an
// unrolled loop): with one or two waves per SIMD the latency has to be hidden inside the wave.
template <int NKC> struct WStage { u32x4 a0[NKC], a1[NKC], b0[NKC], b1[NKC]; };

template <int NBW, int NKC, bool ALIGNED>
__global__ __launch_bounds__(512, 1) void k_conv_blocks_mfma_wlds(
    const float *__restrict__ in, int ci, int64_t in_bytes, float *__restrict__ out, int co, int64_t V_out,
    const int32_t *__restrict__ words, int64_t words_bytes, int vol, int wflip, const float *__restrict__ Wp,
    const float *__restrict__ bias) {
  constexpr int WS = NBW * 16;
  constexpr int TILE = 64 * WS;
  constexpr int RING = 4;
  extern __shared__ __align__(16) float smem[];
  const int NW = blockDim.x >> 6;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int g = lane >> 4, c16 = lane & 15;
  const int nnb = nnb_of(co);
  const int nb0 = blockIdx.y * NBW;
  const int nvalid = (nnb - nb0) < NBW ? (nnb - nb0) : NBW;
  const int wfloats = vol * NKC * NBW * 512;
  float *Wl = smem;
  float *Ct = smem + wfloats + (size_t)wave * TILE;
  // ---- weight copy: global [k][kc][nb][lane][8] -> LDS [k][kc][j = nb - nb0][half][lane][4]
  for (int i = threadIdx.x; i < vol * NKC * NBW * 128; i += blockDim.x) {
    const int f = i >> 7, r = i & 127, ln = r >> 1, h = r & 1;
    const int kk = f / NBW, j = f - kk * NBW; // kk = k*NKC + kc
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (j < nvalid) v = *reinterpret_cast<const float4 *>(Wp + ((size_t)(kk * nnb + nb0 + j) * 128 + r) * 4);
    *reinterpret_cast<float4 *>(Wl + ((kk * NBW + j) * 2 + h) * 256 + ln * 4) = v;
  }
  __syncthreads();
  const int maxb = tb_maxb(vol);
  const int64_t ntiles = tb_ntiles(V_out);
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(in), 0, (int)in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rwords =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t *>(words), 0, (int)words_bytes, 0x00020000);
  const unsigned rowbytes = (unsigned)ci * 4u, g32 = (unsigned)g * 32u, c16x4 = (unsigned)c16 * 4u;
  constexpr int kstride = NKC * NBW * 512; // floats per offset in the LDS image

  for (int64_t tile = (int64_t)blockIdx.x * NW + wave; tile < ntiles; tile += (int64_t)gridDim.x * NW) {
    const int64_t row0 = tile * 64;
    for (int i = lane; i < TILE; i += 64) Ct[i] = 0.0f;
    const int nblk = __builtin_amdgcn_readfirstlane(words[tile]);
    const unsigned kbase = (unsigned)((ntiles + tile * maxb) * 4);
    const unsigned ebase = (unsigned)((ntiles + ntiles * maxb + tile * maxb * 16) * 4);
    const int npairs = (nblk + 1) >> 1;
    auto load_pair = [&](int pr) {
      PairEnt p;
      int bA = 2 * pr;
      if (bA >= nblk) bA = nblk > 0 ? nblk - 1 : 0; // past the end: harmless re-read, never consumed
      int bB = bA + 1;
      const bool hasB = bB < nblk;
      if (!hasB) bB = bA;
      p.eA = (int)__builtin_amdgcn_raw_buffer_load_b32(rwords, c16x4, ebase + (unsigned)bA * 64u, 0);
      p.eB = (int)__builtin_amdgcn_raw_buffer_load_b32(rwords, c16x4, ebase + (unsigned)bB * 64u, 0);
      if (!hasB) p.eB |= (int)0x80000000;
      p.kA = (int)__builtin_amdgcn_raw_buffer_load_b32(rwords, 0u, kbase + (unsigned)bA * 4u, 0);
      p.kB = (int)__builtin_amdgcn_raw_buffer_load_b32(rwords, 0u, kbase + (unsigned)bB * 4u, 0);
      return p;
    };
    auto gather = [&](WStage<NKC> &q, const PairEnt &p) {
      const unsigned va = (((unsigned)p.eA & 0x7fffffffu) >> 6) * rowbytes + g32;
      const unsigned vb = (((unsigned)p.eB & 0x7fffffffu) >> 6) * rowbytes + g32;
#pragma unroll
      for (int kc = 0; kc < NKC; ++kc) {
        const unsigned so = (unsigned)kc * 128u;
        if (ALIGNED) {
          q.a0[kc] = __builtin_amdgcn_raw_buffer_load_b128(rin, va, so, 0);
          q.a1[kc] = __builtin_amdgcn_raw_buffer_load_b128(rin, va + 16u, so, 0);
          q.b0[kc] = __builtin_amdgcn_raw_buffer_load_b128(rin, vb, so, 0);
          q.b1[kc] = __builtin_amdgcn_raw_buffer_load_b128(rin, vb + 16u, so, 0);
        } else {
          // any plane count: element loads, channels past ci read as zero (their weights are zero too)
          const int c0 = kc * kKC + g * 8;
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const bool ok0 = c0 + t < ci, ok1 = c0 + 4 + t < ci;
            q.a0[kc][t] = ok0 ? __builtin_amdgcn_raw_buffer_load_b32(rin, va + 4u * t, so, 0) : 0u;
            q.a1[kc][t] = ok1 ? __builtin_amdgcn_raw_buffer_load_b32(rin, va + 16u + 4u * t, so, 0) : 0u;
            q.b0[kc][t] = ok0 ? __builtin_amdgcn_raw_buffer_load_b32(rin, vb + 4u * t, so, 0) : 0u;
            q.b1[kc][t] = ok1 ? __builtin_amdgcn_raw_buffer_load_b32(rin, vb + 16u + 4u * t, so, 0) : 0u;
          }
        }
      }
    };
    auto compute = [&](const WStage<NKC> &q, const PairEnt &p) {
      int kA = __builtin_amdgcn_readfirstlane(p.kA), kB = __builtin_amdgcn_readfirstlane(p.kB);
      if (wflip) { kA = vol - 1 - kA; kB = vol - 1 - kB; }
      const float *WA0 = Wl + kA * kstride, *WB0 = Wl + kB * kstride;
      f32x4 accA[NBW], accB[NBW];
#pragma unroll
      for (int j = 0; j < NBW; ++j) {
        accA[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        accB[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int kc = 0; kc < NKC; ++kc) {
        GStep t;
        t.a0 = q.a0[kc]; t.a1 = q.a1[kc]; t.b0 = q.b0[kc]; t.b1 = q.b1[kc];
        conv_step_mfma_lds<NBW>(t, WA0 + kc * NBW * 512, WB0 + kc * NBW * 512, lane, accA, accB);
      }
      conv_block_accumulate<NBW, WS>(Ct, p.eA, g, accA);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // A and B may hit the same row
      conv_block_accumulate<NBW, WS>(Ct, p.eB, g, accB);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    };
    PairEnt p[2 * RING];
    WStage<NKC> st[RING];
#pragma unroll
    for (int u = 0; u < 2 * RING - 1; ++u) p[u] = load_pair(u);
#pragma unroll
    for (int u = 0; u < RING - 1; ++u)
      if (u < npairs) gather(st[u], p[u]);
    for (int pr0 = 0; pr0 < npairs; pr0 += 2 * RING) {
#pragma unroll
      for (int u = 0; u < 2 * RING; ++u) {
        const int pr = pr0 + u;
        if (pr < npairs) {
          p[(u + 2 * RING - 1) % (2 * RING)] = load_pair(pr + 2 * RING - 1);
          if (pr + RING - 1 < npairs) gather(st[(u + RING - 1) % RING], p[(u + RING - 1) % (2 * RING)]);
          compute(st[u % RING], p[u]);
        }
      }
    }
    // the wave's own tile goes out once (+ bias, CPU/Convolution.cpp:59-62)
    const int nrows = (int)((V_out - row0) < 64 ? (V_out - row0) : 64);
    const int wcols = (co - nb0 * 16) < NBW * 16 ? (co - nb0 * 16) : NBW * 16;
    if ((co & 3) == 0) {
      const int q = wcols >> 2;
      for (int i = lane; i < nrows * q; i += 64) {
        const int r = i / q, cq = i - r * q;
        float4 v = *reinterpret_cast<const float4 *>(Ct + r * WS + cq * 4);
        if (bias) {
          const float *bb = bias + nb0 * 16 + cq * 4;
          v.x += bb[0]; v.y += bb[1]; v.z += bb[2]; v.w += bb[3];
        }
        *reinterpret_cast<float4 *>(out + (row0 + r) * co + nb0 * 16 + cq * 4) = v;
      }
    } else {
      for (int i = lane; i < nrows * wcols; i += 64) {
        const int r = i / wcols, cc = i - r * wcols;
        float v = Ct[r * WS + cc];
        if (bias) v += bias[nb0 * 16 + cc];
        out[(row0 + r) * co + nb0 * 16 + cc] = v;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // tile reads done before the next tile's zero fill
  }
}

// ------------------------------------------------------------------ bf16 features (extension)
// The reference is fp32-only (sparseconvnet_cuda.cpp instantiates <float>); BASELINE configs 3-5 ask
// for bf16.  Features are stored bf16, weights stay fp32 master parameters and are packed to bf16
// per call, accumulation is fp32 (v_mfma_f32_16x16x32_bf16: one instruction covers the 32-channel
// chunk that takes eight fp32 MFMAs), the LDS output tile is fp32 and is rounded to bf16 once at
// the final store.  Same tile-block streams, same determinism.  Requires ci % 32 == 0.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ inline float bf2f(__bf16 v) { return (float)v; }

// Wp16[k][kc][nb][lane][j] = bf16(Wl[k][kc*32 + (lane>>4)*8 + j][nb*16 + (lane&15)])
__global__ __launch_bounds__(256) void k_pack_weights_bf16(const float *__restrict__ W, int vol, int ci, int co,
                                                           int transpose, __bf16 *__restrict__ Wp) {
  const int nkc = nkc_of(ci), nnb = nnb_of(co);
  int64_t total = (int64_t)vol * nkc * nnb * 512;
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  int s = idx & 7;
  int lane = (idx >> 3) & 63;
  int64_t r = idx >> 9;
  int nb = (int)(r % nnb); r /= nnb;
  int kc = (int)(r % nkc); r /= nkc;
  int k = (int)r;
  int c = kc * kKC + (lane >> 4) * 8 + s;
  int n = nb * 16 + (lane & 15);
  float v = 0.0f;
  if (c < ci && n < co) v = transpose ? W[((int64_t)k * co + n) * ci + c] : W[((int64_t)k * ci + c) * co + n];
  Wp[idx] = (__bf16)v;
}

// KG = channel chunks (of 32) gathered per work item; an item is (pair of blocks, chunk group).
template <int NBW, int WPB, int KG, bool ADJ>
__global__ __launch_bounds__(WPB * 64, 2) void k_conv_blocks_mfma_bf16(
    const __bf16 *__restrict__ in, int ci, int64_t in_bytes, __bf16 *__restrict__ out, int co, int64_t V_out,
    const int32_t *__restrict__ words, int64_t words_bytes, int vol, int wflip, const __bf16 *__restrict__ Wp,
    int64_t wp_bytes, const float *__restrict__ bias) {
  constexpr int WS = NBW * 16;
  constexpr int TILE = 64 * WS;
  extern __shared__ __align__(16) float smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int g = lane >> 4, c16 = lane & 15;
  float *Ct = smem + (size_t)wave * TILE;
  const int maxb = tb_maxb(vol);
  const int nnb = nnb_of(co), nkc = ci / kKC;
  const int ngrp = (nkc + KG - 1) / KG;
  const int nb0 = blockIdx.y * NBW;
  const int64_t tile = blockIdx.x, row0 = tile * 64;
  const int64_t ntiles = tb_ntiles(V_out);
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16 *>(in), 0, (int)in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16 *>(Wp), 0, (int)wp_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rwords =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t *>(words), 0, (int)words_bytes, 0x00020000);
  const int nblk = words[tile];
  const unsigned kbase = (unsigned)((ntiles + tile * maxb) * 4);
  const unsigned ebase = (unsigned)((ntiles + ntiles * maxb + tile * maxb * 16) * 4);
  const unsigned rowbytes = (unsigned)ci * 2u, g16 = (unsigned)g * 16u, lane16 = (unsigned)lane * 16u;
  const unsigned wk_bytes = (unsigned)nkc * (unsigned)nnb * 1024u; // packed bytes per offset
  const unsigned c16x4 = (unsigned)c16 * 4u;
  const int nvalid = (nnb - nb0) < NBW ? (nnb - nb0) : NBW;

  for (int i = lane; i < TILE; i += 64) Ct[i] = 0.0f;

  // ADJ (throughput-bound launches): pairs of adjacent blocks, same-offset pairs share their weights
  const int nmine = wave < nblk ? (nblk - wave + WPB - 1) / WPB : 0;
  const int tpairs = (nblk + 1) >> 1;
  const int npairs = ADJ ? (wave < tpairs ? (tpairs - wave + WPB - 1) / WPB : 0) : ((nmine + 1) >> 1);
  const int nitems = npairs * ngrp;
  struct Item { int eA, eB, kA, kB, kc0; };
  auto load_item = [&](int it) {
    Item p;
    const int pr = it / ngrp;
    p.kc0 = (it - pr * ngrp) * KG;
    int bA = ADJ ? 2 * (wave + pr * WPB) : wave + (2 * pr) * WPB;
    if (bA >= nblk) bA = nblk > 0 ? (ADJ ? ((nblk - 1) & ~1) : nblk - 1) : 0;
    int bB = bA + (ADJ ? 1 : WPB);
    const bool hasB = bB < nblk;
    if (!hasB) bB = bA;
    p.eA = (int)__builtin_amdgcn_raw_buffer_load_b32(rwords, c16x4, ebase + (unsigned)bA * 64u, 0);
    p.eB = (int)__builtin_amdgcn_raw_buffer_load_b32(rwords, c16x4, ebase + (unsigned)bB * 64u, 0);
    if (!hasB) p.eB |= (int)0x80000000;
    p.kA = (int)__builtin_amdgcn_raw_buffer_load_b32(rwords, 0u, kbase + (unsigned)bA * 4u, 0);
    p.kB = (int)__builtin_amdgcn_raw_buffer_load_b32(rwords, 0u, kbase + (unsigned)bB * 4u, 0);
    return p;
  };
  // the KG channel chunks of an item are gathered together (KG x 2 x 16 B per lane in flight)
  struct G16 { u32x4 a[KG], b[KG]; };
  auto gather = [&](G16 &q, const Item &p) {
    const unsigned va = (((unsigned)p.eA & 0x7fffffffu) >> 6) * rowbytes + g16;
    const unsigned vb = (((unsigned)p.eB & 0x7fffffffu) >> 6) * rowbytes + g16;
#pragma unroll
    for (int t = 0; t < KG; ++t) {
      if (p.kc0 + t < nkc) {
        q.a[t] = __builtin_amdgcn_raw_buffer_load_b128(rin, va, (unsigned)(p.kc0 + t) * 64u, 0);
        q.b[t] = __builtin_amdgcn_raw_buffer_load_b128(rin, vb, (unsigned)(p.kc0 + t) * 64u, 0);
      }
    }
  };
  auto compute = [&](const G16 &q, const Item &p) {
    int kA = __builtin_amdgcn_readfirstlane(p.kA), kB = __builtin_amdgcn_readfirstlane(p.kB);
    if (wflip) { kA = vol - 1 - kA; kB = vol - 1 - kB; }
    const unsigned soA0 = (unsigned)kA * wk_bytes + (unsigned)nb0 * 1024u;
    const unsigned soB0 = (unsigned)kB * wk_bytes + (unsigned)nb0 * 1024u;
    f32x4 accA[NBW], accB[NBW];
#pragma unroll
    for (int j = 0; j < NBW; ++j) {
      accA[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      accB[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    // weight fragments of chunk t+1 are in flight while chunk t's MFMAs issue (two register sets, picked by
    // the unrolled index); a pair of adjacent same-offset blocks fetches one set for both chains
    const bool same = ADJ && kA == kB; // wave-uniform
    u32x4 wA[2][NBW], wB[2][NBW];
    auto loadw = [&](int set, int t) {
      const unsigned so = (unsigned)(p.kc0 + t) * (unsigned)nnb * 1024u;
#pragma unroll
      for (int j = 0; j < NBW; ++j) {
        if (j < nvalid) {
          wA[set][j] = __builtin_amdgcn_raw_buffer_load_b128(rw, lane16, soA0 + so + j * 1024, 0);
          if (!same) wB[set][j] = __builtin_amdgcn_raw_buffer_load_b128(rw, lane16, soB0 + so + j * 1024, 0);
        }
      }
    };
    if (p.kc0 < nkc) loadw(0, 0);
#pragma unroll
    for (int t = 0; t < KG; ++t) {
      if (p.kc0 + t < nkc) {
        if (t + 1 < KG && p.kc0 + t + 1 < nkc) loadw((t + 1) & 1, t + 1);
        if (same) {
#pragma unroll
          for (int j = 0; j < NBW; ++j) {
            if (j < nvalid) {
              accA[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wA[t & 1][j]),
                                                                __builtin_bit_cast(bf16x8, q.a[t]), accA[j], 0, 0, 0);
              accB[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wA[t & 1][j]),
                                                                __builtin_bit_cast(bf16x8, q.b[t]), accB[j], 0, 0, 0);
            }
          }
        } else {
#pragma unroll
          for (int j = 0; j < NBW; ++j) {
            if (j < nvalid) {
              accA[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wA[t & 1][j]),
                                                                __builtin_bit_cast(bf16x8, q.a[t]), accA[j], 0, 0, 0);
              accB[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wB[t & 1][j]),
                                                                __builtin_bit_cast(bf16x8, q.b[t]), accB[j], 0, 0, 0);
            }
          }
        }
      }
    }
    conv_block_accumulate<NBW, WS>(Ct, p.eA, g, accA);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    conv_block_accumulate<NBW, WS>(Ct, p.eB, g, accB);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  };
  Item p0 = load_item(0), p1 = load_item(1), p2 = load_item(2);
  G16 s0, s1;
  if (nitems > 0) gather(s0, p0);
  for (int it = 0; it < nitems; it += 2) {
    if (it + 1 < nitems) gather(s1, p1);
    compute(s0, p0);
    Item p3 = load_item(it + 3);
    if (it + 1 < nitems) {
      if (it + 2 < nitems) gather(s0, p2);
      compute(s1, p1);
    }
    Item p4 = load_item(it + 4);
    p0 = p2; p1 = p3; p2 = p4;
  }
  __syncthreads();
  // combine in wave order, add bias, round to bf16 once, write the tile
  const int nrows = (int)((V_out - row0) < 64 ? (V_out - row0) : 64);
  const int wcols = (co - nb0 * 16) < NBW * 16 ? (co - nb0 * 16) : NBW * 16;
  if ((co & 3) == 0) {
    const int q = wcols >> 2;
    for (int i = threadIdx.x; i < nrows * q; i += WPB * 64) {
      int r = i / q, cq = i % q;
      float4 v = *reinterpret_cast<const float4 *>(smem + r * WS + cq * 4);
#pragma unroll
      for (int w = 1; w < WPB; ++w) {
        float4 u = *reinterpret_cast<const float4 *>(smem + (size_t)w * TILE + r * WS + cq * 4);
        v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
      }
      if (bias) {
        const float *bb = bias + nb0 * 16 + cq * 4;
        v.x += bb[0]; v.y += bb[1]; v.z += bb[2]; v.w += bb[3];
      }
      typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
      bf16x4 o = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
      *reinterpret_cast<bf16x4 *>(out + (row0 + r) * co + nb0 * 16 + cq * 4) = o;
    }
  } else {
    for (int i = threadIdx.x; i < nrows * wcols; i += WPB * 64) {
      int r = i / wcols, cc = i % wcols;
      float v = smem[r * WS + cc];
#pragma unroll
      for (int w = 1; w < WPB; ++w) v += smem[(size_t)w * TILE + r * WS + cc];
      if (bias) v += bias[nb0 * 16 + cc];
      out[(row0 + r) * co + nb0 * 16 + cc] = (__bf16)v;
    }
  }
}

// ------------------------------------------------------------------ compiled rule book, part 2
// Offset-major compacted pairs (the reference's RuleBook layout: for offset k the (in, out)
// pairs in ascending out order, Metadata.h:34) for the weight-gradient pass, whose reduction
// runs over the pairs of ONE offset.
//   words: [vol] R_k | [vol+1] first 1024-pair chunk of offset k | [vol+1] first 256-pair chunk |
//          [vol][nb256] block bases | [vol][V][2] pairs
// pairs per weight-gradient chunk (one workgroup = 4 waves x chunk/4).  1024 keeps the partial-sum
// traffic small (a partial is nIn*nOut floats per chunk); 256 gives a small rule book ~4x more workgroups
// than CUs -- at 1024 the S80k launch ran one wave per SIMD and was pure gather latency.  Both chunk tables
// are compiled into the pair list; the launch picks by rule-book size and layer width.
__host__ __device__ inline int dw_chunk(int64_t V, int vol, int n_in, int n_out) {
  return ((int64_t)vol * V <= (1ll << 21) && (int64_t)n_in * n_out <= 64 * 64) ? 256 : 1024;
}
__host__ __device__ inline int64_t op_hdr(int vol) { return (int64_t)vol + 2 * (vol + 1); }
__host__ __device__ inline int64_t op_nb256(int64_t V) { return (V + 255) / 256; }

// one block per offset: exclusive scan of the per-256-row hit counts
__global__ __launch_bounds__(256) void k_offset_bases(const StreamJobs js) {            // common.h: one launch, many books
  __shared__ int ws[4];
  __shared__ int carry_s;
  const int job = stream_job_of(js, blockIdx.x);
  const int32_t *__restrict__ counts = js.j[job].counts;
  int32_t *__restrict__ words = js.j[job].words;
  const int64_t nb = op_nb256(js.j[job].V);
  const int vol = js.j[job].vol;
  const int k = blockIdx.x - js.first[job], lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int32_t *bases = words + op_hdr(vol) + (int64_t)k * nb;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int64_t base = 0; base < nb; base += 256) {
    int64_t i = base + threadIdx.x;
    int v = (i < nb) ? counts[(int64_t)k * nb + i] : 0, inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      int u = __shfl_up(inc, d);
      if (lane >= d) inc += u;
    }
    if (lane == 63) ws[wave] = inc;
    __syncthreads();
    int pre = carry_s;
    for (int j = 0; j < wave; ++j) pre += ws[j];
    if (i < nb) bases[i] = pre + inc - v;
    __syncthreads();
    if (threadIdx.x == 255) carry_s = pre + inc;
    __syncthreads();
  }
  if (threadIdx.x == 0) words[k] = carry_s; // R_k
}


__global__ __launch_bounds__(256) void k_fill_offset_pairs(const StreamJobs js) {       // common.h: one launch, many books
  __shared__ int ws[4];
  const int job = stream_job_of(js, blockIdx.x);
  const int32_t *__restrict__ table = js.j[job].table;
  int32_t *__restrict__ words = js.j[job].words;
  const int64_t V = js.j[job].V;
  const int vol = js.j[job].vol;
  const int64_t nb = op_nb256(V);
  const unsigned lb = blockIdx.x - js.first[job];            // block of this book: (row block, offset), offset-major
  const unsigned bx = (unsigned)(lb % (unsigned)nb);
  const int k = (int)(lb / (unsigned)nb), lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t row = (int64_t)bx * 256 + threadIdx.x;
  // chunk layouts (one thread of the grid): offset k owns chunks [cstart[k], cstart[k+1]) of 1024 resp. 256
  // pairs each; the R_k were written by k_offset_bases, the previous launch on this stream
  if (lb == 0 && threadIdx.x == 0) {
    int c = 0, c4 = 0;
    for (int kk = 0; kk < vol; ++kk) {
      words[vol + kk] = c;
      words[vol + (vol + 1) + kk] = c4;
      c += (words[kk] + 1023) / 1024;
      c4 += (words[kk] + 255) / 256;
    }
    words[vol + vol] = c;
    words[vol + (vol + 1) + vol] = c4;
  }
  const int t = (row < V) ? table[(int64_t)k * V + row] : -1;
  const unsigned long long m = __ballot(t >= 0);
  if (lane == 0) ws[wave] = (int)__popcll(m);
  __syncthreads();
  int pre = words[op_hdr(vol) + (int64_t)k * nb + bx];
  for (int j = 0; j < wave; ++j) pre += ws[j];
  if (t >= 0) {
    int pos = pre + (int)__popcll(m & ((1ull << lane) - 1ull));
    int2 *pairs = reinterpret_cast<int2 *>(words + op_hdr(vol) + (int64_t)vol * nb) + (int64_t)k * V;
    pairs[pos] = make_int2(t, (int)row);
  }
}

// ----------------------------------------------------------------------------- dW
// partial[chunk][c][n] = sum over the chunk's pairs (t, o) of in[t][c] * d_out[o][n]; one workgroup
// per chunk of dw_chunk(V, vol) pairs of one offset (each wave a quarter), MFMA with the pair index as the
// reduction dimension.  Pair indices are loaded 64 at a time (coalesced) and handed to the lane
// groups by shuffles; 16 pairs are gathered per step before their MFMAs issue.  The four waves'
// accumulators are summed through LDS in wave order (deterministic).  CB x NB blocks of 16.
__device__ inline bool dw_chunk_range(const int32_t *__restrict__ words, int vol, int chunk_pairs, int direct, int chunk,
                                      int lane, int &k, int &p0, int &p1) {
  if (direct) {
    k = chunk;
    p0 = 0;
    p1 = words[k];
    return true;
  }
  const int32_t *cstart = words + vol + (chunk_pairs == 256 ? vol + 1 : 0);
  if (chunk >= cstart[vol]) return false;                  // workgroup-uniform
  k = 0;
  for (int k0 = 0; k0 < vol; k0 += 64) {
    int kk = k0 + lane;
    bool mine = kk < vol && cstart[kk] <= chunk && chunk < cstart[kk + 1];
    unsigned long long m = __ballot(mine);
    if (m) { k = k0 + (__ffsll((long long)m) - 1); break; }
  }
  const int rk = words[k];
  p0 = (chunk - cstart[k]) * chunk_pairs;
  p1 = p0 + chunk_pairs;
  if (p1 > rk) p1 = rk;
  return true;
}

__device__ inline float ldf(const float *p, int64_t i) { return p[i]; }
__device__ inline float ldf(const __bf16 *p, int64_t i) { return (float)p[i]; }

template <int CB, int NB, typename T>
__global__ __launch_bounds__(256) void k_conv_dw_pairs(const T *__restrict__ in, int ci,
                                                       const T *__restrict__ d_out, int co, int64_t V,
                                                       const int32_t *__restrict__ words, int vol,
                                                       int chunk_pairs, float *__restrict__ partial, int direct) {
  __shared__ f32x4 red[CB * NB][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int g = lane >> 4, c16 = lane & 15;
  const int nnb = nnb_of(co);
  const int tiles_n = (nnb + NB - 1) / NB;
  const int tile = blockIdx.y;
  const int cb0 = (tile / tiles_n) * CB, nb0 = (tile % tiles_n) * NB;
  const int chunk = blockIdx.x;
  int k, c0, c1;                                           // the chunk's offset and pair range; each wave a quarter
  if (!dw_chunk_range(words, vol, chunk_pairs, direct, chunk, lane, k, c0, c1)) return;
  const int rk = words[k];
  const int p0 = c0 + wave * (chunk_pairs / 4);
  int p1 = p0 + chunk_pairs / 4;
  if (p1 > c1) p1 = c1;
  const int2 *pairs = reinterpret_cast<const int2 *>(words + op_hdr(vol) + (int64_t)vol * op_nb256(V)) +
                      (int64_t)k * V;
  f32x4 acc[CB][NB];
#pragma unroll
  for (int a = 0; a < CB; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // Software pipeline: the rows of the NEXT group of 16 pairs are in flight while the current group's
  // MFMAs issue (two register sets, static alternation: a block of 64 pairs is four groups).  All loads
  // are unconditional from clamped addresses and masked at use, so the waits stay counted.
  const int cA = ci - 1, nA = co - 1;
  int ca[CB], na[NB];
#pragma unroll
  for (int a = 0; a < CB; ++a) { int c = (cb0 + a) * 16 + c16; ca[a] = c < ci ? c : cA; }
#pragma unroll
  for (int b = 0; b < NB; ++b) { int n = (nb0 + b) * 16 + c16; na[b] = n < co ? n : nA; }
  auto gather = [&](T (&av)[4][CB], T (&bv)[4][NB], int2 pr, int q0) {
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      const int src = q0 + st * 4 + g;
      int tq = __shfl(pr.x, src), oq = __shfl(pr.y, src);
      tq = tq < 0 ? 0 : tq; oq = oq < 0 ? 0 : oq;
#pragma unroll
      for (int a = 0; a < CB; ++a) av[st][a] = in[(int64_t)tq * ci + ca[a]];
#pragma unroll
      for (int b = 0; b < NB; ++b) bv[st][b] = d_out[(int64_t)oq * co + na[b]];
    }
  };
  auto mfmas = [&](T (&av)[4][CB], T (&bv)[4][NB], int qbase) {
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      const bool on = qbase + st * 4 + g < p1;
      float fa[CB], fb[NB];
#pragma unroll
      for (int a = 0; a < CB; ++a) fa[a] = (on && (cb0 + a) * 16 + c16 < ci) ? (float)av[st][a] : 0.0f;
#pragma unroll
      for (int b = 0; b < NB; ++b) fb[b] = (on && (nb0 + b) * 16 + c16 < co) ? (float)bv[st][b] : 0.0f;
#pragma unroll
      for (int a = 0; a < CB; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[a], fb[b], acc[a][b], 0, 0, 0);
    }
  };
  if (p0 < p1) {
    T avA[4][CB], bvA[4][NB], avB[4][CB], bvB[4][NB];
    const int last = rk - 1;                                 // rk >= 1 here
    auto load_pr = [&](int q64) {
      int q = q64 + lane;
      int2 v = pairs[q < rk ? q : last];
      return (q < p1) ? v : make_int2(-1, -1);
    };
    int2 pr = load_pr(p0);
    gather(avA, bvA, pr, 0);
    for (int q64 = p0; q64 < p1; q64 += 64) {
      int2 prn = load_pr(q64 + 64);
      __builtin_amdgcn_sched_barrier(0);
      gather(avB, bvB, pr, 16);
      __builtin_amdgcn_sched_barrier(0);
      mfmas(avA, bvA, q64);
      __builtin_amdgcn_sched_barrier(0);
      gather(avA, bvA, pr, 32);
      __builtin_amdgcn_sched_barrier(0);
      mfmas(avB, bvB, q64 + 16);
      __builtin_amdgcn_sched_barrier(0);
      gather(avB, bvB, pr, 48);
      __builtin_amdgcn_sched_barrier(0);
      mfmas(avA, bvA, q64 + 32);
      __builtin_amdgcn_sched_barrier(0);
      gather(avA, bvA, prn, 0);
      __builtin_amdgcn_sched_barrier(0);
      mfmas(avB, bvB, q64 + 48);
      pr = prn;
    }
  }
  // sum the four waves' accumulators in wave order: w0 + w1 + w2 + w3
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int a = 0; a < CB; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          if (w == 0) red[a * NB + b][lane] = acc[a][b];
          else {
            f32x4 t = red[a * NB + b][lane];
            t[0] += acc[a][b][0]; t[1] += acc[a][b][1]; t[2] += acc[a][b][2]; t[3] += acc[a][b][3];
            red[a * NB + b][lane] = t;
          }
        }
    }
    __syncthreads();
  }
  if (wave != 0) return;
  // D[i = c (row of dW) = g*4 + r][j = n = c16]
  float *P = partial + (int64_t)(direct ? k : chunk) * ci * co;
#pragma unroll
  for (int a = 0; a < CB; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      f32x4 t = red[a * NB + b][lane];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int c = (cb0 + a) * 16 + g * 4 + r, n = (nb0 + b) * 16 + c16;
        if (c < ci && n < co) P[(int64_t)c * co + n] = t[r];
      }
    }
}

// bf16 features: the same chunked scheme on v_mfma_f32_16x16x32_bf16 with the PAIR index as K.
// The MFMA wants, per lane, 8 consecutive pairs of ONE channel -- a transpose of the row-major
// feature matrices.  Each wave stages 32 gathered rows (16-byte global loads, 16-byte LDS writes, rows
// padded by 16 B) and reads the operands back with ds_read_b64_tr_b16 (hardware 4x16 transpose: lane L
// of a 16-lane group receives column L of 4 rows): two reads per 16-channel fragment.  The next batch's
// rows are loaded into registers while the current batch's MFMAs run.  CB, NB in {2, 4}.
typedef short s16x4 __attribute__((ext_vector_type(4)));

template <int CB, int NB>
__global__ __launch_bounds__(256) void k_conv_dw_pairs_bf16(const __bf16 *__restrict__ in, int ci,
                                                            const __bf16 *__restrict__ d_out, int co, int64_t V,
                                                            const int32_t *__restrict__ words, int vol,
                                                            int chunk_pairs, float *__restrict__ partial, int direct) {
  constexpr int SX = CB * 32 + 16, SG = NB * 32 + 16; // LDS row strides (bytes)
  constexpr int XCH = CB * 2, GCH = NB * 2;           // 16-byte chunks per row
  constexpr int XIT = (32 * XCH) / 64, GIT = (32 * GCH) / 64; // chunks per lane and batch
  __shared__ f32x4 red[CB * NB][64];
  __shared__ __attribute__((aligned(16))) unsigned char stage[4][32 * (SX + SG)];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int g = lane >> 4, c16 = lane & 15;
  const int nnb = nnb_of(co);
  const int tiles_n = (nnb + NB - 1) / NB;
  const int tile = blockIdx.y;
  const int cb0 = (tile / tiles_n) * CB, nb0 = (tile % tiles_n) * NB;
  const int chunk = blockIdx.x;
  int k, c0, c1;
  if (!dw_chunk_range(words, vol, chunk_pairs, direct, chunk, lane, k, c0, c1)) return;
  const int p0 = c0 + wave * (chunk_pairs / 4);
  int p1 = p0 + chunk_pairs / 4;
  if (p1 > c1) p1 = c1;
  const int2 *pairs = reinterpret_cast<const int2 *>(words + op_hdr(vol) + (int64_t)vol * op_nb256(V)) +
                      (int64_t)k * V;
  unsigned char *xs = stage[wave], *gs = stage[wave] + 32 * SX;
  f32x4 acc[CB][NB];
#pragma unroll
  for (int a = 0; a < CB; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // register staging of one batch: XIT + GIT 16-byte chunks per lane (rows past the range: zeros)
  u32x4 xr[XIT], gr[GIT];
  auto load_batch = [&](int q32) {
    const int q = q32 + (lane & 31);
    const int2 pr = (q < p1) ? pairs[q] : make_int2(-1, -1);
#pragma unroll
    for (int it = 0; it < XIT; ++it) {
      const int id = it * 64 + lane, row = id / XCH, ch = id - row * XCH;
      const int t = __shfl(pr.x, row);
      xr[it] = (u32x4){0u, 0u, 0u, 0u};
      if (t >= 0 && cb0 * 16 + ch * 8 < ci) // planes past the layer width (last tile): zeros, never read out of the row
        xr[it] = *reinterpret_cast<const u32x4 *>(in + (int64_t)t * ci + cb0 * 16 + ch * 8);
    }
#pragma unroll
    for (int it = 0; it < GIT; ++it) {
      const int id = it * 64 + lane, row = id / GCH, ch = id - row * GCH;
      const int o = __shfl(pr.y, row);
      gr[it] = (u32x4){0u, 0u, 0u, 0u};
      if (o >= 0 && nb0 * 16 + ch * 8 < co)
        gr[it] = *reinterpret_cast<const u32x4 *>(d_out + (int64_t)o * co + nb0 * 16 + ch * 8);
    }
  };
  auto store_batch = [&]() {
#pragma unroll
    for (int it = 0; it < XIT; ++it) {
      const int id = it * 64 + lane, row = id / XCH, ch = id - row * XCH;
      *reinterpret_cast<u32x4 *>(xs + row * SX + ch * 16) = xr[it];
    }
#pragma unroll
    for (int it = 0; it < GIT; ++it) {
      const int id = it * 64 + lane, row = id / GCH, ch = id - row * GCH;
      *reinterpret_cast<u32x4 *>(gs + row * SG + ch * 16) = gr[it];
    }
  };
  // transposed operand: rows (pairs) g*8 .. g*8+7 of 16-bit column (blk*16 + c16)
  const int q4 = c16 >> 2, p4 = c16 & 3;
  auto tr_frag = [&](const unsigned char *base, int stride, int blk) {
    const unsigned char *a0 = base + (g * 8 + q4) * stride + (blk * 16 + 4 * p4) * 2;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)a0);
    const s16x4 hi =
        __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(a0 + 4 * stride));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
  };
  if (p0 < p1) { // wave-uniform: the tr reads below always run with all 64 lanes active
    load_batch(p0);
    for (int q32 = p0; q32 < p1; q32 += 32) {
      store_batch();
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      if (q32 + 32 < p1) load_batch(q32 + 32);
      bf16x8 af[CB], bf[NB];
#pragma unroll
      for (int a = 0; a < CB; ++a) af[a] = tr_frag(xs, SX, a);
#pragma unroll
      for (int b = 0; b < NB; ++b) bf[b] = tr_frag(gs, SG, b);
#pragma unroll
      for (int a = 0; a < CB; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a], bf[b], acc[a][b], 0, 0, 0);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // reads done before the next batch overwrites
    }
  }
  // sum the four waves' accumulators in wave order: w0 + w1 + w2 + w3
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int a = 0; a < CB; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          if (w == 0) red[a * NB + b][lane] = acc[a][b];
          else {
            f32x4 t = red[a * NB + b][lane];
            t[0] += acc[a][b][0]; t[1] += acc[a][b][1]; t[2] += acc[a][b][2]; t[3] += acc[a][b][3];
            red[a * NB + b][lane] = t;
          }
        }
    }
    __syncthreads();
  }
  if (wave != 0) return;
  float *P = partial + (int64_t)(direct ? k : chunk) * ci * co;
#pragma unroll
  for (int a = 0; a < CB; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      f32x4 t = red[a * NB + b][lane];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int c = (cb0 + a) * 16 + g * 4 + r, n = (nb0 + b) * 16 + c16;
        if (c < ci && n < co) P[(int64_t)c * co + n] = t[r];
      }
    }
}

// ---- full-tile weight gradient (round 5) -------------------------------------------------------------------------------
// k_conv_dw_pairs[_bf16] give a workgroup ONE 64 x 64 block of dW: a 128 x 128 layer is four workgroups per chunk and
// every gathered row (input features and output gradients) is fetched twice -- at the dominant level 800 MB (bf16) /
// 1.6 GB (fp32) of row gathers per launch, ~5.5 TB/s out of the L2s, which is what those kernels run at.  Here a
// workgroup forms a whole 128 x 128 block for its chunk: the 256 threads gather each pair's two rows ONCE (full 128-plane
// slices, 16-byte loads, whole 256- / 512-byte segments) into LDS, the four waves own one 64 x 64 quadrant each and read
// their operand fragments from there.  All waves walk all pairs of the chunk, so there is no cross-wave sum at the end;
// the summation order (pairs ascending per quadrant element, workgroups ascending in k_conv_dw_reduce_ranges) is fixed.
//
// Work split: a workgroup per 1024-pair chunk is 787 workgroups of equal weight for 768 resident slots at the dominant
// level -- a CU that receives four takes a third longer than the average of 3.07, and every workgroup here is four times as
// heavy as a 64 x 64 one.  So the pairs of ALL offsets are laid end to end (offset k owns [s_k, s_k + R_k) of R = sum R_k)
// and workgroup w takes [w per, (w + 1) per), per = ceil(R / n_wg): every workgroup the same number of pairs, whatever the
// offsets' sizes.  A workgroup whose range crosses offset boundaries writes one partial block per offset it touches, into
// slot w + (number of non-empty offsets before k): along the staircase of (w, k) cells that sum grows by one per cell, so
// slots are unique, at most n_wg + vol of them, and offset k's are contiguous: w_lo(k) + ne_k .. w_hi(k) + ne_k.
constexpr int kDwMinPer = 128;                             // pairs per workgroup at least (tiny rule books: fewer workgroups)
__device__ inline int64_t dw_range_per(int64_t rtot, int n_wg) {
  const int64_t per = (rtot + n_wg - 1) / n_wg;
  return per < kDwMinPer ? kDwMinPer : per;
}
__device__ inline int wave_incl_scan(int v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int u = __shfl_up(v, d);
    if (lane >= d) v += u;
  }
  return v;
}
__device__ inline int64_t dw_total_pairs(const int32_t *__restrict__ words, int vol, int lane) {
  int64_t rtot = 0;
  for (int k0 = 0; k0 < vol; k0 += 64) {
    const int c = k0 + lane < vol ? words[k0 + lane] : 0;
    rtot += __shfl(wave_incl_scan(c, lane), 63);
  }
  return rtot;
}
// calls seg(k, p0, p1, slot) for every offset k the range [lo, hi) of the concatenated pair list touches, k ascending;
// wave-uniform (every wave of a workgroup walks the same segments)
template <typename F>
__device__ inline void dw_for_segments(const int32_t *__restrict__ words, int vol, int lane, int64_t lo, int64_t hi, int w,
                                       F seg) {
  int64_t base = 0;
  int ne_base = 0;
  for (int k0 = 0; k0 < vol && base < hi; k0 += 64) {
    const int c = k0 + lane < vol ? words[k0 + lane] : 0;
    const int incl = wave_incl_scan(c, lane);
    const int64_t s = base + incl - c;
    const unsigned long long nz = __ballot(c > 0);
    unsigned long long m = __ballot(c > 0 && s < hi && s + c > lo);
    while (m) {
      const int j = __ffsll((long long)m) - 1;
      m &= m - 1;
      const int64_t sk = base + __shfl(incl - c, j);
      const int ck = __shfl(c, j);
      const int ne = ne_base + (int)__popcll(nz & ((1ull << j) - 1ull));
      const int64_t a = lo > sk ? lo : sk, b = hi < sk + ck ? hi : sk + ck;
      seg(k0 + j, (int)(a - sk), (int)(b - sk), w + ne);
    }
    base += __shfl(incl, 63);
    ne_base += (int)__popcll(nz);
  }
}
// bf16 storage: batches of 64 pairs; a row slice is 256 bytes = 16 chunks of 16 bytes, stored unpadded with the chunk
// index XOR-swizzled by the row ((row & 3) | (row >> 3 & 1) << 2) << 1, so that the eight rows one half-wave of a
// ds_read_b64_tr_b16 touches land in eight different 32-byte bank groups; the pair -> LDS row assignment is free (the
// pair index is the reduction dimension) and the same for both operands.
__global__ __launch_bounds__(256) void k_conv_dw_full_bf16(const __bf16 *__restrict__ in, int ci,
                                                           const __bf16 *__restrict__ d_out, int co, int64_t V,
                                                           const int32_t *__restrict__ words, int vol,
                                                           float *__restrict__ partial) {
  constexpr int kB = 64;                                   // pairs per batch
  __shared__ __attribute__((aligned(16))) unsigned char xs[kB * 256];
  __shared__ __attribute__((aligned(16))) unsigned char gs[kB * 256];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int g = lane >> 4, c16 = lane & 15;
  const int tiles_n = co >> 7;
  const int tc = (int)blockIdx.y / tiles_n, tn = (int)blockIdx.y % tiles_n;
  const int wc = wave >> 1, wn = wave & 1;
  const int64_t rtot = dw_total_pairs(words, vol, lane);
  const int64_t per = dw_range_per(rtot, (int)gridDim.x);
  const int64_t lo = (int64_t)blockIdx.x * per, hi = lo + per < rtot ? lo + per : rtot;
  if (lo >= hi) return;                                    // workgroup-uniform
  const int2 *pairs0 = reinterpret_cast<const int2 *>(words + op_hdr(vol) + (int64_t)vol * op_nb256(V));
  const __bf16 *inb = in + tc * 128, *gb = d_out + tn * 128;
  auto sw = [](int row) { return (((row & 3) | (((row >> 3) & 1) << 2)) << 1); };
  const int q4 = c16 >> 2, p4 = c16 & 3;
  dw_for_segments(words, vol, lane, lo, hi, (int)blockIdx.x, [&](int k, int p0, int p1, int slot) {
  const int2 *pairs = pairs0 + (int64_t)k * V;
  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  u32x4 xr[4], gr[4];
  // the pair entries of a batch are loaded one batch ahead of its rows (a wave issues in order: waiting for them in
  // front of the row loads would stall the MFMAs behind)
  auto load_pairs = [&](int q0) {
    const int q = q0 + lane;
    return (q < p1) ? pairs[q] : make_int2(-1, -1);
  };
  auto load_batch = [&](int2 pr) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int row = it * 16 + wave * 4 + g;              // this lane's chunk: (row, c16)
      const int t = __shfl(pr.x, row), o = __shfl(pr.y, row);
      xr[it] = (u32x4){0u, 0u, 0u, 0u};
      gr[it] = (u32x4){0u, 0u, 0u, 0u};
      if (t >= 0) {
        xr[it] = *reinterpret_cast<const u32x4 *>(inb + (int64_t)t * ci + c16 * 8);
        gr[it] = *reinterpret_cast<const u32x4 *>(gb + (int64_t)o * co + c16 * 8);
      }
    }
  };
  auto store_batch = [&]() {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int row = it * 16 + wave * 4 + g;
      const int off = row * 256 + ((c16 ^ sw(row)) << 4);
      *reinterpret_cast<u32x4 *>(xs + off) = xr[it];
      *reinterpret_cast<u32x4 *>(gs + off) = gr[it];
    }
  };
  // transposed operand of one K step (32 pairs = LDS rows r0 .. r0+31): pairs g*8 .. g*8+7 of plane blk*16 + c16
  auto tr_frag = [&](const unsigned char *base, int r0, int blk) {
    const int rl = r0 + g * 8 + q4, rh = rl + 4;
    const int ch = blk * 2 + (p4 >> 1), hb = (p4 & 1) << 3;
    const unsigned char *al = base + rl * 256 + ((ch ^ sw(rl)) << 4) + hb;
    const unsigned char *ah = base + rh * 256 + ((ch ^ sw(rh)) << 4) + hb;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)al);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)ah);
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
  };
  {
    load_batch(load_pairs(p0));
    int2 prn = load_pairs(p0 + kB);
    for (int q0 = p0; q0 < p1; q0 += kB) {
      store_batch();
      __syncthreads();
      if (q0 + kB < p1) load_batch(prn);                   // in flight under this batch's MFMAs
      prn = load_pairs(q0 + 2 * kB);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 af[4], bf[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) af[a] = tr_frag(xs, ks * 32, wc * 4 + a);
#pragma unroll
        for (int b = 0; b < 4; ++b) bf[b] = tr_frag(gs, ks * 32, wn * 4 + b);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a], bf[b], acc[a][b], 0, 0, 0);
      }
      __syncthreads();                                     // every wave has read the batch before it is overwritten
    }
  }
  float *P = partial + (int64_t)slot * ci * co;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = tc * 128 + (wc * 4 + a) * 16 + g * 4 + r, n = tn * 128 + (wn * 4 + b) * 16 + c16;
        P[(int64_t)c * co + n] = acc[a][b][r];
      }
  });
}

// fp32: batches of 32 pairs; a row slice is 512 bytes, LDS rows padded to 144 words so that the four rows x 16 planes
// one operand read touches are 64 different banks.
__global__ __launch_bounds__(256) void k_conv_dw_full_f32(const float *__restrict__ in, int ci,
                                                          const float *__restrict__ d_out, int co, int64_t V,
                                                          const int32_t *__restrict__ words, int vol,
                                                          float *__restrict__ partial) {
  constexpr int kB = 32, kS = 144;                         // pairs per batch, LDS row stride in words
  __shared__ __attribute__((aligned(16))) float xs[kB * kS];
  __shared__ __attribute__((aligned(16))) float gs[kB * kS];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int g = lane >> 4, c16 = lane & 15;
  const int tiles_n = co >> 7;
  const int tc = (int)blockIdx.y / tiles_n, tn = (int)blockIdx.y % tiles_n;
  const int wc = wave >> 1, wn = wave & 1;
  const int64_t rtot = dw_total_pairs(words, vol, lane);
  const int64_t per = dw_range_per(rtot, (int)gridDim.x);
  const int64_t lo = (int64_t)blockIdx.x * per, hi = lo + per < rtot ? lo + per : rtot;
  if (lo >= hi) return;                                    // workgroup-uniform
  const int2 *pairs0 = reinterpret_cast<const int2 *>(words + op_hdr(vol) + (int64_t)vol * op_nb256(V));
  const float *inb = in + tc * 128, *gb = d_out + tn * 128;
  const int h = lane >> 5, c32 = lane & 31;                // this lane's chunks: rows it*8 + wave*2 + h, 16-byte chunk c32
  dw_for_segments(words, vol, lane, lo, hi, (int)blockIdx.x, [&](int k, int p0, int p1, int slot) {
  const int2 *pairs = pairs0 + (int64_t)k * V;
  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 xr[4], gr[4];
  auto load_pairs = [&](int q0) {
    const int q = q0 + c32;
    return (q < p1) ? pairs[q] : make_int2(-1, -1);
  };
  auto load_batch = [&](int2 pr) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int row = it * 8 + wave * 2 + h;
      const int t = __shfl(pr.x, row), o = __shfl(pr.y, row);
      xr[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
      gr[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (t >= 0) {
        xr[it] = *reinterpret_cast<const f32x4 *>(inb + (int64_t)t * ci + c32 * 4);
        gr[it] = *reinterpret_cast<const f32x4 *>(gb + (int64_t)o * co + c32 * 4);
      }
    }
  };
  auto store_batch = [&]() {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int row = it * 8 + wave * 2 + h;
      *reinterpret_cast<f32x4 *>(xs + row * kS + c32 * 4) = xr[it];
      *reinterpret_cast<f32x4 *>(gs + row * kS + c32 * 4) = gr[it];
    }
  };
  {
    load_batch(load_pairs(p0));
    int2 prn = load_pairs(p0 + kB);
    for (int q0 = p0; q0 < p1; q0 += kB) {
      store_batch();
      __syncthreads();
      if (q0 + kB < p1) load_batch(prn);
      prn = load_pairs(q0 + 2 * kB);
      const float *xa = xs + g * kS + wc * 64 + c16, *ga = gs + g * kS + wn * 64 + c16;
#pragma unroll
      for (int st = 0; st < 8; ++st) {                     // four pairs per MFMA step: A[i = plane][k = pair g]
        float fa[4], fb[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) fa[a] = xa[st * 4 * kS + a * 16];
#pragma unroll
        for (int b = 0; b < 4; ++b) fb[b] = ga[st * 4 * kS + b * 16];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[a], fb[b], acc[a][b], 0, 0, 0);
      }
      __syncthreads();
    }
  }
  float *P = partial + (int64_t)slot * ci * co;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = tc * 128 + (wc * 4 + a) * 16 + g * 4 + r, n = tn * 128 + (wn * 4 + b) * 16 + c16;
        P[(int64_t)c * co + n] = acc[a][b][r];
      }
  });
}

// dW[k][i] = sum over the workgroups w_lo(k) .. w_hi(k) of the full-tile launch of their partial block for offset k, in
// workgroup order (four slices as below, combined in slice order): deterministic.  Offsets without rules: zeros.
__global__ __launch_bounds__(256) void k_conv_dw_reduce_ranges(const float *__restrict__ partial,
                                                               const int32_t *__restrict__ words, int vol, int n_wg,
                                                               int64_t cico, float *__restrict__ dW) {
  __shared__ float red[4][64];
  const int k = blockIdx.y, col = threadIdx.x & 63, sl = threadIdx.x >> 6, lane = col;
  const int64_t i = (int64_t)blockIdx.x * 64 + col;
  // s_k, R_k, ne_k of this block's offset and the total (every wave the same values)
  int64_t base = 0, sk = 0;
  int ne = 0, ck = 0;
  for (int k0 = 0; k0 < vol; k0 += 64) {
    const int c = k0 + lane < vol ? words[k0 + lane] : 0;
    const int incl = wave_incl_scan(c, lane);
    const unsigned long long nz = __ballot(c > 0);
    if (k >= k0 && k < k0 + 64) {
      const int j = k - k0;
      sk = base + __shfl(incl - c, j);
      ck = __shfl(c, j);
      ne += (int)__popcll(nz & ((1ull << j) - 1ull));
    } else if (k >= k0 + 64) ne += (int)__popcll(nz);
    base += __shfl(incl, 63);
  }
  float s = 0.0f;
  if (ck > 0 && i < cico) {
    const int64_t per = dw_range_per(base, n_wg);
    const int64_t w0 = sk / per, w1 = (sk + ck - 1) / per;
#pragma unroll 4
    for (int64_t w = w0 + sl; w <= w1; w += 4) s += partial[(w + ne) * cico + i];
  }
  red[sl][col] = s;
  __syncthreads();
  if (sl == 0 && i < cico) dW[(int64_t)k * cico + i] = (red[0][col] + red[1][col]) + (red[2][col] + red[3][col]);
}

// dW[k][i] = sum of the partials of offset k's chunks (fixed order => deterministic): a block takes 64
// consecutive elements i and deals the chunks to 4 slices (chunk c0+s, c0+s+4, ...), several loads in
// flight per thread; the slices are combined in slice order through LDS
__global__ __launch_bounds__(256) void k_conv_dw_reduce(const float *__restrict__ partial,
                                                        const int32_t *__restrict__ words, int vol,
                                                        int chunk_pairs, int64_t cico, float *__restrict__ dW) {
  __shared__ float red[4][64];
  const int k = blockIdx.y, col = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * 64 + col;
  const int32_t *cstart = words + vol + (chunk_pairs == 256 ? vol + 1 : 0);
  const int c0 = cstart[k], c1 = cstart[k + 1];
  float s = 0.0f;
  if (i < cico) {
#pragma unroll 8
    for (int c = c0 + sl; c < c1; c += 4) s += partial[(int64_t)c * cico + i];
  }
  red[sl][col] = s;
  __syncthreads();
  if (sl == 0 && i < cico) dW[(int64_t)k * cico + i] = (red[0][col] + red[1][col]) + (red[2][col] + red[3][col]);
}

// d_bias[n] = sum_rows d_out[row][n] (at::sum_out, CPU/Convolution.cpp:100-101), two fixed-order stages:
// S row slices x 64-column blocks of partial sums (rows s, s+S, ... per slice; 4 sub-slices per block
// combined through LDS), then one thread per column adds the S partials in slice order => deterministic.
template <typename T>
__global__ __launch_bounds__(256) void k_col_sum_partial(const T *__restrict__ x, int64_t rows, int co, int S,
                                                         float *__restrict__ part) {
  __shared__ float red[4][64];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), sub = threadIdx.x >> 6, sl = blockIdx.y;
  float s = 0.0f;
  if (col < co)
    for (int64_t r = (int64_t)sl * 4 + sub; r < rows; r += (int64_t)S * 4) s += ldf(x, r * co + col);
  red[sub][threadIdx.x & 63] = s;
  __syncthreads();
  if (sub == 0 && col < co)
    part[(int64_t)sl * co + col] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

__global__ __launch_bounds__(256) void k_col_sum_final(const float *__restrict__ part, int co, int S,
                                                       float *__restrict__ out) {
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col >= co) return;
  float s = 0.0f;
  for (int i = 0; i < S; ++i) s += part[(int64_t)i * co + col];
  out[col] = s;
}

// scratch: the dW partial buffer, free again once the chunk reduction has been enqueued
template <typename T>
static void launch_col_sum(const T *d_out, int64_t rows, int co, float *d_bias, float *scratch, int64_t scratch_floats,
                           hipStream_t st) {
  int64_t S = scratch_floats / co;
  if (S > 128) S = 128;
  if (S > ceil_div(rows, 4)) S = ceil_div(rows, 4);
  if (S < 1) S = 1;
  hipLaunchKernelGGL((k_col_sum_partial<T>), dim3((unsigned)ceil_div(co, 64), (unsigned)S), dim3(256), 0, st, d_out,
                     rows, co, (int)S, scratch);
  hipLaunchKernelGGL(k_col_sum_final, dim3((unsigned)ceil_div(co, 256)), dim3(256), 0, st, scratch, co, (int)S, d_bias);
}

// the full-tile kernels (k_conv_dw_full_*): whole 128 x 128 blocks, 16-byte row loads; knob DW_FULL = 0 keeps the
// 64 x 64-block kernels (A/B)
// Returns the number of workgroups per 128 x 128 block, or 0 when the 64 x 64-block kernels should run.  Measured on
// the bench's rule books (tools/tools_dw_ab.py): the full-tile kernels win from ~250 k rules on and lose below (few, heavy
// workgroups: latency-bound); a workgroup count that is a multiple of the 256 CUs (every CU the same number of equal
// ranges) beats anything in between by 10-25 %; two per CU pay from ~600 k rules.  The rule count is on the device: it
// is estimated from the table's size (a 3^3 submanifold table of a scene is about a third full, vol 1 is full).
static int dw_full_workgroups(int ci, int co, const void *in, const void *d_out, int64_t max_chunks, int vol, int64_t V_out,
                              bool bf) {
  if (ci % 128 || co % 128 || ((uintptr_t)in & 15) || ((uintptr_t)d_out & 15) || knob(K_DW_FULL) == 0) return 0;
  const int tiles = (ci >> 7) * (co >> 7);
  const int64_t slots = max_chunks - vol;                  // n_wg + vol partial blocks must fit the caller's scratch buffer
  const int knob_min = knob(K_DW_FULL_MIN), knob_wgs = knob(K_DW_FULL_WGS);
  if (knob_wgs > 0) return knob_wgs <= slots ? knob_wgs : 0;             // (A/B: a given number of workgroups)
  if (knob_min > 0) {                                                    // (tests: small rule books through the kernel)
    const int64_t n = slots < 256 / tiles ? slots : 256 / tiles;
    return n >= knob_min ? (int)n : 0;
  }
  const int64_t r_est = vol == 1 ? V_out : (int64_t)vol * V_out / 3;
  if (r_est < (bf ? 150000 : 250000)) return 0;            // (bf16: 232 k rules 46 -> 38 us; fp32: 112 -> 116)
  int64_t n = (r_est >= 600000 ? 512 : 256) / tiles;
  if (n > slots) n = 256 / tiles;
  return n >= 1 && n <= slots ? (int)n : 0;
}
static void dw_tiling(int ci, int co, int &cb, int &nb, int &tiles) {
  int ncb = nnb_of(ci), nnb = nnb_of(co);
  cb = ncb >= 4 ? 4 : (ncb >= 2 ? 2 : 1);
  nb = nnb >= 4 ? 4 : (nnb >= 2 ? 2 : 1);
  tiles = (int)(ceil_div(ncb, cb) * ceil_div(nnb, nb));
}


} // namespace aabr

using namespace aabr;

// name of the kernel instance the last conv / dW entry point dispatched on this thread (bench provenance:
// `roofline.kernel` is what actually ran, not a string typed into the bench)
namespace aabr { thread_local const char *g_last_variant = ""; }
extern "C" const char *aabr_conv_last_variant(void) { return aabr::g_last_variant; }

extern "C" int64_t aabr_conv_wpack_floats(int vol, int n_in, int n_out) {
  // sized for either orientation (forward uses ci=n_in, the transposed pass ci=n_out)
  int64_t a = (int64_t)vol * nkc_of(n_in) * nnb_of(n_out) * 512;
  int64_t b = (int64_t)vol * nkc_of(n_out) * nnb_of(n_in) * 512;
  return a > b ? a : b;
}

extern "C" int64_t aabr_tile_blocks_words(int64_t V, int vol) {
  int64_t nt = tb_ntiles(V);
  return nt + nt * tb_maxb(vol) + nt * tb_maxb(vol) * 16;
}

extern "C" int aabr_build_tile_blocks(const int32_t *table, int64_t V, int vol, int32_t *blocks, void *stream_) {
  AABR_CHECK_ARG(V >= 0 && vol > 0 && vol <= 4096, "bad sizes");
  AABR_CHECK_ARG(V < (1ll << 25), "more than 2^25 sites per grid are not supported");
  if (V == 0) return AABR_OK;
  AABR_CHECK_ARG(table && blocks, "null pointer");
  const StreamJob job{table, nullptr, blocks, V, vol, 0};
  return launch_tile_blocks_jobs(&job, 1, (hipStream_t)stream_);
}

// job lists: the grid of a launch = the blocks of all its books back to back (StreamJobs::first)
template <typename F>
static int launch_stream_jobs(const StreamJob *jobs, int n, F blocks_of, void (*kernel)(const StreamJobs), hipStream_t st) {
  for (int j0 = 0; j0 < n;) {
    StreamJobs js;
    uint64_t blocks = 0;
    int m = 0;
    while (j0 + m < n && m < kStreamJobsMax) {
      const uint64_t b = (uint64_t)blocks_of(jobs[j0 + m]);
      if (m && blocks + b >= (1ull << 31)) break;
      js.j[m] = jobs[j0 + m];
      js.first[m] = (uint32_t)blocks;
      blocks += b;
      ++m;
    }
    js.n = m;
    js.first[m] = (uint32_t)blocks;
    AABR_CHECK_ARG(blocks < (1ull << 31), "too many blocks in one job list");
    if (blocks) hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(256), 0, st, js);
    j0 += m;
  }
  AABR_CHECK_LAUNCH();
  return AABR_OK;
}

int aabr::launch_tile_blocks_jobs(const StreamJob *jobs, int n, hipStream_t st) {
  return launch_stream_jobs(jobs, n, [](const StreamJob &j) { return ceil_div(tb_ntiles(j.V), 4); }, k_build_tile_blocks, st);
}

// the pair lists of n books: ONE scan launch (a block per book and offset), then ONE fill launch (the fill reads the
// R_k the scan wrote: stream order)
int aabr::launch_offset_pairs_jobs(const StreamJob *jobs, int n, hipStream_t st) {
  int rc = launch_stream_jobs(jobs, n, [](const StreamJob &j) { return (int64_t)j.vol; }, k_offset_bases, st);
  if (rc != AABR_OK) return rc;
  return launch_stream_jobs(jobs, n, [](const StreamJob &j) { return op_nb256(j.V) * j.vol; }, k_fill_offset_pairs, st);
}

extern "C" int64_t aabr_offset_pairs_words(int64_t V, int vol) {
  return op_hdr(vol) + (int64_t)vol * op_nb256(V) + 2 * (int64_t)vol * V;
}

extern "C" int aabr_build_offset_pairs(const int32_t *table, const int32_t *block_counts, int64_t V, int vol,
                                       int32_t *pairs, void *stream_) {
  hipStream_t st = (hipStream_t)stream_;
  AABR_CHECK_ARG(V >= 0 && vol > 0 && vol <= 65535 && pairs, "bad arguments");
  if (V == 0) {
    hipMemsetAsync(pairs, 0, (size_t)op_hdr(vol) * sizeof(int32_t), st);
    return AABR_OK;
  }
  AABR_CHECK_ARG(table && block_counts, "null pointer");
  const StreamJob job{table, block_counts, pairs, V, vol, 0};
  return launch_offset_pairs_jobs(&job, 1, st);
}

extern "C" int aabr_conv_forward(const float *in_feats, int n_in, int64_t rows_in, float *out_feats, int n_out,
                                 int64_t V_out, const int32_t *blocks, int vol, const float *W, const float *bias,
                                 int flags, float *wpack, void *stream_) {
  hipStream_t st = (hipStream_t)stream_;
  AABR_CHECK_ARG(n_in > 0 && n_out > 0 && vol > 0 && V_out >= 0 && rows_in >= 0, "bad sizes");
  AABR_CHECK_ARG(n_in <= 4096 && n_out <= 4096, "plane count too large");
  if (V_out == 0) return AABR_OK;
  AABR_CHECK_ARG(in_feats && out_feats && blocks && W && wpack && rows_in > 0, "null pointer / empty input");
  AABR_CHECK_ARG(((uintptr_t)in_feats & 15) == 0 && ((uintptr_t)out_feats & 15) == 0 &&
                     ((uintptr_t)wpack & 15) == 0,
                 "feature / scratch pointers must be 16-byte aligned");
  const int transpose = flags & 1, flip = ((flags >> 1) & 1) | ((flags >> 8) << 8);
  const int nkc = nkc_of(n_in), nnb = nnb_of(n_out);
  int64_t total = (int64_t)vol * nkc * nnb * 512;
  if (!(flags & 4)) // bit2: wpack already holds the packed weights of this (W, flags & 1) pair
    hipLaunchKernelGGL(k_pack_weights, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, st, W, vol, n_in,
                       n_out, transpose, 0, wpack);
  const bool aligned = (n_in % kKC) == 0;
  // WPB waves share the blocks of one 64-row tile.  Pick the split that minimises
  // (rounds of resident workgroups) x (blocks per wave): a grid one workgroup larger than what
  // fits on the chip at once would otherwise pay a whole second round.
  // lean buffer-descriptor kernel: aligned channels and every buffer below 2 GiB (32-bit offsets)
  const int64_t in_bytes = rows_in * n_in * 4, wp_bytes = total * 4,
                words_bytes = aabr_tile_blocks_words(V_out, vol) * 4;
  const bool lean_any = in_bytes < (1ll << 31) && wp_bytes < (1ll << 31) && words_bytes < (1ll << 31) && !(flags >> 8);
  const bool lean = aligned && lean_any;
  // LDS-resident weights: the slab's packed filter bank + enough per-wave output tiles fit in 160 KiB
  if (nkc <= 2 && in_bytes < (1ll << 31) && words_bytes < (1ll << 31) && !(flags >> 8)) {
    const int kv = knob(K_CONV_WLDS);          // tuning experiments only: 0 disables, 1/2/4 forces the slab width
    const int forced = kv == kKnobUnset ? -1 : kv;
    const int64_t ntiles = ceil_div(V_out, 64);
    // Measured (tools_conv_bench.py, S80k and 1.5 M points): 16-column slabs with 8 waves per CU beat
    // wider slabs (fewer waves fit beside the larger filter bank) and beat the streaming kernel when the
    // layer is at least 64 planes wide; narrow layers (<= 32 output planes) stay on the streaming kernel.
    int nbw = 0, nw = 0;
    for (int cand = 1; cand <= 4; cand <<= 1) {
      if (forced > 0 ? cand != forced : (cand != 1 || nnb < 4)) continue;
      if (cand > 1 && cand / 2 >= nnb) continue; // no wider than the layer
      const int64_t w_lds = (int64_t)vol * nkc * cand * 2048, tile_lds = (int64_t)64 * cand * 16 * 4;
      int64_t fit = (160 * 1024 - w_lds) / tile_lds;
      if (fit > 8) fit = 8; // 512 threads: two waves per SIMD, 256 VGPRs each
      if (fit < 4) continue;
      nbw = cand; nw = (int)fit;
      break;
    }
    if (nbw > 0 && forced != 0) {
      const int64_t slabs = ceil_div(nnb, nbw);
      const int64_t w_lds = (int64_t)vol * nkc * nbw * 2048, tile_lds = (int64_t)64 * nbw * 16 * 4;
      int64_t wgx = ceil_div(ntiles, nw);
      const int64_t cap = 256 / slabs > 0 ? 256 / slabs : 1; // persistent: about one workgroup per CU
      if (wgx > cap) wgx = cap;
      // a small tile count spreads over more CUs with fewer waves each
      while (nw > 4 && wgx < cap && ceil_div(ntiles, nw - 1) <= cap) { --nw; wgx = ceil_div(ntiles, nw); }
      const size_t lds = (size_t)(w_lds + nw * tile_lds);
#define AABR_LAUNCH_WLDS(NBW, NKC, AL)                                                                   \
  do {                                                                                                   \
    static DynLdsOnce attr_set;                                                                          \
    AABR_CHECK_HIP(dyn_lds_once(attr_set, (const void *)k_conv_blocks_mfma_wlds<NBW, NKC, AL>, 160 * 1024)); \
    g_last_variant = "k_conv_blocks_mfma_wlds<" #NBW "," #NKC "," #AL ">";                                \
    hipLaunchKernelGGL((k_conv_blocks_mfma_wlds<NBW, NKC, AL>), dim3((unsigned)wgx, (unsigned)slabs),    \
                       dim3(64 * nw), lds, st, in_feats, n_in, in_bytes, out_feats, n_out, V_out,        \
                       blocks, words_bytes, vol, flip & 1, wpack, bias);                                 \
  } while (0)
#define AABR_LAUNCH_WLDS_K(NBW)                                                                          \
  do {                                                                                                   \
    if (nkc == 1) { if (aligned) AABR_LAUNCH_WLDS(NBW, 1, true); else AABR_LAUNCH_WLDS(NBW, 1, false); } \
    else { if (aligned) AABR_LAUNCH_WLDS(NBW, 2, true); else AABR_LAUNCH_WLDS(NBW, 2, false); }          \
  } while (0)
      if (nbw == 1) AABR_LAUNCH_WLDS_K(1);
      else if (nbw == 2) AABR_LAUNCH_WLDS_K(2);
      else AABR_LAUNCH_WLDS_K(4);
#undef AABR_LAUNCH_WLDS_K
#undef AABR_LAUNCH_WLDS
      AABR_CHECK_LAUNCH();
      return AABR_OK;
    }
  }
  // tiny rule books with wide layers (coarse FPN scales): (pair, chunk) items over 8 waves x 16-column slabs
  {
    const int smax = knob(K_SMALL_MAX) == kKnobUnset ? 512 : knob(K_SMALL_MAX);
    if (lean && nkc >= 2 && ceil_div(V_out, 64) * nnb < smax && knob(K_CONV_SMALL) != 0) {   // CONV_SMALL=0 disables
      // 16 waves per workgroup when the grid alone cannot fill the chip (each wave's chain of dependent (pair,
      // chunk) items halves; 94 VGPRs: four waves per SIMD fit): SMALL_WPB forces 8 / 16
      int wpb = ceil_div(V_out, 64) * nnb < 1024 ? 16 : 8;
      if (knob(K_SMALL_WPB) == 8 || knob(K_SMALL_WPB) == 16) wpb = knob(K_SMALL_WPB);
      const dim3 grid((unsigned)ceil_div(V_out, 64), (unsigned)nnb);
      if (wpb == 16) {
        g_last_variant = "k_conv_blocks_mfma_small<16>";
        hipLaunchKernelGGL((k_conv_blocks_mfma_small<16>), grid, dim3(64 * 16), (size_t)16 * 64 * 16 * sizeof(float), st,
                           in_feats, n_in, in_bytes, out_feats, n_out, V_out, blocks, words_bytes, vol, flip & 1, wpack,
                           wp_bytes, bias);
      } else {
        g_last_variant = "k_conv_blocks_mfma_small<8>";
        hipLaunchKernelGGL((k_conv_blocks_mfma_small<8>), grid, dim3(64 * 8), (size_t)8 * 64 * 16 * sizeof(float), st,
                           in_feats, n_in, in_bytes, out_feats, n_out, V_out, blocks, words_bytes, vol, flip & 1, wpack,
                           wp_bytes, bias);
      }
      AABR_CHECK_LAUNCH();
      return AABR_OK;
    }
  }
#define AABR_LAUNCH_CONV(NBW, WPB)                                                                      \
  do {                                                                                                  \
    size_t lds = (size_t)(WPB) * 64 * ((NBW)*16) * sizeof(float);                                       \
    dim3 grid((unsigned)ceil_div(V_out, 64), (unsigned)ceil_div(nnb, (NBW)));                           \
    g_last_variant = (lean && (NBW) == 4 && wgs >= 8192) ? "k_conv_blocks_mfma_wpipe<" #NBW "," #WPB ",true,true>" \
                   : (lean && (NBW) == 4) ? "k_conv_blocks_mfma_wpipe<" #NBW "," #WPB ",true,false>"    \
                   : lean ? "k_conv_blocks_mfma_buf<" #NBW "," #WPB ",true,true>"                       \
                   : lean_any ? "k_conv_blocks_mfma_buf<" #NBW "," #WPB ",true,false>"                  \
                   : aligned ? "k_conv_blocks_mfma<" #NBW "," #WPB ",true>"                             \
                             : "k_conv_blocks_mfma<" #NBW "," #WPB ",false>";                           \
    if (lean && (NBW) == 4 && wgs >= 8192)                                                              \
      hipLaunchKernelGGL((k_conv_blocks_mfma_wpipe<NBW, WPB, true, true>), grid, dim3(64 * (WPB)), lds, st, \
                         in_feats, n_in, in_bytes, out_feats, n_out, V_out, blocks, words_bytes, vol,   \
                         flip & 1, wpack, wp_bytes, bias);                                              \
    else if (lean && (NBW) == 4)                                                                        \
      hipLaunchKernelGGL((k_conv_blocks_mfma_wpipe<NBW, WPB, true, false>), grid, dim3(64 * (WPB)), lds, st, \
                         in_feats, n_in, in_bytes, out_feats, n_out, V_out, blocks, words_bytes, vol,   \
                         flip & 1, wpack, wp_bytes, bias);                                              \
    else if (lean) /* adjacent-pair weight sharing pays at every size in the streaming kernel */        \
      hipLaunchKernelGGL((k_conv_blocks_mfma_buf<NBW, WPB, true, true>), grid, dim3(64 * (WPB)), lds, st, \
                         in_feats, n_in, in_bytes, out_feats, n_out, V_out, blocks, words_bytes, vol,   \
                         flip & 1, wpack, wp_bytes, bias);                                              \
    else if (lean_any) /* same kernel with element gathers: plane counts that are not multiples of 32 */ \
      hipLaunchKernelGGL((k_conv_blocks_mfma_buf<NBW, WPB, true, false>), grid, dim3(64 * (WPB)), lds, st, \
                         in_feats, n_in, in_bytes, out_feats, n_out, V_out, blocks, words_bytes, vol,   \
                         flip & 1, wpack, wp_bytes, bias);                                              \
    else if (aligned)                                                                                   \
      hipLaunchKernelGGL((k_conv_blocks_mfma<NBW, WPB, true>), grid, dim3(64 * (WPB)), lds, st,         \
                         in_feats, n_in, out_feats, n_out, V_out, blocks, vol, flip, wpack, bias);      \
    else                                                                                                \
      hipLaunchKernelGGL((k_conv_blocks_mfma<NBW, WPB, false>), grid, dim3(64 * (WPB)), lds, st,        \
                         in_feats, n_in, out_feats, n_out, V_out, blocks, vol, flip, wpack, bias);      \
  } while (0)
  // widest column slab the layer allows (fewest re-gathers of the input rows) -- unless the rule book is so
  // small that the launch would leave most CUs idle (the coarse FPN scales: 1-50 tiles): then narrower
  // slabs, i.e. more and shorter workgroups; at that size the gathers are latency, not bandwidth
  int nbw = nnb <= 1 ? 1 : (nnb == 2 ? 2 : 4);
  while (nbw > 1 && ceil_div(V_out, 64) * ceil_div(nnb, nbw) < 512) nbw >>= 1;
  // throughput-bound launches: 32-column slabs of the streaming kernel (twice the waves per CU beside half
  // the private LDS tile) edge out the 64-column weight-prefetch kernel: 39.4 -> 40.3 % / 41.9 -> 42.8 % of
  // the fp32 MFMA peak at 128 / 256 planes, 1.5 M points
  if (nbw == 4 && ceil_div(V_out, 64) * ceil_div(nnb, 4) >= 8192) nbw = 2;
  {                                              // tuning experiments only
    const int v = knob(K_CONV_NBW);
    if ((v == 1 || v == 2 || v == 4) && v <= nbw) nbw = v;
  }
  const int64_t wgs = ceil_div(V_out, 64) * ceil_div(nnb, nbw);
  int best_wpb = 2;
  int64_t best_cost = -1;
  for (int wpb = 2; wpb <= (nbw == 4 ? 3 : 4); ++wpb) {
    int64_t lds = (int64_t)wpb * 64 * (nbw * 16) * 4;
    int64_t per_cu = (160 * 1024) / lds;
    int64_t wave_cap = (nbw == 4 ? 12 : 20) / wpb; // register budget: 3 (164 VGPRs, weight prefetch) resp. 5 waves per SIMD
    if (per_cu > wave_cap) per_cu = wave_cap;
    if (per_cu < 1) per_cu = 1;
    int64_t rounds = ceil_div(wgs, 256 * per_cu);
    if (lds > 64 * 1024) continue; // default dynamic-LDS limit per workgroup
    int64_t cost = rounds * ceil_div(vol, wpb);
    if (best_cost < 0 || cost <= best_cost) { best_cost = cost; best_wpb = wpb; }
  }
  {                                              // tuning experiments only
    const int v = knob(K_CONV_WPB);
    if (v >= 2 && v <= (nbw == 4 ? 3 : 4)) best_wpb = v;
  }
  if (nbw == 1) {
    if (best_wpb == 2) AABR_LAUNCH_CONV(1, 2);
    else if (best_wpb == 3) AABR_LAUNCH_CONV(1, 3);
    else AABR_LAUNCH_CONV(1, 4);
  } else if (nbw == 2) {
    if (best_wpb == 2) AABR_LAUNCH_CONV(2, 2);
    else if (best_wpb == 3) AABR_LAUNCH_CONV(2, 3);
    else AABR_LAUNCH_CONV(2, 4);
  } else {
    if (best_wpb == 2) AABR_LAUNCH_CONV(4, 2);
    else AABR_LAUNCH_CONV(4, 3);
  }
#undef AABR_LAUNCH_CONV
  AABR_CHECK_LAUNCH();
  return AABR_OK;
}

extern "C" int64_t aabr_conv_dw_scratch_floats(int64_t max_chunks, int n_in, int n_out) {
  return max_chunks * n_in * n_out;
}

template <typename T>
static int conv_backward_weight_t(const T *in_feats, int n_in, const T *d_out, int n_out, int64_t V_out,
                                  const int32_t *pairs, int vol, int64_t max_chunks, float *dW, float *d_bias,
                                  float *scratch, void *stream_) {
  hipStream_t st = (hipStream_t)stream_;
  AABR_CHECK_ARG(n_in > 0 && n_out > 0 && vol > 0 && V_out >= 0 && vol <= 65535, "bad sizes");
  AABR_CHECK_ARG(dW, "null dW");
  int64_t cico = (int64_t)n_in * n_out;
  if (V_out == 0 || max_chunks == 0) {
    hipMemsetAsync(dW, 0, vol * cico * sizeof(float), st);
    if (d_bias) hipMemsetAsync(d_bias, 0, n_out * sizeof(float), st);
    return AABR_OK;
  }
  AABR_CHECK_ARG(in_feats && d_out && pairs && scratch && max_chunks > 0, "null pointer");
  int cb, nb, tiles;
  dw_tiling(n_in, n_out, cb, nb, tiles);
  AABR_CHECK_ARG(tiles <= 65535, "too many tiles");
  const int chunk_pairs = dw_chunk(V_out, vol, n_in, n_out);
  // an offset has at most V_out rules: with V_out <= chunk_pairs every offset is one chunk at most, workgroup x is offset
  // x, writes dW[x] itself and no reduce follows
  const int direct = V_out <= chunk_pairs ? 1 : 0;
  float *dst = direct ? dW : scratch;
  dim3 grid((unsigned)(direct ? vol : max_chunks), (unsigned)tiles);
  const int n_wg = direct ? 0 : dw_full_workgroups(n_in, n_out, in_feats, d_out, max_chunks, vol, V_out, false);
  if (sizeof(T) == 4 && n_wg) {
    g_last_variant = "k_conv_dw_full_f32";
    hipLaunchKernelGGL(k_conv_dw_full_f32, dim3((unsigned)n_wg, (unsigned)((n_in >> 7) * (n_out >> 7))), dim3(256), 0, st,
                       reinterpret_cast<const float *>(in_feats), n_in, reinterpret_cast<const float *>(d_out), n_out,
                       V_out, pairs, vol, scratch);
    hipLaunchKernelGGL(k_conv_dw_reduce_ranges, dim3((unsigned)ceil_div(cico, 64), (unsigned)vol), dim3(256), 0, st,
                       scratch, pairs, vol, n_wg, cico, dW);
    if (d_bias) launch_col_sum<T>(d_out, V_out, n_out, d_bias, scratch, max_chunks * cico, st);
    AABR_CHECK_LAUNCH();
    return AABR_OK;
  }
#define AABR_LAUNCH_DW(CB, NB)                                                                           \
  do {                                                                                                   \
    g_last_variant = sizeof(T) == 4 ? "k_conv_dw_pairs<" #CB "," #NB ",float>" : "k_conv_dw_pairs<" #CB "," #NB ",bf16>"; \
    hipLaunchKernelGGL((k_conv_dw_pairs<CB, NB, T>), grid, dim3(256), 0, st, in_feats, n_in, d_out, n_out, \
                       V_out, pairs, vol, chunk_pairs, dst, direct);                                     \
  } while (0)
  if (cb == 1 && nb == 1) AABR_LAUNCH_DW(1, 1);
  else if (cb == 1 && nb == 2) AABR_LAUNCH_DW(1, 2);
  else if (cb == 1 && nb == 4) AABR_LAUNCH_DW(1, 4);
  else if (cb == 2 && nb == 1) AABR_LAUNCH_DW(2, 1);
  else if (cb == 2 && nb == 2) AABR_LAUNCH_DW(2, 2);
  else if (cb == 2 && nb == 4) AABR_LAUNCH_DW(2, 4);
  else if (cb == 4 && nb == 1) AABR_LAUNCH_DW(4, 1);
  else if (cb == 4 && nb == 2) AABR_LAUNCH_DW(4, 2);
  else AABR_LAUNCH_DW(4, 4);
#undef AABR_LAUNCH_DW
  if (!direct)
    hipLaunchKernelGGL(k_conv_dw_reduce, dim3((unsigned)ceil_div(cico, 64), (unsigned)vol), dim3(256), 0, st,
                       scratch, pairs, vol, chunk_pairs, cico, dW);
  if (d_bias) launch_col_sum<T>(d_out, V_out, n_out, d_bias, scratch, max_chunks * cico, st);
  AABR_CHECK_LAUNCH();
  return AABR_OK;
}

extern "C" int aabr_conv_backward_weight(const float *in_feats, int n_in, const float *d_out, int n_out,
                                         int64_t V_out, const int32_t *pairs, int vol, int64_t max_chunks,
                                         float *dW, float *d_bias, float *scratch, void *stream_) {
  return conv_backward_weight_t<float>(in_feats, n_in, d_out, n_out, V_out, pairs, vol, max_chunks, dW, d_bias,
                                       scratch, stream_);
}

extern "C" int aabr_conv_backward_weight_bf16(const uint16_t *in_feats, int n_in, const uint16_t *d_out,
                                              int n_out, int64_t V_out, const int32_t *pairs, int vol,
                                              int64_t max_chunks, float *dW, float *d_bias, float *scratch,
                                              void *stream_) {
  if (n_in > 0 && n_out > 0 && n_in % 32 == 0 && n_out % 32 == 0 && V_out > 0 && max_chunks > 0 &&
      ((uintptr_t)in_feats & 15) == 0 && ((uintptr_t)d_out & 15) == 0) {
    // bf16 MFMA with LDS-transposed operands
    hipStream_t st = (hipStream_t)stream_;
    AABR_CHECK_ARG(vol > 0 && vol <= 65535 && n_in <= 4096 && n_out <= 4096, "bad sizes");
    AABR_CHECK_ARG(in_feats && d_out && pairs && scratch && dW, "null pointer");
    const __bf16 *in16 = reinterpret_cast<const __bf16 *>(in_feats), *do16 = reinterpret_cast<const __bf16 *>(d_out);
    const int ncb = nnb_of(n_in), nnb = nnb_of(n_out);
    const int cb = ncb >= 4 ? 4 : 2, nb = nnb >= 4 ? 4 : 2; // plane counts are multiples of 32
    const int tiles = (int)(ceil_div(ncb, cb) * ceil_div(nnb, nb));
    AABR_CHECK_ARG(tiles <= 65535, "too many tiles");
    const int chunk_pairs = dw_chunk(V_out, vol, n_in, n_out);
    const int64_t cico = (int64_t)n_in * n_out;
    const int direct = V_out <= chunk_pairs ? 1 : 0;       // (as in conv_backward_weight_t)
    float *dst = direct ? dW : scratch;
    dim3 grid((unsigned)(direct ? vol : max_chunks), (unsigned)tiles);
    const int n_wg = direct ? 0 : dw_full_workgroups(n_in, n_out, in_feats, d_out, max_chunks, vol, V_out, true);
    if (n_wg) {
      g_last_variant = "k_conv_dw_full_bf16";
      hipLaunchKernelGGL(k_conv_dw_full_bf16, dim3((unsigned)n_wg, (unsigned)((n_in >> 7) * (n_out >> 7))), dim3(256), 0,
                         st, in16, n_in, do16, n_out, V_out, pairs, vol, scratch);
      hipLaunchKernelGGL(k_conv_dw_reduce_ranges, dim3((unsigned)ceil_div(cico, 64), (unsigned)vol), dim3(256), 0, st,
                         scratch, pairs, vol, n_wg, cico, dW);
      if (d_bias) launch_col_sum<__bf16>(do16, V_out, n_out, d_bias, scratch, max_chunks * cico, st);
      AABR_CHECK_LAUNCH();
      return AABR_OK;
    }
#define AABR_LAUNCH_DW16(CB, NB)                                                                          \
  do {                                                                                                    \
    g_last_variant = "k_conv_dw_pairs_bf16<" #CB "," #NB ">";                                             \
    hipLaunchKernelGGL((k_conv_dw_pairs_bf16<CB, NB>), grid, dim3(256), 0, st, in16, n_in, do16, n_out, V_out, \
                       pairs, vol, chunk_pairs, dst, direct);                                             \
  } while (0)
    if (cb == 2 && nb == 2) AABR_LAUNCH_DW16(2, 2);
    else if (cb == 2 && nb == 4) AABR_LAUNCH_DW16(2, 4);
    else if (cb == 4 && nb == 2) AABR_LAUNCH_DW16(4, 2);
    else AABR_LAUNCH_DW16(4, 4);
#undef AABR_LAUNCH_DW16
    if (!direct)
      hipLaunchKernelGGL(k_conv_dw_reduce, dim3((unsigned)ceil_div(cico, 64), (unsigned)vol), dim3(256), 0, st,
                         scratch, pairs, vol, chunk_pairs, cico, dW);
    if (d_bias) launch_col_sum<__bf16>(do16, V_out, n_out, d_bias, scratch, max_chunks * cico, st);
    AABR_CHECK_LAUNCH();
    return AABR_OK;
  }
  return conv_backward_weight_t<__bf16>(reinterpret_cast<const __bf16 *>(in_feats), n_in,
                                        reinterpret_cast<const __bf16 *>(d_out), n_out, V_out, pairs, vol,
                                        max_chunks, dW, d_bias, scratch, stream_);
}

extern "C" int64_t aabr_conv_wpack_bf16_elems(int vol, int n_in, int n_out) {
  return aabr_conv_wpack_floats(vol, n_in, n_out);
}

extern "C" int aabr_conv_forward_bf16(const uint16_t *in_feats, int n_in, int64_t rows_in, uint16_t *out_feats,
                                      int n_out, int64_t V_out, const int32_t *blocks, int vol, const float *W,
                                      const float *bias, int flags, uint16_t *wpack, void *stream_) {
  hipStream_t st = (hipStream_t)stream_;
  AABR_CHECK_ARG(n_in > 0 && n_out > 0 && vol > 0 && V_out >= 0 && rows_in >= 0, "bad sizes");
  AABR_CHECK_ARG(n_in <= 4096 && n_out <= 4096, "plane count too large");
  AABR_CHECK_ARG(n_in % kKC == 0 && n_out % kKC == 0, "bf16 features need plane counts divisible by 32");
  if (V_out == 0) return AABR_OK;
  AABR_CHECK_ARG(in_feats && out_feats && blocks && W && wpack && rows_in > 0, "null pointer / empty input");
  AABR_CHECK_ARG(((uintptr_t)in_feats & 15) == 0 && ((uintptr_t)out_feats & 15) == 0 &&
                     ((uintptr_t)wpack & 15) == 0,
                 "feature / scratch pointers must be 16-byte aligned");
  const int transpose = flags & 1, flip = (flags >> 1) & 1;
  const int nkc = nkc_of(n_in), nnb = nnb_of(n_out);
  const int64_t total = (int64_t)vol * nkc * nnb * 512;
  const int64_t in_bytes = rows_in * n_in * 2, wp_bytes = total * 2,
                words_bytes = aabr_tile_blocks_words(V_out, vol) * 4;
  AABR_CHECK_ARG(in_bytes < (1ll << 31) && wp_bytes < (1ll << 31) && words_bytes < (1ll << 31),
                 "bf16 convolution addresses its buffers with 32-bit offsets (each must be < 2 GiB)");
  __bf16 *wp = reinterpret_cast<__bf16 *>(wpack);
  if (!(flags & 4))
    hipLaunchKernelGGL(k_pack_weights_bf16, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, st, W, vol, n_in,
                       n_out, transpose, wp);
  int nbw = nnb == 2 ? 2 : 4;
  if (nbw == 4 && ceil_div(V_out, 64) * ceil_div(nnb, 4) < 512) nbw = 2; // small rule book: more, shorter workgroups
  {                                              // tuning experiments only
    const int v = knob(K_CONV_NBW);
    if ((v == 2 || v == 4) && v <= nbw) nbw = v;
  }
  const int kg = nkc >= 3 ? 4 : nkc;
  const int64_t wgs = ceil_div(V_out, 64) * ceil_div(nnb, nbw);
  int best_wpb = 2;
  int64_t best_cost = -1;
  for (int wpb = 2; wpb <= (nbw == 4 ? 3 : 4); ++wpb) {
    int64_t lds = (int64_t)wpb * 64 * (nbw * 16) * 4;
    int64_t per_cu = (160 * 1024) / lds;
    int64_t wave_cap = 16 / wpb;
    if (per_cu > wave_cap) per_cu = wave_cap;
    if (per_cu < 1) per_cu = 1;
    int64_t rounds = ceil_div(wgs, 256 * per_cu);
    int64_t cost = rounds * ceil_div(vol, wpb);
    if (best_cost < 0 || cost <= best_cost) { best_cost = cost; best_wpb = wpb; }
  }
  {                                              // tuning experiments only
    const int v = knob(K_CONV_WPB);
    if (v >= 2 && v <= (nbw == 4 ? 3 : 4)) best_wpb = v;
  }
#define AABR_LAUNCH_CONV16(NBW, WPB, KG)                                                                 \
  do {                                                                                                   \
    size_t lds = (size_t)(WPB) * 64 * ((NBW)*16) * sizeof(float);                                        \
    dim3 grid((unsigned)ceil_div(V_out, 64), (unsigned)ceil_div(nnb, (NBW)));                            \
    g_last_variant = "k_conv_blocks_mfma_bf16<" #NBW "," #WPB "," #KG ",true>";                          \
    hipLaunchKernelGGL((k_conv_blocks_mfma_bf16<NBW, WPB, KG, true>), grid, dim3(64 * (WPB)), lds, st,   \
                       reinterpret_cast<const __bf16 *>(in_feats), n_in, in_bytes,                       \
                       reinterpret_cast<__bf16 *>(out_feats), n_out, V_out, blocks, words_bytes, vol,    \
                       flip, wp, wp_bytes, bias);                                                        \
  } while (0)
#define AABR_LAUNCH_CONV16_KG(NBW, WPB)                                                                  \
  do {                                                                                                   \
    if (kg == 1) AABR_LAUNCH_CONV16(NBW, WPB, 1);                                                        \
    else if (kg == 2) AABR_LAUNCH_CONV16(NBW, WPB, 2);                                                   \
    else AABR_LAUNCH_CONV16(NBW, WPB, 4);                                                                \
  } while (0)
  if (nbw == 2) {
    if (best_wpb == 2) AABR_LAUNCH_CONV16_KG(2, 2);
    else if (best_wpb == 3) AABR_LAUNCH_CONV16_KG(2, 3);
    else AABR_LAUNCH_CONV16_KG(2, 4);
  } else {
    if (best_wpb == 2) AABR_LAUNCH_CONV16_KG(4, 2);
    else AABR_LAUNCH_CONV16_KG(4, 3);
  }
#undef AABR_LAUNCH_CONV16_KG
#undef AABR_LAUNCH_CONV16
  AABR_CHECK_LAUNCH();
  return AABR_OK;
}

// ---- weight packing as its own entry point (training: pack both orientations once per step) --------
template <typename TO>
static int conv_pack_weights2_t(const float *W, int vol, int n_in, int n_out, TO *wpack_fwd, TO *wpack_t,
                                void *stream_) {
  AABR_CHECK_ARG(n_in > 0 && n_out > 0 && vol > 0 && n_in <= 4096 && n_out <= 4096, "bad sizes");
  AABR_CHECK_ARG(W && wpack_fwd && wpack_t, "null pointer");
  AABR_CHECK_ARG(((uintptr_t)wpack_fwd & 15) == 0 && ((uintptr_t)wpack_t & 15) == 0, "packs must be 16-byte aligned");
  const int64_t total = (int64_t)vol * 512 *
                        ((int64_t)nkc_of(n_in) * nnb_of(n_out) + (int64_t)nkc_of(n_out) * nnb_of(n_in));
  hipLaunchKernelGGL((k_pack_weights2<TO>), dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream_,
                     W, vol, n_in, n_out, wpack_fwd, wpack_t);
  AABR_CHECK_LAUNCH();
  return AABR_OK;
}

// single orientation (inference / stand-alone launches of the 256-row-tile kernel)
extern "C" int aabr_conv_pack_weights(const float *W, int vol, int n_in, int n_out, int transpose, float *wpack,
                                      void *stream_) {
  AABR_CHECK_ARG(n_in > 0 && n_out > 0 && vol > 0 && n_in <= 4096 && n_out <= 4096, "bad sizes");
  AABR_CHECK_ARG(W && wpack && ((uintptr_t)wpack & 15) == 0, "null / misaligned pointer");
  const int64_t total = (int64_t)vol * nkc_of(n_in) * nnb_of(n_out) * 512;
  hipLaunchKernelGGL(k_pack_weights, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream_, W, vol,
                     n_in, n_out, transpose, 0, wpack);
  AABR_CHECK_LAUNCH();
  return AABR_OK;
}

extern "C" int aabr_conv_pack_weights2(const float *W, int vol, int n_in, int n_out, float *wpack_fwd,
                                       float *wpack_t, void *stream_) {
  return conv_pack_weights2_t<float>(W, vol, n_in, n_out, wpack_fwd, wpack_t, stream_);
}

extern "C" int aabr_conv_pack_weights2_bf16(const float *W, int vol, int n_in, int n_out, uint16_t *wpack_fwd,
                                            uint16_t *wpack_t, void *stream_) {
  return conv_pack_weights2_t<__bf16>(W, vol, n_in, n_out, reinterpret_cast<__bf16 *>(wpack_fwd),
                                      reinterpret_cast<__bf16 *>(wpack_t), stream_);
}

// thread blocks of one job of aabr_conv_pack_weights_jobs (both orientations)
extern "C" int64_t aabr_conv_pack_job_blocks(int vol, int n_in, int n_out) {
  return (int64_t)vol * ceil_div(n_in, kPT) * ceil_div(n_out, kPT);   // one workgroup per 64 x 64 tile of every W[k]
}

extern "C" int aabr_conv_pack_weights_jobs(const void *jobs_dev, int n_jobs, int64_t total_blocks, void *stream_) {
  AABR_CHECK_ARG(n_jobs >= 0 && total_blocks >= 0 && total_blocks < (1ll << 31), "bad sizes");
  if (n_jobs == 0 || total_blocks == 0) return AABR_OK;
  AABR_CHECK_ARG(jobs_dev && ((uintptr_t)jobs_dev & 7) == 0, "null / misaligned job table");
  hipLaunchKernelGGL(k_pack_weights_jobs, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream_,
                     reinterpret_cast<const PackJob *>(jobs_dev), n_jobs);
  AABR_CHECK_LAUNCH();
  return AABR_OK;
}

extern "C" int aabr_conv_dw_chunk_pairs(int64_t V_out, int vol, int n_in, int n_out) {
  return dw_chunk(V_out, vol, n_in, n_out);
}
