// conv.hip -- sparse 3-D convolution on gather tables, fp32 MFMA (gfx950).
//
// One formulation serves SubmanifoldConvolution, Convolution and Deconvolution, forward and
// input-gradient (reference: SCN/CPU/Convolution.cpp:45-185, SCN/CPU/Deconvolution.cpp:7-77,
// SCN/CUDA/Convolution.cu:57-667):
//
//     out[o] = bias + sum_k  in[ table[k][o] ] @ Wl[k]          (table entry -1 -> no term)
//
// It is OUTPUT-STATIONARY: a wave owns 64 consecutive output rows and a slab of output
// columns, keeps that tile in LDS for the whole sweep over the filter offsets and writes it to
// HBM exactly once -- no read-modify-write of `out` per offset and no atomics (the reference's
// GPU path re-reads and re-writes `out` 27 times and its rule book crosses PCIe per offset).
// Per offset the wave ballots which of its 64 rows have a partner, compacts those (row,
// partner) pairs, and multiplies them 16 at a time on v_mfma_f32_16x16x4_f32:
//     D^T[out col][pair] += Wl^T[out col][c] * in[partner(pair)][c]
// so each lane ends up with 4 consecutive output columns of one pair (one 16-byte LDS
// read-add-write).  Accumulation order is fixed (offset order, then channel order inside the
// MFMA chain) => results are bit-reproducible run to run.
//
// fp32 MFMA == k-ordered fmaf chain (exact fp32); peak 157 TFLOP/s.
#include "common.h"

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

// One workgroup owns a tile of 64 output rows x (NBW*16) output columns.  Its WPB waves split
// the filter offsets between them (wave w takes k = w, w+WPB, ...) so that WPB latency chains
// (table -> ballot -> gather -> MFMA) are in flight per tile; each wave accumulates into a
// private LDS tile and the tiles are summed in a fixed order at the end (deterministic).
// ALIGNED: ci % 32 == 0 (float4 gathers, no bounds checks).
template <int NBW, int WPB, bool ALIGNED>
__global__ __launch_bounds__(WPB * 64) void k_conv_gather_mfma(const float *__restrict__ in, int ci,
                                                              float *__restrict__ out, int co,
                                                              int64_t V_out, const int32_t *__restrict__ table,
                                                              int vol, const float *__restrict__ Wp,
                                                              const float *__restrict__ bias) {
  constexpr int WS = NBW * 16 + 4;          // C-tile row stride (floats), keeps 16-B alignment
  constexpr int TILE = 65 * WS + 128;       // floats per wave: 64 rows + 1 dummy row + pair lists
  extern __shared__ __align__(16) float smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int g = lane >> 4, c16 = lane & 15;
  float *Ct = smem + (size_t)wave * TILE;
  int32_t *listIn = reinterpret_cast<int32_t *>(Ct + 65 * WS);
  int32_t *listRow = listIn + 64;

  const int nkc = nkc_of(ci), nnb = nnb_of(co);
  const int nb0 = blockIdx.y * NBW;                       // first global n-block of the slab
  const int64_t row0 = (int64_t)blockIdx.x * 64;
  const int64_t myrow = row0 + lane;
  const bool valid = myrow < V_out;

  for (int i = lane; i < 65 * WS; i += 64) Ct[i] = 0.0f;

  int t_next = (valid && wave < vol) ? table[(int64_t)wave * V_out + myrow] : -1;
  for (int k = wave; k < vol; k += WPB) {
    const int t = t_next;
    if (k + WPB < vol) t_next = valid ? table[(int64_t)(k + WPB) * V_out + myrow] : -1;
    const unsigned long long m = __ballot(t >= 0);
    if (m == 0) continue;
    const int cnt = __popcll(m);
    const int pos = __popcll(m & ((1ull << lane) - 1ull));
    if (t >= 0) { listIn[pos] = t; listRow[pos] = lane; }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const float *Wk = Wp + (int64_t)k * nkc * nnb * 512;
    for (int mb = 0; mb * 16 < cnt; ++mb) {
      const int p = mb * 16 + c16;
      const int inrow = (p < cnt) ? listIn[p] : -1;
      const int orow = (p < cnt) ? listRow[p] : 64;
      f32x4 acc[NBW];
#pragma unroll
      for (int j = 0; j < NBW; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      for (int kc = 0; kc < nkc; ++kc) {
        float b8[8];
        if (ALIGNED) {
          if (inrow >= 0) {
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
            b8[s] = (inrow >= 0 && c < ci) ? in[(int64_t)inrow * ci + c] : 0.0f;
          }
        }
#pragma unroll
        for (int j = 0; j < NBW; ++j) {
          if (nb0 + j < nnb) {
            const float4 *wsrc =
                reinterpret_cast<const float4 *>(Wk + (((int64_t)kc * nnb + nb0 + j) * 64 + lane) * 8);
            float4 w0 = wsrc[0], w1 = wsrc[1];
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.x, b8[0], acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.y, b8[1], acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.z, b8[2], acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.w, b8[3], acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.x, b8[4], acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.y, b8[5], acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.z, b8[6], acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.w, b8[7], acc[j], 0, 0, 0);
          }
        }
      }
      // lane holds D^T[out col = j*16 + g*4 + r][pair c16]: 4 consecutive columns of row orow
#pragma unroll
      for (int j = 0; j < NBW; ++j) {
        float4 *dst = reinterpret_cast<float4 *>(Ct + orow * WS + j * 16 + g * 4);
        float4 cur = *dst;
        cur.x += acc[j][0]; cur.y += acc[j][1]; cur.z += acc[j][2]; cur.w += acc[j][3];
        *dst = cur;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  }
  __syncthreads();
  // combine the per-wave tiles in wave order (+ bias, CPU/Convolution.cpp:59-62) and write once
  const int nt = vol < WPB ? vol : WPB;
  const int nrows = (int)((V_out - row0) < 64 ? (V_out - row0) : 64);
  const int wcols = (co - nb0 * 16) < NBW * 16 ? (co - nb0 * 16) : NBW * 16;
  if ((co & 3) == 0) {
    const int q = wcols >> 2; // float4 per row
    for (int i = threadIdx.x; i < nrows * q; i += WPB * 64) {
      int r = i / q, cq = i % q;
      float4 v = *reinterpret_cast<const float4 *>(smem + r * WS + cq * 4);
      for (int w = 1; w < nt; ++w) {
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
      for (int w = 1; w < nt; ++w) v += smem[(size_t)w * TILE + r * WS + cc];
      if (bias) v += bias[nb0 * 16 + cc];
      out[(row0 + r) * co + nb0 * 16 + cc] = v;
    }
  }
}

// ----------------------------------------------------------------------------- dW
// partial[chunk][k][c][n] = sum over the chunk's output rows o with t = table[k][o] >= 0 of
// in[t][c] * d_out[o][n];  MFMA with the reduction (pair) index as K.  CB x NB blocks of 16.
template <int CB, int NB>
__global__ __launch_bounds__(256) void k_conv_dw_partial(const float *__restrict__ in, int ci,
                                                         const float *__restrict__ d_out, int co,
                                                         int64_t V_out, const int32_t *__restrict__ table,
                                                         int vol, int nchunks, int64_t rows_per_chunk,
                                                         float *__restrict__ partial) {
  __shared__ int32_t lists[4][2][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int g = lane >> 4, c16 = lane & 15;
  const int ncb = nnb_of(ci), nnb = nnb_of(co);
  const int tiles_n = (nnb + NB - 1) / NB;
  const int tile = blockIdx.z;
  const int cb0 = (tile / tiles_n) * CB, nb0 = (tile % tiles_n) * NB;
  const int k = blockIdx.y;
  const int chunk = blockIdx.x * 4 + wave;
  if (chunk >= nchunks) return;
  int32_t *lt = lists[wave][0], *lo = lists[wave][1];

  f32x4 acc[CB][NB];
#pragma unroll
  for (int a = 0; a < CB; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int64_t r_begin = (int64_t)chunk * rows_per_chunk;
  int64_t r_end = r_begin + rows_per_chunk;
  if (r_end > V_out) r_end = V_out;
  for (int64_t base = r_begin; base < r_end; base += 64) {
    const int64_t row = base + lane;
    const int t = (row < r_end) ? table[(int64_t)k * V_out + row] : -1;
    const unsigned long long m = __ballot(t >= 0);
    if (m == 0) continue;
    const int cnt = __popcll(m);
    const int pos = __popcll(m & ((1ull << lane) - 1ull));
    if (t >= 0) { lt[pos] = t; lo[pos] = (int32_t)(row - base); }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    for (int q0 = 0; q0 < cnt; q0 += 16) {
      // 16 pairs per step: issue every gather first (4 independent K-steps in flight), then the MFMAs
      float av[4][CB], bv[4][NB];
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        const int q = q0 + st * 4 + g;
        const bool on = q < cnt;
        const int64_t tq = on ? lt[q] : 0;
        const int64_t oq = on ? base + lo[q] : 0;
#pragma unroll
        for (int a = 0; a < CB; ++a) {
          int c = (cb0 + a) * 16 + c16;
          av[st][a] = (on && c < ci) ? in[tq * ci + c] : 0.0f;
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          int n = (nb0 + b) * 16 + c16;
          bv[st][b] = (on && n < co) ? d_out[oq * co + n] : 0.0f;
        }
      }
#pragma unroll
      for (int st = 0; st < 4; ++st)
#pragma unroll
        for (int a = 0; a < CB; ++a)
#pragma unroll
          for (int b = 0; b < NB; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[st][a], bv[st][b], acc[a][b], 0, 0, 0);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  }
  // D[i = c (row of dW) = g*4 + r][j = n = c16]
  float *P = partial + ((int64_t)chunk * vol + k) * ci * co;
#pragma unroll
  for (int a = 0; a < CB; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int c = (cb0 + a) * 16 + g * 4 + r, n = (nb0 + b) * 16 + c16;
        if (c < ci && n < co) P[(int64_t)c * co + n] = acc[a][b][r];
      }
}

__global__ __launch_bounds__(256) void k_conv_dw_reduce(const float *__restrict__ partial, int nchunks,
                                                        int64_t elems, float *__restrict__ dW) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= elems) return;
  float s = 0.0f;
#pragma unroll 8
  for (int c = 0; c < nchunks; ++c) s += partial[(int64_t)c * elems + i]; // fixed order: deterministic
  dW[i] = s;
}

// d_bias[n] = sum_rows d_out[row][n] (at::sum_out, CPU/Convolution.cpp:100-101); one block per
// 64 columns, fixed-order tree => deterministic.
__global__ __launch_bounds__(256) void k_col_sum(const float *__restrict__ x, int64_t rows, int co,
                                                 float *__restrict__ out) {
  __shared__ float red[4][64];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6;
  float s = 0.0f;
  if (col < co)
    for (int64_t r = part; r < rows; r += 4) s += x[r * co + col];
  red[part][threadIdx.x & 63] = s;
  __syncthreads();
  if (part == 0 && col < co)
    out[col] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

static int dw_chunks(int64_t V_out, int vol, int tiles) {
  int64_t want = 4096 / ((int64_t)vol * tiles); // ~4 waves per SIMD over the chip
  if (want < 1) want = 1;
  int64_t maxc = ceil_div(V_out, 256);
  if (maxc < 1) maxc = 1;
  if (want > maxc) want = maxc;
  if (want > 128) want = 128;
  return (int)want;
}
static void dw_tiling(int ci, int co, int &cb, int &nb, int &tiles) {
  int ncb = nnb_of(ci), nnb = nnb_of(co);
  cb = ncb >= 4 ? 4 : (ncb >= 2 ? 2 : 1);
  nb = nnb >= 4 ? 4 : (nnb >= 2 ? 2 : 1);
  tiles = (int)(ceil_div(ncb, cb) * ceil_div(nnb, nb));
}

} // namespace aabr

using namespace aabr;

extern "C" int64_t aabr_conv_wpack_floats(int vol, int n_in, int n_out) {
  // sized for either orientation (forward uses ci=n_in, the transposed pass ci=n_out)
  int64_t a = (int64_t)vol * nkc_of(n_in) * nnb_of(n_out) * 512;
  int64_t b = (int64_t)vol * nkc_of(n_out) * nnb_of(n_in) * 512;
  return a > b ? a : b;
}

extern "C" int aabr_conv_forward(const float *in_feats, int n_in, float *out_feats, int n_out, int64_t V_out,
                                 const int32_t *table, int vol, const float *W, const float *bias, int flags,
                                 float *wpack, void *stream_) {
  hipStream_t st = (hipStream_t)stream_;
  AABR_CHECK_ARG(n_in > 0 && n_out > 0 && vol > 0 && V_out >= 0, "bad sizes");
  AABR_CHECK_ARG(n_in <= 4096 && n_out <= 4096, "plane count too large");
  if (V_out == 0) return AABR_OK;
  AABR_CHECK_ARG(in_feats && out_feats && table && W && wpack, "null pointer");
  AABR_CHECK_ARG(((uintptr_t)in_feats & 15) == 0 && ((uintptr_t)out_feats & 15) == 0 &&
                     ((uintptr_t)wpack & 15) == 0,
                 "feature / scratch pointers must be 16-byte aligned");
  const int transpose = flags & 1, flip = (flags >> 1) & 1;
  const int nkc = nkc_of(n_in), nnb = nnb_of(n_out);
  int64_t total = (int64_t)vol * nkc * nnb * 512;
  if (!(flags & 4)) // bit2: wpack already holds the packed weights of this (W, flags) pair
    hipLaunchKernelGGL(k_pack_weights, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, st, W, vol, n_in,
                       n_out, transpose, flip, wpack);
  const bool aligned = (n_in % kKC) == 0;
  // WPB waves split the filter offsets of one 64-row tile.  Pick the split that minimises
  // (rounds of resident workgroups) x (offsets per wave): a grid one workgroup larger than what
  // fits on the chip at once would otherwise pay a whole second round.
#define AABR_LAUNCH_CONV(NBW, WPB)                                                                      \
  do {                                                                                                  \
    size_t lds = (size_t)(WPB) * (65 * ((NBW)*16 + 4) + 128) * sizeof(float);                           \
    dim3 grid((unsigned)ceil_div(V_out, 64), (unsigned)ceil_div(nnb, (NBW)));                           \
    if (aligned)                                                                                        \
      hipLaunchKernelGGL((k_conv_gather_mfma<NBW, WPB, true>), grid, dim3(64 * (WPB)), lds, st,         \
                         in_feats, n_in, out_feats, n_out, V_out, table, vol, wpack, bias);             \
    else                                                                                                \
      hipLaunchKernelGGL((k_conv_gather_mfma<NBW, WPB, false>), grid, dim3(64 * (WPB)), lds, st,        \
                         in_feats, n_in, out_feats, n_out, V_out, table, vol, wpack, bias);             \
  } while (0)
  const int nbw = nnb <= 1 ? 1 : (nnb == 2 ? 2 : 4);
  const int64_t wgs = ceil_div(V_out, 64) * ceil_div(nnb, nbw);
  int best_wpb = 2;
  int64_t best_cost = -1;
  for (int wpb = 2; wpb <= (nbw == 4 ? 3 : 4); ++wpb) {
    int64_t lds = (int64_t)wpb * (65 * (nbw * 16 + 4) + 128) * 4;
    int64_t per_cu = (160 * 1024) / lds;
    if (per_cu > 32 / wpb) per_cu = 32 / wpb;
    int64_t rounds = ceil_div(wgs, 256 * per_cu);
    int64_t cost = rounds * ceil_div(vol, wpb);
    if (best_cost < 0 || cost <= best_cost) { best_cost = cost; best_wpb = wpb; }
  }
  if (nbw == 1) {
    if (best_wpb == 2) AABR_LAUNCH_CONV(1, 2); else if (best_wpb == 3) AABR_LAUNCH_CONV(1, 3); else AABR_LAUNCH_CONV(1, 4);
  } else if (nbw == 2) {
    if (best_wpb == 2) AABR_LAUNCH_CONV(2, 2); else if (best_wpb == 3) AABR_LAUNCH_CONV(2, 3); else AABR_LAUNCH_CONV(2, 4);
  } else {
    if (best_wpb == 2) AABR_LAUNCH_CONV(4, 2); else AABR_LAUNCH_CONV(4, 3);
  }
#undef AABR_LAUNCH_CONV
  AABR_CHECK_LAUNCH();
  return AABR_OK;
}

extern "C" int64_t aabr_conv_dw_scratch_floats(int64_t V_out, int vol, int n_in, int n_out) {
  int cb, nb, tiles;
  dw_tiling(n_in, n_out, cb, nb, tiles);
  return (int64_t)dw_chunks(V_out, vol, tiles) * vol * n_in * n_out;
}

extern "C" int aabr_conv_backward_weight(const float *in_feats, int n_in, const float *d_out, int n_out,
                                         int64_t V_out, const int32_t *table, int vol, float *dW,
                                         float *d_bias, float *scratch, void *stream_) {
  hipStream_t st = (hipStream_t)stream_;
  AABR_CHECK_ARG(n_in > 0 && n_out > 0 && vol > 0 && V_out >= 0 && vol <= 65535, "bad sizes");
  AABR_CHECK_ARG(dW, "null dW");
  int64_t elems = (int64_t)vol * n_in * n_out;
  if (V_out == 0) {
    hipMemsetAsync(dW, 0, elems * sizeof(float), st);
    if (d_bias) hipMemsetAsync(d_bias, 0, n_out * sizeof(float), st);
    return AABR_OK;
  }
  AABR_CHECK_ARG(in_feats && d_out && table && scratch, "null pointer");
  int cb, nb, tiles;
  dw_tiling(n_in, n_out, cb, nb, tiles);
  AABR_CHECK_ARG(tiles <= 65535, "too many tiles");
  const int nchunks = dw_chunks(V_out, vol, tiles);
  int64_t rpc = ceil_div(V_out, nchunks);
  rpc = ceil_div(rpc, 64) * 64;
  dim3 grid((unsigned)ceil_div(nchunks, 4), (unsigned)vol, (unsigned)tiles);
#define AABR_LAUNCH_DW(CB, NB)                                                                           \
  hipLaunchKernelGGL((k_conv_dw_partial<CB, NB>), grid, dim3(256), 0, st, in_feats, n_in, d_out, n_out,  \
                     V_out, table, vol, nchunks, rpc, scratch)
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
  hipLaunchKernelGGL(k_conv_dw_reduce, dim3((unsigned)ceil_div(elems, 256)), dim3(256), 0, st, scratch, nchunks,
                     elems, dW);
  if (d_bias)
    hipLaunchKernelGGL(k_col_sum, dim3((unsigned)ceil_div(n_out, 64)), dim3(256), 0, st, d_out, V_out, n_out,
                       d_bias);
  AABR_CHECK_LAUNCH();
  return AABR_OK;
}
