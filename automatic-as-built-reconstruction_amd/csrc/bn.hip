// bn.hip -- BatchNormalization + leaky ReLU, forward and backward (gfx950).
//
// Replaces SCN/CPU/BatchNormalization.cpp:12-157 and SCN/CUDA/BatchNormalization.cu:14-238.
// HBM-bound streaming: forward = 2 reads + 1 write of [rows, planes] fp32, backward = 5 passes
// (x, out, d_out twice each for the statistics and the apply, 1 write).  The reference's CUDA
// launch uses <= 16 blocks (BatchNormalization.cu:111); here the statistics pass fills the
// chip with up to kMaxParts blocks and combines their partial sums in a fixed order, so the
// result is bit-reproducible.  Partial sums are kept in fp64 (the reference accumulates
// sum and sum-of-squares sequentially in fp32; fp64 partials are strictly closer to the
// exact statistics).
#include "common.h"

namespace aabr {

// the BatchNorm kernels sit on the critical chain of a pass while the weight-gradient kernels of the second stream
// share the CUs with them: their waves issue at raised priority (bench step 13.50 -> 13.43 ms; -DAABR_BN_NO_PRIO: off)
#ifndef AABR_BN_NO_PRIO
#define AABR_BN_SETPRIO() __builtin_amdgcn_s_setprio(3)
#else
#define AABR_BN_SETPRIO() ((void)0)
#endif

constexpr int kMaxParts = 512;
// finalize: 8 planes x 32 slices of the partial list per 256-thread block (a wave-load reads 64-byte segments of 8 planes).
// Round 6 measured two wider forms on the long lists a wide convolution's write-out leaves (2,502 parts x 64 planes at the
// 282 k-row level, 12 us): 2 planes x 128 slices in 4 x the blocks (16-byte segments: 13.4 us) and 8 planes x 128 slices in
// 1,024-thread blocks (16.4 us; backward 21-40 us): both slower -- the launch is bound by its ~10 dependent batches of
// strided loads and a wider block only adds lines per load or barrier weight.  Kept as it was.
constexpr int kFinSlices = 32;
constexpr int kFinPlanes = 8;

// feature element access: fp32 (reference precision) or bf16 storage (extension, fp32/fp64 maths)
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
__device__ inline float4 ld4(const float *p, int64_t i) { return *reinterpret_cast<const float4 *>(p + i); }
__device__ inline float4 ld4(const __bf16 *p, int64_t i) {
  bf16x4 v = *reinterpret_cast<const bf16x4 *>(p + i);
  return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}
__device__ inline void st4(float *p, int64_t i, float4 v) { *reinterpret_cast<float4 *>(p + i) = v; }
__device__ inline void st4(__bf16 *p, int64_t i, float4 v) {
  bf16x4 o = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
  *reinterpret_cast<bf16x4 *>(p + i) = o;
}
// a value as the storage type would hold it (a fused "+ d_in_add" must see the BatchNorm gradient as the separate add would
// have read it back: rounded once in bf16 storage, untouched in fp32)
__device__ inline float stored(const float *, float v) { return v; }
__device__ inline float stored(const __bf16 *, float v) { return (float)(__bf16)v; }
__device__ inline float ld1(const float *p, int64_t i) { return p[i]; }
__device__ inline float ld1(const __bf16 *p, int64_t i) { return (float)p[i]; }
__device__ inline void st1(float *p, int64_t i, float v) { p[i] = v; }
__device__ inline void st1(__bf16 *p, int64_t i, float v) { p[i] = (__bf16)v; }

// two-quantity column reduction: for every plane p,
//   A[p] = sum_rows fa(row,p),  B[p] = sum_rows fb(row,p)
// MODE 0: fa = x, fb = x*x            (forward statistics)
// MODE 1: fa = d*r, fb = (x-mean)*d*r (backward statistics, r = out>0 ? 1 : leakiness)
// VEC = 4: planes % 4 == 0, one float4 per thread per row (16-B coalesced streams); VEC = 1: any.
template <int MODE, int VEC, typename T>
__global__ __launch_bounds__(256) void k_bn_partials(const T *__restrict__ x, const T *__restrict__ out,
                                                     const T *__restrict__ d_out,
                                                     const float *__restrict__ mean, float leak, int64_t rows,
                                                     int planes, double *__restrict__ part,
                                                     const float *__restrict__ invstd = nullptr,
                                                     const float *__restrict__ weight = nullptr,
                                                     const float *__restrict__ bias = nullptr, int recompute = 0) {
  AABR_BN_SETPRIO();
  __shared__ double ra[256][VEC], rb[256][VEC];
  const int pv = planes / VEC;                 // vector columns
  const int tpr = pv < 256 ? pv : 256;         // threads per row
  const int rpi = 256 / tpr;                   // rows per iteration
  const int tx = threadIdx.x % tpr, ty = threadIdx.x / tpr;
  for (int c0 = 0; c0 < pv; c0 += tpr) {
    const int cv = c0 + tx;
    double a[VEC], b[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) { a[j] = 0.0; b[j] = 0.0; }
    if (cv < pv && ty < rpi) {
      float mu[VEC], wc[VEC], bc[VEC];
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        mu[j] = (MODE == 1) ? mean[cv * VEC + j] : 0.0f;
        wc[j] = bc[j] = 0.0f;
        if (MODE == 1 && recompute) { // the forward pass's affine coefficients, from the same floats by the same operations
          const int p = cv * VEC + j;
          wc[j] = invstd[p] * (weight ? weight[p] : 1.0f);
          bc[j] = -mu[j] * wc[j] + (bias ? bias[p] : 0.0f);
        }
      }
      for (int64_t r = (int64_t)blockIdx.x * rpi + ty; r < rows; r += (int64_t)gridDim.x * rpi) {
        const int64_t i = r * planes + (int64_t)cv * VEC;
        float xv[VEC], ov[VEC], dv[VEC];
        if (VEC == 4) {
          float4 t = ld4(x, i);
          xv[0] = t.x; xv[1] = t.y; xv[2] = t.z; xv[3] = t.w;
          if (MODE == 1) {
            float4 d = ld4(d_out, i);
            dv[0] = d.x; dv[1] = d.y; dv[2] = d.z; dv[3] = d.w;
            if (recompute) {
#pragma unroll
              for (int j = 0; j < 4; ++j) ov[j] = xv[j] * wc[j] + bc[j];
            } else {
              float4 o = ld4(out, i);
              ov[0] = o.x; ov[1] = o.y; ov[2] = o.z; ov[3] = o.w;
            }
          }
        } else {
          xv[0] = ld1(x, i);
          if (MODE == 1) {
            dv[0] = ld1(d_out, i);
            ov[0] = recompute ? xv[0] * wc[0] + bc[0] : ld1(out, i);
          }
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          if (MODE == 0) {
            a[j] += (double)xv[j];
            b[j] += (double)xv[j] * (double)xv[j];
          } else {
            float d = (ov[j] > 0.0f) ? dv[j] : dv[j] * leak;
            a[j] += (double)d;
            b[j] += (double)(xv[j] - mu[j]) * (double)d;
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) { ra[threadIdx.x][j] = a[j]; rb[threadIdx.x][j] = b[j]; }
    __syncthreads();
    if (ty == 0 && cv < pv) {
      for (int q = 1; q < rpi; ++q)
#pragma unroll
        for (int j = 0; j < VEC; ++j) { a[j] += ra[q * tpr + tx][j]; b[j] += rb[q * tpr + tx][j]; }
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        part[((int64_t)blockIdx.x * 2 + 0) * planes + cv * VEC + j] = a[j];
        part[((int64_t)blockIdx.x * 2 + 1) * planes + cv * VEC + j] = b[j];
      }
    }
    __syncthreads();
  }
}

// sum the per-block partials of one plane: 32 threads take interleaved slices of the list (loads
// in flight instead of one dependent chain), then a fixed-order combine => deterministic.
__device__ inline void reduce_partials(const double *__restrict__ part, int nparts, int planes, int p,
                                       double &s0, double &s1) {
  __shared__ double ra[kFinSlices][kFinPlanes], rb[kFinSlices][kFinPlanes];
  const int pl = threadIdx.x % kFinPlanes, sl = threadIdx.x / kFinPlanes;
  double a = 0.0, b = 0.0;
  if (p < planes)
#pragma unroll 8
    for (int j = sl; j < nparts; j += kFinSlices) {
      a += part[((int64_t)j * 2 + 0) * planes + p];
      b += part[((int64_t)j * 2 + 1) * planes + p];
    }
  ra[sl][pl] = a; rb[sl][pl] = b;
  __syncthreads();
  s0 = 0.0; s1 = 0.0;
#pragma unroll
  for (int j = 0; j < kFinSlices; ++j) { s0 += ra[j][pl]; s1 += rb[j][pl]; }
}

// forward finalize (CPU/BatchNormalization.cpp:33-48): coef[p] = {w, b} with y = x*w + b.
__global__ __launch_bounds__(256) void k_bn_fwd_finalize(const double *__restrict__ part, int nparts,
                                                         int64_t rows, int planes, float *save_mean,
                                                         float *save_invstd, float *running_mean,
                                                         float *running_var, const float *weight,
                                                         const float *bias, float eps, float momentum, int train,
                                                         float *coef) {
  AABR_BN_SETPRIO();
  const int p = blockIdx.x * kFinPlanes + (threadIdx.x % kFinPlanes);
  double s = 0.0, ss = 0.0;
  if (train) reduce_partials(part, nparts, planes, p, s, ss);
  if (p >= planes || threadIdx.x >= kFinPlanes) return;
  float mean, invstd;
  if (train) {
    double m = s / (double)rows;
    double var_n = ss - m * m * (double)rows; // == sum (x-mean)^2
    mean = (float)m;
    running_mean[p] = momentum * running_mean[p] + (1 - momentum) * mean;
    running_var[p] = momentum * running_var[p] + (1 - momentum) * (float)(var_n / (double)(rows - 1));
    invstd = powf((float)(var_n / (double)rows) + eps, -0.5f);
  } else {
    mean = running_mean[p];
    invstd = powf(running_var[p] + eps, -0.5f);
  }
  save_mean[p] = mean;
  save_invstd[p] = invstd;
  float w = invstd * (weight ? weight[p] : 1.0f);
  coef[p] = w;
  coef[planes + p] = -mean * w + (bias ? bias[p] : 0.0f);
}

template <typename T>
__global__ __launch_bounds__(256) void k_bn_fwd_apply(const T *__restrict__ x, T *__restrict__ y,
                                                      int64_t total, int planes,
                                                      const float *__restrict__ coef, float leak) {
  AABR_BN_SETPRIO();
  // planes % 4 == 0 path: float4 per thread
  int64_t i4 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t i = i4 * 4;
  if (i >= total) return;
  int p = (int)(i % planes);
  float4 v = ld4(x, i);
  float4 w = *reinterpret_cast<const float4 *>(coef + p);
  float4 b = *reinterpret_cast<const float4 *>(coef + planes + p);
  float4 o;
  o.x = v.x * w.x + b.x; o.y = v.y * w.y + b.y; o.z = v.z * w.z + b.z; o.w = v.w * w.w + b.w;
  o.x = o.x > 0.0f ? o.x : o.x * leak; o.y = o.y > 0.0f ? o.y : o.y * leak;
  o.z = o.z > 0.0f ? o.z : o.z * leak; o.w = o.w > 0.0f ? o.w : o.w * leak;
  st4(y, i, o);
}
template <typename T>
__global__ __launch_bounds__(256) void k_bn_fwd_apply1(const T *__restrict__ x, T *__restrict__ y,
                                                       int64_t total, int planes,
                                                       const float *__restrict__ coef, float leak) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  int p = (int)(i % planes);
  float o = ld1(x, i) * coef[p] + coef[planes + p];
  st1(y, i, o > 0.0f ? o : o * leak);
}

// backward finalize (CPU/BatchNormalization.cpp:85-90,103-106): coef = {gradMean, k, invstd*w}
__global__ __launch_bounds__(256) void k_bn_bwd_finalize(const double *__restrict__ part, int nparts,
                                                         int64_t rows, int planes, const float *save_invstd,
                                                         const float *weight, float *d_weight, float *d_bias,
                                                         float *coef) {
  AABR_BN_SETPRIO();
  const int p = blockIdx.x * kFinPlanes + (threadIdx.x % kFinPlanes);
  double s, dp;
  reduce_partials(part, nparts, planes, p, s, dp);
  if (p >= planes || threadIdx.x >= kFinPlanes) return;
  float is = save_invstd[p];
  if (d_bias) d_bias[p] = (float)s;
  if (d_weight) d_weight[p] = (float)dp * is;
  coef[p] = (float)(s / (double)rows);
  coef[planes + p] = (float)dp * is * is / (float)rows;
  coef[2 * planes + p] = is * (weight ? weight[p] : 1.0f);
}

template <typename T>
__global__ __launch_bounds__(256) void k_bn_bwd_apply(const T *__restrict__ x, T *__restrict__ d_in,
                                                      const T *__restrict__ out,
                                                      const T *__restrict__ d_out, int64_t total,
                                                      int planes, const float *__restrict__ mean,
                                                      const float *__restrict__ coef, float leak,
                                                      const float *__restrict__ bias, int recompute,
                                                      const T *__restrict__ res) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  int p = (int)(i % planes);
  float d = ld1(d_out, i);
  // recompute: the activation's sign from x with the forward pass's coefficients (coef[2P+p] = invstd * weight)
  const float o = recompute ? ld1(x, i) * coef[2 * planes + p] + (-mean[p] * coef[2 * planes + p] + (bias ? bias[p] : 0.0f))
                            : ld1(out, i);
  d = (o > 0.0f) ? d : d * leak;
  float r = (d - coef[p] - (ld1(x, i) - mean[p]) * coef[planes + p]) * coef[2 * planes + p];
  if (res) r = stored(res, r) + ld1(res, i);
  st1(d_in, i, r);
}

template <typename T>
__global__ __launch_bounds__(256) void k_bn_bwd_apply4(const T *__restrict__ x, T *__restrict__ d_in,
                                                       const T *__restrict__ out,
                                                       const T *__restrict__ d_out, int64_t total,
                                                       int planes, const float *__restrict__ mean,
                                                       const float *__restrict__ coef, float leak,
                                                       const float *__restrict__ bias, int recompute,
                                                       const T *__restrict__ res) {
  AABR_BN_SETPRIO();
  int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= total) return;
  int p = (int)(i % planes);
  float4 xv = ld4(x, i), ov;
  float4 dv = ld4(d_out, i), mu = *reinterpret_cast<const float4 *>(mean + p);
  float4 gm = *reinterpret_cast<const float4 *>(coef + p), kk = *reinterpret_cast<const float4 *>(coef + planes + p);
  float4 sw = *reinterpret_cast<const float4 *>(coef + 2 * planes + p), r;
  if (recompute) { // sign of the forward activation from x: y = x*w + b with w = invstd*weight (= sw), b = -mean*w + bias
    float4 bb = bias ? *reinterpret_cast<const float4 *>(bias + p) : make_float4(0.f, 0.f, 0.f, 0.f);
    ov.x = xv.x * sw.x + (-mu.x * sw.x + bb.x); ov.y = xv.y * sw.y + (-mu.y * sw.y + bb.y);
    ov.z = xv.z * sw.z + (-mu.z * sw.z + bb.z); ov.w = xv.w * sw.w + (-mu.w * sw.w + bb.w);
  } else {
    ov = ld4(out, i);
  }
  float d;
  d = ov.x > 0.0f ? dv.x : dv.x * leak; r.x = (d - gm.x - (xv.x - mu.x) * kk.x) * sw.x;
  d = ov.y > 0.0f ? dv.y : dv.y * leak; r.y = (d - gm.y - (xv.y - mu.y) * kk.y) * sw.y;
  d = ov.z > 0.0f ? dv.z : dv.z * leak; r.z = (d - gm.z - (xv.z - mu.z) * kk.z) * sw.z;
  d = ov.w > 0.0f ? dv.w : dv.w * leak; r.w = (d - gm.w - (xv.w - mu.w) * kk.w) * sw.w;
  if (res) { // d_in = BatchNorm gradient + another contribution to the same tensor's gradient
    const float4 q = ld4(res, i);
    r.x = stored(res, r.x) + q.x; r.y = stored(res, r.y) + q.y; r.z = stored(res, r.z) + q.z; r.w = stored(res, r.w) + q.w;
  }
  st4(d_in, i, r);
}


// ---- small maps: ONE launch -------------------------------------------------------------------------------------
// A map of <= 2048 rows (the kernels hold up to 8192 / 6144) is under 2 MB: the three launches above cost it ~4 us each of pure
// launch/drain latency (profiles/r02_bn_bench.txt: 27 us forward / 68 us backward regardless of size below 22k rows).
// Here a workgroup owns FOUR planes (one 16-byte column of every row) and all rows: each thread keeps its <= 16 rows'
// values in registers, the statistics are reduced inside the workgroup (shuffle butterfly, then the 8 waves in order:
// fixed order => bit-reproducible), every thread forms the coefficients from the same sums and applies them to the
// registers -- one read of x (and d_out), one write, no partials in memory.  Same formulas, float for float, as
// k_bn_*_finalize / k_bn_*_apply; only the order of the fp64 additions differs from the three-launch path.
constexpr int kSmallThreads = 512, kSmallRows = 16; // forward: x in registers (64 VGPRs)
constexpr int kSmallDefaultCap = 2048;                // rows up to which the one-launch kernels are used: measured on the
                                                      // bench step, cap 8192: 13.46 ms, 2048: 13.35, 512: 13.34, never: 13.41
constexpr int kSmallThreadsB = 512, kSmallRowsB = 12; // backward: x and d_out in registers (239 VGPRs, nothing spilled; 16 rows spill)

__device__ inline double wave_sum_f64(double v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}
// sums over the workgroup of a[4], b[4]; every thread returns with the totals
template <int NT>
__device__ inline void block_sum8(double (&a)[4], double (&b)[4]) {
  __shared__ double red[NT / 64][8];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int j = 0; j < 4; ++j) { a[j] = wave_sum_f64(a[j]); b[j] = wave_sum_f64(b[j]); }
  if (lane == 0)
#pragma unroll
    for (int j = 0; j < 4; ++j) { red[wave][j] = a[j]; red[wave][4 + j] = b[j]; }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    double s = 0.0, t = 0.0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) { s += red[w][j]; t += red[w][4 + j]; }
    a[j] = s; b[j] = t;
  }
}

template <typename T>
__global__ __launch_bounds__(kSmallThreads) void k_bn_fwd_small(const T *__restrict__ x, T *__restrict__ y, int rows,
                                                                int planes, float *save_mean, float *save_invstd,
                                                                float *running_mean, float *running_var,
                                                                const float *weight, const float *bias, float eps,
                                                                float momentum, float leak) {
  AABR_BN_SETPRIO();
  const int p0 = blockIdx.x * 4;
  float4 xv[kSmallRows];
  double a[4] = {0.0, 0.0, 0.0, 0.0}, b[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int j = 0; j < kSmallRows; ++j) {
    const int r = (int)threadIdx.x + j * kSmallThreads;
    if (r < rows) {
      xv[j] = ld4(x, (int64_t)r * planes + p0);
      a[0] += (double)xv[j].x; b[0] += (double)xv[j].x * (double)xv[j].x;
      a[1] += (double)xv[j].y; b[1] += (double)xv[j].y * (double)xv[j].y;
      a[2] += (double)xv[j].z; b[2] += (double)xv[j].z * (double)xv[j].z;
      a[3] += (double)xv[j].w; b[3] += (double)xv[j].w * (double)xv[j].w;
    }
  }
  block_sum8<kSmallThreads>(a, b);
  float w[4], c[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { // k_bn_fwd_finalize, train branch
    const int p = p0 + j;
    const double m = a[j] / (double)rows;
    const double var_n = b[j] - m * m * (double)rows;
    const float mean = (float)m;
    const float invstd = powf((float)(var_n / (double)rows) + eps, -0.5f);
    if (threadIdx.x == 0) {
      running_mean[p] = momentum * running_mean[p] + (1 - momentum) * mean;
      running_var[p] = momentum * running_var[p] + (1 - momentum) * (float)(var_n / (double)(rows - 1));
      save_mean[p] = mean;
      save_invstd[p] = invstd;
    }
    w[j] = invstd * (weight ? weight[p] : 1.0f);
    c[j] = -mean * w[j] + (bias ? bias[p] : 0.0f);
  }
#pragma unroll
  for (int j = 0; j < kSmallRows; ++j) {
    const int r = (int)threadIdx.x + j * kSmallThreads;
    if (r < rows) {
      float4 o;
      o.x = xv[j].x * w[0] + c[0]; o.y = xv[j].y * w[1] + c[1]; o.z = xv[j].z * w[2] + c[2]; o.w = xv[j].w * w[3] + c[3];
      o.x = o.x > 0.0f ? o.x : o.x * leak; o.y = o.y > 0.0f ? o.y : o.y * leak;
      o.z = o.z > 0.0f ? o.z : o.z * leak; o.w = o.w > 0.0f ? o.w : o.w * leak;
      st4(y, (int64_t)r * planes + p0, o);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(kSmallThreadsB) void k_bn_bwd_small(const T *__restrict__ x, T *__restrict__ d_in,
                                                                const T *__restrict__ out, const T *__restrict__ d_out,
                                                                int rows, int planes, const float *__restrict__ mean,
                                                                const float *__restrict__ save_invstd,
                                                                const float *weight, const float *bias, float *d_weight,
                                                                float *d_bias, float leak, int recompute,
                                                                const T *__restrict__ res) {
  AABR_BN_SETPRIO();
  const int p0 = blockIdx.x * 4;
  float mu[4], is[4], sw[4], bc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    mu[j] = mean[p0 + j];
    is[j] = save_invstd[p0 + j];
    sw[j] = is[j] * (weight ? weight[p0 + j] : 1.0f);
    bc[j] = -mu[j] * sw[j] + (bias ? bias[p0 + j] : 0.0f);
  }
  float xv[kSmallRowsB][4], dv[kSmallRowsB][4];
  double a[4] = {0.0, 0.0, 0.0, 0.0}, b[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int j = 0; j < kSmallRowsB; ++j) {
    const int r = (int)threadIdx.x + j * kSmallThreadsB;
    if (r < rows) {
      const int64_t i = (int64_t)r * planes + p0;
      const float4 t = ld4(x, i), d = ld4(d_out, i);
      float ov[4];
      xv[j][0] = t.x; xv[j][1] = t.y; xv[j][2] = t.z; xv[j][3] = t.w;
      dv[j][0] = d.x; dv[j][1] = d.y; dv[j][2] = d.z; dv[j][3] = d.w;
      if (recompute) {
#pragma unroll
        for (int q = 0; q < 4; ++q) ov[q] = xv[j][q] * sw[q] + bc[q];
      } else {
        const float4 o = ld4(out, i);
        ov[0] = o.x; ov[1] = o.y; ov[2] = o.z; ov[3] = o.w;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        dv[j][q] = (ov[q] > 0.0f) ? dv[j][q] : dv[j][q] * leak;
        a[q] += (double)dv[j][q];
        b[q] += (double)(xv[j][q] - mu[q]) * (double)dv[j][q];
      }
    }
  }
  block_sum8<kSmallThreadsB>(a, b);
  float gm[4], kk[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { // k_bn_bwd_finalize
    if (threadIdx.x == 0) {
      if (d_bias) d_bias[p0 + j] = (float)a[j];
      if (d_weight) d_weight[p0 + j] = (float)b[j] * is[j];
    }
    gm[j] = (float)(a[j] / (double)rows);
    kk[j] = (float)b[j] * is[j] * is[j] / (float)rows;
  }
#pragma unroll
  for (int j = 0; j < kSmallRowsB; ++j) {
    const int r = (int)threadIdx.x + j * kSmallThreadsB;
    if (r < rows) {
      const int64_t i = (int64_t)r * planes + p0;
      float4 o;
      o.x = (dv[j][0] - gm[0] - (xv[j][0] - mu[0]) * kk[0]) * sw[0];
      o.y = (dv[j][1] - gm[1] - (xv[j][1] - mu[1]) * kk[1]) * sw[1];
      o.z = (dv[j][2] - gm[2] - (xv[j][2] - mu[2]) * kk[2]) * sw[2];
      o.w = (dv[j][3] - gm[3] - (xv[j][3] - mu[3]) * kk[3]) * sw[3];
      if (res) {
        const float4 q = ld4(res, i);
        o.x = stored(res, o.x) + q.x; o.y = stored(res, o.y) + q.y; o.z = stored(res, o.z) + q.z; o.w = stored(res, o.w) + q.w;
      }
      st4(d_in, i, o);
    }
  }
}

static bool bn_small(int64_t rows, int planes, bool backward) {
  int64_t cap = backward ? (int64_t)kSmallThreadsB * kSmallRowsB : (int64_t)kSmallThreads * kSmallRows;
  const int v = knob(K_BN_SMALL);          // 0: never; > 1: row cap (tuning experiments)
  if (v == 0) return false;
  if (v > 1 && v < cap) cap = v;
  else if (v < 0 && kSmallDefaultCap < cap) cap = kSmallDefaultCap;
  return rows > 1 && rows <= cap && (planes & 3) == 0;
}

static int bn_parts(int64_t rows, int planes, int vec) {
  int pv = planes / vec, tpr = pv < 256 ? pv : 256, rpi = 256 / tpr;
  int64_t want = ceil_div(rows, (int64_t)rpi * 4); // >= 4 rows per thread
  if (want < 1) want = 1;
  if (want > kMaxParts) want = kMaxParts;
  return (int)want;
}

} // namespace aabr
using namespace aabr;

extern "C" int64_t aabr_bn_scratch_floats(int planes) {
  return (int64_t)kMaxParts * 2 * planes * 2 /* doubles */ + 4 * (int64_t)planes + 8;
}

template <typename T>
static int bn_forward_t(const T *in, T *out, int64_t rows, int planes, float *save_mean, float *save_invstd,
                        float *running_mean, float *running_var, const float *weight, const float *bias,
                        float eps, float momentum, int train, float leakiness, float *scratch, void *stream_,
                        const double *pre_part = nullptr, int pre_nparts = 0) {
  hipStream_t st = (hipStream_t)stream_;
  AABR_CHECK_ARG(rows >= 0 && planes > 0, "bad sizes");
  AABR_CHECK_ARG(save_mean && save_invstd && running_mean && running_var && scratch, "null pointer");
  AABR_CHECK_ARG(((uintptr_t)scratch & 15) == 0, "scratch must be 16-byte aligned");
  if (rows == 0) return AABR_OK; // reference: nActive == 0 leaves everything untouched
  AABR_CHECK_ARG(in && out, "null pointer");
  if (train && !pre_part && bn_small(rows, planes, false) && (((uintptr_t)in | (uintptr_t)out) & 15) == 0) {
    hipLaunchKernelGGL((k_bn_fwd_small<T>), dim3(planes / 4), dim3(kSmallThreads), 0, st, in, out, (int)rows, planes,
                       save_mean, save_invstd, running_mean, running_var, weight, bias, eps, momentum, leakiness);
    AABR_CHECK_LAUNCH();
    return AABR_OK;
  }
  const double *part = reinterpret_cast<double *>(scratch);
  float *coef = scratch + (int64_t)kMaxParts * 2 * planes * 2;
  int nparts = 0;
  if (train && pre_part) { // the statistics' partial sums came with the producing convolution's write-out
    AABR_CHECK_ARG(pre_nparts > 0, "partial count");
    part = pre_part;
    nparts = pre_nparts;
  } else if (train) {
    double *part = reinterpret_cast<double *>(scratch);
    const bool v4 = (planes & 3) == 0 && ((uintptr_t)in & 15) == 0;
    nparts = bn_parts(rows, planes, v4 ? 4 : 1);
    if (v4)
      hipLaunchKernelGGL((k_bn_partials<0, 4, T>), dim3(nparts), dim3(256), 0, st, in, (const T *)nullptr,
                         (const T *)nullptr, (const float *)nullptr, 0.0f, rows, planes, part);
    else
      hipLaunchKernelGGL((k_bn_partials<0, 1, T>), dim3(nparts), dim3(256), 0, st, in, (const T *)nullptr,
                         (const T *)nullptr, (const float *)nullptr, 0.0f, rows, planes, part);
  }
  hipLaunchKernelGGL(k_bn_fwd_finalize, dim3((unsigned)ceil_div(planes, kFinPlanes)), dim3(256), 0, st, part, nparts,
                     rows, planes, save_mean, save_invstd, running_mean, running_var, weight, bias, eps,
                     momentum, train, coef);
  int64_t total = rows * planes;
  if ((planes & 3) == 0 && (((uintptr_t)in | (uintptr_t)out) & 15) == 0)
    hipLaunchKernelGGL((k_bn_fwd_apply<T>), dim3((unsigned)ceil_div(total / 4, 256)), dim3(256), 0, st, in, out,
                       total, planes, coef, leakiness);
  else
    hipLaunchKernelGGL((k_bn_fwd_apply1<T>), dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, st, in, out, total,
                       planes, coef, leakiness);
  AABR_CHECK_LAUNCH();
  return AABR_OK;
}

template <typename T>
static int bn_backward_t(const T *in, T *d_in, const T *out, const T *d_out, int64_t rows, int planes,
                         const float *save_mean, const float *save_invstd, const float *weight, const float *bias,
                         float *d_weight, float *d_bias, float leakiness, float *scratch, void *stream_,
                         const T *d_in_add = nullptr, const double *pre_part = nullptr, int pre_nparts = 0) {
  hipStream_t st = (hipStream_t)stream_;
  AABR_CHECK_ARG(rows >= 0 && planes > 0, "bad sizes");
  AABR_CHECK_ARG(save_mean && save_invstd && scratch, "null pointer");
  if (rows == 0) {
    if (d_weight) hipMemsetAsync(d_weight, 0, planes * sizeof(float), st);
    if (d_bias) hipMemsetAsync(d_bias, 0, planes * sizeof(float), st);
    return AABR_OK;
  }
  // fp32 storage: the sign of the forward activation is recomputed from x with the forward pass's own coefficients
  // (same floats, same operations => the same bits as reading `out`; for leakiness >= 0 out > 0 <=> x*w+b > 0), which
  // saves both passes over `out` -- 2 of the 7 matrix passes of the backward.  bf16 storage keeps reading the
  // rounded `out` (half the bytes, and the rounding could flip a sign at the very bottom of the range).
  const int recompute = (sizeof(T) == 4 && leakiness >= 0.0f) ? 1 : 0;
  AABR_CHECK_ARG(in && d_in && d_out && (out || recompute), "null pointer");
  double *part = reinterpret_cast<double *>(scratch);
  float *coef = scratch + (int64_t)kMaxParts * 2 * planes * 2;
  const bool v4 = (planes & 3) == 0 && (((uintptr_t)in | (uintptr_t)(recompute ? nullptr : out) | (uintptr_t)d_out) & 15) == 0;
  if (!pre_part && v4 && bn_small(rows, planes, true) && (((uintptr_t)d_in | (uintptr_t)d_in_add) & 15) == 0) {
    hipLaunchKernelGGL((k_bn_bwd_small<T>), dim3(planes / 4), dim3(kSmallThreadsB), 0, st, in, d_in, out, d_out, (int)rows,
                       planes, save_mean, save_invstd, weight, bias, d_weight, d_bias, leakiness, recompute, d_in_add);
    AABR_CHECK_LAUNCH();
    return AABR_OK;
  }
  int nparts = bn_parts(rows, planes, v4 ? 4 : 1);
  const double *fin_part = part;
  if (pre_part) {   // the statistics' partial sums came with the write-out of the convolution that produced d_out
    AABR_CHECK_ARG(pre_nparts > 0 && (recompute || sizeof(T) == 2),
                   "precomputed backward statistics: leakiness >= 0 in fp32 storage (the sign is recomputed from x)");
    fin_part = pre_part;
    nparts = pre_nparts;
  } else if (v4)
    hipLaunchKernelGGL((k_bn_partials<1, 4, T>), dim3(nparts), dim3(256), 0, st, in, out, d_out, save_mean,
                       leakiness, rows, planes, part, save_invstd, weight, bias, recompute);
  else
    hipLaunchKernelGGL((k_bn_partials<1, 1, T>), dim3(nparts), dim3(256), 0, st, in, out, d_out, save_mean,
                       leakiness, rows, planes, part, save_invstd, weight, bias, recompute);
  hipLaunchKernelGGL(k_bn_bwd_finalize, dim3((unsigned)ceil_div(planes, kFinPlanes)), dim3(256), 0, st, fin_part, nparts,
                     rows, planes, save_invstd, weight, d_weight, d_bias, coef);
  int64_t total = rows * planes;
  if (v4 && (((uintptr_t)d_in | (uintptr_t)save_mean | (uintptr_t)d_in_add) & 15) == 0)
    hipLaunchKernelGGL((k_bn_bwd_apply4<T>), dim3((unsigned)ceil_div(total / 4, 256)), dim3(256), 0, st, in, d_in, out,
                       d_out, total, planes, save_mean, coef, leakiness, bias,
                       (recompute && (!bias || ((uintptr_t)bias & 15) == 0)) ? 1 : 0, d_in_add);
  else
    hipLaunchKernelGGL((k_bn_bwd_apply<T>), dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, st, in, d_in, out,
                       d_out, total, planes, save_mean, coef, leakiness, bias, recompute, d_in_add);
  AABR_CHECK_LAUNCH();
  return AABR_OK;
}

extern "C" int aabr_bn_forward(const float *in, float *out, int64_t rows, int planes, float *save_mean,
                               float *save_invstd, float *running_mean, float *running_var,
                               const float *weight, const float *bias, float eps, float momentum, int train,
                               float leakiness, float *scratch, void *stream_) {
  return bn_forward_t<float>(in, out, rows, planes, save_mean, save_invstd, running_mean, running_var, weight,
                             bias, eps, momentum, train, leakiness, scratch, stream_);
}

// forward with the training statistics' partial sums already formed (aabr_conv_forward_wide_stats: one [2][planes]
// fp64 pair per output tile, in tile order): the statistics pass over `in` is skipped
extern "C" int aabr_bn_forward_parts(const float *in, float *out, int64_t rows, int planes, float *save_mean,
                                     float *save_invstd, float *running_mean, float *running_var, const float *weight,
                                     const float *bias, float eps, float momentum, float leakiness, const double *parts,
                                     int nparts, float *scratch, void *stream_) {
  AABR_CHECK_ARG(parts && nparts > 0, "partials");
  return bn_forward_t<float>(in, out, rows, planes, save_mean, save_invstd, running_mean, running_var, weight, bias,
                             eps, momentum, 1, leakiness, scratch, stream_, parts, nparts);
}
extern "C" int aabr_bn_forward_parts_bf16(const uint16_t *in, uint16_t *out, int64_t rows, int planes, float *save_mean,
                                          float *save_invstd, float *running_mean, float *running_var,
                                          const float *weight, const float *bias, float eps, float momentum,
                                          float leakiness, const double *parts, int nparts, float *scratch,
                                          void *stream_) {
  AABR_CHECK_ARG(parts && nparts > 0, "partials");
  return bn_forward_t<__bf16>(reinterpret_cast<const __bf16 *>(in), reinterpret_cast<__bf16 *>(out), rows, planes,
                              save_mean, save_invstd, running_mean, running_var, weight, bias, eps, momentum, 1,
                              leakiness, scratch, stream_, parts, nparts);
}

extern "C" int aabr_bn_backward(const float *in, float *d_in, const float *out, const float *d_out,
                                int64_t rows, int planes, const float *save_mean, const float *save_invstd,
                                const float *weight, const float *bias, float *d_weight, float *d_bias,
                                float leakiness, float *scratch, void *stream_) {
  return bn_backward_t<float>(in, d_in, out, d_out, rows, planes, save_mean, save_invstd, weight, bias, d_weight,
                              d_bias, leakiness, scratch, stream_);
}

// d_in = BatchNorm gradient + d_in_add: the gradient sum of a tensor with a second consumer (the identity branch of
// a residual block, a lateral connection) folded into the apply pass (a + b is commutative bit for bit; in bf16 storage
// the BatchNorm gradient is rounded before the sum, as the separate add would have read it: aabr_bn_backward_add_bf16)
extern "C" int aabr_bn_backward_add(const float *in, float *d_in, const float *out, const float *d_out,
                                    int64_t rows, int planes, const float *save_mean, const float *save_invstd,
                                    const float *weight, const float *bias, float *d_weight, float *d_bias,
                                    float leakiness, float *scratch, const float *d_in_add, void *stream_) {
  return bn_backward_t<float>(in, d_in, out, d_out, rows, planes, save_mean, save_invstd, weight, bias, d_weight,
                              d_bias, leakiness, scratch, stream_, d_in_add);
}

// backward with the statistics' partial sums given (aabr_conv_forward_wide_bwd_stats: per tile [2][planes] fp64 sums of
// the masked d_out and of (x - mean) * masked d_out, in tile order): the statistics pass over x and d_out is skipped
extern "C" int aabr_bn_backward_parts(const float *in, float *d_in, const float *out, const float *d_out, int64_t rows,
                                      int planes, const float *save_mean, const float *save_invstd,
                                      const float *weight, const float *bias, float *d_weight, float *d_bias,
                                      float leakiness, const double *parts, int nparts, float *scratch,
                                      const float *d_in_add, void *stream_) {
  AABR_CHECK_ARG(parts && nparts > 0, "partials");
  return bn_backward_t<float>(in, d_in, out, d_out, rows, planes, save_mean, save_invstd, weight, bias, d_weight,
                              d_bias, leakiness, scratch, stream_, d_in_add, parts, nparts);
}

extern "C" int aabr_bn_backward_parts_bf16(const uint16_t *in, uint16_t *d_in, const uint16_t *out, const uint16_t *d_out,
                                           int64_t rows, int planes, const float *save_mean, const float *save_invstd,
                                           const float *weight, const float *bias, float *d_weight, float *d_bias,
                                           float leakiness, const double *parts, int nparts, float *scratch,
                                           void *stream_) {
  AABR_CHECK_ARG(parts && nparts > 0, "partials");
  return bn_backward_t<__bf16>(reinterpret_cast<const __bf16 *>(in), reinterpret_cast<__bf16 *>(d_in),
                               reinterpret_cast<const __bf16 *>(out), reinterpret_cast<const __bf16 *>(d_out), rows,
                               planes, save_mean, save_invstd, weight, bias, d_weight, d_bias, leakiness, scratch,
                               stream_, nullptr, parts, nparts);
}

// bf16 feature storage (extension; statistics in fp64, affine maths in fp32, parameters fp32)
extern "C" int aabr_bn_forward_bf16(const uint16_t *in, uint16_t *out, int64_t rows, int planes,
                                    float *save_mean, float *save_invstd, float *running_mean,
                                    float *running_var, const float *weight, const float *bias, float eps,
                                    float momentum, int train, float leakiness, float *scratch, void *stream_) {
  return bn_forward_t<__bf16>(reinterpret_cast<const __bf16 *>(in), reinterpret_cast<__bf16 *>(out), rows, planes,
                              save_mean, save_invstd, running_mean, running_var, weight, bias, eps, momentum,
                              train, leakiness, scratch, stream_);
}

extern "C" int aabr_bn_backward_bf16(const uint16_t *in, uint16_t *d_in, const uint16_t *out,
                                     const uint16_t *d_out, int64_t rows, int planes, const float *save_mean,
                                     const float *save_invstd, const float *weight, const float *bias,
                                     float *d_weight, float *d_bias, float leakiness, float *scratch,
                                     void *stream_) {
  return bn_backward_t<__bf16>(reinterpret_cast<const __bf16 *>(in), reinterpret_cast<__bf16 *>(d_in),
                               reinterpret_cast<const __bf16 *>(out), reinterpret_cast<const __bf16 *>(d_out),
                               rows, planes, save_mean, save_invstd, weight, bias, d_weight, d_bias, leakiness,
                               scratch, stream_);
}

// bf16 storage: d_in = bf16(bf16(BatchNorm gradient) + d_in_add) -- the gradient sum of a tensor with a second consumer in
// the apply pass, bit for bit what aabr_bn_backward_bf16 followed by aabr_add(bf16) stores; `parts` / `nparts` as
// aabr_bn_backward_parts_bf16, or NULL / 0 for statistics of its own
extern "C" int aabr_bn_backward_add_bf16(const uint16_t *in, uint16_t *d_in, const uint16_t *out, const uint16_t *d_out,
                                         int64_t rows, int planes, const float *save_mean, const float *save_invstd,
                                         const float *weight, const float *bias, float *d_weight, float *d_bias,
                                         float leakiness, const double *parts, int nparts, float *scratch,
                                         const uint16_t *d_in_add, void *stream_) {
  AABR_CHECK_ARG((parts != nullptr) == (nparts > 0), "partials");
  return bn_backward_t<__bf16>(reinterpret_cast<const __bf16 *>(in), reinterpret_cast<__bf16 *>(d_in),
                               reinterpret_cast<const __bf16 *>(out), reinterpret_cast<const __bf16 *>(d_out), rows,
                               planes, save_mean, save_invstd, weight, bias, d_weight, d_bias, leakiness, scratch,
                               stream_, reinterpret_cast<const __bf16 *>(d_in_add), parts, nparts);
}
