// roi.hip -- SparseToDense and rotated 3-D ROI-align, gfx950.
//
// Replaces SCN/CPU/SparseToDense.cpp:7-87 (+ SCN/CUDA/SparseToDense.cu) and
// maskrcnn_benchmark/csrc/cuda/ROIAlignRotated3D_cuda.cu:16-346 (`_C.roi_align_rotated_3d_*`,
// csrc/vision.cpp:19-20; the reference has no CPU implementation, ROIAlignRotated3D.h:25,47).
// Dense layout [B, C, X(h), Y(w), Z]: element (y*width + x)*zsize + z in the reference's naming.
// HBM-bound gathers / scatters; the ROI-align backward accumulates with no-return fp32 atomics
// exactly like the reference (the one place in this library where summation order is not fixed).
#include "common.h"

namespace aabr {

__global__ __launch_bounds__(256) void k_sparse_to_dense(const int32_t *__restrict__ sc, int64_t V,
                                                         const float *__restrict__ in, int planes, int X,
                                                         int Y, int Z, float *__restrict__ out, int backward,
                                                         float *__restrict__ d_in) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= V * planes) return;
  int64_t v = idx / planes;
  int p = (int)(idx - v * planes);
  int4 c = *reinterpret_cast<const int4 *>(sc + 4 * v);
  const int64_t vol = (int64_t)X * Y * Z;
  // RectangularRegion::offset over [0, size): last dimension fastest (ConvolutionRules.h:109-151)
  const int64_t off = ((int64_t)c.x * Y + c.y) * Z + c.z;
  const int64_t o = ((int64_t)c.w * planes + p) * vol + off;
  if (backward) d_in[idx] = out[o];
  else out[o] = in[idx];
}

// forward sampling (ROIAlignRotated3D_cuda.cu:16-84).  NB the reference's forward bound check reads
// `zsize > zsize` for the upper z limit (always false), so z above the map clamps to the last slice
// instead of returning 0 -- reproduced; the gradient version (:187-243) does test `z > zsize`.
__device__ inline float trilinear(const float *__restrict__ d, int height, int width, int zsize, float y,
                                  float x, float z) {
  if (y < -1.0f || y > height || x < -1.0f || x > width || z < -1.0f) return 0.0f;
  if (y <= 0) y = 0;
  if (x <= 0) x = 0;
  if (z <= 0) z = 0;
  int y_low = (int)y, x_low = (int)x, z_low = (int)z, y_high, x_high, z_high;
  if (y_low >= height - 1) { y_high = y_low = height - 1; y = (float)y_low; } else y_high = y_low + 1;
  if (x_low >= width - 1) { x_high = x_low = width - 1; x = (float)x_low; } else x_high = x_low + 1;
  if (z_low >= zsize - 1) { z_high = z_low = zsize - 1; z = (float)z_low; } else z_high = z_low + 1;
  float ly = y - y_low, lx = x - x_low, lz = z - z_low;
  float hy = 1.f - ly, hx = 1.f - lx, hz = 1.f - lz;
  float v1 = d[((int64_t)y_low * width + x_low) * zsize + z_low];
  float v2 = d[((int64_t)y_low * width + x_high) * zsize + z_low];
  float v3 = d[((int64_t)y_high * width + x_low) * zsize + z_low];
  float v4 = d[((int64_t)y_high * width + x_high) * zsize + z_low];
  float v5 = d[((int64_t)y_low * width + x_low) * zsize + z_high];
  float v6 = d[((int64_t)y_low * width + x_high) * zsize + z_high];
  float v7 = d[((int64_t)y_high * width + x_low) * zsize + z_high];
  float v8 = d[((int64_t)y_high * width + x_high) * zsize + z_high];
  float w1 = hy * hx * hz, w2 = hy * lx * hz, w3 = ly * hx * hz, w4 = ly * lx * hz;
  float w5 = hy * hx * lz, w6 = hy * lx * lz, w7 = ly * hx * lz, w8 = ly * lx * lz;
  return (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4 + w5 * v5 + w6 * v6 + w7 * v7 + w8 * v8);
}

struct RoiGeom {
  int channels, height, width, zsize, ph, pw, pz, sampling;
  float scale;
};

template <bool BACKWARD>
__global__ __launch_bounds__(256) void k_roi_align_rot3d(int64_t nthreads, const float *__restrict__ bottom,
                                                         const float *__restrict__ rois, RoiGeom g,
                                                         float *__restrict__ top, const float *__restrict__ top_diff,
                                                         float *bottom_diff) {
  for (int64_t index = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; index < nthreads;
       index += (int64_t)blockDim.x * gridDim.x) {
    int pz = (int)(index % g.pz);
    int pw = (int)((index / g.pz) % g.pw);
    int ph = (int)((index / g.pz / g.pw) % g.ph);
    int c = (int)((index / g.pz / g.pw / g.ph) % g.channels);
    int64_t n = index / g.pz / g.pw / g.ph / g.channels;
    const float *r = rois + n * 8;
    int b = (int)r[0];
    float cw = r[1] * g.scale, ch = r[2] * g.scale, cz = r[3] * g.scale;
    float rw = r[4] * g.scale, rh = r[5] * g.scale, rz = r[6] * g.scale;
    float theta = (float)(r[7] * 3.14159265358979323846 / 180.0);
    rw = fmaxf(rw, 1.f); rh = fmaxf(rh, 1.f); rz = fmaxf(rz, 1.f);
    float bh = rh / (float)g.ph, bw = rw / (float)g.pw, bz = rz / (float)g.pz;
    int gh = g.sampling > 0 ? g.sampling : (int)ceilf(rh / g.ph);
    int gw = g.sampling > 0 ? g.sampling : (int)ceilf(rw / g.pw);
    int gz = g.sampling > 0 ? g.sampling : (int)ceilf(rz / g.pz);
    float sh = -rh / 2.0f, sw = -rw / 2.0f, sz = -rz / 2.0f;
    float ct = cosf(theta), st = sinf(theta);
    const float count = (float)(gh * gw * gz);
    const int64_t plane = (int64_t)g.height * g.width * g.zsize;
    const float *src = BACKWARD ? nullptr : bottom + ((int64_t)b * g.channels + c) * plane;
    float *dst = BACKWARD ? bottom_diff + ((int64_t)b * g.channels + c) * plane : nullptr;
    const float tdiff = BACKWARD ? top_diff[index] : 0.f;
    float acc = 0.f;
    for (int iy = 0; iy < gh; iy++) {
      const float yy = sh + ph * bh + (iy + .5f) * bh / (float)gh;
      for (int ix = 0; ix < gw; ix++) {
        const float xx = sw + pw * bw + (ix + .5f) * bw / (float)gw;
        for (int iz = 0; iz < gz; iz++) {
          const float zz = sz + pz * bz + (iz + .5f) * bz / (float)gz;
          float x = xx * ct + yy * st + cw;
          float y = yy * ct - xx * st + ch;
          float z = zz + cz;
          if (!BACKWARD) {
            acc += trilinear(src, g.height, g.width, g.zsize, y, x, z);
          } else {
            if (y < -1.0f || y > g.height || x < -1.0f || x > g.width || z < -1.0f || z > g.zsize) continue;
            if (y <= 0) y = 0;
            if (x <= 0) x = 0;
            if (z <= 0) z = 0;
            int y_low = (int)y, x_low = (int)x, z_low = (int)z, y_high, x_high, z_high;
            if (y_low >= g.height - 1) { y_high = y_low = g.height - 1; y = (float)y_low; } else y_high = y_low + 1;
            if (x_low >= g.width - 1) { x_high = x_low = g.width - 1; x = (float)x_low; } else x_high = x_low + 1;
            if (z_low >= g.zsize - 1) { z_high = z_low = g.zsize - 1; z = (float)z_low; } else z_high = z_low + 1;
            float ly = y - y_low, lx = x - x_low, lz = z - z_low;
            float hy = 1.f - ly, hx = 1.f - lx, hz = 1.f - lz;
            float w[8] = {hy * hx * hz, hy * lx * hz, ly * hx * hz, ly * lx * hz,
                          hy * hx * lz, hy * lx * lz, ly * hx * lz, ly * lx * lz};
            int64_t o[8] = {((int64_t)y_low * g.width + x_low) * g.zsize + z_low,
                            ((int64_t)y_low * g.width + x_high) * g.zsize + z_low,
                            ((int64_t)y_high * g.width + x_low) * g.zsize + z_low,
                            ((int64_t)y_high * g.width + x_high) * g.zsize + z_low,
                            ((int64_t)y_low * g.width + x_low) * g.zsize + z_high,
                            ((int64_t)y_low * g.width + x_high) * g.zsize + z_high,
                            ((int64_t)y_high * g.width + x_low) * g.zsize + z_high,
                            ((int64_t)y_high * g.width + x_high) * g.zsize + z_high};
#pragma unroll
            for (int q = 0; q < 8; ++q) atomicAdd(dst + o[q], tdiff * w[q] / count);
          }
        }
      }
    }
    if (!BACKWARD) top[index] = acc / count;
  }
}

} // namespace aabr
using namespace aabr;

extern "C" int aabr_sparse_to_dense_forward(const int32_t *site_coords, int64_t V, const float *in_feats,
                                            int planes, const int32_t *spatial_host, int64_t batch_size,
                                            float *out, void *stream_) {
  hipStream_t st = (hipStream_t)stream_;
  AABR_CHECK_ARG(V >= 0 && planes > 0 && spatial_host && batch_size >= 0, "bad arguments");
  const int X = spatial_host[0], Y = spatial_host[1], Z = spatial_host[2];
  AABR_CHECK_ARG(X > 0 && Y > 0 && Z > 0, "bad spatial size");
  int64_t total = batch_size * planes * (int64_t)X * Y * Z;
  if (total > 0) {
    AABR_CHECK_ARG(out, "null output");
    hipMemsetAsync(out, 0, total * sizeof(float), st); // output_features.zero_(), SparseToDense.cpp:49
  }
  if (V == 0) return AABR_OK;
  AABR_CHECK_ARG(site_coords && in_feats, "null pointer");
  hipLaunchKernelGGL(k_sparse_to_dense, dim3((unsigned)ceil_div(V * planes, 256)), dim3(256), 0, st, site_coords,
                     V, in_feats, planes, X, Y, Z, out, 0, (float *)nullptr);
  AABR_CHECK_LAUNCH();
  return AABR_OK;
}

extern "C" int aabr_sparse_to_dense_backward(const int32_t *site_coords, int64_t V, float *d_in_feats,
                                             int planes, const int32_t *spatial_host, const float *d_out,
                                             void *stream_) {
  AABR_CHECK_ARG(V >= 0 && planes > 0 && spatial_host, "bad arguments");
  if (V == 0) return AABR_OK;
  AABR_CHECK_ARG(site_coords && d_in_feats && d_out, "null pointer");
  hipLaunchKernelGGL(k_sparse_to_dense, dim3((unsigned)ceil_div(V * planes, 256)), dim3(256), 0,
                     (hipStream_t)stream_, site_coords, V, (const float *)nullptr, planes, spatial_host[0],
                     spatial_host[1], spatial_host[2], const_cast<float *>(d_out), 1, d_in_feats);
  AABR_CHECK_LAUNCH();
  return AABR_OK;
}

static int roi_geom(RoiGeom &g, int channels, int height, int width, int zsize, int ph, int pw, int pz,
                    int sampling, float scale) {
  if (channels <= 0 || height <= 0 || width <= 0 || zsize <= 0 || ph <= 0 || pw <= 0 || pz <= 0) return -1;
  g.channels = channels; g.height = height; g.width = width; g.zsize = zsize;
  g.ph = ph; g.pw = pw; g.pz = pz; g.sampling = sampling; g.scale = scale;
  return 0;
}

extern "C" int aabr_roi_align_rotated_3d_forward(const float *input, const float *rois, int64_t num_rois,
                                                 float spatial_scale, int channels, int height, int width,
                                                 int zsize, int pooled_h, int pooled_w, int pooled_z,
                                                 int sampling_ratio, float *output, void *stream_) {
  RoiGeom g;
  AABR_CHECK_ARG(num_rois >= 0 &&
                     roi_geom(g, channels, height, width, zsize, pooled_h, pooled_w, pooled_z, sampling_ratio,
                              spatial_scale) == 0,
                 "bad geometry");
  int64_t n = num_rois * channels * pooled_h * pooled_w * pooled_z;
  if (n == 0) return AABR_OK;
  AABR_CHECK_ARG(input && rois && output, "null pointer");
  int64_t blocks = ceil_div(n, 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(k_roi_align_rot3d<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream_, n,
                     input, rois, g, output, (const float *)nullptr, (float *)nullptr);
  AABR_CHECK_LAUNCH();
  return AABR_OK;
}

extern "C" int aabr_roi_align_rotated_3d_backward(const float *grad_output, const float *rois, int64_t num_rois,
                                                  float spatial_scale, int pooled_h, int pooled_w, int pooled_z,
                                                  int batch_size, int channels, int height, int width, int zsize,
                                                  int sampling_ratio, float *grad_input, void *stream_) {
  hipStream_t st = (hipStream_t)stream_;
  RoiGeom g;
  AABR_CHECK_ARG(num_rois >= 0 && batch_size >= 0 &&
                     roi_geom(g, channels, height, width, zsize, pooled_h, pooled_w, pooled_z, sampling_ratio,
                              spatial_scale) == 0,
                 "bad geometry");
  int64_t tot = (int64_t)batch_size * channels * height * width * zsize;
  if (tot > 0) {
    AABR_CHECK_ARG(grad_input, "null grad_input");
    hipMemsetAsync(grad_input, 0, tot * sizeof(float), st); // at::zeros, ROIAlignRotated3D_cuda.cu:418
  }
  int64_t n = num_rois * channels * pooled_h * pooled_w * pooled_z;
  if (n == 0) return AABR_OK;
  AABR_CHECK_ARG(grad_output && rois, "null pointer");
  int64_t blocks = ceil_div(n, 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(k_roi_align_rot3d<true>, dim3((unsigned)blocks), dim3(256), 0, st, n, (const float *)nullptr,
                     rois, g, (float *)nullptr, grad_output, grad_input);
  AABR_CHECK_LAUNCH();
  return AABR_OK;
}

namespace aabr {
// ------------------------------------------------------------------ fused sparse ROI-align
// ROIAlignRotated3D straight from the sparse feature matrix ("a sparse-gather ROI-align
// that never densifies").  The module's contract is unchanged -- densify, crop to the occupied extent,
// trilinear ROI-align (tools_3d_2d.py:7-48 + ROIAlignRotated3D_cuda.cu:16-346) -- but the dense
// [B, C, X, Y, Z] tensor is replaced by an int32 cell map (site row or -1) over the cropped extent:
//   forward : a workgroup owns one ROI x 128 planes; lanes are planes, so every active cell is one coalesced
//             512-byte read of a feature row, inactive cells (83 % of a typical map) cost one map lookup;
//             the arithmetic is the dense kernel's, statement for statement (zeros for inactive cells), so the
//             results are bit-identical to densify + `_C.roi_align_rotated_3d_forward`.  Bins are staged in
//             LDS and leave as contiguous runs per plane.
//   backward: same walk; only ACTIVE cells receive (coalesced, no-return) atomics, directly into the sparse
//             gradient rows: no dense gradient tensor, no fill, ~6x fewer atomics than the dense form
//             (which the reference also accumulates with atomics: summation order is not fixed there either).
constexpr int kRoiPlanes = 128; // planes per workgroup (lanes)
constexpr int kRoiBins = 96;    // bins staged in LDS per pass: 128 x 96 x 4 B = 48 KiB

__global__ __launch_bounds__(256) void k_roi_cellmap(const int32_t *__restrict__ sc, int64_t V, int X, int Y, int Z,
                                                     int B, int32_t *__restrict__ cellmap) {
  int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V) return;
  int4 c = *reinterpret_cast<const int4 *>(sc + 4 * v);
  if (c.x < X && c.y < Y && c.z < Z && c.w < B) cellmap[(((int64_t)c.w * X + c.x) * Y + c.y) * Z + c.z] = (int32_t)v;
}

// geometry of one sample point: the 8 corner cells and weights exactly as the dense kernels compute them
struct RoiCorner { int64_t o[8]; float w[8]; bool ok; };

__device__ inline RoiCorner roi_corners(const RoiGeom &g, float y, float x, float z, bool backward) {
  RoiCorner r;
  r.ok = !(y < -1.0f || y > g.height || x < -1.0f || x > g.width || z < -1.0f || (backward && z > g.zsize));
  if (!r.ok) return r;
  if (y <= 0) y = 0;
  if (x <= 0) x = 0;
  if (z <= 0) z = 0;
  int y_low = (int)y, x_low = (int)x, z_low = (int)z, y_high, x_high, z_high;
  if (y_low >= g.height - 1) { y_high = y_low = g.height - 1; y = (float)y_low; } else y_high = y_low + 1;
  if (x_low >= g.width - 1) { x_high = x_low = g.width - 1; x = (float)x_low; } else x_high = x_low + 1;
  if (z_low >= g.zsize - 1) { z_high = z_low = g.zsize - 1; z = (float)z_low; } else z_high = z_low + 1;
  const float ly = y - y_low, lx = x - x_low, lz = z - z_low;
  const float hy = 1.f - ly, hx = 1.f - lx, hz = 1.f - lz;
  r.w[0] = hy * hx * hz; r.w[1] = hy * lx * hz; r.w[2] = ly * hx * hz; r.w[3] = ly * lx * hz;
  r.w[4] = hy * hx * lz; r.w[5] = hy * lx * lz; r.w[6] = ly * hx * lz; r.w[7] = ly * lx * lz;
  r.o[0] = ((int64_t)y_low * g.width + x_low) * g.zsize + z_low;
  r.o[1] = ((int64_t)y_low * g.width + x_high) * g.zsize + z_low;
  r.o[2] = ((int64_t)y_high * g.width + x_low) * g.zsize + z_low;
  r.o[3] = ((int64_t)y_high * g.width + x_high) * g.zsize + z_low;
  r.o[4] = ((int64_t)y_low * g.width + x_low) * g.zsize + z_high;
  r.o[5] = ((int64_t)y_low * g.width + x_high) * g.zsize + z_high;
  r.o[6] = ((int64_t)y_high * g.width + x_low) * g.zsize + z_high;
  r.o[7] = ((int64_t)y_high * g.width + x_high) * g.zsize + z_high;
  return r;
}

template <bool BACKWARD>
__global__ __launch_bounds__(256) void k_roi_align_rot3d_sparse(const float *__restrict__ feats, int C,
                                                                const int32_t *__restrict__ cellmap, int B,
                                                                const float *__restrict__ rois, RoiGeom g,
                                                                float *__restrict__ top,
                                                                const float *__restrict__ top_diff,
                                                                float *d_feats) {
  extern __shared__ float stage[]; // [kRoiPlanes][kRoiBins] bins of this pass (forward: results; backward: top_diff)
  const int64_t n = blockIdx.x;
  const int c0 = blockIdx.y * kRoiPlanes;
  const int lane_c = threadIdx.x & (kRoiPlanes - 1), half = threadIdx.x >> 7; // two bin streams per workgroup
  const int c = c0 + lane_c;
  const bool c_ok = c < C;
  const int nbins = g.ph * g.pw * g.pz;
  const float *r = rois + n * 8;
  const int b = (int)r[0];
  const float cw = r[1] * g.scale, ch = r[2] * g.scale, cz = r[3] * g.scale;
  float rw = r[4] * g.scale, rh = r[5] * g.scale, rz = r[6] * g.scale;
  const float theta = (float)(r[7] * 3.14159265358979323846 / 180.0);
  rw = fmaxf(rw, 1.f); rh = fmaxf(rh, 1.f); rz = fmaxf(rz, 1.f);
  const float bh = rh / (float)g.ph, bw = rw / (float)g.pw, bz = rz / (float)g.pz;
  const int gh = g.sampling > 0 ? g.sampling : (int)ceilf(rh / g.ph);
  const int gw = g.sampling > 0 ? g.sampling : (int)ceilf(rw / g.pw);
  const int gz = g.sampling > 0 ? g.sampling : (int)ceilf(rz / g.pz);
  const float sh = -rh / 2.0f, sw = -rw / 2.0f, sz = -rz / 2.0f;
  const float ct = cosf(theta), st = sinf(theta);
  const float count = (float)(gh * gw * gz);
  const bool b_ok = b >= 0 && b < B;
  const int32_t *cm = cellmap + (int64_t)(b_ok ? b : 0) * g.height * g.width * g.zsize;
  const int64_t obase = (n * C + c0) * (int64_t)nbins; // top[n][c0 + .][bin]

  for (int bin0 = 0; bin0 < nbins; bin0 += kRoiBins) {
    const int nb = (nbins - bin0) < kRoiBins ? (nbins - bin0) : kRoiBins;
    if (BACKWARD) { // stage this pass's output gradients: contiguous runs per plane
      for (int i = threadIdx.x; i < kRoiPlanes * nb; i += 256) {
        const int pc = i / nb, pb = i - pc * nb;
        stage[pc * kRoiBins + pb] = (c0 + pc < C) ? top_diff[obase + (int64_t)pc * nbins + bin0 + pb] : 0.f;
      }
      __syncthreads();
    }
    for (int lb = half; lb < nb; lb += 2) {
      const int bin = bin0 + lb;
      const int pz = bin % g.pz, pw = (bin / g.pz) % g.pw, ph = bin / g.pz / g.pw;
      const float tdiff = BACKWARD ? stage[lane_c * kRoiBins + lb] : 0.f;
      float acc = 0.f;
      for (int iy = 0; iy < gh; iy++) {
        const float yy = sh + ph * bh + (iy + .5f) * bh / (float)gh;
        for (int ix = 0; ix < gw; ix++) {
          const float xx = sw + pw * bw + (ix + .5f) * bw / (float)gw;
          for (int iz = 0; iz < gz; iz++) {
            const float zz = sz + pz * bz + (iz + .5f) * bz / (float)gz;
            const float x = xx * ct + yy * st + cw;
            const float y = yy * ct - xx * st + ch;
            const float z = zz + cz;
            const RoiCorner q = roi_corners(g, y, x, z, BACKWARD);
            if (!q.ok) continue; // forward: trilinear() returns 0, acc += 0 changes nothing
            if (!BACKWARD) {
              float v[8];
#pragma unroll
              for (int t = 0; t < 8; ++t) {
                const int32_t row = b_ok ? cm[q.o[t]] : -1; // workgroup-uniform address: one broadcast load
                v[t] = (row >= 0 && c_ok) ? feats[(int64_t)row * C + c] : 0.f;
              }
              acc += (q.w[0] * v[0] + q.w[1] * v[1] + q.w[2] * v[2] + q.w[3] * v[3] + q.w[4] * v[4] + q.w[5] * v[5] +
                      q.w[6] * v[6] + q.w[7] * v[7]);
            } else {
#pragma unroll
              for (int t = 0; t < 8; ++t) {
                const int32_t row = b_ok ? cm[q.o[t]] : -1;
                if (row >= 0 && c_ok) atomicAdd(d_feats + (int64_t)row * C + c, tdiff * q.w[t] / count);
              }
            }
          }
        }
      }
      if (!BACKWARD) stage[lane_c * kRoiBins + lb] = acc / count;
    }
    __syncthreads();
    if (!BACKWARD) {
      for (int i = threadIdx.x; i < kRoiPlanes * nb; i += 256) {
        const int pc = i / nb, pb = i - pc * nb;
        if (c0 + pc < C) top[obase + (int64_t)pc * nbins + bin0 + pb] = stage[pc * kRoiBins + pb];
      }
      __syncthreads();
    }
  }
}

} // namespace aabr
using namespace aabr;

extern "C" int aabr_roi_cellmap(const int32_t *site_coords, int64_t V, const int32_t *extent_host, int batch_size,
                                int32_t *cellmap, void *stream_) {
  hipStream_t st = (hipStream_t)stream_;
  AABR_CHECK_ARG(V >= 0 && extent_host && batch_size >= 0, "bad arguments");
  const int X = extent_host[0], Y = extent_host[1], Z = extent_host[2];
  AABR_CHECK_ARG(X > 0 && Y > 0 && Z > 0, "bad extent");
  const int64_t cells = (int64_t)batch_size * X * Y * Z;
  if (cells == 0) return AABR_OK;
  AABR_CHECK_ARG(cellmap, "null cellmap");
  hipMemsetAsync(cellmap, 0xFF, cells * sizeof(int32_t), st);
  if (V == 0) return AABR_OK;
  AABR_CHECK_ARG(site_coords, "null pointer");
  hipLaunchKernelGGL(k_roi_cellmap, dim3((unsigned)ceil_div(V, 256)), dim3(256), 0, st, site_coords, V, X, Y, Z,
                     batch_size, cellmap);
  AABR_CHECK_LAUNCH();
  return AABR_OK;
}

extern "C" int aabr_roi_align_rotated_3d_sparse_forward(const float *feats, int channels, const int32_t *cellmap,
                                                        int batch_size, int height, int width, int zsize,
                                                        const float *rois, int64_t num_rois, float spatial_scale,
                                                        int pooled_h, int pooled_w, int pooled_z, int sampling_ratio,
                                                        float *output, void *stream_) {
  RoiGeom g;
  AABR_CHECK_ARG(num_rois >= 0 && batch_size >= 0 &&
                     roi_geom(g, channels, height, width, zsize, pooled_h, pooled_w, pooled_z, sampling_ratio,
                              spatial_scale) == 0,
                 "bad geometry");
  if (num_rois == 0) return AABR_OK;
  AABR_CHECK_ARG(feats && cellmap && rois && output && batch_size > 0, "null pointer / empty batch");
  AABR_CHECK_ARG(ceil_div(channels, kRoiPlanes) <= 65535, "too many planes");
  hipLaunchKernelGGL(k_roi_align_rot3d_sparse<false>, dim3((unsigned)num_rois, (unsigned)ceil_div(channels, kRoiPlanes)),
                     dim3(256), (size_t)kRoiPlanes * kRoiBins * sizeof(float), (hipStream_t)stream_, feats, channels,
                     cellmap, batch_size, rois, g, output, (const float *)nullptr, (float *)nullptr);
  AABR_CHECK_LAUNCH();
  return AABR_OK;
}

extern "C" int aabr_roi_align_rotated_3d_sparse_backward(const float *grad_output, int channels,
                                                         const int32_t *cellmap, int batch_size, int height,
                                                         int width, int zsize, const float *rois, int64_t num_rois,
                                                         float spatial_scale, int pooled_h, int pooled_w, int pooled_z,
                                                         int sampling_ratio, int64_t V, float *d_feats, void *stream_) {
  hipStream_t st = (hipStream_t)stream_;
  RoiGeom g;
  AABR_CHECK_ARG(num_rois >= 0 && batch_size >= 0 && V >= 0 &&
                     roi_geom(g, channels, height, width, zsize, pooled_h, pooled_w, pooled_z, sampling_ratio,
                              spatial_scale) == 0,
                 "bad geometry");
  if (V > 0) {
    AABR_CHECK_ARG(d_feats, "null d_feats");
    hipMemsetAsync(d_feats, 0, (size_t)V * channels * sizeof(float), st);
  }
  if (num_rois == 0 || V == 0) return AABR_OK;
  AABR_CHECK_ARG(grad_output && cellmap && rois && batch_size > 0, "null pointer / empty batch");
  AABR_CHECK_ARG(ceil_div(channels, kRoiPlanes) <= 65535, "too many planes");
  hipLaunchKernelGGL(k_roi_align_rot3d_sparse<true>, dim3((unsigned)num_rois, (unsigned)ceil_div(channels, kRoiPlanes)),
                     dim3(256), (size_t)kRoiPlanes * kRoiBins * sizeof(float), st, (const float *)nullptr, channels,
                     cellmap, batch_size, rois, g, (float *)nullptr, grad_output, d_feats);
  AABR_CHECK_LAUNCH();
  return AABR_OK;
}
