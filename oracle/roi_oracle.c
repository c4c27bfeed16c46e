/*
 * roi_oracle.c -- CPU restatement of SparseToDense and the rotated 3-D ROI align.
 * TEST INFRASTRUCTURE ONLY (see scn_oracle.c header).
 *
 * Follows SparseConvNet/sparseconvnet/SCN/CPU/SparseToDense.cpp:7-87 (+ the rule layout of
 * SCN/Metadata/ConvolutionRules.h:109-130) and
 * maskrcnn_benchmark/csrc/cuda/ROIAlignRotated3D_cuda.cu:16-346 (the reference has NO CPU version of
 * the ROI align, csrc/ROIAlignRotated3D.h:25,47, and no test vectors: "parity unpinned"; this is a
 * serial restatement of the CUDA kernels' arithmetic, including the forward pass's missing upper-z
 * bound check (:27, `zsize > zsize`)).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

void oracle_sparse_to_dense_fwd(const int64_t *sc, int64_t V, const float *in, int planes, const int64_t *sp,
                                int64_t batch, float *out) {
  int64_t vol = sp[0] * sp[1] * sp[2];
  memset(out, 0, (size_t)(batch * planes * vol) * sizeof(float));
  for (int64_t v = 0; v < V; ++v) {
    int64_t off = (sc[4 * v] * sp[1] + sc[4 * v + 1]) * sp[2] + sc[4 * v + 2];
    for (int p = 0; p < planes; ++p) out[(sc[4 * v + 3] * planes + p) * vol + off] = in[v * planes + p];
  }
}
void oracle_sparse_to_dense_bwd(const int64_t *sc, int64_t V, float *d_in, int planes, const int64_t *sp,
                                const float *d_out) {
  int64_t vol = sp[0] * sp[1] * sp[2];
  for (int64_t v = 0; v < V; ++v) {
    int64_t off = (sc[4 * v] * sp[1] + sc[4 * v + 1]) * sp[2] + sc[4 * v + 2];
    for (int p = 0; p < planes; ++p) d_in[v * planes + p] = d_out[(sc[4 * v + 3] * planes + p) * vol + off];
  }
}

static int corners(int height, int width, int zsize, float *y, float *x, float *z, int *yl, int *yh, int *xl,
                   int *xh, int *zl, int *zh, float *w) {
  if (*y <= 0) *y = 0;
  if (*x <= 0) *x = 0;
  if (*z <= 0) *z = 0;
  *yl = (int)*y; *xl = (int)*x; *zl = (int)*z;
  if (*yl >= height - 1) { *yh = *yl = height - 1; *y = (float)*yl; } else *yh = *yl + 1;
  if (*xl >= width - 1) { *xh = *xl = width - 1; *x = (float)*xl; } else *xh = *xl + 1;
  if (*zl >= zsize - 1) { *zh = *zl = zsize - 1; *z = (float)*zl; } else *zh = *zl + 1;
  float ly = *y - *yl, lx = *x - *xl, lz = *z - *zl, hy = 1.f - ly, hx = 1.f - lx, hz = 1.f - lz;
  w[0] = hy * hx * hz; w[1] = hy * lx * hz; w[2] = ly * hx * hz; w[3] = ly * lx * hz;
  w[4] = hy * hx * lz; w[5] = hy * lx * lz; w[6] = ly * hx * lz; w[7] = ly * lx * lz;
  return 0;
}

void oracle_roi_align_rot3d(const float *bottom, const float *rois, int64_t num_rois, float scale, int channels,
                            int height, int width, int zsize, int PH, int PW, int PZ, int sampling, float *top,
                            const float *top_diff, float *bottom_diff, int backward) {
  int64_t plane = (int64_t)height * width * zsize;
  int64_t n_out = num_rois * channels * PH * PW * PZ;
  for (int64_t index = 0; index < n_out; ++index) {
    int pz = (int)(index % PZ), pw = (int)((index / PZ) % PW), ph = (int)((index / PZ / PW) % PH);
    int c = (int)((index / PZ / PW / PH) % channels);
    int64_t n = index / PZ / PW / PH / channels;
    const float *r = rois + n * 8;
    int b = (int)r[0];
    float cw = r[1] * scale, ch = r[2] * scale, cz = r[3] * scale;
    float rw = r[4] * scale, rh = r[5] * scale, rz = r[6] * scale;
    float theta = (float)(r[7] * 3.14159265358979323846 / 180.0);
    rw = fmaxf(rw, 1.f); rh = fmaxf(rh, 1.f); rz = fmaxf(rz, 1.f);
    float bh = rh / (float)PH, bw = rw / (float)PW, bz = rz / (float)PZ;
    int gh = sampling > 0 ? sampling : (int)ceilf(rh / PH);
    int gw = sampling > 0 ? sampling : (int)ceilf(rw / PW);
    int gz = sampling > 0 ? sampling : (int)ceilf(rz / PZ);
    float sh = -rh / 2.0f, sw = -rw / 2.0f, sz = -rz / 2.0f, ct = cosf(theta), st = sinf(theta);
    float count = (float)(gh * gw * gz), acc = 0.f;
    const float *src = bottom ? bottom + ((int64_t)b * channels + c) * plane : 0;
    float *dst = bottom_diff ? bottom_diff + ((int64_t)b * channels + c) * plane : 0;
    for (int iy = 0; iy < gh; iy++) {
      float yy = sh + ph * bh + (iy + .5f) * bh / (float)gh;
      for (int ix = 0; ix < gw; ix++) {
        float xx = sw + pw * bw + (ix + .5f) * bw / (float)gw;
        for (int iz = 0; iz < gz; iz++) {
          float zz = sz + pz * bz + (iz + .5f) * bz / (float)gz;
          float x = xx * ct + yy * st + cw, y = yy * ct - xx * st + ch, z = zz + cz;
          int yl, yh, xl, xh, zl, zh;
          float w[8];
          if (!backward) {
            if (y < -1.0f || y > height || x < -1.0f || x > width || z < -1.0f) continue;
          } else {
            if (y < -1.0f || y > height || x < -1.0f || x > width || z < -1.0f || z > zsize) continue;
          }
          corners(height, width, zsize, &y, &x, &z, &yl, &yh, &xl, &xh, &zl, &zh, w);
          int64_t o[8] = {((int64_t)yl * width + xl) * zsize + zl, ((int64_t)yl * width + xh) * zsize + zl,
                          ((int64_t)yh * width + xl) * zsize + zl, ((int64_t)yh * width + xh) * zsize + zl,
                          ((int64_t)yl * width + xl) * zsize + zh, ((int64_t)yl * width + xh) * zsize + zh,
                          ((int64_t)yh * width + xl) * zsize + zh, ((int64_t)yh * width + xh) * zsize + zh};
          if (!backward) {
            acc += (w[0] * src[o[0]] + w[1] * src[o[1]] + w[2] * src[o[2]] + w[3] * src[o[3]] + w[4] * src[o[4]] +
                    w[5] * src[o[5]] + w[6] * src[o[6]] + w[7] * src[o[7]]);
          } else {
            for (int q = 0; q < 8; ++q) dst[o[q]] += top_diff[index] * w[q] / count;
          }
        }
      }
    }
    if (!backward) top[index] = acc / count;
  }
}
