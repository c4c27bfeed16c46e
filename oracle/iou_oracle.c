/*
 * iou_oracle.c -- CPU restatement of the reference's rotated-box IoU and the
 * rotated-3D NMS decision rule.
 *
 * TEST INFRASTRUCTURE ONLY (see scn_oracle.c header): never linked, imported
 * or called by the product path.
 *
 * Follows (reference file:line under /root/reference):
 *   second/core/non_max_suppression/nms_gpu.py
 *     :166-179 trangle_area / area         :182-219 sort_vertex_in_convex_polygon
 *     :222-265 line_segment_intersection   :310-326 point_in_quadrilateral
 *     :331-352 quadrilateral_intersection  :355-378 rbbox_to_corners
 *     :381-395 inter                       :552-623 devRotateIoUEval
 *     :626-664 rotate_iou_kernel_eval (argument order: query box first)
 *     :706-717 check_same_boxes
 *   utils3d/rotate_nms_3d_torch.py:7-90   boxes_iou_3d (+ iou_one_dim)
 *   second/core/non_max_suppression/nms_cpu.py:32-44 rotate_nms_3d_cc
 *   second/pytorch/core/box_torch_ops.py:557-582 rotate_nms_3d
 *
 * All arithmetic is float32 like the numba.cuda kernels (signatures are
 * float32[:], fastmath=False); compile with -ffp-contract=off.
 *
 * PINNING: the numba device functions are plain Python bodies; golden IoU
 * matrices produced by executing those bodies (tests/golden/gen_iou_golden.py)
 * pin this file.  The final suppression loop lives in the un-vendored
 * third-party spconv 1.x (`rotate_non_max_suppression_cpu`, docs/Install.md:147-156,
 * no commit pinned, no reference test holds its outputs): that step is
 * "parity unpinned"; the rule implemented here is the published spconv-1.x one:
 * greedy over descending score, j suppressed by a kept i iff the pre-filter
 * matrix entry (i,j) > 0 and polygon IoU(i,j) >= thresh, where the polygon IoU
 * is taken to be the same rotated-rectangle IoU value.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static float tri_area(const float *a, const float *b, const float *c) {
  return ((a[0] - c[0]) * (b[1] - c[1]) - (a[1] - c[1]) * (b[0] - c[0])) / 2.0f;
}

static float poly_area(const float *pts, int n) {
  float s = 0.0f;
  for (int i = 0; i < n - 2; ++i)
    s += fabsf(tri_area(pts, pts + 2 * i + 2, pts + 2 * i + 4));
  return s;
}

/* insertion sort of vertices by a monotone pseudo-angle about the centroid */
static void sort_vertices(float *pts, int n) {
  if (n <= 0) return;
  float cx = 0.0f, cy = 0.0f;
  for (int i = 0; i < n; ++i) { cx += pts[2 * i]; cy += pts[2 * i + 1]; }
  cx /= n; cy /= n;
  float vs[24];
  for (int i = 0; i < n; ++i) {
    float vx = pts[2 * i] - cx, vy = pts[2 * i + 1] - cy;
    float d = sqrtf(vx * vx + vy * vy);
    vx = vx / d; vy = vy / d;
    if (vy < 0) vx = -2 - vx;
    vs[i] = vx;
  }
  for (int i = 1; i < n; ++i) {
    if (vs[i - 1] > vs[i]) {
      float t = vs[i], tx = pts[2 * i], ty = pts[2 * i + 1];
      int j = i;
      while (j > 0 && vs[j - 1] > t) {
        vs[j] = vs[j - 1];
        pts[2 * j] = pts[2 * j - 2];
        pts[2 * j + 1] = pts[2 * j - 1];
        --j;
      }
      vs[j] = t; pts[2 * j] = tx; pts[2 * j + 1] = ty;
    }
  }
}

static int seg_intersect(const float *p1, const float *p2, int i, int j,
                         float *out) {
  float A0 = p1[2 * i], A1 = p1[2 * i + 1];
  float B0 = p1[2 * ((i + 1) % 4)], B1 = p1[2 * ((i + 1) % 4) + 1];
  float C0 = p2[2 * j], C1 = p2[2 * j + 1];
  float D0 = p2[2 * ((j + 1) % 4)], D1 = p2[2 * ((j + 1) % 4) + 1];
  float BA0 = B0 - A0, BA1 = B1 - A1;
  float DA0 = D0 - A0, CA0 = C0 - A0, DA1 = D1 - A1, CA1 = C1 - A1;
  int acd = DA1 * CA0 > CA1 * DA0;
  int bcd = (D1 - B1) * (C0 - B0) > (C1 - B1) * (D0 - B0);
  if (acd != bcd) {
    int abc = CA1 * BA0 > BA1 * CA0;
    int abd = DA1 * BA0 > BA1 * DA0;
    if (abc != abd) {
      float DC0 = D0 - C0, DC1 = D1 - C1;
      float ABBA = A0 * B1 - B0 * A1;
      float CDDC = C0 * D1 - D0 * C1;
      float DH = BA1 * DC0 - BA0 * DC1;
      float Dx = ABBA * DC0 - BA0 * CDDC;
      float Dy = ABBA * DC1 - BA1 * CDDC;
      out[0] = Dx / DH; out[1] = Dy / DH;
      return 1;
    }
  }
  return 0;
}

static int pt_in_quad(float x, float y, const float *c) {
  float ab0 = c[2] - c[0], ab1 = c[3] - c[1];
  float ad0 = c[6] - c[0], ad1 = c[7] - c[1];
  float ap0 = x - c[0], ap1 = y - c[1];
  float abab = ab0 * ab0 + ab1 * ab1, abap = ab0 * ap0 + ab1 * ap1;
  float adad = ad0 * ad0 + ad1 * ad1, adap = ad0 * ap0 + ad1 * ap1;
  return abab >= abap && abap >= 0 && adad >= adap && adap >= 0;
}

/* up to 8 + 16 candidate vertices; the reference's local array holds 8 points
 * (16 floats) and is overrun (undefined behaviour) beyond that -- this
 * restatement keeps all of them in a 24-point buffer. */
static int quad_intersection(const float *p1, const float *p2, float *ip) {
  int n = 0;
  for (int i = 0; i < 4; ++i) {
    if (pt_in_quad(p1[2 * i], p1[2 * i + 1], p2)) {
      ip[2 * n] = p1[2 * i]; ip[2 * n + 1] = p1[2 * i + 1]; ++n;
    }
    if (pt_in_quad(p2[2 * i], p2[2 * i + 1], p1)) {
      ip[2 * n] = p2[2 * i]; ip[2 * n + 1] = p2[2 * i + 1]; ++n;
    }
  }
  float t[2];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j)
      if (seg_intersect(p1, p2, i, j, t)) {
        ip[2 * n] = t[0]; ip[2 * n + 1] = t[1]; ++n;
      }
  return n;
}

static void box_corners(float *c, const float *rb) {
  /* math.cos / math.sin evaluate in double; the product with a float32 operand narrows the
   * factor to float32 first (rbbox_to_corners, nms_gpu.py:358-359,375-378) */
  float ang = rb[4], ac = (float)cos((double)ang), as = (float)sin((double)ang);
  float cx = rb[0], cy = rb[1], xd = rb[2], yd = rb[3];
  float px[4] = {-xd / 2, -xd / 2, xd / 2, xd / 2};
  float py[4] = {-yd / 2, yd / 2, yd / 2, -yd / 2};
  for (int i = 0; i < 4; ++i) {
    c[2 * i] = ac * px[i] + as * py[i] + cx;
    c[2 * i + 1] = -as * px[i] + ac * py[i] + cy;
  }
}

static float inter_area(const float *r1, const float *r2) {
  float c1[8], c2[8], ip[48];
  box_corners(c1, r1);
  box_corners(c2, r2);
  int n = quad_intersection(c1, c2, ip);
  sort_vertices(ip, n);
  return poly_area(ip, n);
}

/* devRotateIoUEval, all criteria (nms_gpu.py:552-623) */
float oracle_rotate_iou_pair(const float *r1, const float *r2, int criterion) {
  float area1 = r1[2] * r1[3], area2 = r2[2] * r2[3];
  float ai = inter_area(r1, r2);
  /* numba types `x ** 2` / `x ** 0.5` on float32 as float64: the distance terms
   * are evaluated in double and narrowed when stored to the float32 output */
  double l1x = r1[0] + r1[2] * 0.5, l1y = r1[1];
  double w1x = r1[0], w1y = r1[1] + r1[3] * 0.5;
  double l2x = r2[0] + r2[2] * 0.5, l2y = r2[1];
  double w2x = r2[0], w2y = r2[1] + r2[3] * 0.5;
  double d0 = (double)(float)(r1[0] - r2[0]), d1 = (double)(float)(r1[1] - r2[1]);
  double dc = sqrt(d0 * d0 + d1 * d1);
  double diag = dc + sqrt((double)r1[2] * r1[2] + (double)r2[0] * r2[0]) * 0.5 +
                sqrt((double)r2[2] * r2[2] + (double)r2[0] * r2[0]) * 0.5;
  double Dl = sqrt((l1x - l2x) * (l1x - l2x) + (l1y - l2y) * (l1y - l2y));
  double Dw = sqrt((w1x - w2x) * (w1x - w2x) + (w1y - w2y) * (w1y - w2y));
  const double pi = 3.14159265358979323846;
  switch (criterion) {
  case -1: return ai / (area1 + area2 - ai);
  case 3: {
    double iou = ai / (area1 + area2 - ai);
    double diou = 1 - dc * dc / (diag * diag);
    double da = atan((double)(r1[2] / r1[3])) - atan((double)(r2[2] / r2[3]));
    double aiou = 1 - (4 / (pi * pi)) * da * da;
    return (float)(iou * 0 + diou * 0.1 + aiou * 0.1);
  }
  case 4: return (float)(2 - (Dl + Dw + 1.5 * dc) / 1);
  case 5: {
    float long1 = r1[2] >= r1[3] ? r1[2] : r1[3];
    float long2 = r2[2] >= r2[3] ? r2[2] : r2[3];
    double da = atan((double)(r1[2] / r1[3])) - atan((double)(r2[2] / r2[3]));
    double aiou = (4 / (pi * pi)) * da * da;
    double dl = (double)(float)(long1 - long2);
    return (float)(1 - (sqrt(dl * dl) + 0 + dc) / 0.5 + 0.2 * aiou);
  }
  case 6: {
    double m = (double)fabsf(r1[2] - r2[2]) + (double)fabsf(r1[3] - r2[3]) + dc;
    return (float)(1 - m / 0.7);
  }
  case 0: return ai / area1;
  case 1: return ai / area2;
  case 2: {
    float mn = r2[2] < r2[3] ? r2[2] : r2[3], mx = r2[2] > r2[3] ? r2[2] : r2[3];
    if (mn / mx < 0.25f) {
      float e = area1 * 0.5f - ai;
      return ai / (area2 + (e > 0 ? e : 0));
    }
    return ai / (area1 + area2 - ai);
  }
  default: return ai;
  }
}

/* rotate_iou_gpu_eval (nms_gpu.py:667-703): iou[n,k] =
 * devRotateIoUEval(query k, box n) then check_same_boxes forces 1.0 where all
 * five parameters differ by < 1e-6. */
void oracle_rotate_iou_eval(const float *boxes, int64_t N, const float *query,
                            int64_t K, int criterion, float *iou) {
#pragma omp parallel for
  for (int64_t n = 0; n < N; ++n)
    for (int64_t k = 0; k < K; ++k) {
      const float *b = boxes + 5 * n, *q = query + 5 * k;
      float v = oracle_rotate_iou_pair(q, b, criterion);
      int same = 1;
      for (int d = 0; d < 5; ++d)
        if (!(fabsf(b[d] - q[d]) < 1e-6f)) same = 0;
      iou[n * K + k] = same ? 1.0f : v;
    }
}

/* boxes_iou_3d (rotate_nms_3d_torch.py:23-90).  Boxes are [.,7] "yx_zb":
 * xc, yc, z_bottom, size_y(thickness), size_x, size_z, yaw.  aug[4] =
 * {target_Y, target_Z, anchor_Y, anchor_Z} minima.  only_xy != 0 returns the
 * 2-D rotated IoU (the reference's module-level DEBUG=1 forces that). */
void oracle_boxes_iou_3d(const float *targets, int64_t M, const float *anchors,
                         int64_t K, const float *aug, int criterion,
                         int only_xy, float *iou) {
  float *t2 = (float *)malloc((size_t)M * 5 * sizeof(float));
  float *a2 = (float *)malloc((size_t)K * 5 * sizeof(float));
  float *tz = (float *)malloc((size_t)M * 2 * sizeof(float));
  float *az = (float *)malloc((size_t)K * 2 * sizeof(float));
  for (int64_t i = 0; i < M; ++i) {
    const float *b = targets + 7 * i;
    float th = b[3] < aug[0] ? aug[0] : b[3], h = b[5] < aug[1] ? aug[1] : b[5];
    t2[5 * i] = b[0]; t2[5 * i + 1] = b[1]; t2[5 * i + 2] = th;
    t2[5 * i + 3] = b[4]; t2[5 * i + 4] = b[6];
    tz[2 * i] = b[2]; tz[2 * i + 1] = b[2] + h;
  }
  for (int64_t i = 0; i < K; ++i) {
    const float *b = anchors + 7 * i;
    float th = b[3] < aug[2] ? aug[2] : b[3], h = b[5] < aug[3] ? aug[3] : b[5];
    a2[5 * i] = b[0]; a2[5 * i + 1] = b[1]; a2[5 * i + 2] = th;
    a2[5 * i + 3] = b[4]; a2[5 * i + 4] = b[6];
    az[2 * i] = b[2]; az[2 * i + 1] = b[2] + h;
  }
  oracle_rotate_iou_eval(t2, M, a2, K, criterion, iou);
  if (!only_xy)
    for (int64_t i = 0; i < M; ++i)
      for (int64_t j = 0; j < K; ++j) {
        float hi = az[2 * j + 1] < tz[2 * i + 1] ? az[2 * j + 1] : tz[2 * i + 1];
        float lo = az[2 * j] > tz[2 * i] ? az[2 * j] : tz[2 * i];
        float HI = az[2 * j + 1] > tz[2 * i + 1] ? az[2 * j + 1] : tz[2 * i + 1];
        float LO = az[2 * j] < tz[2 * i] ? az[2 * j] : tz[2 * i];
        iou[i * K + j] *= (hi - lo) / (HI - LO);
      }
  free(t2); free(a2); free(tz); free(az);
}

/* Greedy suppression over a precomputed IoU matrix (the contract of
 * spconv.utils.rotate_non_max_suppression_cpu as called from nms_cpu.py:43):
 * order[] = indices by descending score; returns number kept. */
int64_t oracle_nms_from_matrix(const float *iou, int64_t n,
                               const int32_t *order, float thresh,
                               int64_t *keep) {
  char *sup = (char *)calloc(n, 1);
  int64_t nk = 0;
  for (int64_t _i = 0; _i < n; ++_i) {
    int64_t i = order[_i];
    if (sup[i]) continue;
    keep[nk++] = i;
    for (int64_t _j = _i + 1; _j < n; ++_j) {
      int64_t j = order[_j];
      if (sup[j]) continue;
      float v = iou[i * n + j];
      if (v <= 0.0f) continue;   /* standup/IoU pre-filter */
      if (v >= thresh) sup[j] = 1;
    }
  }
  free(sup);
  return nk;
}

/* The same loop with the decision taken on a second matrix: `pre` is what the reference hands spconv as the `> 0`
 * pre-filter (boxes_iou_3d, nms_cpu.py:36-43), `dec` the exact polygon IoU spconv 1.x computes itself with
 * boost::geometry and compares with `>= thresh` (restated independently in clip_oracle.c). */
int64_t oracle_nms_prefilter_decide(const float *pre, const double *dec, int64_t n, const int32_t *order,
                                    float thresh, int64_t *keep) {
  char *sup = (char *)calloc(n, 1);
  int64_t nk = 0;
  for (int64_t _i = 0; _i < n; ++_i) {
    int64_t i = order[_i];
    if (sup[i]) continue;
    keep[nk++] = i;
    for (int64_t _j = _i + 1; _j < n; ++_j) {
      int64_t j = order[_j];
      if (sup[j]) continue;
      if (pre[i * n + j] <= 0.0f) continue;
      if (dec[i * n + j] >= (double)thresh) sup[j] = 1;
    }
  }
  free(sup);
  return nk;
}

/* maskrcnn_benchmark/csrc/cpu/nms_cpu.cpp:5-75: axis-aligned NMS with the +1
 * pixel convention; returns kept indices in ascending index order. */
int64_t oracle_nms_axis_aligned(const float *dets, const float *scores,
                                int64_t n, float thresh, int64_t *keep) {
  int64_t *order = (int64_t *)malloc(n * sizeof(int64_t));
  char *sup = (char *)calloc(n, 1);
  for (int64_t i = 0; i < n; ++i) order[i] = i;
  /* stable insertion sort, descending score */
  for (int64_t i = 1; i < n; ++i) {
    int64_t o = order[i], j = i;
    while (j > 0 && scores[order[j - 1]] < scores[o]) { order[j] = order[j - 1]; --j; }
    order[j] = o;
  }
  for (int64_t _i = 0; _i < n; ++_i) {
    int64_t i = order[_i];
    if (sup[i]) continue;
    float ix1 = dets[4 * i], iy1 = dets[4 * i + 1], ix2 = dets[4 * i + 2],
          iy2 = dets[4 * i + 3];
    float iarea = (ix2 - ix1 + 1) * (iy2 - iy1 + 1);
    for (int64_t _j = _i + 1; _j < n; ++_j) {
      int64_t j = order[_j];
      if (sup[j]) continue;
      float xx1 = fmaxf(ix1, dets[4 * j]), yy1 = fmaxf(iy1, dets[4 * j + 1]);
      float xx2 = fminf(ix2, dets[4 * j + 2]), yy2 = fminf(iy2, dets[4 * j + 3]);
      float w = fmaxf(0.f, xx2 - xx1 + 1), h = fmaxf(0.f, yy2 - yy1 + 1);
      float in = w * h;
      float ja = (dets[4 * j + 2] - dets[4 * j] + 1) *
                 (dets[4 * j + 3] - dets[4 * j + 1] + 1);
      if (in / (iarea + ja - in) >= thresh) sup[j] = 1;
    }
  }
  int64_t nk = 0;
  for (int64_t i = 0; i < n; ++i) if (!sup[i]) keep[nk++] = i;
  free(order); free(sup);
  return nk;
}
