// iou_math.h -- rotated-rectangle IoU arithmetic shared by the HIP kernels (device) and the
// host-side self-check build (tests compile this header for the CPU to exercise the exact
// statement sequence the GPU runs).
//
// Behavioural contract: second/core/non_max_suppression/nms_gpu.py:166-403,552-623 of the
// reference (numba.cuda device functions, float32, fastmath=False): corners from
// (xc, yc, size_a, size_b, yaw) with clockwise-positive yaw; candidate vertices = corners of
// either box inside the other (inclusive tests) + edge/edge intersections; vertices ordered by
// a monotone pseudo-angle about their centroid; area by fan triangulation.  Degenerate
// duplicates are NOT removed (identical boxes give 1/3, which is why the reference patches
// such pairs to 1.0 afterwards, nms_gpu.py:706-717 -- done here in the kernel epilogue).
// The reference's 8-point local buffer can be overrun by up to 24 candidates (undefined
// behaviour there); this implementation keeps all 24.
// Compile with -ffp-contract=off: every product/sum is rounded separately, as in the oracle.
#pragma once
#include <math.h>
#ifndef AABR_HD
#ifdef __HIPCC__
#define AABR_HD __host__ __device__ inline
#else
#define AABR_HD static inline
#endif
#endif

namespace aabr_iou {

AABR_HD float tri2(float ax, float ay, float bx, float by, float cx, float cy) {
  return ((ax - cx) * (by - cy) - (ay - cy) * (bx - cx)) / 2.0f;
}

AABR_HD void corners_of(const float *rb, float *cx, float *cy) {
  const float ang = rb[4];
  // double-precision trig narrowed to fp32: identical on host and device (fp32 cosf/sinf differ
  // by an ulp between libm and the device library, which thin 6 m x 0.1 m walls amplify ~60x)
  const float ac = (float)cos((double)ang), as = (float)sin((double)ang);
  const float hx = rb[2] / 2, hy = rb[3] / 2;
  const float px[4] = {-hx, -hx, hx, hx};
  const float py[4] = {-hy, hy, hy, -hy};
  for (int i = 0; i < 4; ++i) {
    cx[i] = ac * px[i] + as * py[i] + rb[0];
    cy[i] = -as * px[i] + ac * py[i] + rb[1];
  }
}

AABR_HD bool inside_quad(float x, float y, const float *qx, const float *qy) {
  const float ab0 = qx[1] - qx[0], ab1 = qy[1] - qy[0];
  const float ad0 = qx[3] - qx[0], ad1 = qy[3] - qy[0];
  const float ap0 = x - qx[0], ap1 = y - qy[0];
  const float abab = ab0 * ab0 + ab1 * ab1, abap = ab0 * ap0 + ab1 * ap1;
  const float adad = ad0 * ad0 + ad1 * ad1, adap = ad0 * ap0 + ad1 * ap1;
  return abab >= abap && abap >= 0 && adad >= adap && adap >= 0;
}

AABR_HD bool edge_cross(const float *px, const float *py, int i, const float *qx, const float *qy, int j,
                        float *ox, float *oy) {
  const float A0 = px[i], A1 = py[i], B0 = px[(i + 1) & 3], B1 = py[(i + 1) & 3];
  const float C0 = qx[j], C1 = qy[j], D0 = qx[(j + 1) & 3], D1 = qy[(j + 1) & 3];
  const float BA0 = B0 - A0, BA1 = B1 - A1;
  const float DA0 = D0 - A0, CA0 = C0 - A0, DA1 = D1 - A1, CA1 = C1 - A1;
  const bool acd = DA1 * CA0 > CA1 * DA0;
  const bool bcd = (D1 - B1) * (C0 - B0) > (C1 - B1) * (D0 - B0);
  if (acd == bcd) return false;
  const bool abc = CA1 * BA0 > BA1 * CA0;
  const bool abd = DA1 * BA0 > BA1 * DA0;
  if (abc == abd) return false;
  const float DC0 = D0 - C0, DC1 = D1 - C1;
  const float ABBA = A0 * B1 - B0 * A1, CDDC = C0 * D1 - D0 * C1;
  const float DH = BA1 * DC0 - BA0 * DC1;
  *ox = (ABBA * DC0 - BA0 * CDDC) / DH;
  *oy = (ABBA * DC1 - BA1 * CDDC) / DH;
  return true;
}

// area of the intersection polygon of two rotated rectangles
AABR_HD float inter_area(const float *r1, const float *r2) {
  // Rectangles whose circumscribed circles are clearly apart share no point: the construction below would
  // collect no vertex and return exactly 0.0f.  Most pairs of a scene end here (0.1 % margin on the
  // squared radius sum; NaNs and negative sizes fail the comparison and take the full path).
  {
    const float dx = r1[0] - r2[0], dy = r1[1] - r2[1];
    const float rr = 0.5f * (sqrtf(r1[2] * r1[2] + r1[3] * r1[3]) + sqrtf(r2[2] * r2[2] + r2[3] * r2[3]));
    if (dx * dx + dy * dy > rr * rr * 1.001f + 1e-12f) return 0.0f;
  }
  float ax[4], ay[4], bx[4], by[4], vx[24], vy[24], key[24];
  corners_of(r1, ax, ay);
  corners_of(r2, bx, by);
  int n = 0;
  for (int i = 0; i < 4; ++i) {
    if (inside_quad(ax[i], ay[i], bx, by)) { vx[n] = ax[i]; vy[n] = ay[i]; ++n; }
    if (inside_quad(bx[i], by[i], ax, ay)) { vx[n] = bx[i]; vy[n] = by[i]; ++n; }
  }
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      float ox, oy;
      if (edge_cross(ax, ay, i, bx, by, j, &ox, &oy)) { vx[n] = ox; vy[n] = oy; ++n; }
    }
  if (n > 0) {
    float mx = 0.0f, my = 0.0f;
    for (int i = 0; i < n; ++i) { mx += vx[i]; my += vy[i]; }
    mx /= n; my /= n;
    for (int i = 0; i < n; ++i) {
      float dx = vx[i] - mx, dy = vy[i] - my;
      const float d = sqrtf(dx * dx + dy * dy);
      dx = dx / d; dy = dy / d;
      if (dy < 0) dx = -2 - dx;
      key[i] = dx;
    }
    for (int i = 1; i < n; ++i) {
      if (key[i - 1] > key[i]) {
        const float t = key[i], tx = vx[i], ty = vy[i];
        int j = i;
        while (j > 0 && key[j - 1] > t) {
          key[j] = key[j - 1]; vx[j] = vx[j - 1]; vy[j] = vy[j - 1];
          --j;
        }
        key[j] = t; vx[j] = tx; vy[j] = ty;
      }
    }
  }
  float s = 0.0f;
  for (int i = 0; i < n - 2; ++i)
    s += fabsf(tri2(vx[0], vy[0], vx[i + 1], vy[i + 1], vx[i + 2], vy[i + 2]));
  return s;
}

// devRotateIoUEval(r1, r2, criterion); the distance-type criteria are evaluated in double
// (numba types float32 ** 2 / ** 0.5 as float64) and narrowed on return.
AABR_HD float rotate_iou(const float *r1, const float *r2, int criterion) {
  const float area1 = r1[2] * r1[3], area2 = r2[2] * r2[3];
  // criteria 4, 5 and 6 are functions of centres and sizes only: the intersection area is not evaluated
  const bool need_ai = !(criterion == 4 || criterion == 5 || criterion == 6);
  const float ai = need_ai ? inter_area(r1, r2) : 0.0f;
  if (criterion == -1) return ai / (area1 + area2 - ai);
  if (criterion == 0) return ai / area1;
  if (criterion == 1) return ai / area2;
  if (criterion == 2) {
    const float mn = r2[2] < r2[3] ? r2[2] : r2[3], mx = r2[2] > r2[3] ? r2[2] : r2[3];
    if (mn / mx < 0.25f) {
      const float e = area1 * 0.5f - ai;
      return ai / (area2 + (e > 0 ? e : 0));
    }
    return ai / (area1 + area2 - ai);
  }
  const double d0 = (double)(float)(r1[0] - r2[0]), d1 = (double)(float)(r1[1] - r2[1]);
  const double dc = sqrt(d0 * d0 + d1 * d1);
  const double pi = 3.14159265358979323846;
  if (criterion == 6) {
    const double m = (double)fabsf(r1[2] - r2[2]) + (double)fabsf(r1[3] - r2[3]) + dc;
    return (float)(1 - m / 0.7);
  }
  if (criterion == 3) {
    const double diag = dc + sqrt((double)r1[2] * r1[2] + (double)r2[0] * r2[0]) * 0.5 +
                        sqrt((double)r2[2] * r2[2] + (double)r2[0] * r2[0]) * 0.5;
    const double iou = ai / (area1 + area2 - ai);
    const double diou = 1 - dc * dc / (diag * diag);
    const double da = atan((double)(r1[2] / r1[3])) - atan((double)(r2[2] / r2[3]));
    const double aiou = 1 - (4 / (pi * pi)) * da * da;
    return (float)(iou * 0 + diou * 0.1 + aiou * 0.1);
  }
  if (criterion == 4) {
    const double l1x = r1[0] + r1[2] * 0.5, l1y = r1[1], w1x = r1[0], w1y = r1[1] + r1[3] * 0.5;
    const double l2x = r2[0] + r2[2] * 0.5, l2y = r2[1], w2x = r2[0], w2y = r2[1] + r2[3] * 0.5;
    const double Dl = sqrt((l1x - l2x) * (l1x - l2x) + (l1y - l2y) * (l1y - l2y));
    const double Dw = sqrt((w1x - w2x) * (w1x - w2x) + (w1y - w2y) * (w1y - w2y));
    return (float)(2 - (Dl + Dw + 1.5 * dc) / 1);
  }
  if (criterion == 5) {
    const float long1 = r1[2] >= r1[3] ? r1[2] : r1[3];
    const float long2 = r2[2] >= r2[3] ? r2[2] : r2[3];
    const double da = atan((double)(r1[2] / r1[3])) - atan((double)(r2[2] / r2[3]));
    const double aiou = (4 / (pi * pi)) * da * da;
    const double dl = (double)(float)(long1 - long2);
    return (float)(1 - (sqrt(dl * dl) + 0 + dc) / 0.5 + 0.2 * aiou);
  }
  return ai;
}

AABR_HD bool same_box(const float *a, const float *b) {
  for (int d = 0; d < 5; ++d)
    if (!(fabsf(a[d] - b[d]) < 1e-6f)) return false;
  return true;
}

// ---- exact polygon IoU (the NMS decision) -----------------------------------------------------------------------------
// The suppression loop of the reference's rotated NMS (spconv 1.x rotate_non_max_suppression_cpu, called from
// second/core/non_max_suppression/nms_cpu.py:43) takes the matrix above only as a `> 0` pre-filter and decides
// `overlap >= thresh` on a boost::geometry polygon intersection / union of the two rectangles.  rotate_iou() is not
// that value where its vertex collection / angular sort breaks down (two thin, nearly parallel walls crossing at
// 0.035 rad: 0.162 against an exact 0.569 -- tests/test_oracle_golden.py::test_nms_decision_iou_disagreements), so the
// decision is made on a Sutherland-Hodgman clip in double: corners of center_to_corner_box2d (box_np_ops.py:374-394;
// corner order x0y0, x0y1, x1y1, x1y0, rotation x' = x cos + y sin, y' = -x sin + y cos), each half-plane of the
// second rectangle applied in turn, area by the shoelace formula.  r = (xc, yc, size_x, size_y, yaw).
AABR_HD void clip_corners(const float *r, double *cx, double *cy) {
  const double a = (double)r[4], s = sin(a), c = cos(a);
  const double hx = 0.5 * (double)r[2], hy = 0.5 * (double)r[3];
  const double px[4] = {-hx, -hx, hx, hx}, py[4] = {-hy, hy, hy, -hy};
  for (int k = 0; k < 4; ++k) {
    cx[k] = px[k] * c + py[k] * s + (double)r[0];
    cy[k] = -px[k] * s + py[k] * c + (double)r[1];
  }
}

AABR_HD double clip_iou_exact(const float *r1, const float *r2) {
  double ax[4], ay[4], bx[4], by[4];
  clip_corners(r1, ax, ay);
  clip_corners(r2, bx, by);
  double px[12], py[12], qx[12], qy[12];
  int n = 4;
  for (int i = 0; i < 4; ++i) { px[i] = ax[i]; py[i] = ay[i]; }
  // the corner order above is clockwise in the (x, y) plane for positive sizes (inside = right of each edge); the
  // sign of the clip rectangle's signed area covers the other orientation
  double sb = 0.0;
  for (int i = 0; i < 4; ++i) sb += bx[i] * by[(i + 1) & 3] - bx[(i + 1) & 3] * by[i];
  const double o = sb <= 0.0 ? 1.0 : -1.0;
  for (int e = 0; e < 4 && n > 0; ++e) {
    const double ex = o * (bx[(e + 1) & 3] - bx[e]), ey = o * (by[(e + 1) & 3] - by[e]);
    int m = 0;
    for (int i = 0; i < n; ++i) {
      const int j = i + 1 < n ? i + 1 : 0;
      const double di = ey * (px[i] - bx[e]) - ex * (py[i] - by[e]);
      const double dj = ey * (px[j] - bx[e]) - ex * (py[j] - by[e]);
      if (di >= 0.0) { qx[m] = px[i]; qy[m] = py[i]; ++m; }
      if ((di > 0.0 && dj < 0.0) || (di < 0.0 && dj > 0.0)) {
        const double t = di / (di - dj);
        qx[m] = px[i] + t * (px[j] - px[i]);
        qy[m] = py[i] + t * (py[j] - py[i]);
        ++m;
      }
    }
    n = m;
    for (int i = 0; i < n; ++i) { px[i] = qx[i]; py[i] = qy[i]; }
  }
  if (n < 3) return 0.0;
  double inter = 0.0;
  for (int i = 0; i < n; ++i) {
    const int j = i + 1 < n ? i + 1 : 0;
    inter += px[i] * py[j] - px[j] * py[i];
  }
  inter = 0.5 * fabs(inter);
  const double uni = fabs((double)r1[2] * (double)r1[3]) + fabs((double)r2[2] * (double)r2[3]) - inter;
  return uni > 0.0 ? inter / uni : 0.0;
}

// pair value of rotate_iou_gpu_eval: iou[n][k] for box n, query k
AABR_HD float iou_eval_entry(const float *box_n, const float *query_k, int criterion) {
  return same_box(box_n, query_k) ? 1.0f : rotate_iou(query_k, box_n, criterion);
}

} // namespace aabr_iou
