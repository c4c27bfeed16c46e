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
// decision is made on an exact clip in double: corners of center_to_corner_box2d (box_np_ops.py:374-394; corner order
// x0y0, x0y1, x1y1, x1y0, rotation x' = x cos + y sin, y' = -x sin + y cos).  r = (xc, yc, size_x, size_y, yaw).
AABR_HD void clip_corners(const float *r, double *cx, double *cy) {
  const double a = (double)r[4], s = sin(a), c = cos(a);
  const double hx = 0.5 * (double)r[2], hy = 0.5 * (double)r[3];
  const double px[4] = {-hx, -hx, hx, hx}, py[4] = {-hy, hy, hy, -hy};
  for (int k = 0; k < 4; ++k) {
    cx[k] = px[k] * c + py[k] * s + (double)r[0];
    cy[k] = -px[k] * s + py[k] * c + (double)r[1];
  }
}

// Intersection area by boundary integration instead of polygon construction (no vertex lists, no dynamic indexing:
// everything stays in registers on the device): the boundary of A n B is made of the pieces of A's edges inside B and
// the pieces of B's edges inside A; each edge is cut down to its piece inside the other rectangle by the four
// half-planes (parametric clip), and the shoelace sum over the directed pieces is twice the area.  A piece of
// boundary the two rectangles SHARE (collinear edges running the same way) is counted once: A's edges are clipped
// against B closed (d >= 0), the copy on B's side is left out.  Agrees with a Sutherland-Hodgman
// clip (oracle/clip_oracle.c, tests/test_oracle_golden.py) to 1e-12.
AABR_HD double clip_edges_inside(const double *ax, const double *ay, const double *bx, const double *by, bool drop_collinear) {
  double sum = 0.0;
  for (int i = 0; i < 4; ++i) {
    const double x0 = ax[i], y0 = ay[i], x1 = ax[(i + 1) & 3], y1 = ay[(i + 1) & 3];
    double t0 = 0.0, t1 = 1.0;
    bool out = false;
    for (int e = 0; e < 4; ++e) {
      const double ex = bx[(e + 1) & 3] - bx[e], ey = by[(e + 1) & 3] - by[e];
      const double d0 = ey * (x0 - bx[e]) - ex * (y0 - by[e]);      // >= 0: on the inner side of B's edge e
      const double d1 = ey * (x1 - bx[e]) - ex * (y1 - by[e]);
      if (d0 < 0.0 && d1 < 0.0) out = true;
      // on the line of B's edge e: a shared piece of boundary when the two edges run the same way (interiors on the
      // same side: counted with the other rectangle's copy), two pieces that cancel when they run against each other
      // (the rectangles touch from opposite sides)
      if (drop_collinear && d0 == 0.0 && d1 == 0.0 && (x1 - x0) * ex + (y1 - y0) * ey > 0.0) out = true;
      if (d0 < 0.0 && d1 >= 0.0) { const double t = d0 / (d0 - d1); t0 = t > t0 ? t : t0; }
      if (d1 < 0.0 && d0 >= 0.0) { const double t = d0 / (d0 - d1); t1 = t < t1 ? t : t1; }
    }
    if (!out && t0 < t1) {
      const double px = x0 + t0 * (x1 - x0), py = y0 + t0 * (y1 - y0);
      const double qx = x0 + t1 * (x1 - x0), qy = y0 + t1 * (y1 - y0);
      sum += px * qy - qx * py;
    }
  }
  return sum;
}

// the same on corners computed ahead (clip_corners): the NMS mask kernel computes a box's corners once per workgroup
AABR_HD double clip_iou_corners(const double *ax0, const double *ay0, const double *bx0, const double *by0, double area_a,
                                double area_b) {
  double ax[4], ay[4], bx[4], by[4];
  for (int i = 0; i < 4; ++i) { ax[i] = ax0[i]; ay[i] = ay0[i]; bx[i] = bx0[i]; by[i] = by0[i]; }
  // the corner order is clockwise for positive sizes (inner side = right of each edge); a rectangle given with a
  // negative size is turned round so that both run the same way
  double sa = 0.0, sb = 0.0;
  for (int i = 0; i < 4; ++i) {
    sa += ax[i] * ay[(i + 1) & 3] - ax[(i + 1) & 3] * ay[i];
    sb += bx[i] * by[(i + 1) & 3] - bx[(i + 1) & 3] * by[i];
  }
  if (sa > 0.0) { double t = ax[1]; ax[1] = ax[3]; ax[3] = t; t = ay[1]; ay[1] = ay[3]; ay[3] = t; }
  if (sb > 0.0) { double t = bx[1]; bx[1] = bx[3]; bx[3] = t; t = by[1]; by[1] = by[3]; by[3] = t; }
  // shift to A's first corner: the shoelace terms lose no digits to the scene's offset
  const double ox = ax[0], oy = ay[0];
  for (int i = 0; i < 4; ++i) { ax[i] -= ox; ay[i] -= oy; bx[i] -= ox; by[i] -= oy; }
  const double twice = clip_edges_inside(ax, ay, bx, by, false) + clip_edges_inside(bx, by, ax, ay, true);
  const double inter = 0.5 * fabs(twice);
  const double uni = area_a + area_b - inter;
  return uni > 0.0 ? inter / uni : 0.0;
}

AABR_HD double clip_iou_exact(const float *r1, const float *r2) {
  double ax[4], ay[4], bx[4], by[4];
  clip_corners(r1, ax, ay);
  clip_corners(r2, bx, by);
  return clip_iou_corners(ax, ay, bx, by, fabs((double)r1[2] * (double)r1[3]), fabs((double)r2[2] * (double)r2[3]));
}

// pair value of rotate_iou_gpu_eval: iou[n][k] for box n, query k
AABR_HD float iou_eval_entry(const float *box_n, const float *query_k, int criterion) {
  return same_box(box_n, query_k) ? 1.0f : rotate_iou(query_k, box_n, criterion);
}

} // namespace aabr_iou
