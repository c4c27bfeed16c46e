/* clip_oracle.c -- exact (fp64) intersection-over-union of rotated rectangles by Sutherland-Hodgman clipping.
 *
 * TEST INFRASTRUCTURE ONLY (tests/, tools/): the product never links this.
 *
 * Why it exists: the suppression step of the reference's rotated NMS lives in un-vendored spconv 1.x
 * (spconv.utils.rotate_non_max_suppression_cpu, called from second/core/non_max_suppression/nms_cpu.py:32-44 with
 * the corners of box_np_ops.center_to_corner_box2d, second/core/box_np_ops.py:374-394).  Its published loop decides
 * `overlap >= thresh` on a boost::geometry polygon intersection / union of the two rectangles -- an EXACT polygon
 * IoU -- whereas the pre-filter matrix it is handed (boxes_iou_3d -> rotate_iou_gpu_eval, nms_gpu.py:552-703) is the
 * numba kernel's vertex-collection IoU, which is known to be wrong where vertices coincide (identical boxes give
 * 1/3 before check_same_boxes, nms_gpu.py:706-717).  This file is an independent statement of the exact value --
 * it shares no code with iou_oracle.c (no vertex collection, no angular sort) -- so that tests can COUNT the pairs
 * on which the two disagree about `>= thresh` (tests/test_oracle_golden.py, tests/test_gpu_parity.py).
 *
 * Geometry: a box row is (xc, yc, z, size_x, size_y, size_z, yaw) as rotate_nms_3d_cc slices it (centers = [:, :2],
 * dims = [:, 3:5], angle = [:, 6]); corners_nd's 2-D corner order (x0y0, x0y1, x1y1, x1y0) about the centre, rotated
 * by rotation_2d (box_np_ops.py:313-326: x' = x cos + y sin, y' = -x sin + y cos -- clockwise for a positive angle)
 * and shifted to the centre.  Everything from the fp32 box parameters on is evaluated in double. */
#include <math.h>
#include <stdint.h>
#include <omp.h>

static void corners_of(const float *b, double c[4][2]) {
  const double dx = b[3], dy = b[4], a = b[6];
  const double s = sin(a), co = cos(a);
  const double nx[4] = {-0.5, -0.5, 0.5, 0.5}, ny[4] = {-0.5, 0.5, 0.5, -0.5};
  for (int k = 0; k < 4; ++k) {
    const double x = nx[k] * dx, y = ny[k] * dy;
    c[k][0] = x * co + y * s + b[0];
    c[k][1] = -x * s + y * co + b[1];
  }
}

static double shoelace(double p[][2], int n) {
  double a = 0.0;
  for (int i = 0; i < n; ++i) {
    const int j = (i + 1) % n;
    a += p[i][0] * p[j][1] - p[j][0] * p[i][1];
  }
  return 0.5 * a;
}

/* area of (subject polygon) clipped by the convex polygon `clip` (Sutherland-Hodgman, one half-plane per edge) */
static double clipped_area(double subj[4][2], double clip[4][2]) {
  double cur[16][2], nxt[16][2];
  int n = 4;
  for (int i = 0; i < 4; ++i) { cur[i][0] = subj[i][0]; cur[i][1] = subj[i][1]; }
  const double orient = shoelace(clip, 4) >= 0.0 ? 1.0 : -1.0;   /* inside = left of a counter-clockwise edge */
  for (int e = 0; e < 4 && n > 0; ++e) {
    const double ax = clip[e][0], ay = clip[e][1], bx = clip[(e + 1) & 3][0], by = clip[(e + 1) & 3][1];
    const double ex = bx - ax, ey = by - ay;
    int m = 0;
    for (int i = 0; i < n; ++i) {
      const int j = (i + 1) % n;
      const double di = orient * (ex * (cur[i][1] - ay) - ey * (cur[i][0] - ax));
      const double dj = orient * (ex * (cur[j][1] - ay) - ey * (cur[j][0] - ax));
      if (di >= 0.0) { nxt[m][0] = cur[i][0]; nxt[m][1] = cur[i][1]; ++m; }
      if ((di > 0.0 && dj < 0.0) || (di < 0.0 && dj > 0.0)) {
        const double t = di / (di - dj);
        nxt[m][0] = cur[i][0] + t * (cur[j][0] - cur[i][0]);
        nxt[m][1] = cur[i][1] + t * (cur[j][1] - cur[i][1]);
        ++m;
      }
    }
    n = m;
    for (int i = 0; i < n; ++i) { cur[i][0] = nxt[i][0]; cur[i][1] = nxt[i][1]; }
  }
  return n >= 3 ? fabs(shoelace(cur, n)) : 0.0;
}

/* exact 2-D IoU of every (i, j): out[n * n] doubles; rows of boxes7 as described above */
void oracle_clip_iou_matrix(const float *boxes7, int64_t n, double *out) {
#pragma omp parallel for schedule(dynamic, 8)
  for (int64_t i = 0; i < n; ++i) {
    double ci[4][2];
    corners_of(boxes7 + 7 * i, ci);
    const double ai = fabs(shoelace(ci, 4));
    for (int64_t j = 0; j < n; ++j) {
      double cj[4][2];
      corners_of(boxes7 + 7 * j, cj);
      const double aj = fabs(shoelace(cj, 4));
      /* quick reject on the circumscribed circles */
      const double dx = (double)boxes7[7 * i] - boxes7[7 * j], dy = (double)boxes7[7 * i + 1] - boxes7[7 * j + 1];
      const double ri = 0.5 * hypot(boxes7[7 * i + 3], boxes7[7 * i + 4]);
      const double rj = 0.5 * hypot(boxes7[7 * j + 3], boxes7[7 * j + 4]);
      double v = 0.0;
      if (dx * dx + dy * dy <= (ri + rj) * (ri + rj)) {
        const double inter = clipped_area(ci, cj);
        const double uni = ai + aj - inter;
        v = uni > 0.0 ? inter / uni : 0.0;
      }
      out[i * n + j] = v;
    }
  }
}
