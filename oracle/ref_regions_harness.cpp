// ref_regions_harness.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Compiles the reference's OWN region geometry
//   /root/reference/SparseConvNet/sparseconvnet/SCN/Metadata/RectangularRegions.h
//   /root/reference/SparseConvNet/sparseconvnet/SCN/Metadata/32bits.h
// from where they lie (nothing is copied) and exposes it through a C ABI so
// tests can pin oracle/scn_oracle.c's restatement of the same arithmetic.
// Only real headers are used (ATen from the installed torch supplies at::Tensor
// which 32bits.h mentions); no stand-in for any missing header is written.
// The rest of the SparseConvNet C++ path (Metadata.h and everything that
// includes it) needs google sparsehash, which this image lacks: unbuildable here.
// Output goes to oracle/_ref/ (git-ignored; travels to the GPU box via gpurun).
#include <ATen/ATen.h>
#include <cstdint>
#include <algorithm>
#include "Metadata/32bits.h"
#include "Metadata/RectangularRegions.h"

extern "C" {

// InputRegionCalculator + RectangularRegion::offset
int ref_input_region_offset(const long *out_pt, const long *in_pt, long *size,
                            long *stride, long *lb_ub /*6*/) {
  Point<3> o = {(Int)out_pt[0], (Int)out_pt[1], (Int)out_pt[2]};
  Point<3> p = {(Int)in_pt[0], (Int)in_pt[1], (Int)in_pt[2]};
  auto r = InputRegionCalculator<3>(o, size, stride);
  for (int i = 0; i < 3; ++i) { lb_ub[i] = r.lb[i]; lb_ub[3 + i] = r.ub[i]; }
  return r.offset(p);
}

// OutputRegionCalculator, then iterate it in RectangularRegionIterator order
long ref_output_region_points(const long *in_pt, long *size, long *stride,
                              long *out_spatial, long *pts, long cap) {
  Point<3> p = {(Int)in_pt[0], (Int)in_pt[1], (Int)in_pt[2]};
  auto r = OutputRegionCalculator<3>(p, size, stride, out_spatial);
  long n = 0;
  for (auto j : r) {
    if (n >= cap) return -1;
    pts[3 * n] = j[0]; pts[3 * n + 1] = j[1]; pts[3 * n + 2] = j[2];
    ++n;
  }
  return n;
}

// iterate an arbitrary region [lb, ub] in the reference's order and report
// each point's offset()
long ref_region_points(const long *lb, const long *ub, long *pts, int *offs,
                       long cap) {
  Point<3> l = {(Int)lb[0], (Int)lb[1], (Int)lb[2]};
  Point<3> u = {(Int)ub[0], (Int)ub[1], (Int)ub[2]};
  RectangularRegion<3> r(l, u);
  long n = 0;
  for (auto j : r) {
    if (n >= cap) return -1;
    pts[3 * n] = j[0]; pts[3 * n + 1] = j[1]; pts[3 * n + 2] = j[2];
    offs[n] = r.offset(j);
    ++n;
  }
  return n;
}

// IntArrayHash<3> (32bits.h:57-66) -- exported for completeness
unsigned long ref_point_hash(const long *pt) {
  Point<3> p = {(Int)pt[0], (Int)pt[1], (Int)pt[2]};
  return IntArrayHash<3>()(p);
}
}
