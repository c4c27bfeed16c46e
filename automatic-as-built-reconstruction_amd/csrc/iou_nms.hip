// iou_nms.hip -- rotated-box IoU matrix and greedy rotated / axis-aligned NMS (gfx950).
//
// Replaces the numba.cuda kernels rotate_iou_kernel_eval / rotate_nms_kernel
// (second/core/non_max_suppression/nms_gpu.py:406-439,626-664), the host post-processing
// (nms_postprocess :109-126, check_same_boxes :706-717) and the third-party CPU suppression
// loop the reference's 3-D path ends in (nms_cpu.py:32-44 -> spconv rotate_non_max_suppression_cpu).
// VALU / latency bound (56 KB of boxes, ~2 M branchy pair evaluations), not HBM bound.
//   mask kernel : one PAIR per lane, one wave per (row, 64-column block) of the upper triangle; the 64
//                 verdicts become the suppression word with one ballot.
//   scan kernel : one workgroup walks the 64-row blocks; the in-block dependency chain is
//                 resolved on the diagonal 64x64 bit block by one wave in registers
//                 (wavefront-level), then the kept rows' words are OR-reduced in parallel.
// Everything stays on the device; the keep list never round-trips through the host.
#include "common.h"
#include "iou_math.h"

namespace aabr {
using namespace aabr_iou;

// one pair per lane: a wave takes one box row and 64 consecutive queries (coalesced row of the output);
// four rows per workgroup.  (A thread that walks 64 pairs in a loop -- the usual CUDA shape of this kernel
// -- is a 64-long serial chain of a ~5 us, branchy, scratch-heavy evaluation: 300 us whatever the size.)
__global__ __launch_bounds__(256) void k_rotate_iou_eval(const float *__restrict__ boxes, int64_t N,
                                                         const float *__restrict__ query, int64_t K,
                                                         int criterion, float *__restrict__ iou) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t k = (int64_t)blockIdx.y * 64 + lane;
  const int64_t n = (int64_t)blockIdx.x * 4 + wave;
  if (n >= N || k >= K) return;
  float b[5], q[5];
#pragma unroll
  for (int d = 0; d < 5; ++d) { b[d] = boxes[n * 5 + d]; q[d] = query[k * 5 + d]; }
  iou[n * K + k] = iou_eval_entry(b, q, criterion);
}

// [.,7] yx_zb -> 5-parameter 2-D box + z interval, with the reference's thickness clamps
// (rotate_nms_3d_torch.py:59-66: columns [0,1,3,4,6], clamp col 3 and col 5)
__global__ __launch_bounds__(256) void k_box7_to_2d(const float *__restrict__ b7, int64_t n, float minY,
                                                    float minZ, float *__restrict__ b5,
                                                    float *__restrict__ zz) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float *b = b7 + 7 * i;
  float th = b[3] < minY ? minY : b[3];
  float h = b[5] < minZ ? minZ : b[5];
  b5[5 * i] = b[0]; b5[5 * i + 1] = b[1]; b5[5 * i + 2] = th; b5[5 * i + 3] = b[4]; b5[5 * i + 4] = b[6];
  if (zz) { zz[2 * i] = b[2]; zz[2 * i + 1] = b[2] + h; }
}

// iou_one_dim (rotate_nms_3d_torch.py:7-21) folded into the matrix
__global__ __launch_bounds__(256) void k_scale_by_z(float *__restrict__ iou, int64_t M, int64_t K,
                                                    const float *__restrict__ tz,
                                                    const float *__restrict__ az) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= M * K) return;
  int64_t i = idx / K, j = idx - i * K;
  float t0 = tz[2 * i], t1 = tz[2 * i + 1], a0 = az[2 * j], a1 = az[2 * j + 1];
  float overlap = fminf(a1, t1) - fmaxf(a0, t0);
  float common = fmaxf(a1, t1) - fminf(a0, t0);
  iou[idx] = iou[idx] * (overlap / common);
}

// RPN glue: anchors from sparse locations + BoxCoder3D.decode_centroid_box for the selected
// (top-k) anchors, fused (reference: modeling/rpn/anchor_generator_sparse3d.py:88-104,
// modeling/box_coder_3d.py:53-80, second/pytorch/core/box_torch_ops.py:118-154 with
// smooth_dim=True, utils3d/geometric_torch.py:4-10).  Flat anchor index t = site * A + yaw.
struct RpnDecodeParams {
  float inv_scale_num;      // voxel_scale
  float stride[3];
  float weights[7];
  float clip;               // bbox_xform_clip
};

__global__ __launch_bounds__(256) void k_rpn_decode(const int32_t *__restrict__ site_coords, int64_t site0,
                                                    const int64_t *__restrict__ sel, int64_t k,
                                                    const float *__restrict__ regression, int64_t reg0,
                                                    const float *__restrict__ base_anchors, int A,
                                                    RpnDecodeParams p, float *__restrict__ boxes) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= k) return;
  const int64_t t = sel[i];
  const int64_t site = site0 + t / A;
  const int a = (int)(t % A);
  const float *ba = base_anchors + 7 * a;
  float an[7];
#pragma unroll
  for (int d = 0; d < 3; ++d)  // (location.float() + 0) / voxel_scale * stride  + base
    an[d] = (float)site_coords[4 * site + d] / p.inv_scale_num * p.stride[d] + ba[d];
#pragma unroll
  for (int d = 3; d < 7; ++d) an[d] = 0.0f + ba[d];
  const float *r = regression + 7 * (reg0 + t);
  float e[7];
#pragma unroll
  for (int d = 0; d < 7; ++d) e[d] = r[d] / p.weights[d];
#pragma unroll
  for (int d = 3; d < 6; ++d) e[d] = e[d] > p.clip ? p.clip : e[d];
  // second_box_decode: anchors split as (xa, ya, za, wa, la, ha, ra)
  const float diagonal = sqrtf(an[4] * an[4] + an[3] * an[3]);
  float o[7];
  o[0] = e[0] * diagonal + an[0];
  o[1] = e[1] * diagonal + an[1];
  o[2] = e[2] * an[5] + an[2];
  o[3] = (e[3] + 1) * an[3];   // wg = (wt + 1) * wa
  o[4] = (e[4] + 1) * an[4];   // lg = (lt + 1) * la
  o[5] = (e[5] + 1) * an[5];
  float rg = e[6] + an[6];
  const float period = 3.14159265358979323846f;
  rg = rg - floorf(rg / period + 0.5f) * period; // limit_period(., 0.5, pi)
  o[6] = rg;
#pragma unroll
  for (int d = 0; d < 7; ++d) boxes[7 * i + d] = o[d];
}

// Cross-scale form (the shape RPNPostProcessor actually runs in: cat_scales_obj_reg regroups the scales
// example-major, rpn_sparse3d.py:19-77, then ONE top-k + decode + NMS per example, rpn/inference_3d.py:95-149).
// `selected[i]` indexes the example's concatenated anchor list [map][site][yaw]; the segment table maps it
// back to (map, site row, yaw) so neither the anchors nor the concatenated regression are ever materialised.
// Also fused: objectness sigmoid of the selected logits and the boxlist_nms_3d thickness clamps
// (structures/boxlist_ops_3d.py:42-44) into a second, NMS-only copy of the boxes.
constexpr int kMaxRpnMaps = 8;
struct RpnMapsParams {
  const int32_t *coords[kMaxRpnMaps];   // [V_m,4] site lists
  const float *logits[kMaxRpnMaps];     // [V_m*A]
  const float *regression[kMaxRpnMaps]; // [V_m*A,7]
  int32_t seg_begin[kMaxRpnMaps + 1];   // first local anchor index of map m in this example's list
  int32_t site_begin[kMaxRpnMaps];      // first site row of this example in map m
  float stride[kMaxRpnMaps][3];
  int n_maps, A;
  float voxel_scale, clip, nms_min_yx, nms_min_z;
  float weights[7];
};

__global__ __launch_bounds__(256) void k_rpn_decode_maps(RpnMapsParams p, const int64_t *__restrict__ sel, int64_t k,
                                                         const float *__restrict__ base_anchors /*[n_maps*A,7]*/,
                                                         float *__restrict__ boxes, float *__restrict__ nms_boxes,
                                                         float *__restrict__ scores) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= k) return;
  const int32_t t = (int32_t)sel[i];
  int m = 0;
#pragma unroll
  for (int q = 1; q < kMaxRpnMaps; ++q)
    if (q < p.n_maps && t >= p.seg_begin[q]) m = q;
  const int32_t r = t - p.seg_begin[m];
  const int64_t site = (int64_t)p.site_begin[m] + r / p.A;
  const int a = r % p.A;
  const int32_t *sc = p.coords[m] + 4 * site;
  const float *ba = base_anchors + 7 * ((int64_t)m * p.A + a);
  float an[7];
#pragma unroll
  for (int d = 0; d < 3; ++d) an[d] = (float)sc[d] / p.voxel_scale * p.stride[m][d] + ba[d];
#pragma unroll
  for (int d = 3; d < 7; ++d) an[d] = 0.0f + ba[d];
  const int64_t row = site * p.A + a;
  const float *rg7 = p.regression[m] + 7 * row;
  float e[7];
#pragma unroll
  for (int d = 0; d < 7; ++d) e[d] = rg7[d] / p.weights[d];
#pragma unroll
  for (int d = 3; d < 6; ++d) e[d] = e[d] > p.clip ? p.clip : e[d];
  const float diagonal = sqrtf(an[4] * an[4] + an[3] * an[3]);
  float o[7];
  o[0] = e[0] * diagonal + an[0];
  o[1] = e[1] * diagonal + an[1];
  o[2] = e[2] * an[5] + an[2];
  o[3] = (e[3] + 1) * an[3];
  o[4] = (e[4] + 1) * an[4];
  o[5] = (e[5] + 1) * an[5];
  float rg = e[6] + an[6];
  const float period = 3.14159265358979323846f;
  o[6] = rg - floorf(rg / period + 0.5f) * period;
#pragma unroll
  for (int d = 0; d < 7; ++d) boxes[7 * i + d] = o[d];
  if (nms_boxes) {
    o[3] = o[3] < p.nms_min_yx ? p.nms_min_yx : o[3];
    o[4] = o[4] < p.nms_min_yx ? p.nms_min_yx : o[4];
    o[5] = o[5] < p.nms_min_z ? p.nms_min_z : o[5];
#pragma unroll
    for (int d = 0; d < 7; ++d) nms_boxes[7 * i + d] = o[d];
  }
  if (scores) scores[i] = 1.0f / (1.0f + expf(-p.logits[m][row]));
}

// KIND 0: rotated 3-D boxes [n,7] (2-D IoU, optionally times z-IoU); KIND 1: axis-aligned [n,4].
// One pair per lane: a wave owns one row i and one block of 64 columns; the 64 verdicts become the
// suppression word with a single ballot (no loop over columns, no shared-memory staging).  Four rows per
// workgroup.  Blocks below the diagonal are never launched with work: a lower-scored box suppresses nothing.
template <int KIND>
__global__ __launch_bounds__(256) void k_nms_mask(const float *__restrict__ boxes, int64_t n, float thresh,
                                                  int only_xy, int colblocks,
                                                  unsigned long long *__restrict__ mask) {
  constexpr int W = KIND == 0 ? 7 : 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cb = blockIdx.y;
  const int64_t i = (int64_t)blockIdx.x * 4 + wave;
  if (i >= n) return;                                  // wave-uniform
  const int64_t j = (int64_t)cb * 64 + lane;
  bool hit = false;
  if (cb >= (int)(i >> 6) && j < n && j > i) {         // strictly above the diagonal
    const float *b = boxes + i * W, *c = boxes + j * W;
    if (KIND == 0) {
      const float bi[5] = {b[0], b[1], b[3], b[4], b[6]};
      const float bj[5] = {c[0], c[1], c[3], c[4], c[6]};
      // matrix entry [i][j] of boxes_iou_3d(dets, dets): box = i, query = j
      float v = iou_eval_entry(bi, bj, -1);
      if (!only_xy) {
        const float z0 = b[2], z1 = b[2] + b[5], a0 = c[2], a1 = c[2] + c[5];
        v = v * ((fminf(a1, z1) - fmaxf(a0, z0)) / (fmaxf(a1, z1) - fminf(a0, z0)));
      }
      // spconv 1.x rotate_non_max_suppression_cpu (behind nms_cpu.py:43): the matrix entry is the `> 0` pre-filter,
      // the decision is an exact polygon IoU of the two rectangles (2-D, whatever only_xy says) >= thresh
      hit = v > 0.0f && clip_iou_exact(bi, bj) >= (double)thresh;
    } else {
      const float xx1 = fmaxf(b[0], c[0]), yy1 = fmaxf(b[1], c[1]);
      const float xx2 = fminf(b[2], c[2]), yy2 = fminf(b[3], c[3]);
      const float w = fmaxf(0.0f, xx2 - xx1 + 1), h = fmaxf(0.0f, yy2 - yy1 + 1);
      const float inter = w * h;
      const float iarea = (b[2] - b[0] + 1) * (b[3] - b[1] + 1), jarea = (c[2] - c[0] + 1) * (c[3] - c[1] + 1);
      hit = inter / (iarea + jarea - inter) >= thresh;
    }
  }
  const unsigned long long bits = __ballot(hit);
  if (lane == 0) mask[i * colblocks + cb] = bits;
}

// Rotated boxes, round 4.  The decision of the reference's loop is `pre-filter matrix > 0 and exact polygon IoU >=
// thresh` (spconv 1.x behind nms_cpu.py:43); evaluated in that order every overlapping pair pays both the numba-style
// IoU (vertex collection + sort, ~4x the cost of the clip) and the clip.  Here: a workgroup owns 16 rows x 64 columns;
// the corners of its 80 boxes are computed ONCE (fp64 sin / cos per box instead of per pair) and shared through LDS;
// per pair: circumscribed circles apart -> no hit; else the exact clip (registers only); only a pair that WOULD suppress
// (clip >= thresh: a few per cent) evaluates the pre-filter value.  Same verdicts as the order above.
constexpr int kNmsRows = 16;
__global__ __launch_bounds__(256) void k_nms_mask_rot(const float *__restrict__ boxes, int64_t n, float thresh,
                                                      int only_xy, int colblocks,
                                                      unsigned long long *__restrict__ mask) {
  __shared__ double s_cx[64 + kNmsRows][4], s_cy[64 + kNmsRows][4];
  __shared__ float s_b[64 + kNmsRows][8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cb = blockIdx.y;
  const int64_t i0 = (int64_t)blockIdx.x * kNmsRows;
  if (cb < (int)(i0 >> 6)) {                           // wholly below the diagonal: a lower-scored box suppresses nothing
    if (threadIdx.x < kNmsRows && i0 + threadIdx.x < n) mask[(i0 + threadIdx.x) * colblocks + cb] = 0ull;
    return;
  }
  if (threadIdx.x < 64 + kNmsRows) {                   // slots 0..63: the column boxes; 64..79: the row boxes
    const int64_t q = threadIdx.x < 64 ? (int64_t)cb * 64 + threadIdx.x : i0 + (threadIdx.x - 64);
    float b7[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (q < n)
#pragma unroll
      for (int d = 0; d < 7; ++d) b7[d] = boxes[q * 7 + d];
    const float r5[5] = {b7[0], b7[1], b7[3], b7[4], b7[6]};
    double cx[4], cy[4];
    clip_corners(r5, cx, cy);
#pragma unroll
    for (int k = 0; k < 4; ++k) { s_cx[threadIdx.x][k] = cx[k]; s_cy[threadIdx.x][k] = cy[k]; }
#pragma unroll
    for (int d = 0; d < 7; ++d) s_b[threadIdx.x][d] = b7[d];
  }
  __syncthreads();
  const int64_t j = (int64_t)cb * 64 + lane;
  float c[7];
#pragma unroll
  for (int d = 0; d < 7; ++d) c[d] = s_b[lane][d];
  double jx[4], jy[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) { jx[k] = s_cx[lane][k]; jy[k] = s_cy[lane][k]; }
  const double area_j = fabs((double)c[3] * (double)c[4]);
  const float rj = 0.5f * sqrtf(c[3] * c[3] + c[4] * c[4]);
  for (int t = 0; t < kNmsRows / 4; ++t) {
    const int rs = wave * (kNmsRows / 4) + t;
    const int64_t i = i0 + rs;
    if (i >= n) break;                                 // wave-uniform
    bool hit = false;
    if (cb >= (int)(i >> 6) && j < n && j > i) {
      const float *b = s_b[64 + rs];
      const float dx = b[0] - c[0], dy = b[1] - c[1];
      const float ri = 0.5f * sqrtf(b[3] * b[3] + b[4] * b[4]);
      const float rr = (ri + rj) * 1.0001f + 1e-6f;    // conservative: never rejects a pair the clip would count
      if (dx * dx + dy * dy <= rr * rr) {
        double ix[4], iy[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { ix[k] = s_cx[64 + rs][k]; iy[k] = s_cy[64 + rs][k]; }
        const double e = clip_iou_corners(ix, iy, jx, jy, fabs((double)b[3] * (double)b[4]), area_j);
        if (e >= (double)thresh) {
          const float bi[5] = {b[0], b[1], b[3], b[4], b[6]};
          const float bj[5] = {c[0], c[1], c[3], c[4], c[6]};
          float v = iou_eval_entry(bi, bj, -1);        // matrix entry [i][j] of boxes_iou_3d(dets, dets)
          if (!only_xy) {
            const float z0 = b[2], z1 = b[2] + b[5], a0 = c[2], a1 = c[2] + c[5];
            v = v * ((fminf(a1, z1) - fmaxf(a0, z0)) / (fmaxf(a1, z1) - fminf(a0, z0)));
          }
          hit = v > 0.0f;
        }
      }
    }
    const unsigned long long bits = __ballot(hit);
    if (lane == 0) mask[i * colblocks + cb] = bits;
  }
}

// Greedy scan.  remv words live in LDS (colblocks <= 8192 -> n <= 524288).
__global__ __launch_bounds__(256) void k_nms_scan(const unsigned long long *__restrict__ mask, int64_t n,
                                                  int colblocks, int64_t post_max,
                                                  int64_t *__restrict__ keep, int32_t *__restrict__ meta) {
  extern __shared__ unsigned long long remv[];
  __shared__ unsigned long long s_kept;
  __shared__ int s_nk;
  const int tid = threadIdx.x;
  for (int w = tid; w < colblocks; w += blockDim.x) remv[w] = 0ull;
  if (tid == 0) s_nk = 0;
  __syncthreads();
  const int wl = tid & 31, slice = tid >> 5;
  for (int rb = 0; rb < colblocks; ++rb) {
    const int64_t r0 = (int64_t)rb * 64;
    const int rows = (int)((n - r0) < 64 ? (n - r0) : 64);
    // speculative fetch, before the block's kept set is known: this thread's 8 rows (b = slice mod 8) of the
    // first 32 words to the right of the diagonal.  The loads overlap the serial chain below; rows that
    // turn out suppressed are simply not OR-ed in.
    unsigned long long pre[8];
    {
      const int w = rb + 1 + wl;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int64_t row = r0 + slice + 8 * i;
        pre[i] = (w < colblocks && row < n) ? mask[row * colblocks + w] : 0ull;
      }
    }
    if (tid < 64) { // first wave: resolve the chain inside the block on the diagonal bits
      unsigned long long diag = (tid < rows) ? mask[(r0 + tid) * colblocks + rb] : 0ull;
      unsigned long long cur = remv[rb], kept = 0ull;
      for (int b = 0; b < rows; ++b) {
        unsigned long long db = __shfl(diag, b); // wave-uniform trip
        if (!((cur >> b) & 1ull)) { kept |= 1ull << b; cur |= db; }
      }
      // kept rows go to the list at their rank (all lanes at once)
      const int nk0 = s_nk;
      if ((kept >> tid) & 1ull) {
        const int64_t pos = nk0 + __popcll(kept & ((1ull << tid) - 1ull));
        if (pos < post_max) keep[pos] = r0 + tid;
      }
      if (tid == 0) {
        s_nk = nk0 + (int)__popcll(kept);
        s_kept = kept;
      }
    }
    __syncthreads();
    const unsigned long long kept = s_kept;
    if (s_nk >= post_max) break; // uniform
    // OR the kept rows' words into remv: 32 words x 8 row slices per pass (rows b = slice mod 8), loads of
    // a thread are independent; slices meet in LDS with a 64-bit atomic OR
    {
      const unsigned long long mine = kept & (0x0101010101010101ull << slice);
      if (mine) {
        unsigned long long acc0 = 0ull;
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if ((mine >> (slice + 8 * i)) & 1ull) acc0 |= pre[i];
        if (acc0) atomicOr(&remv[rb + 1 + wl], acc0);
      }
      if (mine)
        for (int w = rb + 1 + wl + 32; w < colblocks; w += 32) {
          unsigned long long acc = 0ull, kb = mine;
          while (kb) {
            const int b = __ffsll((long long)kb) - 1;
            kb &= kb - 1;
            acc |= mask[(r0 + b) * colblocks + w];
          }
          if (acc) atomicOr(&remv[w], acc);
        }
    }
    __syncthreads();
  }
  if (tid == 0) meta[0] = (int32_t)(s_nk < post_max ? s_nk : post_max);
}

} // namespace aabr
using namespace aabr;

extern "C" int aabr_rotate_iou_eval(const float *boxes, int64_t N, const float *query, int64_t K,
                                    int criterion, float *iou, void *stream_) {
  AABR_CHECK_ARG(N >= 0 && K >= 0, "bad sizes");
  if (N == 0 || K == 0) return AABR_OK;
  AABR_CHECK_ARG(boxes && query && iou, "null pointer");
  AABR_CHECK_ARG(ceil_div(K, 64) <= 65535, "too many query boxes");
  hipLaunchKernelGGL(k_rotate_iou_eval, dim3((unsigned)ceil_div(N, 4), (unsigned)ceil_div(K, 64)), dim3(256), 0,
                     (hipStream_t)stream_, boxes, N, query, K, criterion, iou);
  AABR_CHECK_LAUNCH();
  return AABR_OK;
}

extern "C" int aabr_boxes_iou_3d(const float *targets, int64_t M, const float *anchors, int64_t K,
                                 const float *aug_host, int criterion, int only_xy, float *iou,
                                 void *stream_) {
  hipStream_t st = (hipStream_t)stream_;
  AABR_CHECK_ARG(M >= 0 && K >= 0 && aug_host, "bad arguments");
  if (M == 0 || K == 0) return AABR_OK;
  AABR_CHECK_ARG(targets && anchors && iou, "null pointer");
  AABR_CHECK_ARG(ceil_div(K, 64) <= 65535, "too many anchors");
  // scratch carved from a stream-ordered allocation (small: 7 floats per box)
  float *tmp = nullptr;
  if (hipMallocAsync((void **)&tmp, (size_t)(M + K) * 7 * sizeof(float), st) != hipSuccess) {
    set_error("aabr_boxes_iou_3d: hipMallocAsync failed");
    return AABR_ELAUNCH;
  }
  float *t5 = tmp, *a5 = t5 + 5 * M, *tz = a5 + 5 * K, *az = tz + 2 * M;
  hipLaunchKernelGGL(k_box7_to_2d, dim3((unsigned)ceil_div(M, 256)), dim3(256), 0, st, targets, M, aug_host[0],
                     aug_host[1], t5, tz);
  hipLaunchKernelGGL(k_box7_to_2d, dim3((unsigned)ceil_div(K, 256)), dim3(256), 0, st, anchors, K, aug_host[2],
                     aug_host[3], a5, az);
  hipLaunchKernelGGL(k_rotate_iou_eval, dim3((unsigned)ceil_div(M, 4), (unsigned)ceil_div(K, 64)), dim3(256), 0,
                     st, t5, M, a5, K, criterion, iou);
  if (!only_xy)
    hipLaunchKernelGGL(k_scale_by_z, dim3((unsigned)ceil_div(M * K, 256)), dim3(256), 0, st, iou, M, K, tz, az);
  hipFreeAsync(tmp, st);
  AABR_CHECK_LAUNCH();
  return AABR_OK;
}

extern "C" int aabr_rpn_decode(const int32_t *site_coords, int64_t site_begin, const int64_t *selected,
                               int64_t k, const float *regression, int64_t reg_begin,
                               const float *base_anchors, int num_anchors, float voxel_scale,
                               const float *stride_host, const float *weights_host, float clip, float *boxes,
                               void *stream_) {
  AABR_CHECK_ARG(k >= 0 && num_anchors > 0 && voxel_scale > 0 && stride_host && weights_host, "bad arguments");
  if (k == 0) return AABR_OK;
  AABR_CHECK_ARG(site_coords && selected && regression && base_anchors && boxes, "null pointer");
  RpnDecodeParams p;
  p.inv_scale_num = voxel_scale;
  for (int d = 0; d < 3; ++d) p.stride[d] = stride_host[d];
  for (int d = 0; d < 7; ++d) p.weights[d] = weights_host[d];
  p.clip = clip;
  hipLaunchKernelGGL(k_rpn_decode, dim3((unsigned)ceil_div(k, 256)), dim3(256), 0, (hipStream_t)stream_,
                     site_coords, site_begin, selected, k, regression, reg_begin, base_anchors, num_anchors, p,
                     boxes);
  AABR_CHECK_LAUNCH();
  return AABR_OK;
}

extern "C" int aabr_rpn_decode_maps(int n_maps, const void *const *coords_ptrs, const void *const *logit_ptrs,
                                    const void *const *regression_ptrs, const int32_t *seg_begin_host,
                                    const int32_t *site_begin_host, const float *strides_host,
                                    const float *base_anchors, int num_anchors, float voxel_scale,
                                    const float *weights_host, float clip, float nms_min_yx, float nms_min_z,
                                    const int64_t *selected, int64_t k, float *boxes, float *nms_boxes,
                                    float *scores, void *stream_) {
  AABR_CHECK_ARG(n_maps >= 1 && n_maps <= kMaxRpnMaps && k >= 0 && num_anchors > 0 && voxel_scale > 0,
                 "bad arguments");
  AABR_CHECK_ARG(coords_ptrs && logit_ptrs && regression_ptrs && seg_begin_host && site_begin_host &&
                     strides_host && weights_host, "null host table");
  if (k == 0) return AABR_OK;
  AABR_CHECK_ARG(selected && base_anchors && boxes, "null pointer");
  RpnMapsParams p;
  for (int m = 0; m < kMaxRpnMaps; ++m) {
    const bool on = m < n_maps;
    p.coords[m] = on ? (const int32_t *)coords_ptrs[m] : nullptr;
    p.logits[m] = on ? (const float *)logit_ptrs[m] : nullptr;
    p.regression[m] = on ? (const float *)regression_ptrs[m] : nullptr;
    p.site_begin[m] = on ? site_begin_host[m] : 0;
    for (int d = 0; d < 3; ++d) p.stride[m][d] = on ? strides_host[3 * m + d] : 0.f;
  }
  for (int m = 0; m <= kMaxRpnMaps; ++m) p.seg_begin[m] = seg_begin_host[m <= n_maps ? m : n_maps];
  for (int m = 0; m < n_maps; ++m) {
    AABR_CHECK_ARG(p.seg_begin[m + 1] >= p.seg_begin[m], "segment table must be non-decreasing");
    AABR_CHECK_ARG(p.seg_begin[m + 1] == p.seg_begin[m] || (p.coords[m] && p.logits[m] && p.regression[m]),
                   "null map pointer");
  }
  p.n_maps = n_maps; p.A = num_anchors; p.voxel_scale = voxel_scale; p.clip = clip;
  p.nms_min_yx = nms_min_yx; p.nms_min_z = nms_min_z;
  for (int d = 0; d < 7; ++d) p.weights[d] = weights_host[d];
  hipLaunchKernelGGL(k_rpn_decode_maps, dim3((unsigned)ceil_div(k, 256)), dim3(256), 0, (hipStream_t)stream_, p,
                     selected, k, base_anchors, boxes, nms_boxes, scores);
  AABR_CHECK_LAUNCH();
  return AABR_OK;
}

template <int KIND>
static int nms_sorted_impl(const float *boxes, int64_t n, float thresh, int only_xy, int64_t post_max,
                           uint64_t *mask, int64_t *keep, int32_t *meta, hipStream_t st, const char *fn) {
  if (n < 0 || !meta) { set_error("%s: bad arguments", fn); return AABR_EINVAL; }
  if (post_max < 0) post_max = n;
  hipMemsetAsync(meta, 0, AABR_META_WORDS * sizeof(int32_t), st);
  if (n == 0 || post_max == 0) return AABR_OK;
  if (!boxes || !mask || !keep) { set_error("%s: null pointer", fn); return AABR_EINVAL; }
  int64_t colblocks = ceil_div(n, 64);
  if (colblocks > 8192) { set_error("%s: n too large (max 524288)", fn); return AABR_EINVAL; }
  if constexpr (KIND == 0)
    hipLaunchKernelGGL(k_nms_mask_rot, dim3((unsigned)ceil_div(n, kNmsRows), (unsigned)colblocks), dim3(256), 0, st,
                       boxes, n, thresh, only_xy, (int)colblocks, reinterpret_cast<unsigned long long *>(mask));
  else
  hipLaunchKernelGGL((k_nms_mask<KIND>), dim3((unsigned)ceil_div(n, 4), (unsigned)colblocks), dim3(256), 0, st,
                     boxes, n, thresh, only_xy, (int)colblocks, (unsigned long long *)mask);
  hipLaunchKernelGGL(k_nms_scan, dim3(1), dim3(256), (size_t)colblocks * 8, st,
                     (const unsigned long long *)mask, n, (int)colblocks, post_max, keep, meta);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("%s: HIP error %s", fn, hipGetErrorString(e)); return AABR_ELAUNCH; }
  return AABR_OK;
}

extern "C" int aabr_rotate_nms_sorted(const float *boxes7, int64_t n, float thresh, int only_xy,
                                      int64_t post_max, uint64_t *mask, int64_t *keep, int32_t *meta,
                                      void *stream_) {
  return nms_sorted_impl<0>(boxes7, n, thresh, only_xy, post_max, mask, keep, meta, (hipStream_t)stream_,
                            "aabr_rotate_nms_sorted");
}
extern "C" int aabr_nms_sorted(const float *dets4, int64_t n, float thresh, uint64_t *mask, int64_t *keep,
                               int32_t *meta, void *stream_) {
  return nms_sorted_impl<1>(dets4, n, thresh, 1, -1, mask, keep, meta, (hipStream_t)stream_, "aabr_nms_sorted");
}

// ---- the proposal stage of a whole batch (extension) ---------------------------------------------------------------
// The reference's RPNPostProcessor loops over the examples in Python (rpn/inference_3d.py:95-149).  These two
// entries let a host do the same work with one top-k over a padded matrix and one call here:
//   aabr_rpn_gather_logits: logits[b][j] = the j-th logit of example b's cross-scale anchor list (maps in order),
//                           -inf past its end, so ONE torch.topk(dim=1) selects for every example;
//   aabr_rpn_proposals_batch: per example the decode of the selected anchors (aabr_rpn_decode_maps) and the rotated
//                           NMS of the decoded list (aabr_rotate_nms_sorted), looped HERE; the numbers kept stay on
//                           the device (meta[b][0]) for the caller to read once.
constexpr int kMaxRpnBatch = 16;
struct RpnGatherParams {
  const float *logits[kMaxRpnMaps];
  int32_t seg_begin[kMaxRpnBatch][kMaxRpnMaps + 1]; // in anchors, per example
  int32_t src_begin[kMaxRpnBatch][kMaxRpnMaps];     // first anchor of example b in map m's logit vector
  int n_maps, nb;
};

// RPN label generation, fused (reference: RPNLossComputation.match_targets_to_anchors, modeling/rpn/loss_3d.py:91-100:
// `boxlist_iou_3d(target, anchor, aug_thickness, criterion, flag='rpn_label_generation')` over the anchors of ALL maps
// of an example, the |yaw difference| of every pair (utils3d/geometric_torch.py:4-21), then Matcher.__call__ as
// make_rpn_loss_evaluator builds it, loss_3d.py:338-344 / modeling/matcher.py:50-196: yaw mask, best ground truth
// per anchor, the two thresholds, set_low_quality_matches_ and its ignore-nearby pass).
// One thread per anchor of one example: the anchor is generated from its site (anchor_generator_sparse3d.py:88-104)
// through the same segment table as k_rpn_decode_maps, the example's ground-truth boxes sit in LDS with the
// target-side thickness clamps applied (rotate_nms_3d_torch.py:59-66), every pair goes through the same
// iou_eval_entry as aabr_boxes_iou_3d.  The [G, N] matrix is never stored (only when the caller asks for it):
// set_low_quality_matches_ needs every ground truth's row maximum before any anchor can be labelled, so the pairs are
// evaluated twice -- PASS 0 reduces the row maxima (wave max -> LDS -> one device atomicMax per row and workgroup, on
// order-preserving integer keys: max is order-independent, the result is bit-reproducible), PASS 1 re-evaluates the
// same pairs with the same instructions and labels the anchors.
struct RpnLabelParams {
  const int32_t *coords[kMaxRpnMaps];
  const float *targets[kMaxRpnBatch];               // [G_b, 7] yx_zb
  int32_t n_targets[kMaxRpnBatch];
  int32_t gt_begin[kMaxRpnBatch];                   // first row-maximum key of example b
  int32_t seg_begin[kMaxRpnBatch][kMaxRpnMaps + 1]; // in anchors, per example
  int32_t site_begin[kMaxRpnBatch][kMaxRpnMaps];
  int64_t out_begin[kMaxRpnBatch];                  // first anchor of example b in the concatenated outputs
  int64_t iou_begin[kMaxRpnBatch];                  // first float of example b's [G_b, N_b] matrix
  float stride[kMaxRpnMaps][3];
  float aug[4];                                     // target_Y, target_Z, anchor_Y, anchor_Z
  int n_maps, A, criterion, only_xy, use_yaw, allow_low;
  float voxel_scale, fg, bg, yaw_thr;
  float w[7];                                       // BoxCoder3D weights of the regression targets
};
constexpr int kLabelTgtChunk = 128;

// BoxCoder3D.encode_centroid_box (modeling/box_coder_3d.py:46-51) = second_box_encode(targets, anchors, smooth_dim=True)
// (second/pytorch/core/box_torch_ops.py:82-116; both boxes split positionally as x, y, z, w, l, h, r), the yaw difference
// wrapped by limit_period(., 0.5, pi) (utils3d/geometric_torch.py:4-10), times the coder's weights -- the same fp32
// operations in the same order as the torch expressions (no contraction: -ffp-contract=off)
__device__ __forceinline__ void box_encode7(const float *g, const float *a, const float *w, float *o) {
  const float diagonal = sqrtf(a[4] * a[4] + a[3] * a[3]);
  float e[7];
  e[0] = (g[0] - a[0]) / diagonal;
  e[1] = (g[1] - a[1]) / diagonal;
  e[2] = (g[2] - a[2]) / a[5];
  e[3] = g[3] / a[3] - 1.0f;
  e[4] = g[4] / a[4] - 1.0f;
  e[5] = g[5] / a[5] - 1.0f;
  const float kPi = 3.14159274101257324f;             // (float)math.pi
  const float rt = g[6] - a[6];
  e[6] = rt - floorf(rt / kPi + 0.5f) * kPi;
#pragma unroll
  for (int d = 0; d < 7; ++d) o[d] = e[d] * w[d];
}

// order-preserving float -> uint32 key (0 = below every float): the row maxima are reduced with integer atomicMax
__device__ __forceinline__ uint32_t label_key(float v) {
  const uint32_t u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float label_unkey(uint32_t k) {
  return __uint_as_float((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k);
}

// match-quality entry of (ground truth g, this thread's anchor): the IoU of boxlist_iou_3d and, masked, what the
// Matcher sees (matcher.py:50-55: `match_quality_matrix * (abs(yaw_diff) < yaw_threshold).float()`; yaw_diff =
// limit_period(target_yaw - anchor_yaw, 0.5, pi) evaluated in fp32 like the torch expression)
__device__ __forceinline__ float label_pair(const RpnLabelParams &p, const float *t5, float t0, float t1,
                                            const float *a5, float az0, float az1, float *raw) {
  float v = iou_eval_entry(t5, a5, p.criterion);
  if (!p.only_xy) {
    const float overlap = fminf(az1, t1) - fmaxf(az0, t0);
    const float common = fmaxf(az1, t1) - fminf(az0, t0);
    v = v * (overlap / common);
  }
  *raw = v;
  if (p.use_yaw) {
    const float kPi = 3.14159274101257324f;           // (float)math.pi
    const float dif = t5[4] - a5[4];
    const float wrapped = dif - floorf(dif / kPi + 0.5f) * kPi;
    v = v * (fabsf(wrapped) < p.yaw_thr ? 1.0f : 0.0f);
  }
  return v;
}

template <int PASS>
__global__ __launch_bounds__(256) void k_rpn_label_maps(RpnLabelParams p, const float *__restrict__ base_anchors,
                                                        int64_t *__restrict__ matched_idx,
                                                        float *__restrict__ matched_val, float *__restrict__ iou_out,
                                                        uint32_t *__restrict__ gt_best,
                                                        float *__restrict__ reg_targets) {
  __shared__ float s_t5[kLabelTgtChunk][5];
  __shared__ float s_tz[kLabelTgtChunk][2];
  __shared__ uint32_t s_key[kLabelTgtChunk];        // PASS 0: this workgroup's row maxima
  __shared__ float s_hi[kLabelTgtChunk][2];         // PASS 1: row maximum, ignore threshold
  const int b = blockIdx.y;
  const int64_t N = p.seg_begin[b][p.n_maps];
  if ((int64_t)blockIdx.x * blockDim.x >= N) return;   // grid.x is sized for the largest example
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int G = p.n_targets[b];
  float a5[5] = {0.f, 0.f, 0.f, 0.f, 0.f}, az0 = 0.f, az1 = 0.f;
  float an[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (t < N) {
    int m = 0;
#pragma unroll
    for (int q = 1; q < kMaxRpnMaps; ++q)
      if (q < p.n_maps && t >= p.seg_begin[b][q]) m = q;
    const int32_t r = (int32_t)(t - p.seg_begin[b][m]);
    const int64_t site = (int64_t)p.site_begin[b][m] + r / p.A;
    const int a = r % p.A;
    const int32_t *sc = p.coords[m] + 4 * site;
    const float *ba = base_anchors + 7 * ((int64_t)m * p.A + a);
#pragma unroll
    for (int d = 0; d < 3; ++d) an[d] = (float)sc[d] / p.voxel_scale * p.stride[m][d] + ba[d];
#pragma unroll
    for (int d = 3; d < 7; ++d) an[d] = 0.0f + ba[d];
    const float th = an[3] < p.aug[2] ? p.aug[2] : an[3];
    const float h = an[5] < p.aug[3] ? p.aug[3] : an[5];
    a5[0] = an[0]; a5[1] = an[1]; a5[2] = th; a5[3] = an[4]; a5[4] = an[6];
    az0 = an[2]; az1 = an[2] + h;
  }
  float best = -__builtin_inff();
  int best_g = 0;
  bool tie = false, near = false;
  for (int g0 = 0; g0 < G; g0 += kLabelTgtChunk) {
    const int gn = G - g0 < kLabelTgtChunk ? G - g0 : kLabelTgtChunk;
    __syncthreads();
    if ((int)threadIdx.x < gn) {
      const float *tb = p.targets[b] + 7 * (int64_t)(g0 + threadIdx.x);
      const float th = tb[3] < p.aug[0] ? p.aug[0] : tb[3];
      const float h = tb[5] < p.aug[1] ? p.aug[1] : tb[5];
      s_t5[threadIdx.x][0] = tb[0]; s_t5[threadIdx.x][1] = tb[1]; s_t5[threadIdx.x][2] = th;
      s_t5[threadIdx.x][3] = tb[4]; s_t5[threadIdx.x][4] = tb[6];
      s_tz[threadIdx.x][0] = tb[2]; s_tz[threadIdx.x][1] = tb[2] + h;
      if (PASS == 0) s_key[threadIdx.x] = 0u;
      if (PASS == 1 && p.allow_low) {
        const float hi = label_unkey(gt_best[p.gt_begin[b] + g0 + threadIdx.x]);
        const float thr = hi - 0.05f;                 // matcher.py:166-167
        s_hi[threadIdx.x][0] = hi;
        s_hi[threadIdx.x][1] = 0.02f > thr ? 0.02f : thr;
      }
    }
    __syncthreads();
    for (int g = 0; g < gn; ++g) {
      float t5[5];
#pragma unroll
      for (int d = 0; d < 5; ++d) t5[d] = s_t5[g][d];
      float raw;
      const float v = label_pair(p, t5, s_tz[g][0], s_tz[g][1], a5, az0, az1, &raw);
      if (PASS == 0) {
        uint32_t k = t < N ? label_key(v) : 0u;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
          const uint32_t q = (uint32_t)__shfl_xor((int)k, o, 64);
          k = q > k ? q : k;
        }
        if ((threadIdx.x & 63) == 0) atomicMax(&s_key[g], k);
      } else if (t < N) {
        if (iou_out) iou_out[p.iou_begin[b] + (int64_t)(g0 + g) * N + t] = raw;
        if (v > best) { best = v; best_g = g0 + g; }   // first maximum, like torch.max(dim=0) on the host
        if (p.allow_low) {
          tie = tie || (v == s_hi[g][0]);              // matcher.py:126-128 (== : -0 ties with +0, as in torch)
          near = near || (v > s_hi[g][1]);             // :168
        }
      }
    }
    if (PASS == 0) {
      __syncthreads();
      if ((int)threadIdx.x < gn && s_key[threadIdx.x])
        atomicMax(gt_best + p.gt_begin[b] + g0 + threadIdx.x, s_key[threadIdx.x]);
    }
  }
  if (PASS == 1 && t < N) {
    const int64_t o = p.out_begin[b] + t;
    if (G == 0) {
      matched_idx[o] = -1;
      matched_val[o] = 0.f;
      if (reg_targets) {       // an example without ground truth: matched_targets = anchor (loss_3d.py:91-94)
        float e[7];
        box_encode7(an, an, p.w, e);
#pragma unroll
        for (int d = 0; d < 7; ++d) reg_targets[7 * o + d] = e[d];
      }
      return;
    }
    matched_val[o] = best;
    int64_t mi = best < p.bg ? -1 : (best < p.fg ? -2 : best_g);   // BELOW_LOW_THRESHOLD / BETWEEN_THRESHOLDS
    if (p.allow_low) {
      if (tie) mi = best_g;                            // matches[pred_inds_to_update] = all_matches[...]
      if (mi == -1 && near) mi = -2;                   // IGNORE_HIGHEST_MATCH_NEARBY
    }
    matched_idx[o] = mi;
    if (reg_targets) {
      // regression target of EVERY anchor, as RPNLossComputation.prepare_targets forms it (loss_3d.py:186-196):
      // box_coder.encode(target[matched_idxs.clamp(min=0)], anchor) -- the un-thickened boxes, anchor and ground truth
      const float *tb = p.targets[b] + 7 * (mi < 0 ? 0 : mi);
      float g7[7], e[7];
#pragma unroll
      for (int d = 0; d < 7; ++d) g7[d] = tb[d];
      box_encode7(g7, an, p.w, e);
#pragma unroll
      for (int d = 0; d < 7; ++d) reg_targets[7 * o + d] = e[d];
    }
  }
}

__global__ __launch_bounds__(256) void k_rpn_gather_logits(RpnGatherParams p, int64_t lmax, float *__restrict__ out) {
  const int b = blockIdx.y;
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= lmax) return;
  float v = -__builtin_inff();
  if (j < p.seg_begin[b][p.n_maps]) {
    int m = 0;
    while (m + 1 < p.n_maps && j >= p.seg_begin[b][m + 1]) ++m;
    v = p.logits[m][(int64_t)p.src_begin[b][m] + (j - p.seg_begin[b][m])];
  }
  out[(int64_t)b * lmax + j] = v;
}

extern "C" int aabr_rpn_gather_logits(int n_maps, const void *const *logit_ptrs, int nb, const int32_t *seg_begin_host,
                                      const int32_t *site_begin_host, int num_anchors, int64_t lmax, float *out,
                                      void *stream_) {
  AABR_CHECK_ARG(n_maps >= 1 && n_maps <= kMaxRpnMaps && nb >= 1 && nb <= kMaxRpnBatch && num_anchors > 0 && lmax > 0,
                 "bad arguments (at most 8 maps, 16 examples)");
  AABR_CHECK_ARG(logit_ptrs && seg_begin_host && site_begin_host && out, "null pointer");
  RpnGatherParams p;
  p.n_maps = n_maps; p.nb = nb;
  for (int m = 0; m < kMaxRpnMaps; ++m) p.logits[m] = m < n_maps ? (const float *)logit_ptrs[m] : nullptr;
  for (int b = 0; b < nb; ++b) {
    for (int m = 0; m <= n_maps; ++m) p.seg_begin[b][m] = seg_begin_host[b * (n_maps + 1) + m];
    for (int m = 0; m < n_maps; ++m) {
      p.src_begin[b][m] = site_begin_host[b * n_maps + m] * num_anchors;
      AABR_CHECK_ARG(p.seg_begin[b][m + 1] >= p.seg_begin[b][m], "segment table must be non-decreasing");
      AABR_CHECK_ARG(p.seg_begin[b][m + 1] == p.seg_begin[b][m] || p.logits[m], "null map pointer");
    }
    AABR_CHECK_ARG(p.seg_begin[b][n_maps] <= lmax, "lmax smaller than an example's list");
  }
  hipLaunchKernelGGL(k_rpn_gather_logits, dim3((unsigned)ceil_div(lmax, 256), (unsigned)nb), dim3(256), 0,
                     (hipStream_t)stream_, p, lmax, out);
  AABR_CHECK_LAUNCH();
  return AABR_OK;
}

// ---- fused cross-scale top-k (round 5; replaces torch.topk + torch.cat of the proposal stage) ----------------------------
// The reference's RPNPostProcessor does, per example, objectness.sigmoid() -> topk(pre_nms_top_n, sorted) over the anchors
// of all maps (rpn/inference_3d.py:107-112).  torch.topk is a chain of ~9 rocprim launches per example (radix select +
// merge sort), run per example on a concatenated copy.  Here the examples of a step go through FOUR launches together and
// nothing is concatenated (the j-th logit of an example is read through the segment table, like k_rpn_decode_maps does):
//   K1 histogram of the 12 leading bits of the order-preserving key of every logit (LDS histogram per workgroup);
//   K2 every workgroup finds the bin where the count from the top reaches k, histogram of the NEXT 12 bits inside it;
//   K3 the same one level down, then every logit whose 24-bit prefix is >= the threshold prefix goes to the candidate
//      list (<= 4096: k plus the ties of one 24-bit prefix; more = overflow, reported, the caller falls back);
//   K4 one workgroup per example sorts the candidates (bitonic, in LDS) by (logit descending, index ascending) and
//      writes the first k indices.
// Same selection as torch.topk(sorted=True) except between EXACTLY equal logits, where torch leaves the order (and the
// choice at the cut) open and this picks the lower index -- deterministic.
constexpr int kTopkBins = 4096, kTopkCand = 4096;
struct RpnTopkParams {
  RpnGatherParams g;
  int32_t k[kMaxRpnBatch];
};
// per example: hist1[4096], hist2[4096], then 8 control words, then the candidates (uint64 x kTopkCand)
constexpr int kTopkCtl = 8, kTopkWordsPerEx = 2 * kTopkBins + kTopkCtl + 2 * kTopkCand;
enum { kTkBin1 = 0, kTkAbove1 = 1, kTkCount = 2, kTkOverflow = 3 };
__device__ inline uint32_t topk_key(float f) {
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);          // ascending in the float order (-0 < +0: harmless)
}
__device__ inline float topk_logit(const RpnGatherParams &p, int b, int64_t j) {
  int m = 0;
  while (m + 1 < p.n_maps && j >= p.seg_begin[b][m + 1]) ++m;
  return p.logits[m][(int64_t)p.src_begin[b][m] + (j - p.seg_begin[b][m])];
}
// the bin (from the top) in which the running count reaches `need`: returns it, and the count ABOVE it through `above`
__device__ inline int topk_find_bin(const int32_t *__restrict__ hist, int need, int &above) {
  __shared__ int s_part[256], s_bin, s_above;
  const int t = threadIdx.x;
  int sum = 0;                                                // thread t owns bins [4095 - 16 t - 15, 4095 - 16 t], top first
  for (int q = 0; q < 16; ++q) sum += hist[kTopkBins - 1 - (16 * t + q)];
  s_part[t] = sum;
  if (t == 0) { s_bin = 0; s_above = 0; }
  __syncthreads();
  if (t == 0) {
    int run = 0, bt = 255;
    for (int i = 0; i < 256; ++i) {
      if (run + s_part[i] >= need) { bt = i; break; }
      run += s_part[i];
    }
    int bin = 0, ab = run;
    for (int q = 0; q < 16; ++q) {
      const int bq = kTopkBins - 1 - (16 * bt + q);
      const int c = hist[bq];
      if (ab + c >= need || q == 15) { bin = bq; break; }
      ab += c;
    }
    s_bin = bin;
    s_above = ab;
  }
  __syncthreads();
  above = s_above;
  return s_bin;
}
template <int LEVEL>   // 1: leading 12 bits of every logit; 2: the next 12 bits of the logits inside the level-1 threshold bin
__global__ __launch_bounds__(256) void k_rpn_topk_hist(RpnTopkParams p, int32_t *__restrict__ scratch) {
  __shared__ int32_t s_hist[kTopkBins];
  const int b = blockIdx.y;
  int32_t *sc = scratch + (int64_t)b * kTopkWordsPerEx;
  const int64_t n = p.g.seg_begin[b][p.g.n_maps];
  int bin1 = 0;
  if (LEVEL == 2) {
    int above;
    bin1 = topk_find_bin(sc, p.k[b], above);
    if (blockIdx.x == 0 && threadIdx.x == 0) { sc[2 * kTopkBins + kTkBin1] = bin1; sc[2 * kTopkBins + kTkAbove1] = above; }
  }
  for (int q = threadIdx.x; q < kTopkBins; q += 256) s_hist[q] = 0;
  __syncthreads();
  for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < n; j += (int64_t)gridDim.x * 256) {
    const uint32_t key = topk_key(topk_logit(p.g, b, j));
    if (LEVEL == 1) atomicAdd(&s_hist[key >> 20], 1);
    else if ((int)(key >> 20) == bin1) atomicAdd(&s_hist[(key >> 8) & 4095u], 1);
  }
  __syncthreads();
  int32_t *h = sc + (LEVEL == 1 ? 0 : kTopkBins);
  for (int q = threadIdx.x; q < kTopkBins; q += 256)
    if (s_hist[q]) atomicAdd(&h[q], s_hist[q]);
}
__global__ __launch_bounds__(256) void k_rpn_topk_compact(RpnTopkParams p, int32_t *__restrict__ scratch) {
  const int b = blockIdx.y;
  int32_t *sc = scratch + (int64_t)b * kTopkWordsPerEx;
  int32_t *ctl = sc + 2 * kTopkBins;
  unsigned long long *cand = reinterpret_cast<unsigned long long *>(ctl + kTopkCtl);
  const int64_t n = p.g.seg_begin[b][p.g.n_maps];
  const int bin1 = ctl[kTkBin1];
  int above2;
  const int bin2 = topk_find_bin(sc + kTopkBins, p.k[b] - ctl[kTkAbove1], above2);
  const uint32_t thr = ((uint32_t)bin1 << 12) | (uint32_t)bin2;
  for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < n; j += (int64_t)gridDim.x * 256) {
    const uint32_t key = topk_key(topk_logit(p.g, b, j));
    if ((key >> 8) >= thr) {
      const int pos = atomicAdd(&ctl[kTkCount], 1);
      if (pos < kTopkCand) cand[pos] = ((unsigned long long)(~key) << 32) | (unsigned long long)(uint32_t)j;
      else ctl[kTkOverflow] = 1;
    }
  }
}
__global__ __launch_bounds__(1024) void k_rpn_topk_sort(RpnTopkParams p, int32_t *__restrict__ scratch, int64_t *__restrict__ sel,
                                                        int64_t sel_stride, int32_t *__restrict__ info) {
  __shared__ unsigned long long s[kTopkCand];
  const int b = blockIdx.x;
  int32_t *ctl = scratch + (int64_t)b * kTopkWordsPerEx + 2 * kTopkBins;
  const unsigned long long *cand = reinterpret_cast<const unsigned long long *>(ctl + kTopkCtl);
  int C = ctl[kTkCount];
  C = C < kTopkCand ? C : kTopkCand;
  for (int i = threadIdx.x; i < kTopkCand; i += 1024) s[i] = i < C ? cand[i] : ~0ull;
  for (int size = 2; size <= kTopkCand; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      __syncthreads();
      for (int i = threadIdx.x; i < kTopkCand; i += 1024) {
        const int j = i ^ stride;
        if (j > i) {
          const unsigned long long a = s[i], c = s[j];
          const bool up = (i & size) == 0;
          if ((a > c) == up) { s[i] = c; s[j] = a; }
        }
      }
    }
  __syncthreads();
  const int k = p.k[b];
  for (int i = threadIdx.x; i < k; i += 1024) sel[(int64_t)b * sel_stride + i] = (int64_t)(uint32_t)s[i];
  if (threadIdx.x == 0) {
    info[2 * b] = ctl[kTkCount];
    info[2 * b + 1] = (ctl[kTkOverflow] || ctl[kTkCount] < k) ? 1 : 0;
  }
}

extern "C" int64_t aabr_rpn_topk_scratch_words(int nb) { return (int64_t)(nb > 0 ? nb : 0) * kTopkWordsPerEx; }

extern "C" int aabr_rpn_topk_maps(int n_maps, const void *const *logit_ptrs, int nb, const int32_t *seg_begin_host,
                                  const int32_t *site_begin_host, int num_anchors, const int32_t *k_host, int64_t *selected,
                                  int64_t sel_stride, int32_t *info, int32_t *scratch, void *stream_) {
  hipStream_t st = (hipStream_t)stream_;
  AABR_CHECK_ARG(n_maps >= 1 && n_maps <= kMaxRpnMaps && nb >= 1 && nb <= kMaxRpnBatch && num_anchors > 0,
                 "bad arguments (at most 8 maps, 16 examples)");
  AABR_CHECK_ARG(logit_ptrs && seg_begin_host && site_begin_host && k_host && selected && info && scratch, "null pointer");
  AABR_CHECK_ARG(((uintptr_t)scratch & 7) == 0, "scratch must be 8-byte aligned");
  RpnTopkParams p;
  p.g.n_maps = n_maps; p.g.nb = nb;
  int64_t nmax = 0;
  for (int m = 0; m < kMaxRpnMaps; ++m) p.g.logits[m] = m < n_maps ? (const float *)logit_ptrs[m] : nullptr;
  for (int b = 0; b < nb; ++b) {
    for (int m = 0; m <= n_maps; ++m) p.g.seg_begin[b][m] = seg_begin_host[b * (n_maps + 1) + m];
    for (int m = 0; m < n_maps; ++m) {
      p.g.src_begin[b][m] = site_begin_host[b * n_maps + m] * num_anchors;
      AABR_CHECK_ARG(p.g.seg_begin[b][m + 1] >= p.g.seg_begin[b][m], "segment table must be non-decreasing");
      AABR_CHECK_ARG(p.g.seg_begin[b][m + 1] == p.g.seg_begin[b][m] || p.g.logits[m], "null map pointer");
    }
    const int64_t n = p.g.seg_begin[b][n_maps];
    AABR_CHECK_ARG(k_host[b] >= 0 && k_host[b] <= n && k_host[b] <= kTopkCand / 2 && k_host[b] <= sel_stride,
                   "k: 0 .. min(anchors of the example, 2048, sel_stride)");
    p.k[b] = k_host[b];
    nmax = n > nmax ? n : nmax;
  }
  for (int b = nb; b < kMaxRpnBatch; ++b) p.k[b] = 0;
  AABR_CHECK_HIP(hipMemsetAsync(scratch, 0, (size_t)nb * kTopkWordsPerEx * sizeof(int32_t), st));
  if (nmax == 0) { AABR_CHECK_HIP(hipMemsetAsync(info, 0, (size_t)nb * 2 * sizeof(int32_t), st)); return AABR_OK; }
  int64_t gx = ceil_div(nmax, 256 * 8);
  gx = gx < 1 ? 1 : (gx > 256 ? 256 : gx);
  hipLaunchKernelGGL(k_rpn_topk_hist<1>, dim3((unsigned)gx, (unsigned)nb), dim3(256), 0, st, p, scratch);
  hipLaunchKernelGGL(k_rpn_topk_hist<2>, dim3((unsigned)gx, (unsigned)nb), dim3(256), 0, st, p, scratch);
  hipLaunchKernelGGL(k_rpn_topk_compact, dim3((unsigned)gx, (unsigned)nb), dim3(256), 0, st, p, scratch);
  hipLaunchKernelGGL(k_rpn_topk_sort, dim3((unsigned)nb), dim3(1024), 0, st, p, scratch, selected, sel_stride, info);
  AABR_CHECK_LAUNCH();
  return AABR_OK;
}

extern "C" int aabr_rpn_proposals_batch(int n_maps, const void *const *coords_ptrs, const void *const *logit_ptrs,
                                        const void *const *regression_ptrs, int nb, const int32_t *seg_begin_host,
                                        const int32_t *site_begin_host, const float *strides_host,
                                        const float *base_anchors, int num_anchors, float voxel_scale,
                                        const float *weights_host, float clip, float nms_min_yx, float nms_min_z,
                                        const int64_t *selected, int64_t k, float *boxes, float *nms_boxes,
                                        float *scores, float nms_thresh, int only_xy, int64_t post_max, uint64_t *mask,
                                        int64_t *keep, int32_t *meta, void *stream_) {
  AABR_CHECK_ARG(nb >= 1 && nb <= kMaxRpnBatch && k >= 0 && seg_begin_host && site_begin_host, "bad arguments");
  AABR_CHECK_ARG(selected && boxes && nms_boxes && scores && mask && keep && meta, "null pointer");
  const int64_t cb = ceil_div(k > 0 ? k : 1, (int64_t)64);
  for (int b = 0; b < nb; ++b) {
    int rc = aabr_rpn_decode_maps(n_maps, coords_ptrs, logit_ptrs, regression_ptrs, seg_begin_host + b * (n_maps + 1),
                                  site_begin_host + b * n_maps, strides_host, base_anchors, num_anchors, voxel_scale,
                                  weights_host, clip, nms_min_yx, nms_min_z, selected + (int64_t)b * k, k,
                                  boxes + (int64_t)b * k * 7, nms_boxes + (int64_t)b * k * 7, scores + (int64_t)b * k,
                                  stream_);
    if (rc != AABR_OK) return rc;
    rc = aabr_rotate_nms_sorted(nms_boxes + (int64_t)b * k * 7, k, nms_thresh, only_xy, post_max,
                                mask + (int64_t)b * k * cb, keep + (int64_t)b * k, meta + b * AABR_META_WORDS, stream_);
    if (rc != AABR_OK) return rc;
  }
  return AABR_OK;
}

extern "C" int aabr_rpn_label_generation_targets(int n_maps, const void *const *coords_ptrs, int nb,
                                         const int32_t *seg_begin_host, const int32_t *site_begin_host,
                                         const float *strides_host, const float *base_anchors, int num_anchors,
                                         float voxel_scale, const void *const *target_ptrs,
                                         const int32_t *n_targets_host, const float *aug_host, int criterion,
                                         int only_xy, float fg_iou, float bg_iou, float yaw_threshold,
                                         int allow_low_quality_matches, int64_t *matched_idx, float *matched_val,
                                         float *iou_out, uint32_t *row_max_scratch, const float *weights_host,
                                         float *regression_targets, void *stream_) {
  AABR_CHECK_ARG(n_maps >= 1 && n_maps <= kMaxRpnMaps && nb >= 0 && nb <= kMaxRpnBatch && num_anchors > 0 &&
                     voxel_scale > 0, "bad arguments (<= 8 maps, <= 16 examples)");
  AABR_CHECK_ARG(coords_ptrs && seg_begin_host && site_begin_host && strides_host && target_ptrs && n_targets_host &&
                     aug_host, "null host table");
  if (nb == 0) return AABR_OK;
  RpnLabelParams p;
  for (int m = 0; m < kMaxRpnMaps; ++m) {
    p.coords[m] = m < n_maps ? (const int32_t *)coords_ptrs[m] : nullptr;
    for (int d = 0; d < 3; ++d) p.stride[m][d] = m < n_maps ? strides_host[3 * m + d] : 0.f;
  }
  int64_t out = 0, mat = 0, nmax = 0;
  int32_t gts = 0;
  for (int b = 0; b < kMaxRpnBatch; ++b) {
    const bool on = b < nb;
    for (int m = 0; m <= kMaxRpnMaps; ++m)
      p.seg_begin[b][m] = on ? seg_begin_host[b * (n_maps + 1) + (m <= n_maps ? m : n_maps)] : 0;
    for (int m = 0; m < kMaxRpnMaps; ++m) p.site_begin[b][m] = on && m < n_maps ? site_begin_host[b * n_maps + m] : 0;
    p.targets[b] = on ? (const float *)target_ptrs[b] : nullptr;
    p.n_targets[b] = on ? n_targets_host[b] : 0;
    p.out_begin[b] = out;
    p.iou_begin[b] = mat;
    p.gt_begin[b] = gts;
    if (on) gts += p.n_targets[b];
    if (on) {
      AABR_CHECK_ARG(p.n_targets[b] >= 0 && (p.n_targets[b] == 0 || p.targets[b]), "null target list");
      for (int m = 0; m < n_maps; ++m) {
        AABR_CHECK_ARG(p.seg_begin[b][m + 1] >= p.seg_begin[b][m], "segment table must be non-decreasing");
        AABR_CHECK_ARG(p.seg_begin[b][m + 1] == p.seg_begin[b][m] || p.coords[m], "null map pointer");
      }
      const int64_t n = p.seg_begin[b][n_maps];
      out += n;
      mat += n * p.n_targets[b];
      if (n > nmax) nmax = n;
    }
  }
  if (nmax == 0) return AABR_OK;
  AABR_CHECK_ARG(base_anchors && matched_idx && matched_val, "null pointer");
  for (int d = 0; d < 4; ++d) p.aug[d] = aug_host[d];
  p.n_maps = n_maps; p.A = num_anchors; p.criterion = criterion; p.only_xy = only_xy;
  p.voxel_scale = voxel_scale; p.fg = fg_iou; p.bg = bg_iou;
  // Matcher.yaw_diff_constrain: no mask when the threshold exceeds 1.58 (matcher.py:51-52)
  p.use_yaw = yaw_threshold > 1.58f ? 0 : 1;
  p.yaw_thr = yaw_threshold;
  p.allow_low = allow_low_quality_matches ? 1 : 0;
  AABR_CHECK_ARG(!regression_targets || weights_host, "regression targets need the coder's 7 weights");
  for (int d = 0; d < 7; ++d) p.w[d] = regression_targets ? weights_host[d] : 1.0f;
  const dim3 grid((unsigned)ceil_div(nmax, 256), (unsigned)nb);
  if (p.allow_low && gts > 0) {
    AABR_CHECK_ARG(row_max_scratch, "allow_low_quality_matches needs the row-maximum scratch (sum of n_targets words)");
    AABR_CHECK_HIP(hipMemsetAsync(row_max_scratch, 0, sizeof(uint32_t) * (size_t)gts, (hipStream_t)stream_));
    hipLaunchKernelGGL(k_rpn_label_maps<0>, grid, dim3(256), 0, (hipStream_t)stream_, p, base_anchors, matched_idx,
                       matched_val, iou_out, row_max_scratch, (float *)nullptr);
    AABR_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(k_rpn_label_maps<1>, grid, dim3(256), 0, (hipStream_t)stream_, p, base_anchors, matched_idx,
                     matched_val, iou_out, row_max_scratch, regression_targets);
  AABR_CHECK_LAUNCH();
  return AABR_OK;
}

extern "C" int aabr_rpn_label_generation(int n_maps, const void *const *coords_ptrs, int nb,
                                         const int32_t *seg_begin_host, const int32_t *site_begin_host,
                                         const float *strides_host, const float *base_anchors, int num_anchors,
                                         float voxel_scale, const void *const *target_ptrs,
                                         const int32_t *n_targets_host, const float *aug_host, int criterion,
                                         int only_xy, float fg_iou, float bg_iou, float yaw_threshold,
                                         int allow_low_quality_matches, int64_t *matched_idx, float *matched_val,
                                         float *iou_out, uint32_t *row_max_scratch, void *stream_) {
  return aabr_rpn_label_generation_targets(n_maps, coords_ptrs, nb, seg_begin_host, site_begin_host, strides_host,
                                           base_anchors, num_anchors, voxel_scale, target_ptrs, n_targets_host, aug_host,
                                           criterion, only_xy, fg_iou, bg_iou, yaw_threshold, allow_low_quality_matches,
                                           matched_idx, matched_val, iou_out, row_max_scratch, nullptr, nullptr, stream_);
}

// BoxCoder3D.encode on two [N,7] lists (modeling/box_coder_3d.py:34-51)
struct BoxEncodeW { float w[7]; };
__global__ __launch_bounds__(256) void k_box_encode(const float *__restrict__ targets, const float *__restrict__ anchors,
                                                    int64_t n, BoxEncodeW w, float *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float g[7], a[7], e[7];
#pragma unroll
  for (int d = 0; d < 7; ++d) { g[d] = targets[7 * i + d]; a[d] = anchors[7 * i + d]; }
  box_encode7(g, a, w.w, e);
#pragma unroll
  for (int d = 0; d < 7; ++d) out[7 * i + d] = e[d];
}

extern "C" int aabr_box_encode(const float *targets, const float *anchors, int64_t n, const float *weights_host,
                               float *out, void *stream_) {
  AABR_CHECK_ARG(n >= 0 && weights_host, "bad arguments");
  if (n == 0) return AABR_OK;
  AABR_CHECK_ARG(targets && anchors && out, "null pointer");
  BoxEncodeW w;
  for (int d = 0; d < 7; ++d) w.w[d] = weights_host[d];
  hipLaunchKernelGGL(k_box_encode, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream_, targets, anchors,
                     n, w, out);
  AABR_CHECK_LAUNCH();
  return AABR_OK;
}

// BoxCoder3D.decode_centroid_box on lists (modeling/box_coder_3d.py:53-80): encodings [n, 7 * num_classes] against
// anchors [n, 7] (the anchor of a row serves all its classes) -> boxes [n, 7 * num_classes]
__global__ __launch_bounds__(256) void k_box_decode(const float *__restrict__ enc, const float *__restrict__ anchors,
                                                    int64_t n, int nc, BoxEncodeW w, float clip, float *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * nc) return;
  const float *an = anchors + 7 * (i / nc);
  float e[7];
#pragma unroll
  for (int d = 0; d < 7; ++d) e[d] = enc[7 * i + d] / w.w[d];
#pragma unroll
  for (int d = 3; d < 6; ++d) e[d] = e[d] > clip ? clip : e[d];
  const float diagonal = sqrtf(an[4] * an[4] + an[3] * an[3]);
  float o[7];
  o[0] = e[0] * diagonal + an[0];
  o[1] = e[1] * diagonal + an[1];
  o[2] = e[2] * an[5] + an[2];
  o[3] = (e[3] + 1) * an[3];
  o[4] = (e[4] + 1) * an[4];
  o[5] = (e[5] + 1) * an[5];
  const float period = 3.14159274101257324f;
  const float rg = e[6] + an[6];
  o[6] = rg - floorf(rg / period + 0.5f) * period;
#pragma unroll
  for (int d = 0; d < 7; ++d) out[7 * i + d] = o[d];
}

extern "C" int aabr_box_decode(const float *encodings, const float *anchors, int64_t n, int num_classes,
                               const float *weights_host, float clip, float *out, void *stream_) {
  AABR_CHECK_ARG(n >= 0 && num_classes >= 1 && weights_host, "bad arguments");
  if (n == 0) return AABR_OK;
  AABR_CHECK_ARG(encodings && anchors && out, "null pointer");
  BoxEncodeW w;
  for (int d = 0; d < 7; ++d) w.w[d] = weights_host[d];
  hipLaunchKernelGGL(k_box_decode, dim3((unsigned)ceil_div(n * num_classes, 256)), dim3(256), 0, (hipStream_t)stream_,
                     encodings, anchors, n, num_classes, w, clip, out);
  AABR_CHECK_LAUNCH();
  return AABR_OK;
}
