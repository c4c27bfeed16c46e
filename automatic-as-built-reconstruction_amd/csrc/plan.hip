// plan.hip -- run a compiled list of hot-path launches with ONE call across the C ABI (gfx950).
//
// The reference drives its layers from Python, one pybind call per layer and direction
// (sparseconvnet/{submanifoldConvolution,convolution,deconvolution,batchNormalization}.py ->
// SCN/pybind.cpp:134-221).  Behind the same operator API the host side here can compile the static part of a
// network (sparseconvnet/planExecutor.py: every BatchNorm / convolution / residual add between the input layer
// and the returned feature maps) into a flat array of PlanOp records and hand it over once per pass: the
// per-layer interpreter overhead (tens of microseconds per layer and direction) leaves the critical path and the
// launches go out back to back.  Every record names one of the library's own entry points; nothing here computes
// differently from calling them one by one.
//
// Also the two elementwise launches such a plan needs that the layer API gets from torch: residual add and the
// fp32 <-> bf16 storage casts (round-to-nearest-even, as torch's `.to(torch.bfloat16)`).
#include "common.h"
#include <string.h>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace aabr {

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

template <typename T>
__global__ __launch_bounds__(256) void k_add2(const T *__restrict__ a, const T *__restrict__ b, T *__restrict__ o,
                                              int64_t n) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i + 3 < n) {
    if constexpr (sizeof(T) == 4) {
      const float4 x = *reinterpret_cast<const float4 *>(a + i), y = *reinterpret_cast<const float4 *>(b + i);
      *reinterpret_cast<float4 *>(o + i) = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
    } else {
      const bf16x4 x = *reinterpret_cast<const bf16x4 *>(a + i), y = *reinterpret_cast<const bf16x4 *>(b + i);
      bf16x4 r;
#pragma unroll
      for (int j = 0; j < 4; ++j) r[j] = (__bf16)((float)x[j] + (float)y[j]);
      *reinterpret_cast<bf16x4 *>(o + i) = r;
    }
  } else {
    for (int64_t j = i; j < n; ++j) o[j] = (T)((float)a[j] + (float)b[j]);
  }
}

template <typename TI, typename TO>
__global__ __launch_bounds__(256) void k_cast(const TI *__restrict__ in, TO *__restrict__ out, int64_t n) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i + 3 < n) {
#pragma unroll
    for (int j = 0; j < 4; ++j) out[i + j] = (TO)(float)in[i + j];
  } else {
    for (int64_t j = i; j < n; ++j) out[j] = (TO)(float)in[j];
  }
}

// a run of storage casts in one launch (the fp32 copies of the returned maps at the end of a bf16-storage forward list, the
// bf16 copies of their gradients at the head of the backward list: 10 records each): block b serves job j with
// first[j] <= b < first[j + 1]
constexpr int kCastJobsMax = 16;
struct CastJobs {
  const void *in[kCastJobsMax];
  void *out[kCastJobsMax];
  int64_t n[kCastJobsMax];
  uint32_t first[kCastJobsMax + 1];
  int count;
};
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void k_cast_jobs(const CastJobs js) {
  int j = 0;
  while (j + 1 < js.count && blockIdx.x >= js.first[j + 1]) ++j;
  const TI *__restrict__ in = reinterpret_cast<const TI *>(js.in[j]);
  TO *__restrict__ out = reinterpret_cast<TO *>(js.out[j]);
  const int64_t n = js.n[j];
  const int64_t i = ((int64_t)(blockIdx.x - js.first[j]) * 256 + threadIdx.x) * 4;
  if (i + 3 < n) {
#pragma unroll
    for (int q = 0; q < 4; ++q) out[i + q] = (TO)(float)in[i + q];
  } else {
    for (int64_t q = i; q < n; ++q) out[q] = (TO)(float)in[q];
  }
}

// rule totals of many rule books in one launch: job j sums n[j] int32 counts into the float64 slot out[j]
struct SumJobs {
  const int32_t *c[64];
  int64_t n[64];
  double *o[64];
};

__global__ __launch_bounds__(256) void k_sum_counts(SumJobs jobs) {
  __shared__ long long red[256];
  const int j = blockIdx.x;
  const int32_t *c = jobs.c[j];
  long long s = 0;
  for (int64_t i = threadIdx.x; i < jobs.n[j]; i += 256) s += c[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) *jobs.o[j] = (double)red[0];   // integer sum: exact and order-independent
}

} // namespace aabr
using namespace aabr;

extern "C" int aabr_sum_counts(const int32_t *const *counts_host, const int64_t *n_host, double *const *out_host,
                               int n_jobs, void *stream_) {
  AABR_CHECK_ARG(n_jobs >= 0 && (n_jobs == 0 || (counts_host && n_host && out_host)), "bad arguments");
  for (int j0 = 0; j0 < n_jobs; j0 += 64) {
    SumJobs jobs;
    const int m = n_jobs - j0 < 64 ? n_jobs - j0 : 64;
    for (int j = 0; j < m; ++j) {
      AABR_CHECK_ARG(n_host[j0 + j] >= 0 && out_host[j0 + j] && (counts_host[j0 + j] || n_host[j0 + j] == 0),
                     "bad job");
      jobs.c[j] = counts_host[j0 + j];
      jobs.n[j] = n_host[j0 + j];
      jobs.o[j] = out_host[j0 + j];
    }
    hipLaunchKernelGGL(k_sum_counts, dim3((unsigned)m), dim3(256), 0, (hipStream_t)stream_, jobs);
  }
  AABR_CHECK_LAUNCH();
  return AABR_OK;
}

extern "C" int aabr_add(const void *a, const void *b, void *out, int64_t n, int bf16, void *stream_) {
  AABR_CHECK_ARG(n >= 0, "bad size");
  if (n == 0) return AABR_OK;
  AABR_CHECK_ARG(a && b && out, "null pointer");
  const int es = bf16 ? 2 : 4;
  AABR_CHECK_ARG((((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) & (uintptr_t)(4 * es - 1)) == 0,
                 "operands must be aligned to four elements");
  const dim3 grid((unsigned)((n + 1023) / 1024));
  AABR_CHECK_ARG((n + 1023) / 1024 < (1ll << 31), "too many elements");
  if (bf16)
    hipLaunchKernelGGL((k_add2<__bf16>), grid, dim3(256), 0, (hipStream_t)stream_, (const __bf16 *)a, (const __bf16 *)b,
                       (__bf16 *)out, n);
  else
    hipLaunchKernelGGL((k_add2<float>), grid, dim3(256), 0, (hipStream_t)stream_, (const float *)a, (const float *)b,
                       (float *)out, n);
  AABR_CHECK_LAUNCH();
  return AABR_OK;
}

extern "C" int aabr_cast_storage(const void *in, void *out, int64_t n, int to_bf16, void *stream_) {
  AABR_CHECK_ARG(n >= 0, "bad size");
  if (n == 0) return AABR_OK;
  AABR_CHECK_ARG(in && out, "null pointer");
  AABR_CHECK_ARG((n + 1023) / 1024 < (1ll << 31), "too many elements");
  const dim3 grid((unsigned)((n + 1023) / 1024));
  if (to_bf16)
    hipLaunchKernelGGL((k_cast<float, __bf16>), grid, dim3(256), 0, (hipStream_t)stream_, (const float *)in,
                       (__bf16 *)out, n);
  else
    hipLaunchKernelGGL((k_cast<__bf16, float>), grid, dim3(256), 0, (hipStream_t)stream_, (const __bf16 *)in,
                       (float *)out, n);
  AABR_CHECK_LAUNCH();
  return AABR_OK;
}

// records ops[j .. j + m) are storage casts in the same direction on the caller's stream: one launch
static int plan_cast_run(const AabrPlanOp *ops, int m, void *st) {
  CastJobs js;
  uint64_t blocks = 0;
  int c = 0;
  for (int j = 0; j < m; ++j) {
    const AabrPlanOp &o = ops[j];
    AABR_CHECK_ARG(o.i64[0] >= 0 && (o.i64[0] == 0 || (o.p[0] && o.p[1])), "cast record: bad size / null pointer");
    if (o.i64[0] == 0) continue;
    js.in[c] = o.p[0]; js.out[c] = o.p[1]; js.n[c] = o.i64[0];
    js.first[c] = (uint32_t)blocks;
    blocks += (uint64_t)((o.i64[0] + 1023) / 1024);
    ++c;
  }
  AABR_CHECK_ARG(blocks < (1ull << 31), "too many elements");
  if (c == 0) return AABR_OK;
  js.first[c] = (uint32_t)blocks;
  js.count = c;
  if (ops[0].flags & AABR_PLAN_TO_BF16)
    hipLaunchKernelGGL((k_cast_jobs<float, __bf16>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)st, js);
  else
    hipLaunchKernelGGL((k_cast_jobs<__bf16, float>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)st, js);
  AABR_CHECK_LAUNCH();
  return AABR_OK;
}

// One record = one library call; the field use per kind is listed in include/aabr_hip.h (AabrPlanOp).
static_assert(sizeof(AabrPlanOp) == 176, "AabrPlanOp layout is part of the C ABI");

// Records flagged AABR_PLAN_SIDE run on a second stream of the library's own: they start once everything recorded
// before them on the caller's stream is done (one event each) and the caller's stream waits for all of them before
// aabr_plan_run returns control of it -- or earlier, in front of a record flagged AABR_PLAN_JOIN (the first reader
// of what they produce).  Meant for launches nothing (or nothing soon) in the list reads: the weight gradients of a
// backward pass, the lateral branches of a forward pass.  They fill the CUs that the tails and the small launches
// of the main chain leave idle.  Same kernels on the same operands: the results do not depend on the interleaving.
namespace {
struct SideStream {
  hipStream_t stream = nullptr;
  size_t pending = 0;                      // side launches a held part (plan_run_part(.., hold)) left unjoined
  std::vector<hipEvent_t> events;
  hipEvent_t get(size_t k) {
    while (events.size() <= k) {
      hipEvent_t e = nullptr;
      if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
      events.push_back(e);
    }
    return events[k];
  }
};
thread_local SideStream g_side;
} // namespace

// one record = one entry point of include/aabr_hip.h on stream `st`
static int plan_dispatch(const AabrPlanOp &o, void *st) {
  const bool bf = (o.flags & AABR_PLAN_BF16) != 0;
  void *const *p = o.p;
  int rc = AABR_OK;
  switch (o.kind) {
  case AABR_PLAN_CONV:
    rc = bf ? aabr_conv_forward_bf16((const uint16_t *)p[0], o.i32[0], o.i64[0], (uint16_t *)p[1], o.i32[1], o.i64[1],
                                     (const int32_t *)p[2], o.i32[2], (const float *)p[3], (const float *)p[4],
                                     o.i32[3], (uint16_t *)p[5], st)
            : aabr_conv_forward((const float *)p[0], o.i32[0], o.i64[0], (float *)p[1], o.i32[1], o.i64[1],
                                (const int32_t *)p[2], o.i32[2], (const float *)p[3], (const float *)p[4], o.i32[3],
                                (float *)p[5], st);
    break;
  case AABR_PLAN_CONV_WIDE:
    if (o.i32[5] == 1) { // the write-out forms the BACKWARD statistics of the BatchNorm whose d_out it produces
      if (bf) {          // p7 the BatchNorm's input, p9 its stored output (the sign), p8 save_mean; p3 = residual or NULL
        rc = aabr_conv_forward_wide_bf16_res((const uint16_t *)p[0], o.i32[0], o.i64[0], (uint16_t *)p[1], o.i32[1],
                                             o.i64[1], (const int32_t *)p[2], o.i32[4], o.i32[2], (const float *)p[4],
                                             o.i32[3], (const uint16_t *)p[5], (const uint16_t *)p[3], (double *)p[6],
                                             (const uint16_t *)p[7], (const uint16_t *)p[9], (const float *)p[8], o.f32[0],
                                             st);
        break;
      }
      rc = aabr_conv_forward_wide_bwd_stats((const float *)p[0], o.i32[0], o.i64[0], (float *)p[1], o.i32[1], o.i64[1],
                                            (const int32_t *)p[2], o.i32[4], o.i32[2], (const float *)p[4], o.i32[3],
                                            (const float *)p[5], (const float *)p[3], (double *)p[6],
                                            (const float *)p[7], (const float *)p[8], (const float *)p[9],
                                            (const float *)p[10], (const float *)p[11], o.f32[0], st);
      break;
    }
    rc = bf ? aabr_conv_forward_wide_bf16_res((const uint16_t *)p[0], o.i32[0], o.i64[0], (uint16_t *)p[1], o.i32[1],
                                              o.i64[1], (const int32_t *)p[2], o.i32[4], o.i32[2], (const float *)p[4],
                                              o.i32[3], (const uint16_t *)p[5], (const uint16_t *)p[3], (double *)p[6],
                                              nullptr, nullptr, nullptr, 0.0f, st)
            : aabr_conv_forward_wide_stats((const float *)p[0], o.i32[0], o.i64[0], (float *)p[1], o.i32[1], o.i64[1],
                                           (const int32_t *)p[2], o.i32[4], o.i32[2], (const float *)p[4], o.i32[3],
                                           (const float *)p[5], (const float *)p[3], (double *)p[6], st);
    break;
  case AABR_PLAN_CONV_WIDE_SPLIT:
    rc = bf ? aabr_conv_forward_wide_split_bf16_res((const uint16_t *)p[0], o.i32[0], o.i64[0], (uint16_t *)p[1],
                                                    o.i32[1], o.i64[1], (const int32_t *)p[2], o.i32[4], o.i32[2],
                                                    (const float *)p[4], o.i32[3], (const uint16_t *)p[5], o.i32[5],
                                                    (float *)p[6], (const uint16_t *)p[3], st)
            : aabr_conv_forward_wide_split((const float *)p[0], o.i32[0], o.i64[0], (float *)p[1], o.i32[1], o.i64[1],
                                           (const int32_t *)p[2], o.i32[4], o.i32[2], (const float *)p[4], o.i32[3],
                                           (const float *)p[5], (const float *)p[3], o.i32[5], (float *)p[6], st);
    break;
  case AABR_PLAN_CONV_NARROW:
    if (bf && o.i32[5] == 1) {
      rc = aabr_conv_forward_narrow_bf16_bwd_stats((const uint16_t *)p[0], o.i64[0], (uint16_t *)p[1], o.i64[1],
                                                   (const int32_t *)p[2], o.i32[2], (const float *)p[3], (const float *)p[4],
                                                   o.i32[3], (double *)p[6], (const uint16_t *)p[7], (const uint16_t *)p[9],
                                                   (const float *)p[8], o.f32[0], st);
      break;
    }
    if (bf && p[6]) {
      rc = aabr_conv_forward_narrow_bf16_stats((const uint16_t *)p[0], o.i64[0], (uint16_t *)p[1], o.i64[1],
                                               (const int32_t *)p[2], o.i32[2], (const float *)p[3], (const float *)p[4],
                                               o.i32[3], (double *)p[6], st);
      break;
    }
    rc = bf ? aabr_conv_forward_narrow_bf16((const uint16_t *)p[0], o.i64[0], (uint16_t *)p[1], o.i64[1],
                                            (const int32_t *)p[2], o.i32[2], (const float *)p[3], (const float *)p[4],
                                            o.i32[3], st)
            : aabr_conv_forward_narrow((const float *)p[0], o.i64[0], (float *)p[1], o.i64[1], (const int32_t *)p[2],
                                       o.i32[2], (const float *)p[3], (const float *)p[4], o.i32[3], st);
    break;
  case AABR_PLAN_CONV_DW:
    rc = bf ? aabr_conv_backward_weight_bf16((const uint16_t *)p[0], o.i32[0], (const uint16_t *)p[1], o.i32[1],
                                             o.i64[0], (const int32_t *)p[2], o.i32[2], o.i64[1], (float *)p[3],
                                             (float *)p[4], (float *)p[5], st)
            : aabr_conv_backward_weight((const float *)p[0], o.i32[0], (const float *)p[1], o.i32[1], o.i64[0],
                                        (const int32_t *)p[2], o.i32[2], o.i64[1], (float *)p[3], (float *)p[4],
                                        (float *)p[5], st);
    break;
  case AABR_PLAN_BN_FWD:
    if (p[9]) { // the statistics' partial sums came with the producing convolution (p9, i32[2] of them)
      AABR_CHECK_ARG(o.i32[1], "precomputed statistics in training mode only");
      rc = bf ? aabr_bn_forward_parts_bf16((const uint16_t *)p[0], (uint16_t *)p[1], o.i64[0], o.i32[0], (float *)p[2],
                                           (float *)p[3], (float *)p[4], (float *)p[5], (const float *)p[6],
                                           (const float *)p[7], o.f32[0], o.f32[1], o.f32[2], (const double *)p[9],
                                           o.i32[2], (float *)p[8], st)
              : aabr_bn_forward_parts((const float *)p[0], (float *)p[1], o.i64[0], o.i32[0], (float *)p[2],
                                      (float *)p[3], (float *)p[4], (float *)p[5], (const float *)p[6],
                                      (const float *)p[7], o.f32[0], o.f32[1], o.f32[2], (const double *)p[9], o.i32[2],
                                      (float *)p[8], st);
      break;
    }
    rc = bf ? aabr_bn_forward_bf16((const uint16_t *)p[0], (uint16_t *)p[1], o.i64[0], o.i32[0], (float *)p[2],
                                   (float *)p[3], (float *)p[4], (float *)p[5], (const float *)p[6],
                                   (const float *)p[7], o.f32[0], o.f32[1], o.i32[1], o.f32[2], (float *)p[8], st)
            : aabr_bn_forward((const float *)p[0], (float *)p[1], o.i64[0], o.i32[0], (float *)p[2], (float *)p[3],
                              (float *)p[4], (float *)p[5], (const float *)p[6], (const float *)p[7], o.f32[0],
                              o.f32[1], o.i32[1], o.f32[2], (float *)p[8], st);
    break;
  case AABR_PLAN_BN_BWD:
    if (bf && p[11]) { // bf16 storage with the gradient sum folded in (p11); statistics given (i64[1], i32[1]) or its own
      rc = aabr_bn_backward_add_bf16((const uint16_t *)p[0], (uint16_t *)p[1], (const uint16_t *)p[2],
                                     (const uint16_t *)p[3], o.i64[0], o.i32[0], (const float *)p[4], (const float *)p[5],
                                     (const float *)p[6], (const float *)p[10], (float *)p[7], (float *)p[8], o.f32[2],
                                     (const double *)(uintptr_t)o.i64[1], o.i32[1], (float *)p[9], (const uint16_t *)p[11],
                                     st);
      break;
    }
    if (o.i64[1]) { // the statistics' partial sums came with the producing convolution (i64[1] = address, i32[1] of them)
      if (bf) {
        rc = aabr_bn_backward_parts_bf16((const uint16_t *)p[0], (uint16_t *)p[1], (const uint16_t *)p[2],
                                         (const uint16_t *)p[3], o.i64[0], o.i32[0], (const float *)p[4], (const float *)p[5],
                                         (const float *)p[6], (const float *)p[10], (float *)p[7], (float *)p[8], o.f32[2],
                                         (const double *)(uintptr_t)o.i64[1], o.i32[1], (float *)p[9], st);
        break;
      }
      rc = aabr_bn_backward_parts((const float *)p[0], (float *)p[1], (const float *)p[2], (const float *)p[3], o.i64[0],
                                  o.i32[0], (const float *)p[4], (const float *)p[5], (const float *)p[6],
                                  (const float *)p[10], (float *)p[7], (float *)p[8], o.f32[2],
                                  (const double *)(uintptr_t)o.i64[1], o.i32[1], (float *)p[9], (const float *)p[11], st);
      break;
    }
    rc = bf ? aabr_bn_backward_bf16((const uint16_t *)p[0], (uint16_t *)p[1], (const uint16_t *)p[2],
                                    (const uint16_t *)p[3], o.i64[0], o.i32[0], (const float *)p[4],
                                    (const float *)p[5], (const float *)p[6], (const float *)p[10], (float *)p[7],
                                    (float *)p[8], o.f32[2], (float *)p[9], st)
            : aabr_bn_backward_add((const float *)p[0], (float *)p[1], (const float *)p[2], (const float *)p[3],
                                   o.i64[0], o.i32[0], (const float *)p[4], (const float *)p[5], (const float *)p[6],
                                   (const float *)p[10], (float *)p[7], (float *)p[8], o.f32[2], (float *)p[9],
                                   (const float *)p[11], st);
    break;
  case AABR_PLAN_ADD:
    rc = aabr_add(p[0], p[1], p[2], o.i64[0], bf ? 1 : 0, st);
    break;
  case AABR_PLAN_CAST:
    rc = aabr_cast_storage(p[0], p[1], o.i64[0], (o.flags & AABR_PLAN_TO_BF16) ? 1 : 0, st);
    break;
  default:
    aabr::set_error("aabr_plan_run: a record has unknown kind %d", o.kind);
    return AABR_EINVAL;
  }
  return rc;
}

// `hold`: this list is a PART of a pass whose next part follows on the same thread -- the side stream is left unjoined at
// the end (its launches keep running beside the next part's); the part that ends the pass (hold = 0) joins.
static int plan_run_part(const AabrPlanOp *ops, int n_ops, void *st_, int hold) {
  AABR_CHECK_ARG(n_ops >= 0 && (ops || n_ops == 0), "bad plan");
  hipStream_t main_stream = (hipStream_t)st_;
  size_t n_events = 0, n_pending = g_side.pending;   // events used by this call / side launches not yet joined
  g_side.pending = 0;
  // AABR_PLAN_SIDE records can be handed to the second stream in BATCHES (PLAN_SIDE_BATCH knob): one event on the
  // caller's stream per batch instead of one per record (an event is a marker packet in the queue, ~5 us of it).
  // Measured on the bench step: 13.47 ms with an event per record, 13.60 in batches of 4, 13.68 of 8 -- starting the
  // weight gradients late costs more overlap than the markers cost queue time, so the default stays 1.
  int batch = knob(K_PLAN_SIDE_BATCH);
  if (batch < 1 || batch > 64) batch = 1;
  std::vector<int> deferred;
  auto fail = [&](int rc) {               // the failing entry point has set the error text; never leave the side stream unjoined
    if (n_pending) hipStreamSynchronize(g_side.stream);
    return rc;
  };
  // the caller's stream waits for everything issued on the second stream so far; every error leaves through fail()
  auto join_side = [&]() -> int {
    hipEvent_t e = g_side.get(n_events++);
    if (e == nullptr) { aabr::set_error("aabr_plan_run: event creation failed"); return AABR_ELAUNCH; }
    hipError_t he = hipEventRecord(e, g_side.stream);
    if (he == hipSuccess) he = hipStreamWaitEvent(main_stream, e, 0);
    if (he != hipSuccess) { aabr::set_error("aabr_plan_run: HIP error %s", hipGetErrorString(he)); return AABR_ELAUNCH; }
    return AABR_OK;
  };
  auto flush = [&]() -> int {
    if (deferred.empty()) return AABR_OK;
    if (!g_side.stream) {
      // PLAN_SIDE_PRIO knob (experiment): 1 = lowest queue priority for the second stream (its launches are not on the
      // critical chain), 2 = highest
      int lo = 0, hi = 0;
      const int v = knob(K_PLAN_SIDE_PRIO);
      if ((v == 1 || v == 2) && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess)
        AABR_CHECK_HIP(hipStreamCreateWithPriority(&g_side.stream, hipStreamNonBlocking, v == 1 ? lo : hi));
      else
        AABR_CHECK_HIP(hipStreamCreateWithFlags(&g_side.stream, hipStreamNonBlocking));
    }
    hipEvent_t e = g_side.get(n_events++);
    AABR_CHECK_ARG(e != nullptr, "event creation failed");
    AABR_CHECK_HIP(hipEventRecord(e, main_stream));
    AABR_CHECK_HIP(hipStreamWaitEvent(g_side.stream, e, 0));
    for (int idx : deferred) {
      ++n_pending;
      const int rc = plan_dispatch(ops[idx], (void *)g_side.stream);
      if (rc != AABR_OK) { deferred.clear(); return rc; }
    }
    deferred.clear();
    return AABR_OK;
  };
  for (int j = 0; j < n_ops; ++j) {
    const AabrPlanOp &o = ops[j];
    if (o.flags & AABR_PLAN_JOIN) {        // this record reads what the side stream produced
      const int rc = flush();
      if (rc != AABR_OK) return fail(rc);
      if (n_pending) {
        const int jr = join_side();
        if (jr != AABR_OK) return fail(jr);
        n_pending = 0;
      }
    }
    if (o.flags & AABR_PLAN_SIDE) {
      deferred.push_back(j);
      if ((int)deferred.size() >= batch) {
        const int rc = flush();
        if (rc != AABR_OK) return fail(rc);
      }
      continue;
    }
    if (o.kind == AABR_PLAN_CAST) {        // a run of casts in one direction: one launch (plan_cast_run)
      // The records of a list mean "one after the other": a pass whose arena is packed by liveness hands a cast's input
      // on to a LATER record's output.  A record joins the run only if what it writes overlaps nothing an earlier
      // record of the run reads or writes.
      const size_t es_in = (o.flags & AABR_PLAN_TO_BF16) ? 4 : 2, es_out = (o.flags & AABR_PLAN_TO_BF16) ? 2 : 4;
      auto overlaps = [](const void *a, size_t na, const void *b, size_t nb) {
        const uintptr_t a0 = (uintptr_t)a, b0 = (uintptr_t)b;
        return a0 < b0 + nb && b0 < a0 + na;
      };
      int m = 1;
      while (j + m < n_ops && m < kCastJobsMax && ops[j + m].kind == AABR_PLAN_CAST &&
             !(ops[j + m].flags & (AABR_PLAN_SIDE | AABR_PLAN_JOIN)) &&
             ((ops[j + m].flags ^ o.flags) & AABR_PLAN_TO_BF16) == 0) {
        const AabrPlanOp &c = ops[j + m];
        bool clash = false;
        for (int q = 0; q < m && !clash; ++q) {
          const AabrPlanOp &e = ops[j + q];
          clash = overlaps(c.p[1], (size_t)c.i64[0] * es_out, e.p[0], (size_t)e.i64[0] * es_in) ||
                  overlaps(c.p[1], (size_t)c.i64[0] * es_out, e.p[1], (size_t)e.i64[0] * es_out) ||
                  overlaps(c.p[0], (size_t)c.i64[0] * es_in, e.p[1], (size_t)e.i64[0] * es_out);
        }
        if (clash) break;
        ++m;
      }
      if (m > 1) {
        const int rc = plan_cast_run(ops + j, m, st_);
        if (rc != AABR_OK) return fail(rc);
        j += m - 1;
        continue;
      }
    }
    const int rc = plan_dispatch(o, st_);
    if (rc != AABR_OK) return fail(rc);
  }
  {
    const int rc = flush();
    if (rc != AABR_OK) return fail(rc);
  }
  if (hold) {
    g_side.pending = n_pending;
    return AABR_OK;
  }
  if (n_pending) {
    const int jr = join_side();
    if (jr != AABR_OK) return fail(jr);
  }
  return AABR_OK;
}

static bool launcher_has_work();
extern "C" int aabr_plan_run(const AabrPlanOp *ops, int n_ops, void *st_) {
  // the second stream's join state and event pool are shared with the launcher thread: a pass handed over in parts must
  // have been drained before another list is run directly
  AABR_CHECK_ARG(!launcher_has_work(), "parts submitted with aabr_plan_submit are still queued: call aabr_plan_drain first");
  return plan_run_part(ops, n_ops, st_, 0);
}

// ---- pipelined submission ---------------------------------------------------------------------------------------------
// A pass's list costs the host twice: the caller fills the records (interpreter time) and this library issues their
// launches (3.4 us per launch and 10 us per cross-stream event pair on this stack, tools/dev/launch_cost.hip: 1.7 ms
// per training step, ~450 launches).  aabr_plan_submit hands a PART of the list to a launcher thread and returns; the
// caller fills the next part meanwhile.  Parts are issued strictly in submission order, each with the semantics of
// aabr_plan_run except that only the LAST part (hold_side = 0) joins the second stream.  aabr_plan_drain returns when
// every submitted part has been issued (not when the device has run it) with the first failing part's code: call it
// before anything else is enqueued on the streams involved.  One launcher per process; the records are copied.
namespace {
struct Launcher {
  struct Job { std::vector<AabrPlanOp> ops; void *st; int hold; int dev; };
  std::mutex m;
  std::condition_variable cv_work, cv_idle;
  std::deque<Job> q;
  bool busy = false, started = false;
  int rc = AABR_OK;          // first failure since the last drain: STICKY -- every part behind it, of this pass or a later
  std::string err;           // one, is dropped until aabr_plan_drain has reported it
  const char *variant = "";  // aabr_conv_last_variant() as the launcher thread saw it after its last part
  long long busy_ns = 0, jobs = 0, sleeps = 0;   // (tools: time spent issuing, parts issued, times the queue ran dry)
  void run() {
    int cur_dev = -1;
    for (;;) {
      Job j;
      {
        std::unique_lock<std::mutex> l(m);
        if (q.empty()) ++sleeps;
        cv_work.wait(l, [&] { return !q.empty(); });
        j = std::move(q.front());
        q.pop_front();
        busy = true;
      }
      int r = AABR_OK;
      bool skip;
      const auto t0 = std::chrono::steady_clock::now();
      { std::lock_guard<std::mutex> l(m); skip = rc != AABR_OK; }   // after a failure the rest of the pass is dropped
      if (!skip) {
        if (j.dev != cur_dev) { hipSetDevice(j.dev); cur_dev = j.dev; }
        r = plan_run_part(j.ops.data(), (int)j.ops.size(), j.st, j.hold);
      } else if (!j.hold && g_side.pending) {                        // ... but its side stream is never left unjoined
        hipStreamSynchronize(g_side.stream);
        g_side.pending = 0;
      }
      {
        std::lock_guard<std::mutex> l(m);
        if (r != AABR_OK && rc == AABR_OK) { rc = r; err = aabr_last_error(); }
        if (!skip) variant = aabr_conv_last_variant();
        busy_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
        ++jobs;
        busy = false;
        if (q.empty()) cv_idle.notify_all();
      }
    }
  }
};
Launcher *g_launcher = nullptr;           // created on first use and never destroyed (its thread outlives static destruction)
std::once_flag g_launcher_once;
} // namespace
static bool launcher_has_work() {
  if (!g_launcher) return false;
  std::lock_guard<std::mutex> l(g_launcher->m);
  return !g_launcher->q.empty() || g_launcher->busy;
}
namespace aabr { extern thread_local const char *g_last_variant; }   // conv.hip

extern "C" int aabr_plan_submit(const AabrPlanOp *ops, int n_ops, void *st, int hold_side) {
  AABR_CHECK_ARG(n_ops >= 0 && (ops || n_ops == 0), "bad plan");
  int dev = 0;
  AABR_CHECK_HIP(hipGetDevice(&dev));
  std::call_once(g_launcher_once, [] {
    g_launcher = new Launcher();
    std::thread([] { g_launcher->run(); }).detach();
  });
  Launcher::Job j;
  j.ops.assign(ops, ops + n_ops);
  j.st = st;
  j.hold = hold_side;
  j.dev = dev;
  {
    std::lock_guard<std::mutex> l(g_launcher->m);
    g_launcher->q.push_back(std::move(j));
  }
  g_launcher->cv_work.notify_one();
  return AABR_OK;
}

extern "C" void aabr_plan_launcher_stats(int64_t *busy_ns, int64_t *parts, int64_t *sleeps) {
  *busy_ns = *parts = *sleeps = 0;
  if (!g_launcher) return;
  std::lock_guard<std::mutex> l(g_launcher->m);
  *busy_ns = g_launcher->busy_ns; *parts = g_launcher->jobs; *sleeps = g_launcher->sleeps;
}

extern "C" int aabr_plan_drain(void) {
  if (!g_launcher) return AABR_OK;
  std::unique_lock<std::mutex> l(g_launcher->m);
  g_launcher->cv_idle.wait(l, [&] { return g_launcher->q.empty() && !g_launcher->busy; });
  const int rc = g_launcher->rc;
  if (g_launcher->variant[0]) aabr::g_last_variant = g_launcher->variant;   // what the drained parts dispatched last
  if (rc != AABR_OK) {
    aabr::set_error("%s", g_launcher->err.c_str());
    g_launcher->rc = AABR_OK;
    g_launcher->err.clear();
  }
  return rc;
}

// ---- geometry plan: the rule-book builders of a pass as one list ------------------------------------------------
// The geometry pre-pass of a network (sparseconvnet/fpn_net.py: every rule table, block stream and offset-pair list
// of a pass, ~250 launches) went out one ctypes call at a time: 2.8 ms of interpreter time per step.  A record names
// one of the builders declared in include/aabr_hip.h and carries its arguments; nothing is computed differently.
static_assert(sizeof(AabrGeomOp) == 144, "AabrGeomOp layout is part of the C ABI");

// brick levels in a record: dims packed sbx | sby << 16 | sbz << 32 | nb << 48; one allocation = directory, then bricks
static inline void unpack_dims(int64_t v, int32_t dims[4]) {
  for (int i = 0; i < 4; ++i) dims[i] = (int32_t)(((uint64_t)v >> (16 * i)) & 0xffffu);
}
static inline int64_t dir_bytes(const int32_t dims[4]) { return (int64_t)dims[0] * dims[1] * dims[2] * dims[3] * 16; }

extern "C" int aabr_geom_run(const AabrGeomOp *ops, int n_ops, void *st) {
  AABR_CHECK_ARG(n_ops >= 0 && (ops || n_ops == 0), "bad plan");
  // The stream builders (tile blocks, wide tile blocks, pair lists) read finished gather tables and feed nothing else in
  // the list: they are collected here and issued behind the last record as ONE launch per kind over all the books of the
  // list (common.h StreamJobs) instead of one to two launches per book -- same words, a third of a step's launches gone.
  std::vector<StreamJob> tile_jobs, wide_jobs, pair_jobs;
  std::vector<SampleJob> sample_jobs;      // the per-sample row offsets of the levels built in this list: read by the host only
  const bool per_book = knob(K_GEOM_JOBS) == 0;     // A/B knob: 0 = issue every builder where it stands, book by book (round 5)
  for (int j = 0; j < n_ops; ++j) {
    const AabrGeomOp &o = ops[j];
    void *const *p = o.p;
    int rc = AABR_OK;
    switch (o.kind) {
    case AABR_GEOM_SUBM_TABLE:
      rc = aabr_submanifold_table((const int32_t *)p[0], o.i64[0], (const uint64_t *)p[1], o.i64[1], &o.i32[0],
                                  (int32_t *)p[2], (int32_t *)p[3], st);
      break;
    case AABR_GEOM_CONV_TABLES:
      rc = aabr_convolution_tables2((const int32_t *)p[0], o.i64[0], (const uint64_t *)p[1], o.i64[1],
                                    (const int32_t *)p[2], o.i64[2], (const uint64_t *)p[3], o.i64[3],
                                    &o.i32[0], &o.i32[3], &o.i32[6], (int32_t *)p[4], (int32_t *)p[5], (int32_t *)p[6],
                                    (int32_t *)p[7], st);
      break;
    case AABR_GEOM_TILE_BLOCKS:
      AABR_CHECK_ARG(o.i64[0] >= 0 && o.i64[0] < (1ll << 25) && o.i32[0] > 0 && o.i32[0] <= 4096, "tile blocks: bad sizes");
      if (o.i64[0] == 0) break;
      AABR_CHECK_ARG(p[0] && p[1], "tile blocks: null pointer");
      if (per_book) { rc = aabr_build_tile_blocks((const int32_t *)p[0], o.i64[0], o.i32[0], (int32_t *)p[1], st); break; }
      tile_jobs.push_back(StreamJob{(const int32_t *)p[0], nullptr, (int32_t *)p[1], o.i64[0], o.i32[0], 0});
      break;
    case AABR_GEOM_WIDE_BLOCKS:
      AABR_CHECK_ARG(o.i64[0] >= 0 && o.i32[0] > 0 && o.i32[0] <= 63, "wide blocks: bad sizes (vol <= 63)");
      AABR_CHECK_ARG(o.i32[1] >= 16 && o.i32[1] <= 240 && (o.i32[1] & 15) == 0, "wide blocks: tile_rows a multiple of 16, <= 240");
      if (o.i64[0] == 0) break;
      AABR_CHECK_ARG(p[0] && p[1], "wide blocks: null pointer");
      if (per_book) { rc = aabr_build_wide_blocks((const int32_t *)p[0], o.i64[0], o.i32[0], o.i32[1], (int32_t *)p[1], st); break; }
      wide_jobs.push_back(StreamJob{(const int32_t *)p[0], nullptr, (int32_t *)p[1], o.i64[0], o.i32[0], o.i32[1]});
      break;
    case AABR_GEOM_OFFSET_PAIRS:
      if (o.i64[0] <= 0 || per_book) {   // an empty book: its header is cleared at once
        rc = aabr_build_offset_pairs((const int32_t *)p[0], (const int32_t *)p[1], o.i64[0], o.i32[0], (int32_t *)p[2], st);
        break;
      }
      AABR_CHECK_ARG(o.i32[0] > 0 && o.i32[0] <= 65535 && p[0] && p[1] && p[2], "pair list: bad arguments");
      pair_jobs.push_back(StreamJob{(const int32_t *)p[0], (const int32_t *)p[1], (int32_t *)p[2], o.i64[0], o.i32[0], 0});
      break;
    case AABR_GEOM_CONV_SITES:
      rc = aabr_convolution_sites((const int32_t *)p[0], o.i64[0], &o.i32[0], &o.i32[3], &o.i32[6], (uint64_t *)p[1],
                                  o.i64[1], (int32_t *)p[2], (int32_t *)p[3], (int32_t *)p[4], st);
      break;
    case AABR_GEOM_SAMPLE_OFFSETS:
      AABR_CHECK_ARG(p[0] && p[1] && p[2] && o.i64[0] >= 0 && o.i32[0] >= 1 && o.i32[0] <= 4096, "sample offsets: bad arguments");
      if (per_book) { rc = aabr_sample_offsets((const int32_t *)p[0], (const int32_t *)p[1], o.i64[0], o.i32[0], (int32_t *)p[2], st); break; }
      sample_jobs.push_back(SampleJob{(const int32_t *)p[0], (const int32_t *)p[1], (int32_t *)p[2], o.i64[0], o.i32[0]});
      break;
    case AABR_GEOM_BRICK_BUILD: {
      int32_t dims[4];
      unpack_dims(o.i64[1], dims);
      char *lvl = (char *)p[2];
      rc = aabr_brick_build((const int32_t *)p[0], o.i64[0], (const int32_t *)p[1], &o.i32[0], &o.i32[3], &o.i32[6], dims,
                            lvl, lvl + dir_bytes(dims), o.i64[2], (int32_t *)p[3], (int32_t *)p[4], o.i64[3],
                            (int32_t *)p[5], (int32_t *)p[6], o.i32[9], st);
      break;
    }
    case AABR_GEOM_BRICK_SUBM: {
      int32_t dims[4];
      unpack_dims(o.i64[1], dims);
      const char *lvl = (const char *)p[1];
      rc = aabr_brick_submanifold_table((const int32_t *)p[0], o.i64[0], dims, lvl, lvl + dir_bytes(dims), &o.i32[0],
                                        (int32_t *)p[2], (int32_t *)p[3], st);
      break;
    }
    case AABR_GEOM_BRICK_TABLES: {
      int32_t di[4], dout[4];
      unpack_dims(o.i64[2], di);
      unpack_dims(o.i64[3], dout);
      const char *li = (const char *)p[1], *lo = (const char *)p[3];
      rc = aabr_brick_convolution_tables((const int32_t *)p[0], o.i64[0], di, li, li + dir_bytes(di), (const int32_t *)p[2],
                                         o.i64[1], dout, lo, lo + dir_bytes(dout), &o.i32[0], &o.i32[3], &o.i32[6],
                                         (int32_t *)p[4], (int32_t *)p[5], (int32_t *)p[6], (int32_t *)p[7], st);
      break;
    }
    default:
      aabr::set_error("aabr_geom_run: op %d has unknown kind %d", j, o.kind);
      return AABR_EINVAL;
    }
    if (rc != AABR_OK) return rc;
  }
  int rc = AABR_OK;
  if (!sample_jobs.empty()) rc = launch_sample_offsets_jobs(sample_jobs.data(), (int)sample_jobs.size(), (hipStream_t)st);
  if (rc == AABR_OK && !wide_jobs.empty()) rc = launch_wide_blocks_jobs(wide_jobs.data(), (int)wide_jobs.size(), (hipStream_t)st);
  if (rc == AABR_OK && !tile_jobs.empty()) rc = launch_tile_blocks_jobs(tile_jobs.data(), (int)tile_jobs.size(), (hipStream_t)st);
  if (rc == AABR_OK && !pair_jobs.empty()) rc = launch_offset_pairs_jobs(pair_jobs.data(), (int)pair_jobs.size(), (hipStream_t)st);
  return rc;
}

// ---- mailbox: a small result handed to the host WITHOUT a stream / event wait ----------------------------------------
// The host side of the path reads a few small counts back per step (sites per grid, proposals kept per scene).  With
// hipStreamSynchronize / hipEventSynchronize -- and even with hipEventQuery polling -- on the stream that produced
// them, such a read was measured to return only when the OTHER streams of the process had drained too: the proposal
// stage's read, complete on the device at 8.3 ms of the bench step, came back at 13.6 ms, behind the last kernel of the
// backward pass (tools/tools_step_timeline.py, profiles/r03_step_timeline.txt).  A mailbox is host memory the device
// writes directly (hipHostMalloc, coherent): the posting kernel copies the payload, fences at system scope and then
// stores the caller's sequence number; the host spins on that word.  No HIP call on the waiting side.
namespace aabr {
__global__ void k_mailbox_post(const uint32_t *__restrict__ src, int words, uint32_t *box, uint32_t seq) {
  for (int i = threadIdx.x; i < words; i += blockDim.x) box[2 + i] = src[i];
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    box[1] = (uint32_t)wall_clock64();     // device real-time counter (100 MHz), low word: when the post ran (tools/)
    __threadfence_system();
    __hip_atomic_store(box, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
} // namespace aabr

// capacity of every live mailbox: aabr_mailbox_post refuses a payload larger than what aabr_mailbox_create allocated
// (the posting kernel writes into host memory: an oversized post would run past the block)
#include <mutex>
#include <unordered_map>
static std::mutex g_mb_mu;
static std::unordered_map<void *, int64_t> g_mb_cap;

extern "C" int aabr_mailbox_create(int64_t payload_bytes, void **box) {
  AABR_CHECK_ARG(box && payload_bytes > 0 && payload_bytes <= (1 << 20) && (payload_bytes & 3) == 0, "payload: 4..1 MiB, % 4");
  void *p = nullptr;
  AABR_CHECK_HIP(hipHostMalloc(&p, (size_t)payload_bytes + 8, hipHostMallocCoherent | hipHostMallocMapped));
  memset(p, 0, (size_t)payload_bytes + 8);
  {
    std::lock_guard<std::mutex> lk(g_mb_mu);
    g_mb_cap[p] = payload_bytes;
  }
  *box = p;
  return AABR_OK;
}

extern "C" int aabr_mailbox_destroy(void *box) {
  if (box) {
    {
      std::lock_guard<std::mutex> lk(g_mb_mu);
      g_mb_cap.erase(box);
    }
    AABR_CHECK_HIP(hipHostFree(box));
  }
  return AABR_OK;
}

extern "C" int aabr_mailbox_post(const void *src, int64_t bytes, void *box, uint32_t seq, void *stream_) {
  AABR_CHECK_ARG(src && box && bytes > 0 && (bytes & 3) == 0 && ((uintptr_t)src & 3) == 0, "payload");
  AABR_CHECK_ARG(seq != 0, "sequence numbers start at 1 (a fresh mailbox reads 0)");
  {
    std::lock_guard<std::mutex> lk(g_mb_mu);
    auto it = g_mb_cap.find(box);
    AABR_CHECK_ARG(it != g_mb_cap.end(), "not a mailbox of aabr_mailbox_create (or already destroyed)");
    AABR_CHECK_ARG(bytes <= it->second, "payload larger than the mailbox was created for");
  }
  void *dbox = nullptr;
  AABR_CHECK_HIP(hipHostGetDevicePointer(&dbox, box, 0));
  hipLaunchKernelGGL(k_mailbox_post, dim3(1), dim3(256), 0, (hipStream_t)stream_, (const uint32_t *)src, (int)(bytes / 4),
                     (uint32_t *)dbox, seq);
  AABR_CHECK_LAUNCH();
  return AABR_OK;
}
