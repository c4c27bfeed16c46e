// ref_kernels_harness.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Compiles the reference's OWN CPU kernels for the hot path from where they lie
// (nothing is copied; the four files below are #included by absolute path):
//
//   SCN/CPU/BatchNormalization.cpp:12-107   BatchNormalization_ForwardPass / _BackwardPass
//   SCN/CPU/IOLayers.cpp:11-47              InputLayer_ForwardPass / _BackwardPass
//   SCN/CPU/Convolution.cpp:8-43            rule_index_select / rule_index_add_
//   SCN/CPU/SparseToDense.cpp:7-33          SparseToDense_ForwardPass / _BackwardPass
//
// and exposes them through a C ABI so tests/test_oracle_ref_kernels.py can pin the
// numerical half of oracle/scn_oracle.c against the reference's own arithmetic.
//
// What is REAL here: every header is a real one (ATen from the installed torch, the
// reference's Metadata/32bits.h); the raw-pointer kernels above are instantiated
// unmodified.  Three of those files ALSO define `cpu_*<T, Dimension>(Metadata<Dimension>&, ...)`
// driver templates.  They are never instantiated here -- they need Metadata.h, which
// needs google sparsehash (absent; no stand-in may be written) -- so the only thing this
// harness supplies for them is two bare DECLARATIONS of the reference's own names, so that the
// uninstantiated driver templates parse: `template <Int> class Metadata;` (Metadata/Metadata.h:44)
// and `template <typename T> T *OptionalTensorData(at::Tensor);` (Metadata/Metadata.h:165).
// Neither is defined here and nothing that would need a definition is instantiated.  No
// substitute Metadata.h, no dense_hash_map stand-in.
//
// What is NOT the reference: the ~10-line per-offset driver loops in ref_rule_conv_fwd/bwd
// below.  The reference's drivers (CPU/Convolution.cpp:45-185, CPU/Deconvolution.cpp:7-77)
// take their rule lists from Metadata; here the rule lists come from the caller (the tests
// pass oracle/scn_oracle.c's rule books, whose geometry is pinned separately by
// libref_regions.so) and each loop body issues the same reference calls in the same order:
// rule_index_select -> at::matmul(_out) -> rule_index_add_.
//
// Still unpinned after this harness: the rule-book builders (sparsehash) and the NMS
// suppression loop (spconv, un-vendored) -- see DESIGN.md section 4.
//
// Output: oracle/_ref/libref_kernels.so (git-ignored; travels to the GPU box via gpurun).
#include <ATen/ATen.h>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>
#include "Metadata/32bits.h"

template <Int dimension> class Metadata;                     // declared only (Metadata.h:44)
template <typename T> T *OptionalTensorData(at::Tensor tensor);  // declared only (Metadata.h:165)

#include "CPU/BatchNormalization.cpp"
#include "CPU/IOLayers.cpp"
#include "CPU/Convolution.cpp"
#include "CPU/SparseToDense.cpp"

namespace {
at::Tensor wrap(float *p, int64_t r, int64_t c) {
  return at::from_blob(p, {r, c}, at::TensorOptions().dtype(at::kFloat));
}
}  // namespace

extern "C" {

// BatchNormalization_ForwardPass<float> (CPU/BatchNormalization.cpp:12-61); strides = nPlanes
// exactly as cpu_BatchNormalization_updateOutput passes them (:120-127)
void ref_bn_fwd(float *in, float *out, int nPlanes, long nActive, float *saveMean,
                float *saveInvStd, float *runningMean, float *runningVar, float *weight,
                float *bias, float eps, float momentum, int train, float leakiness) {
  BatchNormalization_ForwardPass<float>(in, out, nPlanes, nPlanes, nPlanes, (Int)nActive,
                                        saveMean, saveInvStd, runningMean, runningVar, weight,
                                        bias, eps, momentum, train != 0, leakiness);
}

// BatchNormalization_BackwardPass<float> (CPU/BatchNormalization.cpp:63-107); d_out is
// modified in place (activation mask), as in the reference
void ref_bn_bwd(float *in, float *d_in, float *out, float *d_out, int nPlanes, long nActive,
                float *saveMean, float *saveInvStd, float *runningMean, float *runningVar,
                float *weight, float *bias, float *d_weight, float *d_bias, float leakiness) {
  BatchNormalization_BackwardPass<float>(in, d_in, out, d_out, nPlanes, nPlanes, nPlanes,
                                         (Int)nActive, saveMean, saveInvStd, runningMean,
                                         runningVar, weight, bias, d_weight, d_bias, leakiness);
}

// InputLayer_ForwardPass<float> (CPU/IOLayers.cpp:11-29); `out` must be zeroed by the caller
void ref_input_layer_fwd(float *in, float *out, long nRows, int maxActive, int nPlanes,
                         int *rules, int average) {
  InputLayer_ForwardPass<float>(in, out, (Int)nRows, maxActive, nPlanes, rules, average != 0);
}

// InputLayer_BackwardPass<float> (CPU/IOLayers.cpp:30-47); `d_in` must be zeroed by the caller
void ref_input_layer_bwd(float *d_in, float *d_out, long nRows, int maxActive, int nPlanes,
                         int *rules, int average) {
  InputLayer_BackwardPass<float>(d_in, d_out, (Int)nRows, maxActive, nPlanes, rules,
                                 average != 0);
}

// Per-offset contraction out[R[:,oc]] += in[R[:,ic]] @ W[k], the loop body of
// cpu_Convolution_updateOutput (CPU/Convolution.cpp:66-79; ic=0, oc=1) and of
// cpu_Deconvolution_updateOutput (CPU/Deconvolution.cpp:27-40; ic=1, oc=0).
// rules: int32 [vol][cap][2], counts: int64 [vol].  `out` pre-filled by the caller
// (zero or bias).  Returns the reference's multiply-add count.
double ref_rule_conv_fwd(float *in, long nInRows, int nIn, float *out, long nOutRows, int nOut,
                         float *weight /*[vol][nIn][nOut]*/, int *rules, long *counts, long vol,
                         long cap, int deconv) {
  auto input_features = wrap(in, nInRows, nIn);
  auto output_features = wrap(out, nOutRows, nOut);
  auto W = at::from_blob(weight, {vol, 1, nIn, nOut}, at::TensorOptions().dtype(at::kFloat));
  double flops = 0;
  const Int groups = 1;
  int ic = deconv ? 1 : 0, oc = deconv ? 0 : 1;
  for (long i = 0; i < vol; ++i) {
    Int nRules = (Int)counts[i];
    Int *r = rules + i * cap * 2;
    if (nRules) {
      flops += (double)nRules * nIn * nOut * groups;
      auto w = W.select(0, i);
      auto input_rows = rule_index_select<float>(input_features, nRules, &r[ic], groups);
      auto output_rows = at::matmul(input_rows, w);
      rule_index_add_<float>(output_features, output_rows, nRules, &r[oc], groups);
    }
  }
  return flops;
}

// Loop body of cpu_Convolution_backward (CPU/Convolution.cpp:101-114) / cpu_Deconvolution_backward
// (CPU/Deconvolution.cpp:63-76).  d_in zeroed by the caller; dW[k] is overwritten per offset
// (matmul_out), offsets without rules keep the caller's value.
void ref_rule_conv_bwd(float *in, float *d_in, long nInRows, int nIn, float *d_out,
                       long nOutRows, int nOut, float *weight, float *d_weight, int *rules,
                       long *counts, long vol, long cap, int deconv) {
  auto input_features = wrap(in, nInRows, nIn);
  auto d_input_features = wrap(d_in, nInRows, nIn);
  auto d_output_features = wrap(d_out, nOutRows, nOut);
  auto W = at::from_blob(weight, {vol, 1, nIn, nOut}, at::TensorOptions().dtype(at::kFloat));
  auto dW = at::from_blob(d_weight, {vol, 1, nIn, nOut}, at::TensorOptions().dtype(at::kFloat));
  const Int groups = 1;
  int ic = deconv ? 1 : 0, oc = deconv ? 0 : 1;
  for (long i = 0; i < vol; ++i) {
    Int nRules = (Int)counts[i];
    Int *r = rules + i * cap * 2;
    if (nRules) {
      auto w = W.select(0, i);
      auto dw = dW.select(0, i);
      auto input_rows = rule_index_select<float>(input_features, nRules, &r[ic], groups);
      auto d_output_rows = rule_index_select<float>(d_output_features, nRules, &r[oc], groups);
      at::matmul_out(dw, input_rows.transpose(1, 2), d_output_rows);
      auto d_input_rows = at::matmul(d_output_rows, w.transpose(1, 2));
      rule_index_add_<float>(d_input_features, d_input_rows, nRules, &r[ic], groups);
    }
  }
}

// SparseToDense_ForwardPass<float> (CPU/SparseToDense.cpp:7-19) for ONE sample: rules =
// (input row, offset inside the sample's [nPlanes, spatialVolume] block) pairs
void ref_sparse_to_dense_fwd(float *in, float *out_sample, int nPlanes, long spatialVolume,
                             int *rules, int nHot) {
  SparseToDense_ForwardPass<float>(in, out_sample, nPlanes, (Int)spatialVolume, rules, nHot);
}
void ref_sparse_to_dense_bwd(float *d_in, float *d_out_sample, int nPlanes, long spatialVolume,
                             int *rules, int nHot) {
  SparseToDense_BackwardPass<float>(d_in, d_out_sample, nPlanes, (Int)spatialVolume, rules,
                                    nHot);
}
}
