// Compiles the PRODUCT's device arithmetic (automatic-as-built-reconstruction_amd/csrc/iou_math.h) for the host so the
// exact statement sequence the HIP kernels run can be compared with the oracle on the CPU.
#include <stdint.h>
#include "../automatic-as-built-reconstruction_amd/csrc/iou_math.h"
extern "C" void host_iou_eval(const float *boxes, int64_t N, const float *query, int64_t K, int criterion,
                              float *iou) {
  for (int64_t n = 0; n < N; ++n)
    for (int64_t k = 0; k < K; ++k)
      iou[n * K + k] = aabr_iou::iou_eval_entry(boxes + 5 * n, query + 5 * k, criterion);
}
extern "C" void host_clip_iou(const float *boxes5, int64_t N, double *iou) {
  for (int64_t n = 0; n < N; ++n)
    for (int64_t k = 0; k < N; ++k) iou[n * N + k] = aabr_iou::clip_iou_exact(boxes5 + 5 * n, boxes5 + 5 * k);
}
