// Dev probe: host cost of the HIP launch paths on this stack (us per call, queue kept short by periodic syncs).
// build: hipcc --offload-arch=gfx950 -O2 -o /tmp/launch_cost tools/dev/launch_cost.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
struct Big { void *p[12]; int64_t a[4]; int i[6]; float f[4]; };
__global__ void k_small(float *p, int n) { if (n < 0) p[0] = 1.f; }
__global__ void k_big(Big b) { if (b.i[0] < 0) ((float *)b.p[0])[0] = 1.f; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  float *d; hipMalloc(&d, 4096);
  hipStream_t s, s2; hipStreamCreateWithFlags(&s, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  hipEvent_t ev[64]; for (auto &e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
  const int N = 4000;
  Big b{}; b.p[0] = d; b.i[0] = 1;
  for (int rep = 0; rep < 2; ++rep) {
    double t0 = now();
    for (int i = 0; i < N; ++i) { hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, s, d, 1); if ((i & 255) == 255) hipStreamSynchronize(s); }
    double t1 = now();
    for (int i = 0; i < N; ++i) { hipLaunchKernelGGL(k_big, dim3(1), dim3(64), 0, s, b); if ((i & 255) == 255) hipStreamSynchronize(s); }
    double t2 = now();
    void *args[] = {&d, (void *)&b.i[0]};
    for (int i = 0; i < N; ++i) { hipLaunchKernel((const void *)k_small, dim3(1), dim3(64), args, 0, s); if ((i & 255) == 255) hipStreamSynchronize(s); }
    double t3 = now();
    hipFunction_t f = nullptr;
    hipError_t e = hipGetFuncBySymbol(&f, (const void *)k_small);
    double t4 = t3, t5 = t3;
    if (e == hipSuccess && f) {
      t4 = now();
      for (int i = 0; i < N; ++i) { hipModuleLaunchKernel(f, 1, 1, 1, 64, 1, 1, 0, s, args, nullptr); if ((i & 255) == 255) hipStreamSynchronize(s); }
      t5 = now();
    }
    double t6 = now();
    for (int i = 0; i < N; ++i) { hipEventRecord(ev[i & 63], s); if ((i & 255) == 255) hipStreamSynchronize(s); }
    double t7 = now();
    for (int i = 0; i < N; ++i) { hipEventRecord(ev[i & 63], s); hipStreamWaitEvent(s2, ev[i & 63], 0); if ((i & 255) == 255) { hipStreamSynchronize(s); hipStreamSynchronize(s2); } }
    double t8 = now();
    for (int i = 0; i < N; ++i) { hipMemsetAsync(d, 0, 1024, s); if ((i & 255) == 255) hipStreamSynchronize(s); }
    double t9 = now();
    // the same launches with the device busy (queue never empty): a long kernel first
    for (int i = 0; i < N; ++i) { hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, s, d, 1); hipGetLastError(); if ((i & 255) == 255) hipStreamSynchronize(s); }
    double t10 = now();
    if (rep == 1)
      printf("us per call: <<<small>>> %.2f  <<<176-byte struct>>> %.2f  hipLaunchKernel %.2f  hipModuleLaunchKernel %.2f (%s)\n"
             "  hipEventRecord %.2f  record+wait %.2f  hipMemsetAsync %.2f  <<<small>>> + hipGetLastError %.2f\n",
             (t1 - t0) / N * 1e6, (t2 - t1) / N * 1e6, (t3 - t2) / N * 1e6, (t5 - t4) / N * 1e6, f ? "ok" : "n/a",
             (t7 - t6) / N * 1e6, (t8 - t7) / N * 1e6, (t9 - t8) / N * 1e6, (t10 - t9) / N * 1e6);
  }
  return 0;
}
