// Dev probe: cost of one s_barrier step in a workgroup of NW waves, with and without an LDS hand-off per step.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE>
__global__ void k(int n, float *out, long long *clk) {
  __shared__ float buf[2][4096];
  const int wave = threadIdx.x >> 6;
  float acc = 0;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int s = 0; s < n; ++s) {
    if (MODE >= 1) {
      if (wave >= (int)(blockDim.x >> 7)) buf[(s + 1) & 1][threadIdx.x] = (float)s;       // "loaders" write next buffer
      else acc += buf[s & 1][threadIdx.x];                                                  // "consumers" read this one
    }
    if (MODE >= 2) {  // some MFMA work in consumers
      if (wave < (int)(blockDim.x >> 7)) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        typedef __bf16 b8 __attribute__((ext_vector_type(8)));
        f4 c = {acc, 0, 0, 0}; b8 a = {}, b = {};
        for (int i = 0; i < 32; ++i) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
        acc += c[0];
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}
int main() {
  float *out; long long *clk; hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&clk, 256 * 8);
  const int n = 2000;
  for (int nt : {256, 512}) for (int mode = 0; mode < 3; ++mode) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(a);
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(nt), 0, 0, n, out, clk);
      if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(nt), 0, 0, n, out, clk);
      if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(nt), 0, 0, n, out, clk);
      hipEventRecord(b); hipEventSynchronize(b);
    }
    float ms; hipEventElapsedTime(&ms, a, b);
    long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
    printf("threads %d mode %d: %.3f us per step (wall), %.0f memtime ticks per step\n", nt, mode, ms * 1e3 / n, (double)c / n);
  }
  return 0;
}
