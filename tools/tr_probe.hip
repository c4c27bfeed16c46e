// dev probe: what does ds_read_b64_tr_b16 deliver to each lane?  LDS image [16 rows][32 cols] of shorts,
// value = row*100 + col, row stride 32 shorts.  Lane l: group g = l>>4, L = l&15, q = L>>2, p = L&3;
// address = row (g*4 + q), cols 4p..4p+3 (+ column block 0).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(short *out) {
  __shared__ __attribute__((aligned(16))) short lds[16 * 32];
  for (int i = threadIdx.x; i < 16 * 32; i += 64) lds[i] = (short)((i / 32) * 100 + (i % 32));
  __syncthreads();
  const int l = threadIdx.x, g = l >> 4, L = l & 15, q = L >> 2, p = L & 3;
  const short *addr = lds + (g * 4 + q) * 32 + 4 * p;
  v4s v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s *)addr);
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
  short *d; hipMalloc(&d, 64 * 4 * 2);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
  return 0;
}
