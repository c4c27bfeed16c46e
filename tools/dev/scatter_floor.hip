// scatter_floor.hip -- what the primitives of ANY points -> voxels dedup cost on this chip (measure the
// floor instead of arguing it).  For N = 320 k and 1.5 M random items (the bench's batch and configs[4]) and tables of
// 4 MB (fits an XCD's L2) and 64 MB (does not), device time with HIP events, best of 20:
//   gather16   : N random 16-byte loads (a probe / a brick look-up)
//   atomic_or  : N fire-and-forget 64-bit atomicOr on random words (a mark round)
//   atomic_min : N fire-and-forget 32-bit atomicMin on random words (first-point selection)
//   atomic_ret : N RETURNING 64-bit atomicCAS on random words (a hash insert)
//   stream     : read N x 32 B + write N x 16 B, coalesced (the point list once through)
//   scan       : exclusive prefix sum of N int32 by chunk scan + decoupled look-back (site numbering)
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/dev/scatter_floor.hip -o /tmp/scatter_floor && /tmp/scatter_floor
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ inline uint64_t mix(uint64_t k) { k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33; return k; }

__global__ void k_gather16(const uint4 *t, uint64_t mask, int64_t n, uint32_t *out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint4 v = t[mix(i) & mask];
  if (v.x == 0x12345u) out[i] = v.y;          // (never true: keeps the load)
}
__global__ void k_atomic_or(unsigned long long *t, uint64_t mask, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) atomicOr(&t[mix(i) & mask], 1ull << (i & 63));
}
__global__ void k_atomic_min(uint32_t *t, uint64_t mask, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) atomicMin(&t[mix(i) & mask], (uint32_t)i);
}
__global__ void k_atomic_ret(unsigned long long *t, uint64_t mask, int64_t n, uint32_t *out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned long long prev = atomicCAS(&t[mix(i) & mask], ~0ull, (unsigned long long)i);
  out[i] = (uint32_t)prev;
}
__global__ void k_stream(const uint4 *in, uint4 *out, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint4 a = in[2 * i], b = in[2 * i + 1];
  out[i] = make_uint4(a.x, a.z, b.x, b.z);
}
__global__ __launch_bounds__(256) void k_scan(const int32_t *in, int32_t *out, int64_t n, unsigned long long *status, int *ticket) {
  __shared__ int s_chunk, s_w[4];
  __shared__ unsigned s_excl;
  if (threadIdx.x == 0) s_chunk = atomicAdd(ticket, 1);
  __syncthreads();
  const int chunk = s_chunk, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t base = (int64_t)chunk * 1024 + threadIdx.x * 4;
  int v[4], a = 0;
  for (int j = 0; j < 4; ++j) { v[j] = base + j < n ? in[base + j] : 0; a += v[j]; }
  int incl = a;
  for (int d = 1; d < 64; d <<= 1) { int o = __shfl_up(incl, d); if (lane >= d) incl += o; }
  if (lane == 63) s_w[wave] = incl;
  __syncthreads();
  int wpre = 0, total = 0;
  for (int w = 0; w < 4; ++w) { if (w < wave) wpre += s_w[w]; total += s_w[w]; }
  if (wave == 0) {
    if (lane == 0) __hip_atomic_store(&status[chunk], ((unsigned long long)(chunk == 0 ? 2u : 1u) << 62) | (unsigned)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned excl = 0;
    int j = chunk - 1;
    while (j >= 0) {
      const int idx = j - lane;
      unsigned long long st = 2ull << 62;
      if (idx >= 0) do { st = __hip_atomic_load(&status[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while ((st >> 62) == 0ull);
      const unsigned long long m2 = __ballot((st >> 62) == 2ull);
      const int stop = m2 ? __ffsll((long long)m2) - 1 : 63;
      unsigned vv = lane <= stop ? (unsigned)(st & 0xffffffffull) : 0u;
      for (int d = 32; d >= 1; d >>= 1) vv += __shfl_xor(vv, d);
      excl += vv;
      if (m2) break;
      j -= 64;
    }
    if (lane == 0) {
      if (chunk != 0) __hip_atomic_store(&status[chunk], (2ull << 62) | (excl + (unsigned)total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_excl = excl;
    }
  }
  __syncthreads();
  int run = (int)s_excl + wpre + incl - a;
  for (int j = 0; j < 4; ++j) if (base + j < n) { out[base + j] = run; run += v[j]; }
}

int main() {
  const int64_t Ns[2] = {320000, 1500000};
  const size_t Ts[2] = {4u << 20, 64u << 20};
  void *table, *buf_in, *buf_out, *status;
  int *ticket;
  CK(hipMalloc(&table, 64u << 20));
  CK(hipMalloc(&buf_in, 1500000 * 32));
  CK(hipMalloc(&buf_out, 1500000 * 16));
  CK(hipMalloc(&status, 8 * 2048));
  CK(hipMalloc((void **)&ticket, 4));
  CK(hipMemset(buf_in, 1, 1500000 * 32));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto best = [&](auto fn, auto prep) -> float {
    std::vector<float> ms;
    for (int r = 0; r < 20; ++r) {
      prep();
      hipEventRecord(e0, 0);
      fn();
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float t; hipEventElapsedTime(&t, e0, e1);
      ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    return ms[2] * 1000.f;
  };
  printf("%-12s %10s %12s %12s\n", "primitive", "N", "table 4 MB", "table 64 MB");
  for (int64_t n : Ns) {
    const unsigned g = (unsigned)((n + 255) / 256);
    float r[4][2];
    for (int ti = 0; ti < 2; ++ti) {
      const size_t T = Ts[ti];
      auto clr = [&]() { hipMemsetAsync(table, 0xFF, T, 0); };
      auto nop = [&]() {};
      r[0][ti] = best([&]() { hipLaunchKernelGGL(k_gather16, dim3(g), dim3(256), 0, 0, (const uint4 *)table, (uint64_t)(T / 16 - 1), n, (uint32_t *)buf_out); }, nop);
      r[1][ti] = best([&]() { hipLaunchKernelGGL(k_atomic_or, dim3(g), dim3(256), 0, 0, (unsigned long long *)table, (uint64_t)(T / 8 - 1), n); }, nop);
      r[2][ti] = best([&]() { hipLaunchKernelGGL(k_atomic_min, dim3(g), dim3(256), 0, 0, (uint32_t *)table, (uint64_t)(T / 4 - 1), n); }, clr);
      r[3][ti] = best([&]() { hipLaunchKernelGGL(k_atomic_ret, dim3(g), dim3(256), 0, 0, (unsigned long long *)table, (uint64_t)(T / 8 - 1), n, (uint32_t *)buf_out); }, clr);
    }
    const char *names[4] = {"gather16", "atomic_or", "atomic_min", "atomic_ret"};
    for (int q = 0; q < 4; ++q) printf("%-12s %10lld %9.1f us %9.1f us\n", names[q], (long long)n, r[q][0], r[q][1]);
    float ts = best([&]() { hipLaunchKernelGGL(k_stream, dim3(g), dim3(256), 0, 0, (const uint4 *)buf_in, (uint4 *)buf_out, n); }, [&]() {});
    printf("%-12s %10lld %9.1f us   (%.0f GB/s of 48 B per item)\n", "stream", (long long)n, ts, n * 48.0 / ts / 1e3);
    const unsigned gc = (unsigned)((n + 1023) / 1024);
    float tc = best([&]() { hipLaunchKernelGGL(k_scan, dim3(gc), dim3(256), 0, 0, (const int32_t *)buf_in, (int32_t *)buf_out, n, (unsigned long long *)status, ticket); },
                    [&]() { hipMemsetAsync(status, 0, 8 * 2048, 0); hipMemsetAsync(ticket, 0, 4, 0); });
    printf("%-12s %10lld %9.1f us   (%u chunks of 1024)\n", "scan", (long long)n, tc, gc);
  }
  return 0;
}
