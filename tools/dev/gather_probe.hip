// gather_probe.hip -- how fast can one wave per SIMD pull 256-byte rows by index into REGISTERS, by lane layout?
//   layout A: lane (n, g) reads 16 bytes at 16 g + 64 i of row n of a 16-row group (the MFMA operand layout: 16 rows x 64 B
//             per instruction)
//   layout B: lane (rr, pp) reads piece pp of row 4 j + rr (4 rows x 256 B per instruction: quads are contiguous)
// Both move 16 KiB per wave and "offset" (64 rows) in 16 instructions; SETS of them in flight.
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/dev/gather_probe.hip -o /tmp/gp && /tmp/gp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int LAYOUT, int SETS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void k_gather(const unsigned char *tab, long long tab_bytes, const int *idx, int iters, unsigned *sink) {
  extern __shared__ unsigned char lds[];   // only to force one workgroup per CU
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(tab), 0, (int)tab_bytes, 0x00020000);
  // XCD-contiguous logical workgroup id (workgroups are dealt round-robin over the 8 XCDs)
  const unsigned lw = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int *my = idx + ((long long)lw * 4 + wave) * iters * 64;
  u32x4 q[SETS][16];
#pragma unroll
  for (int s = 0; s < SETS; ++s)
#pragma unroll
    for (int j = 0; j < 16; ++j) q[s][j] = (u32x4){0u, 0u, 0u, 0u};
  unsigned acc = 0;
  auto issue = [&](int it, u32x4 (&d)[16]) {
    const int *ix = my + (long long)it * 64;
    if (LAYOUT == 0) {
      const int n = lane & 15, g = lane >> 4;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const unsigned vo = (unsigned)ix[p * 16 + n] * 256u + (unsigned)g * 16u;
#pragma unroll
        for (int i = 0; i < 4; ++i) d[p * 4 + i] = __builtin_amdgcn_raw_buffer_load_b128(r, vo + 64u * i, 0, 0);
      }
    } else if (LAYOUT == 1) {
      const int rr = lane >> 4, pp = lane & 15;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const unsigned vo = (unsigned)ix[j * 4 + rr] * 256u + (unsigned)pp * 16u;
        d[j] = __builtin_amdgcn_raw_buffer_load_b128(r, vo, 0, 0);
      }
    } else {
      const int rr = lane >> 4, pp = lane & 15;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int t = ix[j * 4 + rr];
        const unsigned vo = (unsigned)t * 256u + (unsigned)pp * 16u;
        if (t < (1 << 23)) asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "+v"(d[j]) : "v"(vo), "s"(r) : "memory");
      }
    }
  };
  auto eat = [&](const u32x4 (&d)[16]) {
    if (LAYOUT == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SETS == 2 ? 16 : 32) : "memory");   // (an upper bound of what may stay in flight)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc ^= d[j][0] ^ d[j][1] ^ d[j][2] ^ d[j][3];
  };
#pragma unroll
  for (int s = 0; s + 1 < SETS; ++s) issue(s, q[s]);
  for (int it0 = 0; it0 < iters; it0 += SETS) {
#pragma unroll
    for (int s = 0; s < SETS; ++s) {
      const int it = it0 + s;
      if (it < iters) {
        const int nx = it + SETS - 1 < iters ? it + SETS - 1 : iters - 1;
        issue(nx, q[(s + SETS - 1) % SETS]);
        __builtin_amdgcn_sched_barrier(0);
        eat(q[s]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

int main() {
  const int rows = 84077, iters = 54, wgs = 256;      // 54 "offsets" per wave
  const long long tab_bytes = (long long)rows * 256;
  unsigned char *tab;
  int *idx;
  unsigned *sink;
  hipMalloc(&tab, tab_bytes);
  hipMemset(tab, 1, tab_bytes);
  hipMalloc(&sink, 4);
  const long long nidx = (long long)wgs * 4 * iters * 64;
  hipMalloc(&idx, nidx * 4);
  std::vector<int> h(nidx);
  for (int mode = 0; mode < 3; ++mode) {
    srand(1);
    for (long long w = 0; w < (long long)wgs * 4; ++w) {
      // a wave's rows: mode 0 uniformly random; mode 1 within +-1500 rows of the wave's own position (spatially local)
      const long long centre = w * rows / (wgs * 4);
      for (int i = 0; i < iters * 64; ++i) {
        long long v = mode == 0 ? rand() % rows : centre + (rand() % 3000) - 1500;
        if (v < 0) v = 0;
        if (v >= rows) v = rows - 1;
        if (mode == 2) {                       // sparse: group (i / 16) inactive with p = 0.5, else a row valid with p = 0.57
          srand((unsigned)(w * 7919 + i / 16)); const bool off = rand() & 1; srand((unsigned)(w * 104729 + i) * 31u + 7u);
          if (off || (rand() % 100) >= 57) v = 1 << 23;   // * 256 = 2^31: out of the descriptor's range
        }
        h[w * iters * 64 + i] = (int)v;
      }
    }
    hipMemcpy(idx, h.data(), nidx * 4, hipMemcpyHostToDevice);
    auto run = [&](const char *name, auto kern) {
      hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
      hipEvent_t a, b;
      hipEventCreate(&a);
      hipEventCreate(&b);
      for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), 100 * 1024, 0, tab, tab_bytes, idx, iters, sink);
      hipEventRecord(a);
      const int N = 20;
      for (int w = 0; w < N; ++w) hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), 100 * 1024, 0, tab, tab_bytes, idx, iters, sink);
      hipEventRecord(b);
      hipEventSynchronize(b);
      float ms;
      hipEventElapsedTime(&ms, a, b);
      const double us = ms * 1e3 / N, bytes = (double)nidx * 256;
      printf("%-7s %-28s %8.1f us  %6.2f TB/s chip  %5.1f GB/s per CU  (%.2f us per 64-row set)\n", mode == 0 ? "random" : mode == 1 ? "local" : "sparse", name, us,
             bytes / us / 1e6, bytes / us / 1e3 / 256, us / iters);
    };
    run("A lane-per-row, 2 sets", k_gather<0, 2>);
    run("A lane-per-row, 3 sets", k_gather<0, 3>);
    run("B 16 lanes per row, 2 sets", k_gather<1, 2>);
    run("B 16 lanes per row, 3 sets", k_gather<1, 3>);
    run("C = B, EXEC-masked, 2 sets", k_gather<2, 2>);
  }
  return 0;
}
