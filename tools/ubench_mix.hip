// Does a mixed stream of v_mad_u64_u32 and plain 32-bit VALU instructions issue at the sum of the two classes' own
// times?  msm_accumulate's common path is 1558 multiply-adds and ~560 plain instructions per mixed addition
// (profiles/isa_mix_r03.json); capgpu_ubench_issue_rates measures each class alone (mad ~36 T, plain ~68 T lane-ops/s).
// Here: the same instruction counts, (a) each class alone, (b) interleaved 3 : 1 (one plain instruction after every third
// multiply-add, as the column-wise Montgomery product has them), (c) the same mix with the plain instructions in pairs.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_mix.hip -o tools/ubench_mix.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define MAD(acc, a, b) asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(acc), "=s"(carry) : "v"(a), "v"(b))
#define AND(x, m) asm volatile("v_and_b32 %0, %1, %0" : "+v"(x) : "v"(m))
#define ADD(x, m) asm volatile("v_add_u32 %0, %1, %0" : "+v"(x) : "v"(m))

// MODE 0: 24 mads   1: 8 plain   2: (3 mads, 1 plain) x 8   3: (6 mads, 2 plain) x 4   4: 24 mads then 8 plain
template <int MODE>
__global__ __launch_bounds__(64) void k(uint64_t* io, int iters) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t acc = io[i], carry;
  const uint32_t a = (uint32_t)io[i] | 1u, b = (uint32_t)(io[i] >> 32) | 1u;
  uint32_t x = a ^ 0x1234567u, m = b | 0x10001u;
  for (int it = 0; it < iters; it++) {
    if (MODE == 0) {
#pragma unroll
      for (int r = 0; r < 24; r++) MAD(acc, a, b);
    } else if (MODE == 1) {
#pragma unroll
      for (int r = 0; r < 4; r++) { AND(x, m); ADD(x, m); }
    } else if (MODE == 2) {
#pragma unroll
      for (int r = 0; r < 8; r++) {
        MAD(acc, a, b); MAD(acc, a, b); MAD(acc, a, b);
        if (r & 1) AND(x, m); else ADD(x, m);
      }
    } else if (MODE == 3) {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        MAD(acc, a, b); MAD(acc, a, b); MAD(acc, a, b); MAD(acc, a, b); MAD(acc, a, b); MAD(acc, a, b);
        AND(x, m); ADD(x, m);
      }
    } else {
#pragma unroll
      for (int r = 0; r < 24; r++) MAD(acc, a, b);
#pragma unroll
      for (int r = 0; r < 4; r++) { AND(x, m); ADD(x, m); }
    }
  }
  io[i] = acc ^ x;
}

template <int MODE>
double run(int waves_per_simd, uint64_t* d) {  // ns per loop iteration per wave slot (the SIMD's time / iterations)
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int blocks = prop.multiProcessorCount * 4 * waves_per_simd;
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, 10);
  double best = 1e30;
  for (int rep = 0; rep < 3; rep++) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double ns = ms * 1e6 / iters / waves_per_simd;  // SIMD time per iteration of ONE wave
    if (ns < best) best = ns;
  }
  return best;
}

int main() {
  uint64_t* d;
  hipMalloc(&d, 8 * 64 * 256 * 4 * 16);
  hipMemset(d, 0x5a, 8 * 64 * 256 * 4 * 16);
  printf("SIMD time per loop iteration of one wave, ns  (24 multiply-adds and/or 8 plain instructions per iteration)\n");
  printf("waves  mad_only  plain_only  sum   mix_3:1  mix_6:2  24_then_8\n");
  for (int w : {1, 2, 3, 4, 8}) {
    const double m = run<0>(w, d), p = run<1>(w, d);
    printf("%5d %9.2f %10.2f %6.2f %8.2f %8.2f %9.2f\n", w, m, p, m + p, run<2>(w, d), run<3>(w, d), run<4>(w, d));
  }
  return 0;
}
