// v_mad_u64_u32 issue rate as a function of the parallelism a kernel offers the SIMD: W waves per SIMD x C independent
// dependency chains per lane.  msm_accumulate runs 3 waves per SIMD (168 VGPRs) and its column-wise multiplication is one
// long chain of multiply-adds (each column's first multiply-add takes the previous column's carry): (W, C) = (3, 1).
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_chain.hip -o tools/ubench_chain.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

template <int C>
__global__ __launch_bounds__(64) void k(uint64_t* io, int iters) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t acc[C];
  const uint32_t a = (uint32_t)io[i] | 1u, b = (uint32_t)(io[i] >> 32) | 1u;
#pragma unroll
  for (int c = 0; c < C; c++) acc[c] = io[i] + c;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int rep = 0; rep < 64 / C; rep++)
#pragma unroll
      for (int c = 0; c < C; c++) {
        uint64_t r, carry;
        asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(r), "=s"(carry) : "v"(a), "v"(b + c), "v"(acc[c]));
        acc[c] = r;
      }
  }
  uint64_t x = 0;
#pragma unroll
  for (int c = 0; c < C; c++) x ^= acc[c];
  io[i] = x;
}

template <int C>
double run(int waves_per_simd, uint64_t* d) {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int blocks = prop.multiProcessorCount * 4 * waves_per_simd;  // one-wave workgroups: W per SIMD
  const int iters = 4000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k<C>, dim3(blocks), dim3(64), 0, 0, d, 10);
  double best = 0;
  for (int rep = 0; rep < 3; rep++) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<C>, dim3(blocks), dim3(64), 0, 0, d, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double rate = (double)blocks * 64 * iters * 64 / (ms * 1e-3);
    if (rate > best) best = rate;
  }
  return best / 1e12;
}

int main() {
  uint64_t* d;
  hipMalloc(&d, 8 * 64 * 256 * 4 * 16);
  hipMemset(d, 0x5a, 8 * 64 * 256 * 4 * 16);
  printf("v_mad_u64_u32, T lane-operations/s   (rows: waves per SIMD; columns: independent chains per lane)\n");
  printf("waves      C=1      C=2      C=4      C=8\n");
  for (int w : {1, 2, 3, 4, 6, 8})
    printf("%5d %8.2f %8.2f %8.2f %8.2f\n", w, run<1>(w, d), run<2>(w, d), run<4>(w, d), run<8>(w, d));
  return 0;
}
