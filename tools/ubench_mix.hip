// Does a mixed stream of v_mad_u64_u32 and plain 32-bit VALU instructions issue at the sum of the two classes' own
// times, and what do waves per SIMD and independent chains per lane do to it?  msm_accumulate's common path is 1558
// multiply-adds and ~560 plain instructions per mixed addition (profiles/isa_mix_r03.json), 84 % of the multiply-adds
// directly behind one into the same accumulator (a column-wise product is ONE dependency chain), at three waves per SIMD.
// Workgroups of 256 threads (one wave per SIMD), W of them per CU by way of the dynamic LDS size - one-wave workgroups
// are not spread evenly over the SIMDs (tools/ubench_chain.hip's odd 3-wave row).
// Per loop iteration and chain: 3 multiply-adds, then one plain instruction (and / add alternating) on a register of
// the same chain.  Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_mix.hip -o tools/ubench_mix.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

template <int C, int MODE, int U = 1>  // MODE 0: mads only   1: plain only   2: 3 mads + 1 plain;  U: body copies per iteration
__global__ __launch_bounds__(256) void k(uint64_t* io, int iters) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t acc[C], carry;
  uint32_t x[C];
  const uint32_t a = (uint32_t)io[i] | 1u, b = (uint32_t)(io[i] >> 32) | 1u;
#pragma unroll
  for (int c = 0; c < C; c++) {
    acc[c] = io[i] + c;
    x[c] = a ^ (0x1234567u + c);
  }
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int rep = 0; rep < U * 16 / C; rep++) {
#pragma unroll
      for (int m = 0; m < 3; m++)
#pragma unroll
        for (int c = 0; c < C; c++)
          if (MODE != 1) asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(acc[c]), "=s"(carry) : "v"(a), "v"(b));
#pragma unroll
      for (int c = 0; c < C; c++)
        if (MODE != 0) {
          if (rep & 1) asm volatile("v_and_b32 %0, %1, %0" : "+v"(x[c]) : "v"(b));
          else asm volatile("v_add_u32 %0, %1, %0" : "+v"(x[c]) : "v"(a));
        }
    }
  }
  uint64_t r = 0;
#pragma unroll
  for (int c = 0; c < C; c++) r ^= acc[c] ^ x[c];
  io[i] = r;
}

template <int C, int MODE, int U = 1>
double run(int waves_per_simd, uint64_t* d) {  // ns of SIMD time per wave-instruction
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const size_t lds = waves_per_simd >= 8 ? 0 : (size_t)(160 * 1024 / waves_per_simd) & ~(size_t)1023;
  hipFuncSetAttribute((const void*)k<C, MODE, U>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int blocks = prop.multiProcessorCount * waves_per_simd * 4;  // four rounds of resident workgroups
  const int iters = 4000 / U;
  const int per_iter = U * (MODE == 0 ? 48 : MODE == 1 ? 16 : 64);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k<C, MODE, U>), dim3(blocks), dim3(256), lds, 0, d, 10);
  double best = 1e30;
  for (int rep = 0; rep < 3; rep++) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<C, MODE, U>), dim3(blocks), dim3(256), lds, 0, d, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // a SIMD ran blocks / CUs waves of iters * per_iter instructions each
    const double ns = ms * 1e6 / ((double)blocks / prop.multiProcessorCount * iters * per_iter);
    if (ns < best) best = ns;
  }
  return best;
}

int main() {
  uint64_t* d;
  hipMalloc(&d, 8 * 256 * 256 * 4 * 8 * 4);
  hipMemset(d, 0x5a, 8 * 256 * 256 * 4 * 8 * 4);
  printf("ns of SIMD time per wave-instruction (rows: waves per SIMD; C = independent chains per lane)\n");
  printf("waves  mad C=1  mad C=2  mad C=8 | plain C=1 plain C=8 | mix C=1  mix C=2  mix C=4  mix C=8\n");
  for (int w : {1, 2, 3, 4, 8})
    printf("%5d %8.3f %8.3f %8.3f | %9.3f %9.3f | %7.3f %8.3f %8.3f %8.3f\n", w, run<1, 0>(w, d), run<2, 0>(w, d),
           run<8, 0>(w, d), run<1, 1>(w, d), run<8, 1>(w, d), run<1, 2>(w, d), run<2, 2>(w, d), run<4, 2>(w, d),
           run<8, 2>(w, d));
  // the same mixed stream as straight-line code of msm_accumulate's size: 32 copies of the body = 2048 instructions,
  // ~16 KB per loop iteration (the instruction cache is 64 KB per two CUs)
  printf("\nmix C=1, loop body of 64 / 512 / 2048 / 8192 instructions\n");
  for (int w : {2, 3, 4, 8})
    printf("%5d %8.3f %8.3f %8.3f %8.3f\n", w, run<1, 2, 1>(w, d), run<1, 2, 8>(w, d), run<1, 2, 32>(w, d),
           run<1, 2, 128>(w, d));
  return 0;
}
