// Random 64-byte gathers over a table of the given size: how many G gathers/s does HBM + the TLBs sustain?
// (sizing experiment for a precomputed-multiples MSM table; not part of the product)
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_gather.hip -o /tmp/ubench_gather && /tmp/ubench_gather 42 172000
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

struct pt { uint4 a, b, c, d; };  // 64 B

__global__ __launch_bounds__(256) void gather(const pt* __restrict__ t, uint64_t n_pts, int iters, uint32_t* out) {
  uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  x = x * 0x9E3779B97F4A7C15ull + 12345;
  uint32_t acc = 0;
  for (int i = 0; i < iters; i++) {
    x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
    uint64_t idx = x % n_pts;
    pt p = t[idx];
    acc ^= p.a.x ^ p.b.y ^ p.c.z ^ p.d.w;
  }
  if (acc == 0x12345678u) out[0] = acc;
}

int main(int argc, char** argv) {
  for (int a = 1; a < argc; a++) {
    double mb = atof(argv[a]);
    uint64_t bytes = (uint64_t)(mb * 1e6);
    uint64_t n_pts = bytes / 64;
    pt* t = nullptr;
    if (hipMalloc(&t, n_pts * 64) != hipSuccess) { printf("%.0f MB: alloc failed\n", mb); continue; }
    hipMemset(t, 1, n_pts * 64 > (1ull << 33) ? (1ull << 33) : n_pts * 64);
    uint32_t* out; hipMalloc(&out, 4);
    const int iters = 64, blocks = 256 * 64;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    gather<<<blocks, 256>>>(t, n_pts, 4, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    gather<<<blocks, 256>>>(t, n_pts, iters, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double g = (double)blocks * 256 * iters;
    printf("table %.0f MB: %.2f G gathers/s  (%.2f TB/s of 64-B reads)\n", mb, g / ms / 1e6, g * 64 / ms / 1e9);
    hipFree(t); hipFree(out);
  }
  return 0;
}
