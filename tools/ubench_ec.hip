// EC mixed-add throughput: inlined vs called field multiplication (instruction-cache effect).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DCAP_NOINLINE_MUL] tools/ubench_ec.hip -o tools/ubench_ec[_call]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../cap_amd/csrc/curve.hpp"
using namespace cap;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ __launch_bounds__(256) void k_madd(const g1_affine* pts, g1_xyzz* out, int iters, int npts) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  g1_xyzz acc = G1::from_affine(pts[t % npts]);
  for (int k = 0; k < iters; k++) {
    g1_affine p = pts[(t * 7 + k * 13 + 1) % npts];
    acc = G1::add_mixed(acc, p);
  }
  out[t] = acc;
}
__global__ __launch_bounds__(256) void k_mulchain(fe* io, int iters) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  fe x = io[i], y = io[i ^ 1];
  for (int k = 0; k < iters; k++) { x = Fq::mul(x, y); y = Fq::mul(y, x); }
  io[i] = x;
}
int main() {
  const int npts = 4096;
  std::vector<g1_affine> h(npts);
  // points k*G computed on host with the same header
  g1_affine g; g.x = Fq::one(); g.y = Fq::dbl(Fq::one());
  g1_xyzz acc = G1::from_affine(g);
  for (int i = 0; i < npts; i++) { h[i] = G1::to_affine(acc); acc = G1::add_mixed(acc, g); }
  g1_affine* d_pts; g1_xyzz* d_out;
  CK(hipMalloc(&d_pts, sizeof(g1_affine) * npts));
  CK(hipMemcpy(d_pts, h.data(), sizeof(g1_affine) * npts, hipMemcpyHostToDevice));
  CK(hipMalloc(&d_out, sizeof(g1_xyzz) * 256 * 8 * 256));
  fe* d_io; CK(hipMalloc(&d_io, sizeof(fe) * 256 * 8 * 256));
  CK(hipMemcpy(d_io, h.data(), sizeof(fe) * 2 * npts, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w : {1, 2, 4}) {
    int grid = 256 * w, iters = 200;
    hipLaunchKernelGGL(k_madd, dim3(grid), dim3(256), 0, 0, d_pts, d_out, 4, npts);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_madd, dim3(grid), dim3(256), 0, 0, d_pts, d_out, iters, npts);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double adds = (double)grid * 256 * iters;
    printf("add_mixed waves/SIMD=%d %8.3f ms  %7.2f G adds/s  (= %.1f G mul/s at 10 mul/add)\n", w, ms, adds / ms * 1e-6, adds * 10 / ms * 1e-6);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_mulchain, dim3(grid), dim3(256), 0, 0, d_io, 1000);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("mul chain waves/SIMD=%d %8.3f ms  %7.2f G mul/s\n", w, ms, (double)grid * 256 * 2000 / ms * 1e-6);
  }
  return 0;
}
