// Mixed-addition throughput of the lazy 29-bit field in isolation (no gathers): how far is msm_accumulate from it?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_madd29.hip -o tools/ubench_madd29.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../cap_amd/csrc/curve29.hpp"
using namespace cap;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// the common path of G1L::add_mixed alone (no infinity / doubling / cancellation handling): instruction-count probe
__device__ __forceinline__ g1x madd_main(const g1x& a, const g1a& q_in, bool negate) {
  using F = Fq29;
  g1a q = q_in;
  if (negate) q.y = F::neg(q.y);
  fl u2 = F::mul(q.x, a.zz);
  fl s2 = F::mul(q.y, a.zzz);
  fl p = F::sub(u2, a.x);
  fl r = F::sub(s2, a.y);
  fl pp = F::sqr(p);
  fl ppp = F::mul(p, pp);
  fl qq = F::mul(a.x, pp);
  g1x o;
  o.x = F::weak_reduce(F::sub(F::sub(F::sqr(r), ppp), F::add(qq, qq)));
  o.y = F::weak_reduce(F::mul_add_mul(r, F::sub(qq, o.x), F::neg(a.y), ppp));
  o.zz = F::mul(a.zz, pp);
  o.zzz = F::mul(a.zzz, ppp);
  return o;
}
__global__ __launch_bounds__(256) void k_madd_main(const g1_affine* pts, g1_xyzz* out, int iters, int npts) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  g1a p0 = G1L::load(pts[t % npts]), p1 = G1L::load(pts[(t * 7 + 3) % npts]);
  g1x acc = G1L::add_mixed(G1L::inf(), p0, false);
  acc = G1L::add_mixed(acc, p1, false);
  for (int k = 0; k < iters; k++) acc = madd_main(acc, (k & 1) ? p1 : p0, (k & 2) != 0);
  out[t] = G1L::store(acc);
}
__global__ __launch_bounds__(256) void k_madd(const g1_affine* pts, g1_xyzz* out, int iters, int npts) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  g1x acc = G1L::inf();
  g1a p0 = G1L::load(pts[t % npts]), p1 = G1L::load(pts[(t * 7 + 3) % npts]);
  for (int k = 0; k < iters; k++) {
    acc = G1L::add_mixed(acc, (k & 1) ? p1 : p0, (k & 2) != 0);
  }
  out[t] = G1L::store(acc);
}
int main() {
  const int npts = 4096;
  std::vector<g1_affine> h(npts);
  g1_affine g; g.x = Fq::one(); g.y = Fq::dbl(Fq::one());
  g1_xyzz acc = G1::from_affine(g);
  for (int i = 0; i < npts; i++) { h[i] = G1::to_affine(acc); acc = G1::add_mixed(acc, g); }
  // to the internal form of the lazy field
  for (int i = 0; i < npts; i++) { h[i].x = Fq29::pack(Fq29::canonical(Fq29::from_ext(h[i].x))); h[i].y = Fq29::pack(Fq29::canonical(Fq29::from_ext(h[i].y))); }
  g1_affine* d_pts; g1_xyzz* d_out;
  CK(hipMalloc(&d_pts, sizeof(g1_affine) * npts));
  CK(hipMemcpy(d_pts, h.data(), sizeof(g1_affine) * npts, hipMemcpyHostToDevice));
  CK(hipMalloc(&d_out, sizeof(g1_xyzz) * 256 * 8 * 256));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w : {1, 2, 3, 4}) {
    int grid = 256 * w, iters = 400;
    hipLaunchKernelGGL(k_madd, dim3(grid), dim3(256), 0, 0, d_pts, d_out, 4, npts);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_madd, dim3(grid), dim3(256), 0, 0, d_pts, d_out, iters, npts);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double adds = (double)grid * 256 * iters;
    printf("add_mixed29 (all cases)  waves/SIMD=%d %8.3f ms  %7.2f G adds/s  %.1f ps/add\n", w, ms, adds / ms * 1e-6, ms * 1e9 / adds);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_madd_main, dim3(grid), dim3(256), 0, 0, d_pts, d_out, iters, npts);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("madd common path only    waves/SIMD=%d %8.3f ms  %7.2f G adds/s  %.1f ps/add\n", w, ms, adds / ms * 1e-6, ms * 1e9 / adds);
  }
  return 0;
}
