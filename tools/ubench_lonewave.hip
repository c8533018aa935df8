// What does a point addition cost a wave that has a SIMD (almost) to itself?  The small-batch MSM kernels - the bucket
// tree, the bit-plane reduction, the finishing kernels - are chains of dependent additions on a few hundred waves.
// K dependent general additions (curve29.hpp, both field schedules) per wave, W waves per SIMD (grid = 1024 * W
// workgroups of one wave):  time per addition = kernel time / K.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I cap_amd/csrc tools/ubench_lonewave.hip -o tools/ubench_lonewave.bin
#include <hip/hip_runtime.h>
#include <stdio.h>

#include "curve29.hpp"
using namespace cap;

template <class G, int MODE>
__global__ __launch_bounds__(64) void chain(const g1_xyzz* __restrict__ in, g1_xyzz* __restrict__ out, int K) {
  const uint32_t t = blockIdx.x * 64 + threadIdx.x;
  g1x a = G::load(in[t & 1023]);
  g1x b = G::load(in[(t * 7 + 3) & 1023]);
  for (int k = 0; k < K; k++) {
    if (MODE == 0) a = G::add(a, b);              // general addition (special cases in branches)
    else if (!G::add_acc(a, b)) a = G::add(a, b);  // lean addition, general one as the fallback
  }
  out[t] = G::store(a);
}

int main() {
  const int N = 1024;
  g1_xyzz* d_in;
  g1_xyzz* d_out;
  hipMalloc(&d_in, sizeof(g1_xyzz) * N);
  hipMalloc(&d_out, sizeof(g1_xyzz) * 1024 * 64 * 4);
  // points k * G in XYZZ form with zz = 1: (x, y) of small multiples, built on the host by repeated addition
  {
    std::vector<g1_xyzz> h(N);
    g1_affine g;
    g.x = Fq::one();
    g.y = Fq::dbl(Fq::one());
    g1_xyzz acc = G1::from_affine(g);
    for (int i = 0; i < N; i++) {
      g1_affine p = G1::to_affine(acc);
      h[i] = G1::from_affine(p);
      acc = G1::add_mixed(acc, g);
      if (i % 3 == 0) acc = G1::dbl(acc);
    }
    // to the lazy field's memory image (internal Montgomery form): convert through G1L on the host
    for (int i = 0; i < N; i++) {
      g1a q;
      g1_affine p = G1::to_affine(h[i]);
      q.x = Fq29::canonical(Fq29::from_ext(p.x));
      q.y = Fq29::canonical(Fq29::from_ext(p.y));
      g1x x = G1L::add_mixed(G1L::inf(), q);
      h[i] = G1L::store(x);
    }
    hipMemcpy(d_in, h.data(), sizeof(g1_xyzz) * N, hipMemcpyHostToDevice);
  }
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int K = 64;
  auto run = [&](const char* name, void (*kern)(const g1_xyzz*, g1_xyzz*, int), int waves_per_simd) {
    const int grid = 1024 * waves_per_simd;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64), 0, 0, d_in, d_out, K);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64), 0, 0, d_in, d_out, K);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s %d wave(s)/SIMD: %7.2f us per addition  (%6.2f G additions/s)\n", name, waves_per_simd, ms * 1e3 / K,
           (double)grid * 64 * K / ms / 1e6);
  };
  for (int w = 1; w <= 3; w++) run("column-wise, general add", chain<G1LT<1>, 0>, w);
  for (int w = 1; w <= 3; w++) run("column-wise, lean add + fallback", chain<G1LT<1>, 1>, w);
  for (int w = 1; w <= 3; w++) run("row-wise, general add", chain<G1LT<0>, 0>, w);
  for (int w = 1; w <= 3; w++) run("row-wise, lean add + fallback", chain<G1LT<0>, 1>, w);
  return 0;
}
