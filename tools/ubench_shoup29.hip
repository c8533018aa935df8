// Go / no-go for the constant-multiplicand product in the NTT (round-5 VERDICT item 4): Fl::mul_shoup (143 multiply-adds,
// no serial digit chain, a second table word per twiddle) against the Montgomery product the passes use (Fl::mul, row-wise
// schedule: 171 multiply-adds) and its column-wise schedule - in the setting of the passes' radix-4 rounds: four values
// per lane, three twiddle loads per round from a table (36 B entries, L1 / L2 resident like tw_small), the round's four
// products and eight additions / subtractions, 256-thread workgroups, the NTT's occupancy (4 workgroups per CU by way of
// a 36 KiB LDS reservation) and at 2 workgroups per CU.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I cap_amd/csrc tools/ubench_shoup29.hip -o tools/ubench_shoup29.bin
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#include "field29.hpp"

using namespace cap;
using FR = Fl<FrP29, 0>;  // row-wise Montgomery (the NTT passes' schedule)
using FC = Fl<FrP29, 1>;  // column-wise Montgomery

struct Tw {
  fl w;   // Montgomery variants: w * 2^261 mod p (normalized); Shoup: plain w
  fl wq;  // Shoup: floor(w 2^261 / p)
};

// the same constant-multiplicand product with INDEPENDENT column accumulators (the row-wise schedule of Fl::mul: every
// product lands in its own 64-bit column, the carries walk up afterwards) instead of one running accumulator
__device__ __forceinline__ fl mul_shoup_row(const fl& a, const fl& w, const fl& wq) {
  constexpr uint32_t M = 0x1fffffffu;
  uint64_t c[10];
#pragma unroll
  for (int k = 0; k < 10; k++) c[k] = 0;
#pragma unroll
  for (int i = 0; i < 9; i++)
#pragma unroll
    for (int j = 0; j < 9; j++)
      if (i + j >= 7) c[i + j - 7] += (uint64_t)a.v[i] * wq.v[j];
  uint32_t q[9];
  c[1] += c[0] >> 29;
  c[2] += c[1] >> 29;
#pragma unroll
  for (int k = 2; k < 10; k++) {
    q[k - 2] = (uint32_t)c[k] & M;
    if (k < 9) c[k + 1] += c[k] >> 29;
  }
  q[8] = (uint32_t)(c[9] >> 29);
  uint64_t d[9];
#pragma unroll
  for (int k = 0; k < 9; k++) d[k] = 0;
#pragma unroll
  for (int i = 0; i < 9; i++)
#pragma unroll
    for (int j = 0; i + j < 9; j++) d[i + j] += (uint64_t)a.v[i] * w.v[j] + (uint64_t)q[i] * FrP29::NEGP[j];
  fl r;
#pragma unroll
  for (int k = 0; k < 9; k++) {
    r.v[k] = (uint32_t)d[k] & M;
    if (k < 8) d[k + 1] += d[k] >> 29;
  }
  return r;
}
// the row-wise Montgomery product with Fr's special digit: -r^-1 mod 2^29 = 2^28 - 1, so m = (c << 28) - c needs no
// multiplication (Fq has no such luck)
__device__ __forceinline__ fl mul_mont_fr_special(const fl& a, const fl& b) {
  constexpr uint32_t M = 0x1fffffffu;
  uint64_t c[18];
#pragma unroll
  for (int k = 0; k < 18; k++) c[k] = 0;
#pragma unroll
  for (int i = 0; i < 9; i++)
#pragma unroll
    for (int j = 0; j < 9; j++) c[i + j] += (uint64_t)a.v[i] * b.v[j];
#pragma unroll
  for (int k = 0; k < 9; k++) {
    const uint32_t lo = (uint32_t)c[k];
    const uint32_t m = ((lo << 28) - lo) & M;
#pragma unroll
    for (int j = 0; j < 9; j++) c[k + j] += (uint64_t)m * FrP29::MOD[j];
    c[k + 1] += c[k] >> 29;
  }
  fl r;
#pragma unroll
  for (int k = 9; k < 17; k++) {
    r.v[k - 9] = (uint32_t)c[k] & M;
    c[k + 1] += c[k] >> 29;
  }
  r.v[8] = (uint32_t)c[17];
  return r;
}

// V = 0 row-wise Montgomery, 1 column-wise Montgomery, 2 Shoup (running accumulator), 3 Shoup with independent column
// accumulators, 4 row-wise Montgomery with Fr's multiplication-free digit
template <int V>
__device__ __forceinline__ fl tw_mul(const fl& a, const Tw* __restrict__ tab, uint32_t idx) {
  if constexpr (V == 0) return FR::mul(a, tab[idx].w);
  if constexpr (V == 1) return FC::mul(a, tab[idx].w);
  if constexpr (V == 2) return FR::mul_shoup(a, tab[idx].w, tab[idx].wq);
  if constexpr (V == 3) return mul_shoup_row(a, tab[idx].w, tab[idx].wq);
  return mul_mont_fr_special(a, tab[idx].w);
}
template <int V>
__device__ __forceinline__ fl bf_sub(const fl& u, const fl& t) {
  if constexpr (V == 2 || V == 3) return FR::sub8p(u, t);
  return FR::sub2p(u, t);
}
template <int V>
__device__ __forceinline__ fl bf_sub_lazy(const fl& u, const fl& t) {
  if constexpr (V == 2 || V == 3) return FR::sub8p_lazy(u, t);
  return FR::sub2p_lazy(u, t);
}

template <int V>
__global__ __launch_bounds__(256) void k_rounds(fl* io, const Tw* __restrict__ tab, uint32_t tab_mask, int rounds) {
  extern __shared__ unsigned char occupancy_pad[];  // (only reserves LDS: sets the workgroups per CU)
  const uint32_t g = blockIdx.x * 256 + threadIdx.x;
  fl a = io[4 * g], b = io[4 * g + 1], cc = io[4 * g + 2], d = io[4 * g + 3];
  uint32_t pos = g * 2654435761u;
  for (int r = 0; r < rounds; r++) {
    pos = pos * 1664525u + 1013904223u;
    const uint32_t i1 = (pos >> 8) & tab_mask, i2 = (pos >> 9) & tab_mask, i3 = (i2 + (tab_mask >> 1)) & tab_mask;
    fl t = tw_mul<V>(b, tab, i1);
    fl a1 = FR::add(a, t), b1 = bf_sub_lazy<V>(a, t);
    t = tw_mul<V>(d, tab, i1);
    fl c1 = FR::add(cc, t), d1 = bf_sub<V>(cc, t);
    t = tw_mul<V>(c1, tab, i2);
    // (the real round stores these four to LDS normalized; the weak reduction stands for the values coming back small)
    a = FR::weak_reduce(FR::normalize(FR::add(a1, t)));
    cc = FR::weak_reduce(bf_sub<V>(a1, t));
    t = tw_mul<V>(d1, tab, i3);
    b = FR::weak_reduce(FR::normalize(FR::add(b1, t)));
    d = FR::weak_reduce(FR::sub_from_lazy(b1, t));
  }
  io[4 * g] = a;
  io[4 * g + 1] = b;
  io[4 * g + 2] = cc;
  io[4 * g + 3] = d;
}

// products only: 4 independent chains per lane, no butterfly arithmetic (the multiplication's own issue cost)
template <int V>
__global__ __launch_bounds__(256) void k_mul_only(fl* io, const Tw* __restrict__ tab, uint32_t tab_mask, int rounds) {
  extern __shared__ unsigned char occupancy_pad[];
  const uint32_t g = blockIdx.x * 256 + threadIdx.x;
  fl x[4];
  for (int k = 0; k < 4; k++) x[k] = io[4 * g + k];
  uint32_t pos = g * 2654435761u;
  for (int r = 0; r < rounds; r++) {
    pos = pos * 1664525u + 1013904223u;
#pragma unroll
    for (int k = 0; k < 4; k++) x[k] = tw_mul<V>(x[k], tab, ((pos >> 8) + 97u * k) & tab_mask);
  }
  for (int k = 0; k < 4; k++) io[4 * g + k] = x[k];
}

template <class K>
static double run(K kern, fl* d, const Tw* tab, uint32_t mask, int wg_per_cu, size_t lds, int rounds) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int blocks = 256 * wg_per_cu;
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, d, tab, mask, 8);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, d, tab, mask, rounds);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return (double)blocks * 256 * 4 * rounds / ms * 1e-6;  // G multiplications / s
}

int main() {
  const uint32_t tab_n = 512;  // tw_small of a 1024-element tile
  std::vector<Tw> h(tab_n);
  std::vector<fl> hv((size_t)256 * 8 * 256 * 4);
  uint64_t sd = 0x9E3779B97F4A7C15ull;
  auto rnd = [&] {
    sd ^= sd << 13;
    sd ^= sd >> 7;
    sd ^= sd << 17;
    return (uint32_t)(sd >> 20);
  };
  for (auto& t : h)
    for (int j = 0; j < 9; j++) {
      t.w.v[j] = rnd() & (j == 8 ? 0x1fffffu : 0x1fffffffu);  // < 2^253: a canonical value either way
      t.wq.v[j] = rnd() & 0x1fffffffu;                          // (timing only: any normalized word)
    }
  for (auto& v : hv)
    for (int j = 0; j < 9; j++) v.v[j] = rnd() & (j == 8 ? 0x1fffffu : 0x1fffffffu);
  Tw* d_tab;
  fl* d;
  hipMalloc(&d_tab, sizeof(Tw) * tab_n);
  hipMalloc(&d, sizeof(fl) * hv.size());
  hipMemcpy(d_tab, h.data(), sizeof(Tw) * tab_n, hipMemcpyHostToDevice);
  const char* names[5] = {"montgomery row-wise (shipped)", "montgomery column-wise", "shoup, running accumulator",
                          "shoup, independent columns", "montgomery row-wise, Fr digit by shift"};
  for (int wg : {4, 2}) {
    const size_t lds = wg == 4 ? 36 * 1024 : 72 * 1024;
    for (int pass = 0; pass < 2; pass++) {
      double g[5];
      for (int v = 0; v < 5; v++) {
        hipMemcpy(d, hv.data(), sizeof(fl) * hv.size(), hipMemcpyHostToDevice);
        if (pass == 0)
          g[v] = v == 0   ? run(k_rounds<0>, d, d_tab, tab_n - 1, wg, lds, 400)
                 : v == 1 ? run(k_rounds<1>, d, d_tab, tab_n - 1, wg, lds, 400)
                 : v == 2 ? run(k_rounds<2>, d, d_tab, tab_n - 1, wg, lds, 400)
                 : v == 3 ? run(k_rounds<3>, d, d_tab, tab_n - 1, wg, lds, 400)
                          : run(k_rounds<4>, d, d_tab, tab_n - 1, wg, lds, 400);
        else
          g[v] = v == 0   ? run(k_mul_only<0>, d, d_tab, tab_n - 1, wg, lds, 400)
                 : v == 1 ? run(k_mul_only<1>, d, d_tab, tab_n - 1, wg, lds, 400)
                 : v == 2 ? run(k_mul_only<2>, d, d_tab, tab_n - 1, wg, lds, 400)
                 : v == 3 ? run(k_mul_only<3>, d, d_tab, tab_n - 1, wg, lds, 400)
                          : run(k_mul_only<4>, d, d_tab, tab_n - 1, wg, lds, 400);
      }
      for (int v = 0; v < 5; v++)
        printf("%-16s wg/CU=%d  %-40s %8.2f G mul/s  (x%.3f of shipped)\n", pass == 0 ? "radix-4 rounds" : "products only", wg,
               names[v], g[v], g[v] / g[0]);
    }
  }
  return 0;
}
