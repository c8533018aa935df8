// Schedules of the lazy 9 x 29-bit Montgomery multiplication on gfx950 (which one goes into field29.hpp?).
//   V0  row-wise: 18 column accumulators, carries by v_lshrrev_b64 + v_lshl_add_u64          (round-1 code)
//   V1  column-wise, one accumulator per column started from the carry of the column below: the carry add is the
//       addend of a v_mad_u64_u32 (inline asm: LLVM re-associates the C form back into V0)
//   V2  V1 with the 64-bit shift done as v_alignbit_b32 + v_lshrrev_b32
//   V3  V1 with two accumulators per column (products / reduction terms) joined by one 64-bit add: shorter chains
//   V4  column-wise in plain C with an empty input-only asm after every step: each partial sum gets a second use, which
//       stops LLVM's reassociation from rebuilding V0; compiler-generated multiply-adds, no hazard nops
// Also the raw issue rate of v_mad_u64_u32 (independent chains), which prices the multiplication ceiling:
//   ceiling [mul/s] = mad rate / 171  (81 product + 81 reduction + 9 digit multiply-adds).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_mulcol.hip -o tools/ubench_mulcol.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
struct fe29 { uint32_t v[9]; };
#define M29 0x1fffffffu
static constexpr uint32_t MOD29[9] = {0x187cfd47u, 0x010460b6u, 0x1c72a34fu, 0x02d522d0u, 0x1585d978u, 0x02db40c0u, 0x00a6e141u, 0x0e5c2634u, 0x0030644eu};
static constexpr uint32_t NINV = 0x04866389u;

__device__ __forceinline__ uint64_t mad(uint32_t a, uint32_t b, uint64_t c) {
  uint64_t r, carry;
  asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(r), "=s"(carry) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ uint64_t mad_k(uint32_t a, uint32_t k, uint64_t c) {  // k: compile-time constant -> SGPR
  uint64_t r, carry;
  asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(r), "=s"(carry) : "v"(a), "s"(k), "v"(c));
  return r;
}
__device__ __forceinline__ uint64_t mad0(uint32_t a, uint32_t b) {
  uint64_t r, carry;
  asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(r), "=s"(carry) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ uint64_t shr29(uint64_t x) { return x >> 29; }
__device__ __forceinline__ uint64_t shr29_align(uint64_t x) {
  uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32), rl, rh;
  asm("v_alignbit_b32 %0, %1, %2, 29" : "=v"(rl) : "v"(hi), "v"(lo));
  rh = hi >> 29;
  return ((uint64_t)rh << 32) | rl;
}

template <int VAR>
__device__ __forceinline__ fe29 mul29(const fe29& a, const fe29& b) {
  fe29 r;
  if constexpr (VAR == 4) {
#define KEEP(x) asm volatile("" ::"v"(x))
    uint32_t m[9];
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 17; k++) {
#pragma unroll
      for (int i = 0; i < 9; i++) {
        const int j = k - i;
        if (j >= 0 && j < 9) { acc += (uint64_t)a.v[i] * b.v[j]; KEEP(acc); }
      }
      if (k < 9) {
#pragma unroll
        for (int i = 0; i < k; i++) { acc += (uint64_t)m[i] * MOD29[k - i]; KEEP(acc); }
        m[k] = (uint32_t)mad0((uint32_t)acc, NINV) & M29;
        acc += (uint64_t)m[k] * MOD29[0];
      } else {
#pragma unroll
        for (int i = k - 8; i < 9; i++) { acc += (uint64_t)m[i] * MOD29[k - i]; KEEP(acc); }
        r.v[k - 9] = (uint32_t)acc & M29;
      }
      acc >>= 29;
    }
    r.v[8] = (uint32_t)acc;
  } else if constexpr (VAR == 0) {
    uint64_t c[18];
#pragma unroll
    for (int k = 0; k < 18; k++) c[k] = 0;
#pragma unroll
    for (int i = 0; i < 9; i++)
#pragma unroll
      for (int j = 0; j < 9; j++) c[i + j] += (uint64_t)a.v[i] * b.v[j];
#pragma unroll
    for (int k = 0; k < 9; k++) {
      uint32_t m = (uint32_t)mad0((uint32_t)c[k], NINV) & M29;
#pragma unroll
      for (int j = 0; j < 9; j++) c[k + j] += (uint64_t)m * MOD29[j];
      c[k + 1] += c[k] >> 29;
    }
#pragma unroll
    for (int k = 9; k < 17; k++) { r.v[k - 9] = (uint32_t)c[k] & M29; c[k + 1] += c[k] >> 29; }
    r.v[8] = (uint32_t)c[17];
  } else {
    uint32_t m[9];
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 17; k++) {
      uint64_t acc2 = 0;
      bool first2 = true;
#pragma unroll
      for (int i = 0; i < 9; i++) {
        const int j = k - i;
        if (j >= 0 && j < 9) acc = mad(a.v[i], b.v[j], acc);
      }
      auto red = [&](uint32_t mi, uint32_t pj) {
        if constexpr (VAR == 3) {
          acc2 = first2 ? mad_k(mi, pj, 0) : mad_k(mi, pj, acc2);
          first2 = false;
        } else {
          acc = mad_k(mi, pj, acc);
        }
      };
      if (k < 9) {
#pragma unroll
        for (int i = 0; i < k; i++) red(m[i], MOD29[k - i]);
        if constexpr (VAR == 3) { if (!first2) acc += acc2; }
        m[k] = (uint32_t)mad0((uint32_t)acc, NINV) & M29;
        acc = mad_k(m[k], MOD29[0], acc);
      } else {
#pragma unroll
        for (int i = k - 8; i < 9; i++) red(m[i], MOD29[k - i]);
        if constexpr (VAR == 3) { if (!first2) acc += acc2; }
        r.v[k - 9] = (uint32_t)acc & M29;
      }
      acc = (VAR == 2) ? shr29_align(acc) : shr29(acc);
    }
    r.v[8] = (uint32_t)acc;
  }
  return r;
}
template <int VAR>
__global__ __launch_bounds__(256) void k_mul(fe29* io, int iters) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  fe29 x = io[i], y = io[i ^ 1];
  for (int k = 0; k < iters; k++) { x = mul29<VAR>(x, y); y = mul29<VAR>(y, x); }
  io[i] = x;
}
// two independent multiplication streams per thread (what the mixed addition offers the scheduler)
template <int VAR>
__global__ __launch_bounds__(256) void k_mul2(fe29* io, int iters) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  fe29 x = io[i], y = io[i ^ 1], u = io[i ^ 2], w = io[i ^ 3];
  for (int k = 0; k < iters; k++) { x = mul29<VAR>(x, y); u = mul29<VAR>(u, w); y = mul29<VAR>(y, x); w = mul29<VAR>(w, u); }
  for (int j = 0; j < 9; j++) x.v[j] ^= u.v[j];
  io[i] = x;
}
// raw v_mad_u64_u32 issue rate: CH independent accumulate chains per thread
template <int CH>
__global__ __launch_bounds__(256) void k_madrate(uint64_t* io, int iters) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t acc[CH];
  uint32_t a = (uint32_t)io[i], b = (uint32_t)(io[i] >> 32) | 1;
#pragma unroll
  for (int c = 0; c < CH; c++) acc[c] = io[i] + c;
  for (int k = 0; k < iters; k++) {
#pragma unroll
    for (int rep = 0; rep < 8; rep++)
#pragma unroll
      for (int c = 0; c < CH; c++) acc[c] = mad(a, b + c, acc[c]);
  }
  uint64_t s = 0;
#pragma unroll
  for (int c = 0; c < CH; c++) s ^= acc[c];
  io[i] = s;
}
// 64-bit shift / add rates next to it (are they full rate?)
template <int OP>
__global__ __launch_bounds__(256) void k_oprate(uint64_t* io, int iters) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t acc[8];
#pragma unroll
  for (int c = 0; c < 8; c++) acc[c] = io[i] + c * 0x9E3779B97F4A7C15ull;
  for (int k = 0; k < iters; k++) {
#pragma unroll
    for (int rep = 0; rep < 8; rep++)
#pragma unroll
      for (int c = 0; c < 8; c++) {
        if (OP == 0) asm("v_lshrrev_b64 %0, 3, %1" : "=v"(acc[c]) : "v"(acc[c] | (1ull << 63)));
        if (OP == 1) asm("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(acc[c]) : "v"(acc[c]), "v"(acc[(c + 1) & 7]));
        if (OP == 2) { uint32_t lo = (uint32_t)acc[c], hi = (uint32_t)(acc[c] >> 32), r; asm("v_alignbit_b32 %0, %1, %2, 29" : "=v"(r) : "v"(hi), "v"(lo)); acc[c] = ((uint64_t)hi << 32) | r; }
        if (OP == 3) { uint32_t lo = (uint32_t)acc[c], r; asm("v_and_b32 %0, 0x1fffffff, %1" : "=v"(r) : "v"(lo + c)); acc[c] = (acc[c] & 0xffffffff00000000ull) | r; }
      }
  }
  uint64_t s = 0;
#pragma unroll
  for (int c = 0; c < 8; c++) s ^= acc[c];
  io[i] = s;
}

#define TIME(ms, ...) do { hipEventRecord(e0); __VA_ARGS__; hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); } while (0)

int main() {
  size_t n = 256 * 8 * 256; std::vector<fe29> h(n), ref(512), got(512);
  for (size_t i = 0; i < n; i++) for (int j = 0; j < 9; j++) h[i].v[j] = (uint32_t)(i * 2654435761u + j * 40503u) & (j == 8 ? 0xfffffu : M29);
  fe29* d; hipMalloc(&d, n * sizeof(fe29));
  uint64_t* d64; hipMalloc(&d64, n * sizeof(uint64_t)); hipMemset(d64, 0x5a, n * sizeof(uint64_t));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms;
  auto run = [&](int var, int w, int iters, bool two) {
    dim3 g(256 * w), b(256);
    switch (var * 2 + (two ? 1 : 0)) {
      case 0: hipLaunchKernelGGL(k_mul<0>, g, b, 0, 0, d, iters); break;
      case 1: hipLaunchKernelGGL(k_mul2<0>, g, b, 0, 0, d, iters); break;
      case 2: hipLaunchKernelGGL(k_mul<1>, g, b, 0, 0, d, iters); break;
      case 3: hipLaunchKernelGGL(k_mul2<1>, g, b, 0, 0, d, iters); break;
      case 4: hipLaunchKernelGGL(k_mul<2>, g, b, 0, 0, d, iters); break;
      case 5: hipLaunchKernelGGL(k_mul2<2>, g, b, 0, 0, d, iters); break;
      case 6: hipLaunchKernelGGL(k_mul<3>, g, b, 0, 0, d, iters); break;
      case 7: hipLaunchKernelGGL(k_mul2<3>, g, b, 0, 0, d, iters); break;
      case 8: hipLaunchKernelGGL(k_mul<4>, g, b, 0, 0, d, iters); break;
      case 9: hipLaunchKernelGGL(k_mul2<4>, g, b, 0, 0, d, iters); break;
    }
  };
  for (int var = 0; var < 5; var++) {
    hipMemcpy(d, h.data(), n * sizeof(fe29), hipMemcpyHostToDevice);
    run(var, 2, 8, false);
    hipDeviceSynchronize();
    hipMemcpy((var ? got : ref).data(), d, 512 * sizeof(fe29), hipMemcpyDeviceToHost);
    if (var) { int same = 1; for (int i = 0; i < 512; i++) for (int j = 0; j < 9; j++) same &= ref[i].v[j] == got[i].v[j]; printf("V%d results identical to V0: %d\n", var, same); }
    for (int two = 0; two < 2; two++) for (int w : {1, 2, 3, 4, 8}) {
      hipMemcpy(d, h.data(), n * sizeof(fe29), hipMemcpyHostToDevice);
      TIME(ms, run(var, w, 1000, two));
      double muls = (double)256 * w * 256 * 2000 * (two ? 2 : 1);
      printf("V%d %s waves/SIMD=%d %8.3f ms %8.2f G mul/s\n", var, two ? "2 streams" : "1 stream ", w, ms, muls / ms * 1e-6);
    }
  }
  for (int w : {1, 2, 4, 8}) {
    dim3 g(256 * w), b(256);
    const int iters = 2000;
    TIME(ms, hipLaunchKernelGGL(k_madrate<1>, g, b, 0, 0, d64, iters));
    printf("v_mad_u64_u32 1 chain  waves/SIMD=%d %8.3f ms %8.2f T lane-ops/s\n", w, ms, (double)256 * w * 256 * iters * 8 * 1 / ms * 1e-9);
    TIME(ms, hipLaunchKernelGGL(k_madrate<4>, g, b, 0, 0, d64, iters));
    printf("v_mad_u64_u32 4 chains waves/SIMD=%d %8.3f ms %8.2f T lane-ops/s\n", w, ms, (double)256 * w * 256 * iters * 8 * 4 / ms * 1e-9);
    TIME(ms, hipLaunchKernelGGL(k_madrate<8>, g, b, 0, 0, d64, iters));
    printf("v_mad_u64_u32 8 chains waves/SIMD=%d %8.3f ms %8.2f T lane-ops/s\n", w, ms, (double)256 * w * 256 * iters * 8 * 8 / ms * 1e-9);
  }
  const char* names[4] = {"v_lshrrev_b64", "v_lshl_add_u64", "v_alignbit_b32", "v_and_b32"};
  for (int op = 0; op < 4; op++) for (int w : {4, 8}) {
    dim3 g(256 * w), b(256);
    const int iters = 2000;
    switch (op) {
      case 0: TIME(ms, hipLaunchKernelGGL(k_oprate<0>, g, b, 0, 0, d64, iters)); break;
      case 1: TIME(ms, hipLaunchKernelGGL(k_oprate<1>, g, b, 0, 0, d64, iters)); break;
      case 2: TIME(ms, hipLaunchKernelGGL(k_oprate<2>, g, b, 0, 0, d64, iters)); break;
      case 3: TIME(ms, hipLaunchKernelGGL(k_oprate<3>, g, b, 0, 0, d64, iters)); break;
    }
    printf("%-15s 8 chains waves/SIMD=%d %8.3f ms %8.2f T lane-ops/s\n", names[op], w, ms, (double)256 * w * 256 * iters * 64 / ms * 1e-9);
  }
  return 0;
}
