// Is v_mul_lo_u32 slower than v_mad_u64_u32 on gfx950?  The Montgomery digit m = (c * NINV) mod 2^29 needs only the
// low half of a 32 x 32 product; the compiler emits v_mul_lo_u32 for it.  Variant B computes it with v_mad_u64_u32.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
struct fe29 { uint32_t v[9]; };
#define M29 0x1fffffffu
static constexpr uint32_t MOD29[9] = {0x187cfd47u, 0x010460b6u, 0x1c72a34fu, 0x02d522d0u, 0x1585d978u, 0x02db40c0u, 0x00a6e141u, 0x0e5c2634u, 0x0030644eu};
__device__ __forceinline__ uint32_t mullo_mad(uint32_t a, uint32_t b) {
  uint64_t r, carry;
  asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(r), "=s"(carry) : "v"(a), "v"(b));
  return (uint32_t)r;
}
template <int VAR>
__device__ __forceinline__ fe29 mul29(const fe29& a, const fe29& b, uint32_t ninv) {
  uint64_t c[18];
#pragma unroll
  for (int k = 0; k < 18; k++) c[k] = 0;
#pragma unroll
  for (int i = 0; i < 9; i++)
#pragma unroll
    for (int j = 0; j < 9; j++) c[i + j] += (uint64_t)a.v[i] * b.v[j];
#pragma unroll
  for (int k = 0; k < 9; k++) {
    uint32_t m = (VAR ? mullo_mad((uint32_t)c[k], ninv) : (uint32_t)c[k] * ninv) & M29;
#pragma unroll
    for (int j = 0; j < 9; j++) c[k + j] += (uint64_t)m * MOD29[j];
    c[k + 1] += c[k] >> 29;
  }
  fe29 r;
#pragma unroll
  for (int k = 9; k < 17; k++) { r.v[k - 9] = (uint32_t)c[k] & M29; c[k + 1] += c[k] >> 29; }
  r.v[8] = (uint32_t)c[17];
  return r;
}
template <int VAR>
__global__ void k_mul(fe29* io, int iters, uint32_t ninv) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  fe29 x = io[i], y = io[i ^ 1];
  for (int k = 0; k < iters; k++) { x = mul29<VAR>(x, y, ninv); y = mul29<VAR>(y, x, ninv); }
  io[i] = x;
}
int main() {
  size_t n = 256 * 8 * 256; std::vector<fe29> h(n), r0(n), r1(n);
  for (size_t i = 0; i < n; i++) for (int j = 0; j < 9; j++) h[i].v[j] = (uint32_t)(i * 2654435761u + j * 40503u) & (j == 8 ? 0xfffffu : M29);
  fe29* d; hipMalloc(&d, n * sizeof(fe29));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int var = 0; var < 2; var++) for (int w : {2, 4, 8}) {
    hipMemcpy(d, h.data(), n * sizeof(fe29), hipMemcpyHostToDevice);
    if (var) hipLaunchKernelGGL(k_mul<1>, dim3(256 * w), dim3(256), 0, 0, d, 8, 0x04866389u); else hipLaunchKernelGGL(k_mul<0>, dim3(256 * w), dim3(256), 0, 0, d, 8, 0x04866389u);
    hipDeviceSynchronize();
    hipMemcpy((var ? r1 : r0).data(), d, 256 * 2 * sizeof(fe29), hipMemcpyDeviceToHost);
    hipEventRecord(e0);
    if (var) hipLaunchKernelGGL(k_mul<1>, dim3(256 * w), dim3(256), 0, 0, d, 1000, 0x04866389u); else hipLaunchKernelGGL(k_mul<0>, dim3(256 * w), dim3(256), 0, 0, d, 1000, 0x04866389u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("variant %s waves/SIMD=%d %8.3f ms %8.2f G mul/s\n", var ? "mad" : "mul_lo", w, ms, (double)256 * w * 256 * 2000 / ms * 1e-6);
  }
  int same = 1; for (int i = 0; i < 512; i++) for (int j = 0; j < 9; j++) same &= r0[i].v[j] == r1[i].v[j];
  printf("results identical: %d\n", same);
}
