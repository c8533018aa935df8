// Instruction / field-op throughput microbenchmark for gfx950 (MI355X).
// Answers the design questions in DESIGN.md "K1": what does v_mad_u64_u32 cost
// relative to full-rate VALU, and how many Montgomery multiplications per
// second does the chip sustain at 1/2/4/8 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench.hip -o tools/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../cap_amd/csrc/field.hpp"
using namespace cap;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int OP>
__global__ void k_inst(uint32_t* out, int iters) {
  uint32_t a = threadIdx.x * 2654435761u + 12345u, b = blockIdx.x * 40503u + 7u;
  uint64_t acc[8];
  double d[8];
#pragma unroll
  for (int i = 0; i < 8; i++) { acc[i] = a + i; d[i] = (double)(a + i); }
  for (int k = 0; k < iters; k++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if (OP == 0) {  // v_mad_u64_u32
        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc");
      } else if (OP == 1) {  // v_mul_lo_u32
        uint32_t x = (uint32_t)acc[i];
        asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(b));
        acc[i] = x;
      } else if (OP == 2) {  // v_mul_hi_u32
        uint32_t x = (uint32_t)acc[i];
        asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x) : "v"(b));
        acc[i] = x;
      } else if (OP == 3) {  // v_add_u32 (full-rate reference)
        uint32_t x = (uint32_t)acc[i];
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(b));
        acc[i] = x;
      } else if (OP == 4) {  // v_addc_co_u32 chain element
        uint32_t x = (uint32_t)acc[i];
        asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(x) : "v"(b) : "vcc");
        acc[i] = x;
      } else if (OP == 5) {  // v_fma_f64
        asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
      } else if (OP == 6) {  // v_mad_u32_u24
        uint32_t x = (uint32_t)acc[i];
        asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(x) : "v"(b));
        acc[i] = x;
      } else if (OP == 7) {  // v_lshl_add_u64
        asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[i]) : "v"(acc[(i + 1) & 7]));
      } else if (OP == 8) {  // v_mul_hi_u32_u24
        uint32_t x = (uint32_t)acc[i];
        asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(x) : "v"(b));
        acc[i] = x;
      } else if (OP == 9) {  // v_mov_b32
        uint32_t x = (uint32_t)acc[i];
        asm volatile("v_mov_b32 %0, %1" : "+v"(x) : "v"(b));
        acc[i] = x;
      }
    }
  }
  uint64_t s = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) s += acc[i] + (uint64_t)d[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s;
}

template <class F, int OP>
__global__ void k_field(fe* io, int iters) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  fe x = io[i], y = io[i ^ 1];
  for (int k = 0; k < iters; k++) {
    if (OP == 0) x = F::mul(x, y);
    if (OP == 1) { x = F::add(x, y); y = F::sub(y, x); }
    if (OP == 2) { x = F::mul(x, y); y = F::mul(y, x); }  // 2 independent-ish chains
  }
  io[i] = OP == 1 ? F::add(x, y) : x;
}

static const char* opn[] = {"v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_add_u32", "v_addc_co_u32",
                            "v_fma_f64", "v_mad_u32_u24", "v_lshl_add_u64", "v_mul_hi_u32_u24", "v_mov_b32"};

template <int OP>
void run_inst(uint32_t* d_out, int waves_per_simd) {
  int block = 256;                      // 4 waves: one per SIMD
  int grid = 256 * waves_per_simd;      // 256 CUs
  int iters = 262144;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k_inst<OP>, dim3(grid), dim3(block), 0, 0, d_out, 64);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(k_inst<OP>, dim3(grid), dim3(block), 0, 0, d_out, iters);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  double inst = (double)grid * (block / 64) * iters * 8;  // wave-instructions
  double per_simd_per_s = inst / (ms * 1e-3) / (256.0 * 4);
  printf("%-18s waves/SIMD=%d  %8.3f ms  %7.3f G wave-inst/s/SIMD  => %5.2f cycles/wave-inst @2.4GHz  (%.2f T lane-ops/s chip)\n",
         opn[OP], waves_per_simd, ms, per_simd_per_s * 1e-9, 2.4e9 / per_simd_per_s, inst * 64 / (ms * 1e-3) * 1e-12);
}

template <class F, int OP>
void run_field(fe* d_io, const char* name, int waves_per_simd, int opsPerIter) {
  int block = 256;
  int grid = 256 * waves_per_simd;
  int iters = 2048;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k_field<F, OP>), dim3(grid), dim3(block), 0, 0, d_io, 16);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((k_field<F, OP>), dim3(grid), dim3(block), 0, 0, d_io, iters);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  double ops = (double)grid * block * iters * opsPerIter;
  printf("%-18s waves/SIMD=%d  %8.3f ms  %8.2f G field-ops/s chip   latency/op/wave %.0f ns\n", name, waves_per_simd, ms,
         ops / (ms * 1e-3) * 1e-9, ms * 1e6 / (iters * opsPerIter));
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("device: %s  CUs=%d  clock=%d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
  uint32_t* d_out; CK(hipMalloc(&d_out, 256 * 8 * 256 * 4));
  for (int w : {1, 2, 4, 8}) {
    run_inst<0>(d_out, w); run_inst<1>(d_out, w); run_inst<2>(d_out, w); run_inst<3>(d_out, w); run_inst<4>(d_out, w);
    run_inst<5>(d_out, w); run_inst<6>(d_out, w); run_inst<7>(d_out, w); run_inst<8>(d_out, w); run_inst<9>(d_out, w);
  }
  size_t n = 256 * 8 * 256;
  std::vector<fe> h(n);
  for (size_t i = 0; i < n; i++) for (int j = 0; j < 8; j++) h[i].v[j] = (uint32_t)(i * 2654435761u + j * 40503u) & (j == 7 ? 0x1fffffffu : 0xffffffffu);
  fe* d_io; CK(hipMalloc(&d_io, n * sizeof(fe)));
  for (int w : {1, 2, 4, 8}) {
    CK(hipMemcpy(d_io, h.data(), n * sizeof(fe), hipMemcpyHostToDevice));
    run_field<Fq, 0>(d_io, "Fq::mul chain", w, 1);
    run_field<Fq, 2>(d_io, "Fq::mul 2chains", w, 2);
    run_field<Fq, 1>(d_io, "Fq::add+sub", w, 2);
  }
  return 0;
}
