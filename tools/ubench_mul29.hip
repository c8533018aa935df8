#include <hip/hip_runtime.h>
#include <stdint.h>
struct fe29 { uint32_t v[9]; };
#define M29 0x1fffffffu
__device__ __constant__ const uint32_t dummy[1]={0};
static constexpr uint32_t MOD29[9] = {0x187cfd47u, 0x010460b6u, 0x1c72a34fu, 0x02d522d0u, 0x1585d978u, 0x02db40c0u, 0x00a6e141u, 0x0e5c2634u, 0x0030644eu};
static constexpr uint32_t NINV29 = 0;  // patched below
template<uint32_t NINV>
__device__ __forceinline__ fe29 mul29(const fe29& a, const fe29& b){
  uint64_t c[18];
#pragma unroll
  for(int k=0;k<18;k++) c[k]=0;
#pragma unroll
  for(int i=0;i<9;i++)
#pragma unroll
    for(int j=0;j<9;j++) c[i+j] += (uint64_t)a.v[i]*b.v[j];
  // Montgomery reduction, digit by digit
#pragma unroll
  for(int k=0;k<9;k++){
    uint32_t m = ((uint32_t)c[k] * NINV) & M29;
#pragma unroll
    for(int j=0;j<9;j++) c[k+j] += (uint64_t)m*MOD29[j];
    c[k+1] += c[k]>>29;
  }
  fe29 r;
#pragma unroll
  for(int k=9;k<17;k++){ r.v[k-9] = (uint32_t)c[k] & M29; c[k+1] += c[k]>>29; }
  r.v[8] = (uint32_t)c[17];
  return r;
}
extern "C" __global__ void k_mul29(fe29* io, int iters){
  int i = blockIdx.x*blockDim.x+threadIdx.x;
  fe29 x = io[i], y = io[i^1];
  for(int k=0;k<iters;k++){ x = mul29<0x04866389u>(x,y); y = mul29<0x04866389u>(y,x);}
  io[i]=x;
}

#include <cstdio>
#include <vector>
int main(){
  size_t n=256*8*256; std::vector<fe29> h(n);
  for(size_t i=0;i<n;i++) for(int j=0;j<9;j++) h[i].v[j]=(uint32_t)(i*2654435761u+j*40503u)&(j==8?0xfffffu:M29);
  fe29* d; hipMalloc(&d,n*sizeof(fe29)); hipMemcpy(d,h.data(),n*sizeof(fe29),hipMemcpyHostToDevice);
  hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for(int w: {1,2,4,8}){
    hipLaunchKernelGGL(k_mul29,dim3(256*w),dim3(256),0,0,d,8); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(k_mul29,dim3(256*w),dim3(256),0,0,d,1000); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms,e0,e1);
    printf("mul29 waves/SIMD=%d %8.3f ms %8.2f G mul/s\n",w,ms,(double)256*w*256*2000/ms*1e-6);
  }
}
