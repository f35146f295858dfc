// Microbenchmark: FP32 VALU issue rate on gfx950 — v_fma_f32 (VGPR and SGPR coefficient) vs v_pk_fma_f32.
// Used to fix the roofline denominator stated in DESIGN.md. Not part of the product path.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)

typedef float float2v __attribute__((ext_vector_type(2)));

__global__ void __launch_bounds__(256) k_fma(float* out, const float* __restrict__ coef, int iters) {
  float a0=threadIdx.x, a1=a0+1, a2=a0+2, a3=a0+3, a4=a0+4, a5=a0+5, a6=a0+6, a7=a0+7;
  float x = 0.999f + 1e-6f*threadIdx.x;
  for (int i = 0; i < iters; ++i) {
    float c = coef[i & 63];            // wave-uniform -> SGPR
    a0 = __builtin_fmaf(c, x, a0); a1 = __builtin_fmaf(c, x, a1);
    a2 = __builtin_fmaf(c, x, a2); a3 = __builtin_fmaf(c, x, a3);
    a4 = __builtin_fmaf(c, x, a4); a5 = __builtin_fmaf(c, x, a5);
    a6 = __builtin_fmaf(c, x, a6); a7 = __builtin_fmaf(c, x, a7);
    a0 = __builtin_fmaf(c, a1, a0); a2 = __builtin_fmaf(c, a3, a2);
    a4 = __builtin_fmaf(c, a5, a4); a6 = __builtin_fmaf(c, a7, a6);
    a1 = __builtin_fmaf(c, a0, a1); a3 = __builtin_fmaf(c, a2, a3);
    a5 = __builtin_fmaf(c, a4, a5); a7 = __builtin_fmaf(c, a6, a7);
  }
  out[blockIdx.x*blockDim.x+threadIdx.x] = a0+a1+a2+a3+a4+a5+a6+a7;
}

__global__ void __launch_bounds__(256) k_pkfma(float* out, const float* __restrict__ coef, int iters) {
  float t = threadIdx.x;
  float2v a0={t,t+1}, a1={t+2,t+3}, a2={t+4,t+5}, a3={t+6,t+7}, a4={t+8,t+9}, a5={t+10,t+11}, a6={t+12,t+13}, a7={t+14,t+15};
  float2v x = {0.999f + 1e-6f*t, 0.998f + 1e-6f*t};
  for (int i = 0; i < iters; ++i) {
    float cs = coef[i & 63];
    float2v c = {cs, cs};
    a0 = __builtin_elementwise_fma(c, x, a0); a1 = __builtin_elementwise_fma(c, x, a1);
    a2 = __builtin_elementwise_fma(c, x, a2); a3 = __builtin_elementwise_fma(c, x, a3);
    a4 = __builtin_elementwise_fma(c, x, a4); a5 = __builtin_elementwise_fma(c, x, a5);
    a6 = __builtin_elementwise_fma(c, x, a6); a7 = __builtin_elementwise_fma(c, x, a7);
    a0 = __builtin_elementwise_fma(c, a1, a0); a2 = __builtin_elementwise_fma(c, a3, a2);
    a4 = __builtin_elementwise_fma(c, a5, a4); a6 = __builtin_elementwise_fma(c, a7, a6);
    a1 = __builtin_elementwise_fma(c, a0, a1); a3 = __builtin_elementwise_fma(c, a2, a3);
    a5 = __builtin_elementwise_fma(c, a4, a5); a7 = __builtin_elementwise_fma(c, a6, a7);
  }
  float2v s = a0+a1+a2+a3+a4+a5+a6+a7;
  out[blockIdx.x*blockDim.x+threadIdx.x] = s.x+s.y;
}

// min3 / cmp mix similar to the ray filter: 9 fma + 2 sub + min3 + 2 cmp per "test"
__global__ void __launch_bounds__(256) k_mix(float* out, const float* __restrict__ coef, int iters) {
  float dx = 0.1f + 1e-4f*threadIdx.x, dy = -0.2f + 1e-4f*threadIdx.x, dz = -0.5f;
  int cnt = 0;
  for (int i = 0; i < iters; ++i) {
    const float* c = coef + (i & 3) * 12;
    float U = __builtin_fmaf(c[2], dz, __builtin_fmaf(c[1], dy, c[0]*dx));
    float V = __builtin_fmaf(c[5], dz, __builtin_fmaf(c[4], dy, c[3]*dx));
    float N = __builtin_fmaf(c[8], dz, __builtin_fmaf(c[7], dy, c[6]*dx));
    float W = N - U - V;
    float mn = __builtin_fminf(__builtin_fminf(U, V), W);
    if ((mn >= c[10]) & (N >= c[9])) cnt++;
  }
  out[blockIdx.x*blockDim.x+threadIdx.x] = cnt;
}

int main() {
  int dev = 0; CK(hipSetDevice(dev));
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, dev));
  printf("device %s CUs %d clock %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
  float* out; float* coef; 
  int blocks = p.multiProcessorCount * 8, threads = 256;
  CK(hipMalloc(&out, sizeof(float)*blocks*threads)); CK(hipMalloc(&coef, sizeof(float)*64));
  float h[64]; for (int i=0;i<64;i++) h[i] = (i%12==10)? 1e30f : 1e-3f*(i%7) * ((i&1)?-1:1);
  CK(hipMemcpy(coef, h, sizeof(h), hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int iters = 1<<16;
  for (int variant = 0; variant < 3; ++variant) {
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0));
      if (variant==0) k_fma<<<blocks,threads>>>(out, coef, iters);
      else if (variant==1) k_pkfma<<<blocks,threads>>>(out, coef, iters);
      else k_mix<<<blocks,threads>>>(out, coef, iters);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      double lanes = (double)blocks*threads;
      if (variant==0) printf("v_fma_f32 (sgpr coef): %.3f ms  %.1f TFLOP/s\n", ms, lanes*iters*16*2/ms*1e-9);
      else if (variant==1) printf("v_pk_fma_f32       : %.3f ms  %.1f TFLOP/s\n", ms, lanes*iters*16*4/ms*1e-9);
      else printf("filter mix (15 valu)  : %.3f ms  %.2f Ttests/s  (%.1f T lane-instr/s)\n", ms, lanes*iters/ms*1e-9, lanes*iters*15/ms*1e-9);
    }
  }
  return 0;
}
