// microbenchmark: fp32 MFMA issue rate with / without interleaved LDS reads
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_vp;
template <int OFF> __device__ __forceinline__ float ld(unsigned a) {
  float v; asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(a), "i"(OFF)); return v;
}
#define NA 14
typedef const __attribute__((address_space(1))) void* gbl_vp;
template <int V>
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned long long* cyc, const float* gin) {
  __shared__ float sm[16384];
  for (int i = threadIdx.x; i < 16384; i += 256) sm[i] = (float)(i & 7);
  __syncthreads();
  f32x4 acc[NA];
  for (int i = 0; i < NA; ++i) acc[i] = (f32x4){0, 0, 0, 0};
  float a[NA], b = 1.0f;
  for (int i = 0; i < NA; ++i) a[i] = (float)i;
  unsigned base = (unsigned)(uintptr_t)(lds_vp)sm + 4u * (threadIdx.x & 63);
  float r[NA];
  for (int i = 0; i < NA; ++i) r[i] = 0.f;
  float4 pend0 = {0,0,0,0}, pend1 = {0,0,0,0};
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      if (V == 2) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(r[i], b, acc[i], 0, 0, 0);
      else acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b, acc[i], 0, 0, 0);
      if (V >= 1 && V <= 4) {
        float t;
        asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(t) : "v"(base), "i"(256 * (i % 16)));
        if (V == 1) a[i] = a[i];  // unused
        r[i] = t;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (V == 3) {   // 2 LDS-DMA (16 B/lane) per 14 MFMAs
      const float* g = gin + ((size_t)blockIdx.x * 65536 + (it & 63) * 1024 + threadIdx.x * 4);
      __builtin_amdgcn_global_load_lds((gbl_vp)g, (lds_vp)(sm + 8192 + (threadIdx.x >> 6) * 512), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gbl_vp)(g + 256), (lds_vp)(sm + 8192 + (threadIdx.x >> 6) * 512 + 256), 16, 0, 0);
    }
    if (V == 4) {   // 2 plain 16 B loads, written to LDS one iteration later
      const float* g = gin + ((size_t)blockIdx.x * 65536 + (it & 63) * 1024 + threadIdx.x * 4);
      *(float4*)(sm + 8192 + threadIdx.x * 8) = pend0;
      *(float4*)(sm + 8192 + threadIdx.x * 8 + 4) = pend1;
      pend0 = *(const float4*)g;
      pend1 = *(const float4*)(g + 1024);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    base += (it & 1) ? 4 : -4;
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < NA; ++i) s += acc[i][0] + acc[i][3] + r[i];
  s += pend0.x + pend1.y + sm[8192 + threadIdx.x];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[V] = t1 - t0;
}
int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 4 * 256 * 1024); hipMalloc(&cyc, 64); hipMemset(cyc, 0, 64);
  int iters = 2000;
  for (int rep = 0; rep < 2; ++rep) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms[5]; float* gin; hipMalloc(&gin, 256ull*65536*4 + 65536); hipMemset(gin, 0, 256ull*65536*4);
    hipEventRecord(e0); hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, out, iters, cyc, gin); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms[0], e0, e1);
    hipEventRecord(e0); hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, out, iters, cyc, gin); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms[1], e0, e1);
    hipEventRecord(e0); hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, out, iters, cyc, gin); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms[2], e0, e1);
    hipEventRecord(e0); hipLaunchKernelGGL(k<3>, dim3(256), dim3(256), 0, 0, out, iters, cyc, gin); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms[3], e0, e1);
    hipEventRecord(e0); hipLaunchKernelGGL(k<4>, dim3(256), dim3(256), 0, 0, out, iters, cyc, gin); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms[4], e0, e1);
    unsigned long long h[8]; hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    for (int v = 0; v < 5; ++v) {
      double nm = (double)iters * NA;
      double tf = 256.0 * 4 * nm * 2048 / (ms[v] * 1e-3) / 1e12;
      printf("V%d: %.3f ms, %.1f TF/s, %.1f cycles/MFMA (memtime)\n", v, ms[v], tf, (double)h[v] / nm);
    }
  }
  return 0;
}
