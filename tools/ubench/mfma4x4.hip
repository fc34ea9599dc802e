// microbenchmark: v_mfma_f32_4x4x1_16B_f32 with the A operand broadcast from ONE block
// (CBSZ = 4, ABID = b): effectively a 4 (channels) x 64 (lanes = positions) x 1 (k)
// outer-product step -- M granularity 4 instead of 16.  Question: does a stream of them
// (8 cycles each) sustain the f32 matrix rate next to the LDS reads it needs?
//   hipcc -O3 --offload-arch=gfx950 -o mfma4x4 mfma4x4.hip && ./mfma4x4
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <utility>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_vp;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// ---- 1. semantics ---------------------------------------------------------------
template <int CBSZ, int ABID>
__global__ void sem_kernel(const float* a, const float* b, float* d) {
  const int lane = threadIdx.x;
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[lane], b[lane], c, CBSZ, ABID, 0);
#pragma unroll
  for (int r = 0; r < 4; ++r) d[lane * 4 + r] = c[r];
}

// ---- 2. throughput --------------------------------------------------------------
template <int OFF> __device__ __forceinline__ float lds_ld(unsigned a) {
  float v; asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(a), "i"(OFF)); return v;
}
template <int I, int MG, int KU, int NA>
__device__ __forceinline__ void steps(const float (&a)[NA], const float (&b)[KU], f32x4 (&acc)[MG]) {
  constexpr int k = I / MG, g = I % MG;
  constexpr int slot = k * MG + g;
  // (inline asm with the accumulator tied: hipcc otherwise rotates the accumulators
  // through v_accvgpr moves in this loop)
  asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0 cbsz:4 abid:%3"
               : "+v"(acc[g]) : "v"(a[slot / 16]), "v"(b[k]), "n"(slot % 16));
  if constexpr (I + 1 < MG * KU) steps<I + 1, MG, KU, NA>(a, b, acc);
}
// MODE: 0 = operands fixed in registers, 1 = B from LDS each k (asm ds_read, waited once per
// iteration, next iteration's reads issued before this iteration's MFMAs), 2 = 1 + A from
// global memory each iteration
template <int I, int KU>
__device__ __forceinline__ void read_b(float (&b)[KU], unsigned addr) {
  b[I] = lds_ld<I * 1024>(addr);
  if constexpr (I + 1 < KU) read_b<I + 1, KU>(b, addr);
}
template <int MG, int KU, int MODE>
__global__ __launch_bounds__(256) void thr_kernel(float* out, const float* gin, int iters, int zero) {
  __shared__ float smem[16 * 1024];
  constexpr int NA = (MG * KU + 15) / 16;
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 16 * 1024; i += 256) {
    unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u; h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
    smem[i] = zero ? 0.f : ((float)(h & 0xffffff) / 16777216.0f - 0.5f);
  }
  __syncthreads();
  f32x4 acc[MG];
#pragma unroll
  for (int g = 0; g < MG; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float a[NA], b0[KU], b1[KU];
#pragma unroll
  for (int i = 0; i < NA; ++i) a[i] = smem[(lane * 7 + i * 64) & 16383];
#pragma unroll
  for (int i = 0; i < KU; ++i) { b0[i] = smem[(lane + i * 256) & 16383]; b1[i] = b0[i]; }
  unsigned addr = (unsigned)(uintptr_t)(lds_vp)smem + 4u * lane + 256u * (tid >> 6);
  const float* ga = gin + lane;
  for (int it = 0; it < iters; it += 2) {
    if constexpr (MODE >= 1) { read_b<0, KU>(b1, addr + ((it & 7) << 4)); }
    if constexpr (MODE >= 2) {
#pragma unroll
      for (int i = 0; i < NA; ++i) a[i] = ga[((it & 15) * NA + i) * 64];
    }
    __builtin_amdgcn_sched_barrier(0);
    steps<0, MG, KU, NA>(a, b0, acc);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (MODE >= 1) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < KU; ++i) asm volatile("" : "+v"(b1[i]));
      __builtin_amdgcn_sched_barrier(0);
      read_b<0, KU>(b0, addr + (((it + 1) & 7) << 4));
    }
    __builtin_amdgcn_sched_barrier(0);
    steps<0, MG, KU, NA>(a, b1, acc);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (MODE >= 1) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < KU; ++i) asm volatile("" : "+v"(b0[i]));
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int g = 0; g < MG; ++g) s += acc[g][0] + acc[g][1] + acc[g][2] + acc[g][3];
  out[blockIdx.x * 256 + tid] = s;
}

// reference point: the 16x16x4 form, same structure (MT x 1 accumulators, fixed operands)
template <int MT>
__global__ __launch_bounds__(256) void thr16_kernel(float* out, int iters, int zero) {
  const int tid = threadIdx.x, lane = tid & 63;
  f32x4 acc[MT];
#pragma unroll
  for (int g = 0; g < MT; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
  unsigned h = (unsigned)(tid + 1) * 2654435761u + blockIdx.x * 40503u; h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
  float a = zero ? 0.f : ((float)(h & 0xffffff) / 16777216.0f - 0.5f), b = zero ? 0.f : a * 0.37f + 0.1f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < MT; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[g], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int g = 0; g < MT; ++g) s += acc[g][0] + acc[g][1] + acc[g][2] + acc[g][3];
  out[blockIdx.x * 256 + tid] = s;
}

template <typename F>
static double time_ms(F&& launch, int reps) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) launch();
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) launch();
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

template <int MG, int KU, int MODE>
static void run_thr(float* out, const float* gin, int zero) {
  const int blocks = 256 * 4, iters = 2048;
  double ms = time_ms([&] { hipLaunchKernelGGL((thr_kernel<MG, KU, MODE>), dim3(blocks), dim3(256), 0, 0, out, gin, iters, zero); }, 10);
  double flop = (double)blocks * 4 * iters * MG * KU * 512.0;
  printf("4x4x1 bcast  MG=%2d KU=%d mode=%d zero=%d : %.3f ms  %.1f TF/s (%.1f%% of 157.3)\n", MG, KU, MODE, zero, ms,
         flop / ms * 1e-9, flop / ms * 1e-9 / 157.3 * 100);
}
template <int MT>
static void run_thr16(float* out, int zero) {
  const int blocks = 256 * 4, iters = 4096;
  double ms = time_ms([&] { hipLaunchKernelGGL((thr16_kernel<MT>), dim3(blocks), dim3(256), 0, 0, out, iters, zero); }, 10);
  double flop = (double)blocks * 4 * iters * MT * 2048.0;
  printf("16x16x4      MT=%2d zero=%d : %.3f ms  %.1f TF/s (%.1f%%)\n", MT, zero, ms, flop / ms * 1e-9, flop / ms * 1e-9 / 157.3 * 100);
}

int main() {
  // semantics
  std::vector<float> ha(64), hb(64), hd(256);
  for (int i = 0; i < 64; ++i) { ha[i] = (float)(i + 1); hb[i] = (float)(100 + 3 * i); }
  float *da, *db, *dd;
  CHECK(hipMalloc(&da, 256)); CHECK(hipMalloc(&db, 256)); CHECK(hipMalloc(&dd, 1024));
  CHECK(hipMemcpy(da, ha.data(), 256, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(db, hb.data(), 256, hipMemcpyHostToDevice));
  auto check = [&](const char* name, int cbsz, int abid) {
    CHECK(hipMemcpy(hd.data(), dd, 1024, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane)
      for (int r = 0; r < 4; ++r) {
        const int blk = cbsz ? abid : lane / 4;
        const float ref = ha[4 * blk + r] * hb[lane];
        if (hd[lane * 4 + r] != ref) { if (bad < 4) printf("  lane %d r %d: got %g want %g\n", lane, r, hd[lane * 4 + r], ref); ++bad; }
      }
    printf("semantics %s: %s\n", name, bad ? "MISMATCH" : "ok (D[lane][r] = A[4*blk + r] * B[lane])");
  };
  hipLaunchKernelGGL((sem_kernel<0, 0>), dim3(1), dim3(64), 0, 0, da, db, dd); CHECK(hipDeviceSynchronize()); check("cbsz=0", 0, 0);
  hipLaunchKernelGGL((sem_kernel<4, 0>), dim3(1), dim3(64), 0, 0, da, db, dd); CHECK(hipDeviceSynchronize()); check("cbsz=4 abid=0", 4, 0);
  hipLaunchKernelGGL((sem_kernel<4, 5>), dim3(1), dim3(64), 0, 0, da, db, dd); CHECK(hipDeviceSynchronize()); check("cbsz=4 abid=5", 4, 5);
  hipLaunchKernelGGL((sem_kernel<4, 15>), dim3(1), dim3(64), 0, 0, da, db, dd); CHECK(hipDeviceSynchronize()); check("cbsz=4 abid=15", 4, 15);

  float *out, *gin;
  CHECK(hipMalloc(&out, 4 * 256 * 1024 * 4)); CHECK(hipMalloc(&gin, 4 << 20));
  CHECK(hipMemset(gin, 0, 4 << 20));
  for (int zero = 0; zero < 2; ++zero) {
    run_thr16<4>(out, zero);
    run_thr16<8>(out, zero);
    run_thr<5, 3, 0>(out, gin, zero);
    run_thr<10, 3, 0>(out, gin, zero);
    run_thr<25, 3, 0>(out, gin, zero);
    run_thr<5, 3, 1>(out, gin, zero);
    run_thr<10, 3, 1>(out, gin, zero);
    run_thr<25, 3, 1>(out, gin, zero);
    run_thr<5, 3, 2>(out, gin, zero);
    run_thr<10, 3, 2>(out, gin, zero);
    run_thr<25, 3, 2>(out, gin, zero);
    run_thr<38, 1, 1>(out, gin, zero);
    run_thr<50, 1, 2>(out, gin, zero);
  }
  CHECK(hipDeviceSynchronize());
  return 0;
}
