// microbenchmark (round 5): f32-accurate GEMM steps on the bf16 matrix cores.
// An f32 value splits EXACTLY into three bf16 pieces a = a1 + a2 + a3 (8 + 8 + 8 mantissa bits);
// a * b = sum of nine piece products, of which the six with i + j <= 4 carry everything above
// 2^-24 relative: (1,1) (1,2) (2,1) (1,3) (3,1) (2,2).  Each piece product is exact in f32 and the
// MFMA accumulates in f32, so six v_mfma_f32_16x16x16_bf16 (8 cycles each at the 2.5 PF dense
// rate) stand where four v_mfma_f32_16x16x4_f32 (32 cycles each) stood: 48 against 128 matrix-pipe
// cycles for the same 16 x 16 x 16 block -- IF the VALU work of the splitting (5.5 operations
// per operand value) hides next to them.  Questions: (1) how exact is it, (2) what does a wave's
// MT x NT register tile sustain per step, splitting included, against the f32 form?
//   hipcc -O3 --offload-arch=gfx950 -o bf16x3 bf16x3.hip && ./bf16x3
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned cvt_pk(float lo, float hi) {
  union { bf16x2 v; unsigned u; } r;
  r.v = (bf16x2){(__bf16)lo, (__bf16)hi};
  return r.u;
}
// four f32 values (a lane's k slots of one 16-row block) -> three fragments of four bf16 each
struct Frag3 { s16x4 p[3]; };
__device__ __forceinline__ Frag3 split3(float v0, float v1, float v2, float v3) {
  Frag3 f;
  float r[4] = {v0, v1, v2, v3};
#pragma unroll
  for (int piece = 0; piece < 3; ++piece) {
    const unsigned u0 = cvt_pk(r[0], r[1]), u1 = cvt_pk(r[2], r[3]);
    union { unsigned u[2]; s16x4 v; } w;
    w.u[0] = u0; w.u[1] = u1;
    f.p[piece] = w.v;
    if (piece < 2) {
      r[0] -= __uint_as_float(u0 << 16); r[1] -= __uint_as_float(u0 & 0xffff0000u);
      r[2] -= __uint_as_float(u1 << 16); r[3] -= __uint_as_float(u1 & 0xffff0000u);
    }
  }
  return f;
}
__device__ __forceinline__ f32x4 mfma6(const Frag3& a, const Frag3& b, f32x4 c) {
  // smallest terms first
  c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a.p[1], b.p[1], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a.p[0], b.p[2], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a.p[2], b.p[0], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a.p[0], b.p[1], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a.p[1], b.p[0], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a.p[0], b.p[0], c, 0, 0, 0);
  return c;
}

// the six products TERM-major: consecutive MFMAs go to different accumulators (back-to-back MFMAs
// into ONE accumulator wait for each other's result: the first cut measured 16 cycles per MFMA)
template <int MT, int NT>
__device__ __forceinline__ void terms(const Frag3 (&a)[MT], const Frag3 (&b)[NT], f32x4 (&acc)[MT][NT]) {
  constexpr int TA[6] = {1, 0, 2, 0, 1, 0}, TB[6] = {1, 2, 0, 1, 0, 0};
#pragma unroll
  for (int t = 0; t < 6; ++t)
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n)
        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a[m].p[TA[t]], b[n].p[TB[t]], acc[m][n], 0, 0, 0);
}

// ---- 1. accuracy: C (16 x 16) = A (16 x K) * B (K x 16), three ways ---------------------------
// A[i][k], B[k][j] row-major in memory; lane (l15, q) holds k = 16 s + 4 q + e of row / column l15
__global__ void acc_kernel(const float* A, const float* B, int K, float* Cf32, float* Cx3, float* Cbf) {
  const int lane = threadIdx.x, l15 = lane & 15, q = lane >> 4;
  f32x4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0}, c2 = {0, 0, 0, 0};
  for (int s = 0; s < K / 16; ++s) {
    float a[4], b[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { a[e] = A[l15 * K + 16 * s + 4 * q + e]; b[e] = B[(16 * s + 4 * q + e) * 16 + l15]; }
#pragma unroll
    for (int e = 0; e < 4; ++e) c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[e], c0, 0, 0, 0);
    const Frag3 fa = split3(a[0], a[1], a[2], a[3]), fb = split3(b[0], b[1], b[2], b[3]);
    c1 = mfma6(fa, fb, c1);
    c2 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(fa.p[0], fb.p[0], c2, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    Cf32[(4 * q + r) * 16 + l15] = c0[r]; Cx3[(4 * q + r) * 16 + l15] = c1[r]; Cbf[(4 * q + r) * 16 + l15] = c2[r];
  }
}

// ---- 2. throughput: one wave per SIMD, MT x NT blocks per wave --------------------------------
// MODE 0: f32 form (4 x MT x NT MFMAs 16x16x4 per step); 1: split + 6 x MT x NT bf16 MFMAs;
// 2: the six MFMAs on fragments split ONCE outside the loop (matrix pipe alone); 3: the splitting
// alone (its results folded into one accumulator by a cheap xor so that nothing is dead)
template <int MT, int NT, int MODE>
__global__ __launch_bounds__(256) void thr_kernel(float* out, const float* in, int iters) {
  const int lane = threadIdx.x & 63;
  float av[MT][4], bv[NT][4];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int e = 0; e < 4; ++e) av[m][e] = in[(m * 4 + e) * 64 + lane];
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int e = 0; e < 4; ++e) bv[n][e] = in[(64 + n * 4 + e) * 64 + lane];
  f32x4 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[m][n] = (f32x4){0, 0, 0, 0};
  Frag3 fa0[MT], fb0[NT];
#pragma unroll
  for (int m = 0; m < MT; ++m) fa0[m] = split3(av[m][0], av[m][1], av[m][2], av[m][3]);
#pragma unroll
  for (int n = 0; n < NT; ++n) fb0[n] = split3(bv[n][0], bv[n][1], bv[n][2], bv[n][3]);
  unsigned sink = 0;
  for (int it = 0; it < iters; ++it) {
    // (new operand values every step, as loads would deliver them: one cheap op per register)
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int e = 0; e < 4; ++e) asm volatile("" : "+v"(av[m][e]));
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int e = 0; e < 4; ++e) asm volatile("" : "+v"(bv[n][e]));
    if constexpr (MODE == 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int n = 0; n < NT; ++n)
            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m][e], bv[n][e], acc[m][n], 0, 0, 0);
    } else if constexpr (MODE == 1) {
      Frag3 fb[NT];
#pragma unroll
      for (int n = 0; n < NT; ++n) fb[n] = split3(bv[n][0], bv[n][1], bv[n][2], bv[n][3]);
      Frag3 fa[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) fa[m] = split3(av[m][0], av[m][1], av[m][2], av[m][3]);
      terms<MT, NT>(fa, fb, acc);
    } else if constexpr (MODE == 2) {
      terms<MT, NT>(fa0, fb0, acc);
    } else {
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const Frag3 f = split3(bv[n][0], bv[n][1], bv[n][2], bv[n][3]);
        sink ^= (unsigned)f.p[0][0] ^ (unsigned)f.p[1][1] ^ (unsigned)f.p[2][2];
      }
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const Frag3 f = split3(av[m][0], av[m][1], av[m][2], av[m][3]);
        sink ^= (unsigned)f.p[0][0] ^ (unsigned)f.p[1][1] ^ (unsigned)f.p[2][2];
      }
    }
  }
  float s = (float)sink;
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) s += acc[m][n][0] + acc[m][n][1] + acc[m][n][2] + acc[m][n][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MT, int NT, int MODE>
double run(float* out, const float* in, int iters) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((thr_kernel<MT, NT, MODE>), dim3(256), dim3(256), 0, 0, out, in, iters);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL((thr_kernel<MT, NT, MODE>), dim3(256), dim3(256), 0, 0, out, in, iters);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e-3 / iters;                            // seconds per step
}

template <int MT, int NT>
void report(float* out, const float* in) {
  const int iters = 20000;
  const double flop = 2.0 * 16 * MT * 16 * NT * 16 * 1024;   // per step, all 1024 waves
  const double t0 = run<MT, NT, 0>(out, in, iters), t1 = run<MT, NT, 1>(out, in, iters);
  const double t2 = run<MT, NT, 2>(out, in, iters), t3 = run<MT, NT, 3>(out, in, iters);
  printf("%2d x %d blocks per wave: f32 form %6.1f ns / step = %6.1f TF | bf16x3 %6.1f ns = %6.1f TF-equivalent (x%.2f) | "
         "its MFMAs alone %6.1f ns, its splitting alone %6.1f ns\n", MT, NT, t0 * 1e9, flop / t0 / 1e12, t1 * 1e9,
         flop / t1 / 1e12, t0 / t1, t2 * 1e9, t3 * 1e9);
}

int main() {
  // ---- accuracy ---------------------------------------------------------------------------------
  const int K = 3200;
  std::vector<float> A(16 * K), B(K * 16);
  srand(1);
  for (auto& v : A) v = (float)rand() / RAND_MAX - 0.3f;
  for (auto& v : B) v = ((float)rand() / RAND_MAX - 0.5f) * 0.1f;
  float *dA, *dB, *dC;
  CHECK(hipMalloc(&dA, A.size() * 4)); CHECK(hipMalloc(&dB, B.size() * 4)); CHECK(hipMalloc(&dC, 3 * 256 * 4));
  CHECK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(acc_kernel, dim3(1), dim3(64), 0, 0, dA, dB, K, dC, dC + 256, dC + 512);
  std::vector<float> C(3 * 256);
  CHECK(hipMemcpy(C.data(), dC, 3 * 256 * 4, hipMemcpyDeviceToHost));
  double emax[3] = {0, 0, 0}, cmax = 0;
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j) {
      double r = 0;
      for (int k = 0; k < K; ++k) r += (double)A[i * K + k] * (double)B[k * 16 + j];
      cmax = fmax(cmax, fabs(r));
      for (int w = 0; w < 3; ++w) emax[w] = fmax(emax[w], fabs((double)C[w * 256 + i * 16 + j] - r));
    }
  printf("K = %d dot products, error / max |C| against float64: f32 MFMA %.2e | bf16x3 (six products) %.2e | one bf16 product %.2e\n",
         K, emax[0] / cmax, emax[1] / cmax, emax[2] / cmax);
  // ---- throughput ---------------------------------------------------------------------------------
  float *out, *in;
  CHECK(hipMalloc(&out, 256 * 256 * 4)); CHECK(hipMalloc(&in, 128 * 64 * 4));
  std::vector<float> hin(128 * 64);
  for (auto& v : hin) v = (float)rand() / RAND_MAX - 0.5f;
  CHECK(hipMemcpy(in, hin.data(), hin.size() * 4, hipMemcpyHostToDevice));
  report<7, 2>(out, in);
  report<3, 4>(out, in);
  report<13, 1>(out, in);
  report<5, 2>(out, in);
  report<2, 4>(out, in);
  return 0;
}
