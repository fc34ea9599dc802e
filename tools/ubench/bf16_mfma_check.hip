// checks the operand layout assumed by the bf16 operand form: lane (l15, qd) holds
// k = 4*qd + j of row / column l15; D col = lane & 15, row = 4*(lane >> 4) + reg
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ inline unsigned cvt_pk(float lo, float hi) {
  union { bf16x2 v; unsigned u; } r;
  r.v = __builtin_convertvector((f32x2){lo, hi}, bf16x2);
  return r.u;
}
__global__ void k(const float* A, const float* B, float* D) {   // A[16][16] row-major (i,k), B[16][16] (k,j)
  const int lane = threadIdx.x, l15 = lane & 15, qd = lane >> 4;
  float a[4], b[4];
  for (int j = 0; j < 4; ++j) { a[j] = A[l15 * 16 + 4 * qd + j]; b[j] = B[(4 * qd + j) * 16 + l15]; }
  union { unsigned u[2]; s16x4 v; } Af, Bf;
  Af.u[0] = cvt_pk(a[0], a[1]); Af.u[1] = cvt_pk(a[2], a[3]);
  Bf.u[0] = cvt_pk(b[0], b[1]); Bf.u[1] = cvt_pk(b[2], b[3]);
  f32x4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(Af.v, Bf.v, acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[(4 * qd + r) * 16 + l15] = acc[r];
}
int main() {
  float hA[256], hB[256], hD[256];
  for (int i = 0; i < 256; ++i) { hA[i] = (float)((i * 7) % 13 - 6) / 4.f; hB[i] = (float)((i * 5) % 11 - 5) / 8.f; }
  float *dA, *dB, *dD;
  hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dD, 1024);
  hipMemcpy(dA, hA, 1024, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
  double maxerr = 0;
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
    double s = 0; for (int kk = 0; kk < 16; ++kk) s += (double)hA[i * 16 + kk] * hB[kk * 16 + j];
    maxerr = fmax(maxerr, fabs(s - hD[i * 16 + j]));
  }
  printf("max err %g  D[0][0..3] = %g %g %g %g\n", maxerr, hD[0], hD[1], hD[2], hD[3]);
  return maxerr < 1e-5 ? 0 : 1;
}
