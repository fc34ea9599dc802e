// tr_read_check.hip -- what ds_read_b64_tr_b16 hands to each lane (gfx950), checked against
// the description in cdna_hip_programming.md T10: per group of 16 lanes, lane 4q+p supplies the
// address of row q, columns 4p..4p+3 of a 4 x 16 block of 16-bit elements; lane i receives
// column i, row q in element q.   hipcc --offload-arch=gfx950 tr_read_check.hip -o tr_read_check
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

__global__ void k(unsigned short* out) {
  __shared__ unsigned short img[64][64];          // value = 256 * row + col
  for (int i = threadIdx.x; i < 64 * 64; i += 64) img[i / 64][i % 64] = (unsigned short)((i / 64) * 256 + (i % 64));
  __syncthreads();
  const int lane = threadIdx.x, g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
  // group g reads the block with first row 4*g + 8, first column 16*(g & 1)
  const int r0 = 4 * g + 8, c0 = 16 * (g & 1);
  const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)&img[r0 + q][c0 + 4 * p];
  typedef unsigned v2 __attribute__((ext_vector_type(2)));
  v2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr));
  out[lane * 4 + 0] = (unsigned short)(v[0] & 0xffff);
  out[lane * 4 + 1] = (unsigned short)(v[0] >> 16);
  out[lane * 4 + 2] = (unsigned short)(v[1] & 0xffff);
  out[lane * 4 + 3] = (unsigned short)(v[1] >> 16);
}

int main() {
  unsigned short* d;
  hipMalloc(&d, 256 * 2);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  std::vector<unsigned short> h(256);
  hipMemcpy(h.data(), d, 512, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int lane = 0; lane < 64; ++lane) {
    const int g = lane >> 4, i = lane & 15, r0 = 4 * g + 8, c0 = 16 * (g & 1);
    for (int e = 0; e < 4; ++e) {
      const int want = (r0 + e) * 256 + c0 + i;            // column i, row e
      if (h[lane * 4 + e] != want) {
        if (bad < 8) printf("lane %d elem %d: got row %d col %d, expected row %d col %d\n", lane, e,
                            h[lane * 4 + e] >> 8, h[lane * 4 + e] & 255, r0 + e, c0 + i);
        ++bad;
      }
    }
  }
  printf(bad ? "tr read: %d mismatches\n" : "tr read semantics as described: ok\n", bad);
  return bad != 0;
}
