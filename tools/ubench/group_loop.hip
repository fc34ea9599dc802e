// microbenchmark: the igemm (cg,ty)-group loop in isolation (LDS reads + fp32 MFMA),
// to find where the compute phase loses matrix-pipe density.
//   hipcc -O3 --offload-arch=gfx950 -o group_loop group_loop.hip && ./group_loop
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <utility>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_vp;
template <int OFF> __device__ __forceinline__ float lds_ld(unsigned a) {
  float v; asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(a), "i"(OFF)); return v;
}
template <int MT, int NT, int KW, int ASTRIDE>
struct GroupRegs {
  float a[KW][MT];
  float b[KW][NT];
  __device__ __forceinline__ void touch() {
#pragma unroll
    for (int t = 0; t < KW; ++t) {
#pragma unroll
      for (int mb = 0; mb < MT; ++mb) asm volatile("" : "+v"(a[t][mb]));
#pragma unroll
      for (int nb = 0; nb < NT; ++nb) asm volatile("" : "+v"(b[t][nb]));
    }
  }
};
struct GA { const float* base; unsigned voff[8]; };
__device__ GA g_ga_dummy;
template <int OFF> __device__ __forceinline__ float gl_ld(const float* sbase, unsigned voff) {
  float v; asm volatile("global_load_dword %0, %1, %2 offset:%3" : "=v"(v) : "v"(voff), "s"(sbase), "i"(OFF)); return v;
}
template <int MT, int NT, int KW, int ASTRIDE, int R, bool GLB>
__device__ __forceinline__ void group_read(GroupRegs<MT, NT, KW, ASTRIDE>& g, unsigned addrA,
                                           const unsigned (&addrB)[NT], const GA& ga) {
  constexpr int tx = R / (MT + NT), k = R % (MT + NT);
  if constexpr (k < MT) {
    if constexpr (GLB) g.a[tx][k] = gl_ld<k * 64>(ga.base, ga.voff[tx]);
    else g.a[tx][k] = lds_ld<tx * ASTRIDE + k * 64>(addrA);
  } else g.b[tx][k - MT] = lds_ld<tx * 4>(addrB[k - MT]);
}
template <int MT, int NT, int KW, int ASTRIDE, int R0, int R1, bool GLB>
__device__ __forceinline__ void group_reads(GroupRegs<MT, NT, KW, ASTRIDE>& g, unsigned addrA,
                                            const unsigned (&addrB)[NT], const GA& ga) {
  if constexpr (R0 < R1) {
    group_read<MT, NT, KW, ASTRIDE, R0, GLB>(g, addrA, addrB, ga);
    group_reads<MT, NT, KW, ASTRIDE, R0 + 1, R1, GLB>(g, addrA, addrB, ga);
  }
}
// MODE bits: 1 = no LDS reads, 2 = no lgkmcnt wait, 4 = barrier per chunk, 8 = reads spread over ALL mfmas (not 3/4)
template <int MT, int NT, int KW, int ASTRIDE, int MODE, int I>
__device__ __forceinline__ void group_steps(const GroupRegs<MT, NT, KW, ASTRIDE>& cur,
                                            GroupRegs<MT, NT, KW, ASTRIDE>& nxt, f32x4 (&acc)[MT][NT],
                                            unsigned addrA, const unsigned (&addrB)[NT], const GA& ga) {
  constexpr int M = KW * MT * NT, R = KW * (MT + NT);
  constexpr int tx = I / (MT * NT), mb = (I / NT) % MT, nb = I % NT;
  acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.a[tx][mb], cur.b[tx][nb], acc[mb][nb], 0, 0, 0);
  constexpr int num = (MODE & 8) ? 1 : 4, den = (MODE & 8) ? 1 : 3;
  constexpr int r0 = (I * R * num) / (den * M) < R ? (I * R * num) / (den * M) : R;
  constexpr int r1 = ((I + 1) * R * num) / (den * M) < R ? ((I + 1) * R * num) / (den * M) : R;
  if constexpr (!(MODE & 1)) group_reads<MT, NT, KW, ASTRIDE, r0, r1, (MODE & 128) != 0>(nxt, addrA, addrB, ga);
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (I + 1 < M) group_steps<MT, NT, KW, ASTRIDE, MODE, I + 1>(cur, nxt, acc, addrA, addrB, ga);
}
constexpr int bmpad(int MT) { return ((16 * MT) & 31) == 16 ? 16 * MT : 16 * MT + 16; }

template <int MT, int NT, int KW, int MODE, int NWAVE>
__global__ __launch_bounds__(64 * NWAVE) void kb(float* out, int nChunk, int nG, int kh, int isY, int Lpad,
                                           unsigned long long* cyc, int zero_data, const float* gin, int ndma) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int BMpad = bmpad(MT);
  const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3;
  const int l15 = lane & 15, qd = lane >> 4;
  for (int i = tid; i < 36 * 1024; i += 64 * NWAVE) {
    unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u; h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
    smem[i] = zero_data ? 0.f : ((float)(h & 0xffffff) / 16777216.0f - 0.5f);
  }
  __syncthreads();
  f32x4 acc[MT][NT];
#pragma unroll
  for (int mb = 0; mb < MT; ++mb)
#pragma unroll
    for (int nb = 0; nb < NT; ++nb) acc[mb][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
  constexpr int ASTR = 4 * BMpad * 4;
  const unsigned xbase = (unsigned)(uintptr_t)(lds_vp)smem;
  const unsigned wbase = xbase + 4u * 8 * Lpad;
  int posoff[NT];
#pragma unroll
  for (int nb = 0; nb < NT; ++nb) posoff[nb] = wave * 16 * NT + nb * 16 + l15 + qd * Lpad;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  typedef const __attribute__((address_space(1))) void* gbl_vp;
  const int wave8 = tid >> 6;
  const bool producer = (MODE & 16) && wave8 >= 4;
  // DMA destination: upper LDS region [100 KB, 140 KB), never read by the compute loop
  float* dmabase = smem + 25 * 1024;
  const float* gsrc = gin + (size_t)blockIdx.x * 16384 + (size_t)lane * 4;
  for (int ch = 0; ch < nChunk; ++ch) {
    if ((MODE & 48) && (producer || (MODE & 32))) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (MODE & 4) __syncthreads();
    if ((MODE & 16) && producer) {
      for (int d = wave8 - 4; d < ndma; d += 4)
        __builtin_amdgcn_global_load_lds((gbl_vp)(gsrc + ((d + ch) & 63) * 256), (lds_vp)(dmabase + (d % 40) * 256), 16, 0, 0);
      continue;
    }
    if (MODE & 32) {
      for (int d = wave8; d < ndma; d += 4)
        __builtin_amdgcn_global_load_lds((gbl_vp)(gsrc + ((d + ch) & 63) * 256), (lds_vp)(dmabase + (d % 40) * 256), 16, 0, 0);
    }
    if (MODE & 64) {   // plain loads -> regs -> ds_write_b128 one chunk later
      for (int d = wave8; d < ndma; d += 4) {
        float4 v = *(const float4*)(gsrc + ((d + ch) & 63) * 256);
        *(float4*)(dmabase + (d % 40) * 256 + lane * 4) = v;
      }
    }
    GroupRegs<MT, NT, KW, ASTR> g0, g1;
    unsigned addrA = wbase + 4u * (qd * BMpad + l15);
    GA ga; ga.base = gin + 1024 * 1024 + (ch % 25) * (6 * KW * 4 * 224);   // weights image: rows of 224 floats, L2 resident
#pragma unroll
    for (int t = 0; t < 8; ++t) ga.voff[t] = 4u * (unsigned)((t * 4 + qd) * 224 + l15);
    unsigned addrB[NT];
#pragma unroll
    for (int nb = 0; nb < NT; ++nb) addrB[nb] = xbase + 4u * (unsigned)posoff[nb];
    const unsigned stepY = 4u * (unsigned)isY;
    const unsigned wrapCg = 4u * (unsigned)(4 * Lpad) - (unsigned)kh * stepY;
    int ty = 0;
#define NEXT() { addrA += KW * ASTR; ga.base += KW * 4 * 224; ++ty; const unsigned d = (ty == kh) ? (stepY + wrapCg) : stepY; \
      ty = (ty == kh) ? 0 : ty; _Pragma("unroll") for (int nb = 0; nb < NT; ++nb) addrB[nb] += d; }
#define WAIT() if (!(MODE & 2)) { if (MODE & 128) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); } __builtin_amdgcn_sched_barrier(0);
    group_reads<MT, NT, KW, ASTR, 0, KW*(MT+NT), (MODE & 128) != 0>(g0, addrA, addrB, ga);
    if (MODE & 128) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
    g0.touch();
    if (MODE & 1) { g1 = g0; }
    for (int g = 0; g + 1 < nG; g += 2) {
      NEXT()
      __builtin_amdgcn_sched_barrier(0);
      group_steps<MT, NT, KW, ASTR, MODE, 0>(g0, g1, acc, addrA, addrB, ga);
      WAIT()
      g1.touch();
      NEXT()
      __builtin_amdgcn_sched_barrier(0);
      group_steps<MT, NT, KW, ASTR, MODE, 0>(g1, g0, acc, addrA, addrB, ga);
      WAIT()
      g0.touch();
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
#pragma unroll
  for (int mb = 0; mb < MT; ++mb)
#pragma unroll
    for (int nb = 0; nb < NT; ++nb) s += acc[mb][nb][0] + acc[mb][nb][3];
  if (producer) return;
  out[blockIdx.x * 64 * NWAVE + threadIdx.x] = s + dmabase[tid];
  if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = r1 - r0; }
}

template <int MT, int NT, int KW, int MODE, int NWAVE>
void run(const char* name, int nChunk, int nG, int kh, float* out, unsigned long long* cyc, int zero_data = 0, int ndma = 0) {
  static float* gin = nullptr; if (!gin) { hipMalloc(&gin, 256 * 16384 * 4 + (1 << 24)); hipMemset(gin, 0, 256 * 16384 * 4 + (1 << 24)); }
  hipFuncSetAttribute(reinterpret_cast<const void*>(&kb<MT, NT, KW, MODE, NWAVE>),
                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 6; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((kb<MT, NT, KW, MODE, NWAVE>), dim3(256), dim3(64 * NWAVE), 150 * 1024, 0, out, nChunk, nG, kh, 41, 528, cyc, zero_data, gin, ndma);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
  }
  unsigned long long hh[2]; hipMemcpy(hh, cyc, 16, hipMemcpyDeviceToHost); unsigned long long h = hh[0];
  double nm = (double)nChunk * nG * KW * MT * NT;
  double ghz = (double)hh[0] / ((double)hh[1] * 10.0);
  double tf = 256.0 * 4 * ((MODE & 16) ? 1 : NWAVE / 4) * nm * 2048 / (ms * 1e-3) / 1e12;
  printf("%-40s %dx%dx%d w=%d nG=%2d %s: %.1f cyc/MFMA  %.3f ms  density %.1f%%  clock %.2f GHz  %.1f TF/s\n", name, MT, NT, KW, NWAVE, nG, zero_data ? "ZERO" : "RAND",
         (double)h / nm, ms, 100.0 * nm * 32 / (double)h, ghz, tf);
}
int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 4 * 512 * 1024); hipMalloc(&cyc, 64);
  run<7, 2, 3, 4, 4>("loop + barrier/chunk", 4000, 6, 3, out, cyc, 0);
  run<7, 2, 3, 4 + 32, 4>("+ 40 DMA16/chunk by compute waves", 4000, 6, 3, out, cyc, 0, 40);
  run<7, 2, 3, 4 + 128, 4>("A from global (dword)", 4000, 6, 3, out, cyc, 0);
  run<7, 2, 3, 4 + 128 + 32, 4>("A from global + 8 DMA16/chunk", 4000, 6, 3, out, cyc, 0, 8);
  run<7, 2, 3, 4 + 128, 8>("A from global, 8 waves", 2000, 6, 3, out, cyc, 0);
  run<3, 4, 3, 4 + 128, 4>("3x4 A from global", 4000, 6, 3, out, cyc, 0);
  run<2, 4, 4, 4 + 128, 4>("2x4x4 A from global", 4000, 4, 4, out, cyc, 0);
  run<13, 2, 3, 4 + 128, 4>("13x2 A from global", 2000, 6, 3, out, cyc, 0);
  run<7, 4, 3, 4 + 128, 4>("7x4 A from global", 2000, 6, 3, out, cyc, 0);
  return 0;
}
