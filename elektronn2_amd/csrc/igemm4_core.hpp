// igemm4_core.hpp -- implicit-GEMM convolution on v_mfma_f32_4x4x1_16b_f32 with the A
// operand BROADCAST from one block (cbsz:4 abid:b): one instruction is the outer product
//      D[4 channels][64 lanes] += A[4 channels] (x) B[64 lanes]
// i.e. a 4 (M) x 64 (N) x 1 (K) step, 8 cycles, the same 64 FLOP/clk/SIMD as the 16x16x4
// form (measured: tools/ubench/mfma4x4.hip, profiles/r02_a_ubench_mfma4x4.txt -- 96 % of
// the f32 matrix peak at 25 accumulators per wave, at a HIGHER clock than the 16x16x4 loop).
// What it buys: the channel dimension is padded to 4 instead of 16 (Cout = 20 -> 20, not
// 32; 40 -> 40, not 48; 200 -> 200, not 224) and K has no granularity at all; the executed
// FLOPs of the neuro3d_lite step were 1.29x the algorithmic ones in the 16x16 form
// (profiles/r02_a_bench_lite183_pmc_mfma.csv).
//
// Operand flow (same GEMM view and the same packed weight image as igemm_core.hpp):
//   lane l of a compute wave owns output positions q = q_wave + 64*nb + l (nb < NT);
//   B[k] for position block nb is ONE ds_read_b32 per lane from the staged input span
//        (address = per-lane position offset + wave-uniform (channel, tap) offset);
//   A[k] for 64 consecutive output channels is ONE coalesced global_load_dword of a row
//        of the packed image Wp[dz][cg][t][qd][oc] (lane l <- oc = m + l); `abid:g%16`
//        then picks channels 4g..4g+3 of it, so one A register feeds up to 16 MFMAs per
//        position block;
//   accumulator g,nb: 4 registers = channels 4g..4g+3 at the lane's position.
// A step = U consecutive input channels x the KW taps of one tap row; the operands of step
// s+1 are fetched while step s computes (two register sets), retired by one s_waitcnt.
// Work-group: 4 compute waves (WM along the channels x 4/WM along the positions) + 4
// producer waves that stage the input spans of the next channel chunk by LDS-DMA, exactly
// as in igemm_core.hpp.
#pragma once
#include "igemm_core.hpp"

template <int MG, int NT, int KW, int U>
struct G4Regs {
  static constexpr int NA = (4 * MG + 63) / 64;
  float a[U][KW][NA];
  float b[U][KW][NT];
  __device__ __forceinline__ void touch() {
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int j = 0; j < KW; ++j) {
#pragma unroll
        for (int i = 0; i < NA; ++i) asm volatile("" : "+v"(a[u][j][i]));
#pragma unroll
        for (int i = 0; i < NT; ++i) asm volatile("" : "+v"(b[u][j][i]));
      }
  }
};

template <int NT, int KW, int U>
struct G4Addr {
  const float* abase[U];        // SGPR pairs: image row of tap 0 of pair u (+ the wave's first channel)
  unsigned voff[KW];            // per-lane byte offset of tap j: 4 * (lane + 4*j*coP)
  unsigned b[U][NT];            // LDS byte address of tap 0 of pair u, position block nb
};

// read R of a step: all weight (global) loads first -- the longer latency -- then the LDS reads
template <int MG, int NT, int KW, int U, int R>
__device__ __forceinline__ void g4_read(G4Regs<MG, NT, KW, U>& g, const G4Addr<NT, KW, U>& ad) {
  constexpr int NA = G4Regs<MG, NT, KW, U>::NA;
  constexpr int RA = U * KW * NA;
  if constexpr (R < RA) {
    constexpr int u = R / (KW * NA), j = (R / NA) % KW, i = R % NA;
    g.a[u][j][i] = gl_ld<i * 256>(ad.abase[u], ad.voff[j]);
  } else {
    constexpr int rb = R - RA;
    constexpr int u = rb / (KW * NT), j = (rb / NT) % KW, nb = rb % NT;
    g.b[u][j][nb] = lds_ld<j * 4>(ad.b[u][nb]);
  }
}
template <int MG, int NT, int KW, int U, int R0, int R1>
__device__ __forceinline__ void g4_reads(G4Regs<MG, NT, KW, U>& g, const G4Addr<NT, KW, U>& ad) {
  if constexpr (R0 < R1) {
    g4_read<MG, NT, KW, U, R0>(g, ad);
    g4_reads<MG, NT, KW, U, R0 + 1, R1>(g, ad);
  }
}
template <int MG, int NT, int KW, int U, int I>
__device__ __forceinline__ void g4_mfma(const G4Regs<MG, NT, KW, U>& cur, f32x4 (&acc)[MG][NT]) {
  constexpr int u = I / (KW * NT * MG), j = (I / (NT * MG)) % KW, nb = (I / MG) % NT, g = I % MG;
  acc[g][nb] = __builtin_amdgcn_mfma_f32_4x4x1f32(cur.a[u][j][g / 16], cur.b[u][j][nb], acc[g][nb],
                                                  4, g % 16, 0);
}
// one step, hand-scheduled: MFMA i of the CURRENT set, then reads [r0, r1) of the NEXT set,
// spread over the first 3/4 of the MFMAs
template <int MG, int NT, int KW, int U, int I>
__device__ __forceinline__ void g4_steps(const G4Regs<MG, NT, KW, U>& cur, G4Regs<MG, NT, KW, U>& nxt,
                                         f32x4 (&acc)[MG][NT], const G4Addr<NT, KW, U>& ad) {
  constexpr int NA = G4Regs<MG, NT, KW, U>::NA;
  constexpr int M = U * KW * MG * NT, R = U * KW * (NA + NT);
  g4_mfma<MG, NT, KW, U, I>(cur, acc);
  constexpr int r0 = (I * R * 4) / (3 * M) < R ? (I * R * 4) / (3 * M) : R;
  constexpr int r1 = ((I + 1) * R * 4) / (3 * M) < R ? ((I + 1) * R * 4) / (3 * M) : R;
  g4_reads<MG, NT, KW, U, r0, r1>(nxt, ad);
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (I + 1 < M) g4_steps<MG, NT, KW, U, I + 1>(cur, nxt, acc, ad);
}
template <int MG, int NT, int KW, int U, int I>
__device__ __forceinline__ void g4_only(const G4Regs<MG, NT, KW, U>& cur, f32x4 (&acc)[MG][NT]) {
  g4_mfma<MG, NT, KW, U, I>(cur, acc);
  if constexpr (I + 1 < U * KW * MG * NT) g4_only<MG, NT, KW, U, I + 1>(cur, acc);
}

// input channels per step: enough MFMAs (>= ~96, i.e. ~800 cycles) to cover an L2 round
// trip of the weight loads, as far as two operand sets fit the register budget
template <int MG, int NT, int KW>
constexpr int g4_pairs() {
  constexpr int NA = (4 * MG + 63) / 64;
  int u = 1;
  while (u < 8 && u * KW * MG * NT < 96 && 2 * (2 * u) * KW * (NA + NT) + 4 * MG * NT <= 200) u *= 2;
  return u;
}

struct Igemm4Extra {
  int WM;                 // compute waves along the channels (1, 2 or 4)
};

template <int MG, int NT, int KW>
__global__ __launch_bounds__(512, 1) void igemm4_kernel(IgemmP p, Igemm4Extra x) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int U = g4_pairs<MG, NT, KW>();
  const int WM = x.WM, WN = 4 / WM;
  const int BM = 4 * MG * WM, BN = 64 * NT * WN;
  const int CC = p.CC;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave = wave8 & 3;
  const bool producer = wave8 >= 4;

  int bid = blockIdx.x;
  const int pt = bid % p.nPT; bid /= p.nPT;
  const int z = bid % p.Do;  bid /= p.Do;
  const int mt = bid % p.nMT; bid /= p.nMT;
  const int ks = bid % p.splitK;
  const int n = bid / p.splitK;

  const int q0 = pt * BN;
  const int qlast = min(q0 + BN, p.Q) - 1;
  const int r0 = q0 / p.Wo, c0 = q0 - r0 * p.Wo;
  const int rl = qlast / p.Wo, cl = qlast - rl * p.Wo;
  const int isY = (int)p.isY;
  const long span_lo = (long)r0 * p.isY + c0;
  const int L = (rl - r0) * isY + (cl - c0) + (p.kh - 1) * isY + p.kw;
  const int Lpad = p.Lpad;

  const int nChunks = p.kd * p.nChunkC;
  const int per = (nChunks + p.splitK - 1) / p.splitK;
  const int cb = ks * per, ce = min(cb + per, nChunks);

  if (producer) {
    // ---- producers: input spans of chunk ch -> LDS buffer (ch - cb) & 1.  Every chunk
    // stages CC channels; channels past Cin repeat the last one (their weights are zero).
    const int pw = wave8 - 4;
    const int nJ = (L + 63) >> 6;
    const int nJ16 = (L + 255) >> 8;
    const float* in_n = p.in + (long)n * p.isN + (long)z * p.isZ + span_lo;
    auto stage = [&](int ch, int buf) {
      const int dz = ch / p.nChunkC;
      const int cbase = (ch - dz * p.nChunkC) * CC;
      float* xl = smem + buf * p.bufFloats;
      const float* xb = in_n + (long)dz * p.isZ;
      for (int cc = pw; cc < CC; cc += 4) {
        const int ci = min(cbase + cc, p.Cin - 1);
        const float* src = xb + (long)ci * p.isC;
        float* dst = xl + cc * Lpad;
        const bool tail_row = (ci == p.Cin - 1) && (z + dz == p.Din - 1) && (n == p.N - 1);
        if (!tail_row) {
          for (int j = 0; j < nJ16; ++j) {
            const int u = 256 * j + 4 * lane;
            if (u < L) glds16(src + u, dst + 256 * j);
          }
        } else {
          for (int j = 0; j < nJ; ++j)
            if (64 * j + lane < L) glds4(src + 64 * j + lane, dst + 64 * j);
        }
      }
    };
    if (cb < ce) stage(cb, 0);
    for (int ch = cb; ch < ce; ++ch) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();                       // chunk ch is in LDS; buffer of ch-1 is free
      if (ch + 1 < ce) stage(ch + 1, ((ch - cb) & 1) ^ 1);
    }
    return;
  }

  // ---- compute waves ----------------------------------------------------------
  const int wm = wave % WM, wn = wave / WM;
  const int m0w = mt * BM + wm * (4 * MG);            // the wave's first output channel
  const int qw = q0 + wn * (64 * NT);                 // ... and first position
  unsigned posoffB[NT];
#pragma unroll
  for (int nb = 0; nb < NT; ++nb) {
    const int q = min(qw + nb * 64 + lane, p.Q - 1);
    const int r = q / p.Wo, c = q - r * p.Wo;
    posoffB[nb] = 4u * (unsigned)((r - r0) * isY + (c - c0));
  }
  f32x4 acc[MG][NT];
#pragma unroll
  for (int g = 0; g < MG; ++g)
#pragma unroll
    for (int nb = 0; nb < NT; ++nb) acc[g][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

  G4Addr<NT, KW, U> ad;
#pragma unroll
  for (int j = 0; j < KW; ++j) ad.voff[j] = 4u * (unsigned)(lane + 4 * j * p.coP);
  const int nCGp = p.ciP >> 2;
  const int nCblk = CC / U;                           // channel blocks per chunk (CC % U == 0)
  const int nSteps = nCblk * p.kh;
  const unsigned stepY = 4u * (unsigned)isY, stepC = 4u * (unsigned)Lpad;
  const long cgStride = (long)p.THW * 4 * p.coP;      // floats between two channel groups' rows

  // operands' addresses of the step (channel block cblk, tap row ty) of chunk (dz, cbase):
  // channels cbase + cblk*U + u; their image rows are row0 + (u >> 2) * THW*4 + (u & 3)
  auto set_addr = [&](int cblk, int ty, int dz, int cbase, unsigned xbase) {
    const int cc0 = cbase + cblk * U;
    const long row0 = ((long)(dz * nCGp + (cc0 >> 2)) * p.THW + ty * p.kw) * 4 + (cc0 & 3);
    const float* a0 = p.wp + row0 * p.coP + m0w;
    const unsigned so0 = xbase + (unsigned)(cblk * U) * stepC + (unsigned)ty * stepY;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      ad.abase[u] = a0 + (long)(u >> 2) * cgStride + (long)(u & 3) * p.coP;
#pragma unroll
      for (int nb = 0; nb < NT; ++nb) ad.b[u][nb] = posoffB[nb] + so0 + (unsigned)u * stepC;
    }
  };
  constexpr int NA = G4Regs<MG, NT, KW, U>::NA;
  constexpr int RALL = U * KW * (NA + NT), RA = U * KW * NA;
  G4Regs<MG, NT, KW, U> g0, g1;
#define E2_WAIT()                                                         \
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");             \
  __builtin_amdgcn_sched_barrier(0);
  // (fc, fty): the step whose operands are fetched next; past the chunk's end the last
  // step is fetched again (never used)
#define E2_NEXT()                                                         \
  {                                                                       \
    ++fty;                                                                \
    const bool wrap = (fty == p.kh);                                      \
    fty = wrap ? 0 : fty;                                                 \
    fc += wrap ? 1 : 0;                                                   \
    const bool end = (fc == nCblk);                                       \
    fc = end ? nCblk - 1 : fc;                                            \
    fty = end ? p.kh - 1 : fty;                                           \
  }
  for (int ch = cb; ch < ce; ++ch) {
    const int cur = (ch - cb) & 1;
    const int dz = ch / p.nChunkC;
    const int cbase = (ch - dz * p.nChunkC) * CC;
    const unsigned xbase = lds_addr(smem + cur * p.bufFloats);
    int fc = 0, fty = 0;
    // weights of the first step: independent of the LDS contents, start them early
    set_addr(0, 0, dz, cbase, xbase);
    g4_reads<MG, NT, KW, U, 0, RA>(g0, ad);
    __syncthreads();                         // the producers saw their DMA land
    g4_reads<MG, NT, KW, U, RA, RALL>(g0, ad);
    E2_WAIT()
    g0.touch();
    int s = 0;
    for (; s + 1 < nSteps; s += 2) {
      E2_NEXT()
      set_addr(fc, fty, dz, cbase, xbase);
      __builtin_amdgcn_sched_barrier(0);
      g4_steps<MG, NT, KW, U, 0>(g0, g1, acc, ad);     // compute s, fetch s+1
      E2_WAIT()
      g1.touch();
      E2_NEXT()
      set_addr(fc, fty, dz, cbase, xbase);
      __builtin_amdgcn_sched_barrier(0);
      g4_steps<MG, NT, KW, U, 0>(g1, g0, acc, ad);     // compute s+1, fetch s+2
      E2_WAIT()
      g0.touch();
    }
    if (s < nSteps) g4_only<MG, NT, KW, U, 0>(g0, acc);
  }
#undef E2_NEXT
#undef E2_WAIT
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  g0.touch();
  g1.touch();

  // ---- epilogue: register r of accumulator (g, nb) = channel m0w + 4g + r at the lane's
  // position; every store instruction writes 64 consecutive positions of one channel ----
  const int R = p.upz * p.upy * p.upx;
  long ooff[NT];
  bool ok[NT];
#pragma unroll
  for (int nb = 0; nb < NT; ++nb) {
    const int q = qw + nb * 64 + lane;
    ok[nb] = q < p.Q;
    const int qq = min(q, p.Q - 1);
    const int r = qq / p.Wo, c = qq - r * p.Wo;
    ooff[nb] = (long)(r * p.upy) * p.osY + (long)c * p.upx;
  }
  float* ob = p.out + (long)n * p.osN + (long)(z * p.upz) * p.osZ;
#pragma unroll
  for (int g = 0; g < MG; ++g) {
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int co = m0w + 4 * g + rr;                 // wave-uniform
      if (co >= p.Cout) continue;
      float* dst;
      if (R == 1) {
        dst = ob + (long)co * p.osC;
      } else {
        const int cr = co / R, sub = co - cr * R;
        const int rz = sub / (p.upy * p.upx);
        const int rem = sub - rz * (p.upy * p.upx);
        const int ry = rem / p.upx, rx = rem - ry * p.upx;
        dst = ob + (long)cr * p.osC + (long)rz * p.osZ + (long)ry * p.osY + rx;
      }
      const float bv = p.bias ? p.bias[co] : 0.f;
#pragma unroll
      for (int nb = 0; nb < NT; ++nb) {
        if (!ok[nb]) continue;
        float v = acc[g][nb][rr];
        if (p.bias) {
          v += bv;
          if (p.act == E2_ACT_RELU) v = (v > 0.f) ? v : ((v == 0.f) ? 0.f : -0.f);
        }
        if (p.atomic) unsafeAtomicAdd(dst + ooff[nb], v);
        else dst[ooff[nb]] = v;
      }
    }
  }
}

// ---- launch helpers ----------------------------------------------------------
template <int MG, int NT, int KW>
static int igemm4_launch(e2_ctx* ctx, const IgemmP& p, const Igemm4Extra& x, int grid, size_t lds) {
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm4_kernel<MG, NT, KW>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) { e2_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return 1; }
    attr_done = true;
  }
  hipLaunchKernelGGL((igemm4_kernel<MG, NT, KW>), dim3(grid), dim3(512), lds, ctx->stream, p, x);
  E2_CHECK_HIP(hipGetLastError());
  return 0;
}

// the (MG, NT) instances: channels per wave 4*MG in {20, 32, 40, 52, 64, 76, 80, 100}
#define E2_IGEMM4_INSTANCES(X) \
  X(5, 2) X(5, 4) X(8, 2) X(8, 4) X(10, 2) X(10, 4) X(13, 1) X(13, 2) X(16, 1) X(16, 2) \
  X(19, 1) X(19, 2) X(20, 1) X(20, 2) X(25, 1) X(25, 2)

template <int KW>
static int igemm4_dispatch(e2_ctx* ctx, const IgemmP& p, const Igemm4Extra& x, int MG, int NT,
                           int grid, size_t lds) {
#define E2_X(M, N) if (MG == M && NT == N) return igemm4_launch<M, N, KW>(ctx, p, x, grid, lds);
  E2_IGEMM4_INSTANCES(E2_X)
#undef E2_X
  e2_set_error("igemm4: no instance MG=%d NT=%d", MG, NT);
  return 2;
}
template <int KW>
static int igemm4_pairs(int MG, int NT) {
#define E2_X(M, N) if (MG == M && NT == N) return g4_pairs<M, N, KW>();
  E2_IGEMM4_INSTANCES(E2_X)
#undef E2_X
  return 0;
}

// entry points of the per-width translation units (conv_igemm4_k*.hip)
int e2i_igemm4_launch_k1(e2_ctx*, const IgemmP&, const Igemm4Extra&, int MG, int NT, int grid, size_t lds);
int e2i_igemm4_launch_k3(e2_ctx*, const IgemmP&, const Igemm4Extra&, int MG, int NT, int grid, size_t lds);
int e2i_igemm4_launch_k4(e2_ctx*, const IgemmP&, const Igemm4Extra&, int MG, int NT, int grid, size_t lds);
int e2i_igemm4_launch_k5(e2_ctx*, const IgemmP&, const Igemm4Extra&, int MG, int NT, int grid, size_t lds);
int e2i_igemm4_pairs_k1(int MG, int NT);          // input channels per step of an instance (0: none)
int e2i_igemm4_pairs_k3(int MG, int NT);
int e2i_igemm4_pairs_k4(int MG, int NT);
int e2i_igemm4_pairs_k5(int MG, int NT);
