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
// Work-group: WM x WN <= 12 compute waves (WM along the channels, WN along the positions)
// + 4 producer waves that stage the input spans of the coming channel chunks by LDS-DMA.
// Unlike the 32-cycle 16x16x4 MFMA, an 8-cycle 4x4x1 MFMA leaves no room to issue anything
// else from the SAME wave for free (measured: one compute wave per SIMD reaches 40-60 % of
// the pipe, profiles/r02_b_igemm4_sweeps.txt), so the tiles per wave are small (<= 64
// accumulator registers, <= 128 VGPRs) and up to three compute waves share a SIMD: while
// one issues its loads or address arithmetic the others keep the matrix pipe busy.
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
// E2_G4_ABLATE (timing experiments only, never in a release build): 1 = no weight loads
// inside the step loop, 2 = no LDS reads, 4 = no MFMAs
#ifndef E2_G4_ABLATE
#define E2_G4_ABLATE 0
#endif
template <int MG, int NT, int KW, int U, int R>
__device__ __forceinline__ void g4_read(G4Regs<MG, NT, KW, U>& g, const G4Addr<NT, KW, U>& ad) {
  constexpr int NA = G4Regs<MG, NT, KW, U>::NA;
  constexpr int RA = U * KW * NA;
  if constexpr (R < RA) {
    constexpr int u = R / (KW * NA), j = (R / NA) % KW, i = R % NA;
    if constexpr (!(E2_G4_ABLATE & 1)) g.a[u][j][i] = gl_ld<i * 256>(ad.abase[u], ad.voff[j]);
  } else {
    constexpr int rb = R - RA;
    constexpr int u = rb / (KW * NT), j = (rb / NT) % KW, nb = rb % NT;
    if constexpr (!(E2_G4_ABLATE & 2)) g.b[u][j][nb] = lds_ld<j * 4>(ad.b[u][nb]);
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
  if constexpr (!(E2_G4_ABLATE & 4))
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

// input channels per step: >= ~40 MFMAs per step (the other compute waves of the SIMD cover
// the rest of a weight load's round trip), as far as two operand sets fit 128 registers
template <int MG, int NT, int KW>
constexpr int g4_pairs() {
  constexpr int NA = (4 * MG + 63) / 64;
  int u = 1;
  while (u < 8 && u * KW * MG * NT < 40 && 2 * (2 * u) * KW * (NA + NT) + 4 * MG * NT <= 88) u *= 2;
  return u;
}

struct Igemm4Extra {
  int WM, WN;             // compute waves along the channels / the positions (WM * WN <= 12)
  int tilesTotal;         // N * splitK * nMT * Do * nPT
};
constexpr int kG4Producers = 4;

// one (n, k-split, channel tile, z-plane, position tile) unit of work
struct G4Tile {
  int z, mt, ks, n;
  int q0, r0, c0, L;      // first position, its row / column, span length in floats
  int cb, ce;             // chunk range of this tile's K split
  long span_lo;
};

// The kernel is PERSISTENT: work-group w walks the tiles w, w + grid, w + 2*grid, ... and
// treats the (tile, channel chunk) pairs as ONE sequence of items staged through a ring of
// three LDS buffers.  The producer waves run two items ahead, so at the barrier that opens
// item i the data of item i+1 has landed as well: the compute waves fetch the first
// operands of a chunk during the last step of the chunk before it and the step pipeline
// runs through chunk boundaries without a bubble (the two-buffer form paid ~1.5 k cycles
// per chunk); a work-group pays the DMA latency once, not once per tile, and tile counts
// per work-group differ by at most one (the one-tile-per-work-group grid left up to half
// of the last round idle).
template <int MG, int NT, int KW>
__global__ __launch_bounds__(1024, 1) void igemm4_kernel(IgemmP p, Igemm4Extra x) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int U = g4_pairs<MG, NT, KW>();
  constexpr int NBUF = 3;
  const int WM = x.WM, WN = x.WN;
  const int BM = 4 * MG * WM, BN = 64 * NT * WN;
  const int CC = p.CC;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave = wave8;
  const bool producer = wave8 >= WM * WN;
  const int isY = (int)p.isY;
  const int Lpad = p.Lpad;
  // (with a clipped K range -- IgemmP::zpad -- a split of a border plane may be EMPTY: such a
  // tile has no items; both the staging cursor and the compute loop skip it)

  auto decode = [&](int t, G4Tile& T) {
    const int pt = t % p.nPT; t /= p.nPT;
    T.z = t % p.Do; t /= p.Do;
    T.mt = t % p.nMT; t /= p.nMT;
    T.ks = t % p.splitK;
    T.n = t / p.splitK;
    T.q0 = pt * BN;
    const int qlast = min(T.q0 + BN, p.Q) - 1;
    T.r0 = T.q0 / p.Wo; T.c0 = T.q0 - T.r0 * p.Wo;
    const int rl = qlast / p.Wo, cl = qlast - rl * p.Wo;
    T.span_lo = (long)T.r0 * p.isY + T.c0;
    T.L = (rl - T.r0) * isY + (cl - T.c0) + (p.kh - 1) * isY + p.kw;
    int c_lo, c_hi;
    e2_chunk_range(p, T.z, c_lo, c_hi);
    const int per = (c_hi - c_lo + p.splitK - 1) / p.splitK;
    T.cb = c_lo + T.ks * per; T.ce = max(T.cb, min(T.cb + per, c_hi));
  };

  if (producer) {
    // ---- producers: input spans of item i -> LDS buffer i % 3.  Every chunk stages CC
    // channels; channels past Cin repeat the last one (their weights are zero).
    const int pw = wave8 - WM * WN;
    auto stage = [&](const G4Tile& T, int ch, int buf) {
      const int nJ = (T.L + 63) >> 6;
      const int nJ16 = (T.L + 255) >> 8;
      const int dz = ch / p.nChunkC;
      const int cbase = (ch - dz * p.nChunkC) * CC;
      float* xl = smem + buf * p.bufFloats;
      const float* xb = p.in + (long)T.n * p.isN + (long)(T.z + dz) * p.isZ + T.span_lo;
      for (int cc = pw; cc < ((E2_G4_ABLATE & 32) ? 0 : CC); cc += kG4Producers) {
        const int ci = min(cbase + cc, p.Cin - 1);
        const float* src = xb + (long)ci * p.isC;
        float* dst = xl + cc * Lpad;
        // 16-byte pieces; the straddling lane over-reads <= 12 bytes, which stays inside
        // the tensor except on its very last row: that row goes by dwords
        const bool tail_row = (ci == p.Cin - 1) && (T.z + dz == p.Din - 1) && (T.n == p.N - 1);
        if (!tail_row) {
          for (int j = 0; j < nJ16; ++j) {
            const int u = 256 * j + 4 * lane;
            if (u < T.L) glds16(src + u, dst + 256 * j);
          }
        } else {
          for (int j = 0; j < nJ; ++j)
            if (64 * j + lane < T.L) glds4(src + 64 * j + lane, dst + 64 * j);
        }
      }
    };
    // item counts: barriers to take (all of this work-group's items) / items staged
    int nItems = 0;
    for (int t = blockIdx.x; t < x.tilesTotal; t += gridDim.x) {
      G4Tile T; decode(t, T);
      nItems += T.ce - T.cb;
    }
    G4Tile S; int st = blockIdx.x, sch = 0, staged = 0;      // the staging cursor
    auto seek = [&]() {                                      // ... rests on a tile that has items
      for (; st < x.tilesTotal; st += gridDim.x) {
        decode(st, S); sch = S.cb;
        if (S.cb < S.ce) break;
      }
    };
    seek();
    auto stage_next = [&]() {
      if (staged >= nItems) return;
      stage(S, sch, staged % NBUF);
      ++staged;
      if (++sch == S.ce) { st += gridDim.x; seek(); }
    };
    stage_next();
    stage_next();
    for (int i = 0; i < nItems; ++i) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();            // items <= i+1 are in LDS; the buffer of item i-1 is free
      stage_next();               // item i+2 -> the buffer of item i-1
    }
    return;
  }

  // ---- compute waves ----------------------------------------------------------
  const int wm = wave % WM, wn = wave / WM;
  G4Addr<NT, KW, U> ad;
#pragma unroll
  for (int j = 0; j < KW; ++j) ad.voff[j] = 4u * (unsigned)(lane + 4 * j * p.coP);
  const int nCGp = p.ciP >> 2;
  const int nCblk = CC / U;                           // channel blocks per chunk (CC % 2U == 0)
  const int nSteps = nCblk * p.kh;                    // even
  const unsigned stepY = 4u * (unsigned)isY, stepC = 4u * (unsigned)Lpad;
  const long cgStride = (long)p.THW * 4 * p.coP;      // floats between two channel groups' rows
  const int R = p.upz * p.upy * p.upx;
  constexpr int NA = G4Regs<MG, NT, KW, U>::NA;
  constexpr int RALL = U * KW * (NA + NT);
  G4Regs<MG, NT, KW, U> g0, g1;
  int item = 0;                                       // index of the tile's first item (ring position)
#if (E2_G4_ABLATE & 8)       // (timing experiment: results are garbage)
#define E2_WAIT() __builtin_amdgcn_sched_barrier(0);
#else
#define E2_WAIT()                                                         \
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");             \
  __builtin_amdgcn_sched_barrier(0);
#endif

  for (int t = blockIdx.x; t < x.tilesTotal; t += gridDim.x) {
    G4Tile T; decode(t, T);
    const bool has_items = T.cb < T.ce;
    if (!has_items && !p.parts) continue;             // empty split of a clipped K range
    const int m0w = T.mt * BM + wm * (4 * MG);        // the wave's first output channel
    const int qw = T.q0 + wn * (64 * NT);             // ... and first position
    unsigned posoffB[NT];
#pragma unroll
    for (int nb = 0; nb < NT; ++nb) {
      const int q = min(qw + nb * 64 + lane, p.Q - 1);
      const int r = q / p.Wo, c = q - r * p.Wo;
      posoffB[nb] = 4u * (unsigned)((r - T.r0) * isY + (c - T.c0));
    }
    f32x4 acc[MG][NT];
#pragma unroll
    for (int g = 0; g < MG; ++g)
#pragma unroll
      for (int nb = 0; nb < NT; ++nb) acc[g][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nCh = T.ce - T.cb;
    const int total = nCh * nSteps;
    // fetch cursor: chunk (relative to the tile), channel block, tap row + what follows from
    // the chunk: kernel plane dz, first channel, LDS buffer
    int fch = 0, fc = 0, fty = 0;
    int fdz = T.cb / p.nChunkC;
    int fcb = (T.cb - fdz * p.nChunkC) * CC;
    int fbuf = item % NBUF;
    // operands' addresses of the step under the cursor: channels fcb + fc*U + u; their image
    // rows are row0 + (u >> 2) * THW*4 + (u & 3)
    auto set_addr = [&]() {
      const int cc0 = fcb + fc * U;
      const long row0 = ((long)(fdz * nCGp + (cc0 >> 2)) * p.THW + fty * p.kw) * 4 + (cc0 & 3);
      const float* a0 = p.wp + row0 * p.coP + m0w;
      const unsigned so0 = lds_addr(smem + fbuf * p.bufFloats) + (unsigned)(fc * U) * stepC +
                           (unsigned)fty * stepY;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        ad.abase[u] = a0 + (long)(u >> 2) * cgStride + (long)(u & 3) * p.coP;
#pragma unroll
        for (int nb = 0; nb < NT; ++nb) ad.b[u][nb] = posoffB[nb] + so0 + (unsigned)u * stepC;
      }
    };
    // advance the cursor by one step; past the tile's end it stays on the last step (that
    // fetch is never used)
#define E2_NEXT()                                                         \
    {                                                                     \
      int nty = fty + 1, nfc = fc, nch = fch;                             \
      if (nty == p.kh) { nty = 0; ++nfc; }                                \
      if (nfc == nCblk) { nfc = 0; ++nch; }                               \
      if (nch < nCh) {                                                    \
        if (nch != fch) {                                                 \
          fcb += CC;                                                      \
          if (fcb >= p.nChunkC * CC) { fcb = 0; ++fdz; }                  \
          fbuf = (fbuf + 1 == NBUF) ? 0 : fbuf + 1;                       \
        }                                                                 \
        fty = nty; fc = nfc; fch = nch;                                   \
      }                                                                   \
    }
    if (has_items) {                         // (partial-sum stores: an empty split stores zeros)
    __syncthreads();                         // the tile's first item (and the one after it) landed
    set_addr();
    g4_reads<MG, NT, KW, U, 0, RALL>(g0, ad);
    E2_WAIT()
    g0.touch();
    }
    int cs = 0;                              // steps computed in the current chunk
    for (int s = 0; s < total; s += 2) {
      E2_NEXT()
      set_addr();
      __builtin_amdgcn_sched_barrier(0);
      g4_steps<MG, NT, KW, U, 0>(g0, g1, acc, ad);     // compute s, fetch s+1
      E2_WAIT()
      g1.touch();
      E2_NEXT()
      set_addr();
      __builtin_amdgcn_sched_barrier(0);
      g4_steps<MG, NT, KW, U, 0>(g1, g0, acc, ad);     // compute s+1, fetch s+2
      E2_WAIT()
      g0.touch();
      cs += 2;
      if (cs == nSteps) {                    // chunk done (its successor's first operands are in g0)
        cs = 0;
        if (s + 2 < total) __syncthreads();  // opens the next item of this tile
      }
    }
#undef E2_NEXT
    item += nCh;

    // ---- epilogue: register r of accumulator (g, nb) = channel m0w + 4g + r at the lane's
    // position; every store instruction writes 64 consecutive positions of one channel ----
    unsigned ooff[NT];          // (offset inside one output plane: fits 32 bits)
    bool ok[NT];
#pragma unroll
    for (int nb = 0; nb < NT; ++nb) {
      const int q = qw + nb * 64 + lane;
      ok[nb] = q < p.Q;
      const int qq = min(q, p.Q - 1);
      const int r = qq / p.Wo, c = qq - r * p.Wo;
      ooff[nb] = (unsigned)(r * p.upy) * (unsigned)p.osY + (unsigned)(c * p.upx);
    }
    float* ob = p.out + (long)T.ks * p.partStride + (long)T.n * p.osN + (long)(T.z * p.upz) * p.osZ;
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
          if (!ok[nb] || ((E2_G4_ABLATE & 64) && co > 0)) continue;
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
#undef E2_WAIT
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
  hipLaunchKernelGGL((igemm4_kernel<MG, NT, KW>), dim3(grid), dim3(64 * (x.WM * x.WN + kG4Producers)),
                     lds, ctx->stream, p, x);
  E2_CHECK_HIP(hipGetLastError());
  return 0;
}

// the (MG, NT) instances: channels per wave 4*MG in {16, 20, 28, 32, 40, 52, 64}.
// (Only tiles whose accumulators + two operand sets fit 128 registers WITHOUT scratch:
// the build fails on a non-zero ScratchSize of these kernels, csrc/check_scratch.py -- a
// spilled operand register is copied while the asm load that fills it is in flight.)
#define E2_IGEMM4_INSTANCES(X) \
  X(4, 2) X(5, 1) X(5, 2) X(7, 1) X(7, 2) X(8, 1) X(10, 1) X(13, 1) X(16, 1)

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
