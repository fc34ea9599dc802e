// conv_igemm.hip -- 3-D "valid" correlation as an implicit GEMM on the gfx950
// fp32 matrix cores (v_mfma_f32_16x16x4_f32: exact f32 fma chain).
//
//   out[n][oc][z][y][x] (+)= sum_{dz,ty,tx,ic} Wp[dz][t][ic][oc] *
//                             in[n][ic][z+dz][y+ty][x+tx]
//
// One kernel serves
//   * Conv forward  (reference: computations.py:386-428; flip folded into Wp)
//   * Conv dgrad    (Theano's ConvGradI born at model.py:182): the same
//     correlation run over the zero-padded dy with (oc,ic) swapped, no flip
//   * UpConv forward as a 1x1x1 GEMM with a depth-to-space scatter epilogue.
//
// GEMM view: M = out channels (A operand = packed weights), N = output
// positions of one z-plane (B operand = input), K = (dz, ic, ty, tx).
// Work-group = 4 waves; each wave owns MT x NT 16x16 accumulator blocks; the
// four waves sit side by side along N, so a work-group covers BM = 16*MT
// channels x BN = 64*NT consecutive plane positions (q = y*Wo + x, tiles never
// cross a z-plane).  Because the positions are consecutive in the plane, the
// input window the tile touches is, per (ic, dz), ONE contiguous span of the
// input plane: [in_off(q_first), in_off(q_last) + (kh-1)*sY + kw-1].  Those
// spans are staged into LDS with fully coalesced dword loads and every tap
// (ty,tx) of every channel is then served from LDS -- each input element is
// fetched once per tile instead of kh*kw times.
//
// MFMA operand maps (cdna_hip_programming.md §3): 16x16x4 f32, lane l:
//   A[i = l&15][k = l>>4], B[k = l>>4][j = l&15]; D: col = l&15,
//   row = 4*(l>>4) + reg.  Lane quarter qd = l>>4 picks channel ic = 4*cg+qd of
//   the staged chunk, so per k-step the tap offset is wave-uniform and the
//   per-lane part of both LDS addresses is loop invariant.
// LDS rows are padded so that rows qd and qd+1 sit 16 banks apart
// (stride == 16 mod 32): ds_read_b32 of a 32-lane half is conflict free.
#include "common.hpp"
#ifndef E2_INTERLEAVE
#define E2_INTERLEAVE 1
#endif
#include <stdlib.h>
#include <algorithm>
#include <utility>

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct IgemmP {
  const float* in;
  const float* wp;
  float* out;
  int Cin, Cout, kd, kh, kw, THW;
  int Do, Ho, Wo, Q;
  long isN, isC, isZ, isY;
  long osN, osC, osZ, osY;
  int ciP, coP;
  int Lpad, CC, Din, N;
  int dbg;                // timing ablations only (E2_IGEMM_DBG): 1 = stage once, 2 = no MFMA, 4 = no barrier
  int nPT, nMT, splitK, nChunkC;
  int atomic;
  int upz, upy, upx;
  int bufFloats;          // floats per LDS buffer (x region + w region + slack)
};

typedef __attribute__((address_space(3))) void* lds_vp;
typedef const __attribute__((address_space(1))) void* gbl_vp;

// async global -> LDS copies (no VGPR destination); LDS address is
// wave-uniform base + lane*size, the global source is per lane.
__device__ __forceinline__ void glds4(const float* g, float* l) {
  __builtin_amdgcn_global_load_lds((gbl_vp)g, (lds_vp)l, 4, 0, 0);
}
__device__ __forceinline__ void glds16(const float* g, float* l) {
  __builtin_amdgcn_global_load_lds((gbl_vp)g, (lds_vp)l, 16, 0, 0);
}

// ds_read_b32 with an immediate offset, invisible to hipcc's waitcnt
// bookkeeping: the kernel waits with its own "s_waitcnt lgkmcnt(0)" AFTER the
// MFMAs of the previous group have been issued (cdna guide §5.7 form iii).
template <int OFF>
__device__ __forceinline__ float lds_ld(unsigned addr) {
  float v;
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF));
  return v;
}
__device__ __forceinline__ unsigned lds_addr(const float* p) {
  return (unsigned)(uintptr_t)(lds_vp)p;
}
// operands of one (cg,ty) group: KW taps x (MT weight blocks + NT position blocks)
template <int MT, int NT, int KW, int ASTRIDE>
struct GroupRegs {
  float a[KW][MT];
  float b[KW][NT];
  template <int... I>
  __device__ __forceinline__ void load_a(unsigned addr, std::integer_sequence<int, I...>) {
    ((a[I / MT][I % MT] = lds_ld<(I / MT) * ASTRIDE + (I % MT) * 64>(addr)), ...);
  }
  template <int NB, int... I>
  __device__ __forceinline__ void load_b1(unsigned addr, std::integer_sequence<int, I...>) {
    ((b[I][NB] = lds_ld<I * 4>(addr)), ...);
  }
  template <int... NB>
  __device__ __forceinline__ void load_b(const unsigned (&addr)[NT],
                                         std::integer_sequence<int, NB...>) {
    (load_b1<NB>(addr[NB], std::make_integer_sequence<int, KW>{}), ...);
  }
  __device__ __forceinline__ void load(unsigned addrA, const unsigned (&addrB)[NT]) {
    load_a(addrA, std::make_integer_sequence<int, KW * MT>{});
    load_b(addrB, std::make_integer_sequence<int, NT>{});
  }
  // keep every asm-load destination allocated until the s_waitcnt that retires
  // the loads: a destination that is never read again (prefetch past the end of
  // a chunk) could otherwise be reused while its LDS data is still in flight.
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

// One group step, hand-scheduled: MFMA i of the CURRENT group, then LDS reads
// [R0, R1) of the NEXT group.  The reads are spread over the first ~3/4 of the
// MFMAs: a wave may have at most 15 LDS operations outstanding (lgkmcnt is 4
// bits), so a burst of KW*(MT+NT) reads stalls the wave -- and the matrix pipe
// behind it -- while 1-2 reads per 32-cycle MFMA slot are free.
template <int MT, int NT, int KW, int ASTRIDE, int R>
__device__ __forceinline__ void group_read(GroupRegs<MT, NT, KW, ASTRIDE>& g, unsigned addrA,
                                           const unsigned (&addrB)[NT]) {
  constexpr int tx = R / (MT + NT), k = R % (MT + NT);
  if constexpr (k < MT) g.a[tx][k] = lds_ld<tx * ASTRIDE + k * 64>(addrA);
  else g.b[tx][k - MT] = lds_ld<tx * 4>(addrB[k - MT]);
}
template <int MT, int NT, int KW, int ASTRIDE, int R0, int R1>
__device__ __forceinline__ void group_reads(GroupRegs<MT, NT, KW, ASTRIDE>& g, unsigned addrA,
                                            const unsigned (&addrB)[NT]) {
  if constexpr (R0 < R1) {
    group_read<MT, NT, KW, ASTRIDE, R0>(g, addrA, addrB);
    group_reads<MT, NT, KW, ASTRIDE, R0 + 1, R1>(g, addrA, addrB);
  }
}
template <int MT, int NT, int KW, int ASTRIDE, int I>
__device__ __forceinline__ void group_steps(const GroupRegs<MT, NT, KW, ASTRIDE>& cur,
                                            GroupRegs<MT, NT, KW, ASTRIDE>& nxt, f32x4 (&acc)[MT][NT],
                                            unsigned addrA, const unsigned (&addrB)[NT]) {
  constexpr int M = KW * MT * NT, R = KW * (MT + NT);
  constexpr int tx = I / (MT * NT), mb = (I / NT) % MT, nb = I % NT;
  acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.a[tx][mb], cur.b[tx][nb], acc[mb][nb],
                                                     0, 0, 0);
  constexpr int r0 = (I * R * 4) / (3 * M) < R ? (I * R * 4) / (3 * M) : R;
  constexpr int r1 = ((I + 1) * R * 4) / (3 * M) < R ? ((I + 1) * R * 4) / (3 * M) : R;
  group_reads<MT, NT, KW, ASTRIDE, r0, r1>(nxt, addrA, addrB);
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (I + 1 < M) group_steps<MT, NT, KW, ASTRIDE, I + 1>(cur, nxt, acc, addrA, addrB);
}
template <int MT, int NT, int KW, int ASTRIDE, int I>
__device__ __forceinline__ void group_mfma_only(const GroupRegs<MT, NT, KW, ASTRIDE>& cur,
                                                f32x4 (&acc)[MT][NT]) {
  constexpr int tx = I / (MT * NT), mb = (I / NT) % MT, nb = I % NT;
  acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.a[tx][mb], cur.b[tx][nb], acc[mb][nb],
                                                     0, 0, 0);
  if constexpr (I + 1 < KW * MT * NT) group_mfma_only<MT, NT, KW, ASTRIDE, I + 1>(cur, acc);
}

constexpr int igemm_bmpad(int MT) {          // row stride == 16 (mod 32)
  return ((16 * MT) & 31) == 16 ? 16 * MT : 16 * MT + 16;
}

// Pipeline: two LDS buffers.  While the MFMAs of chunk k run out of buffer
// k&1, the LDS-DMA of chunk k+1 lands in the other one; one barrier per chunk
// (wait vmcnt(0) -> barrier -> issue next DMA -> compute).
//
// K order inside a chunk: channel group cg (4 channels = the 4 lane quarters),
// then tap row ty, then tap column tx.  The packed weight image and the LDS A
// tile use the same order, so (a) the weight staging is a linear copy, (b) the
// A rows of one (cg,ty) group are contiguous, and (c) with the kernel width KW
// a template parameter the tx loop is fully unrolled: every LDS address of a
// group is "loop-invariant VGPR + immediate", which keeps the scalar / vector
// overhead per MFMA small (the matrix pipe, not instruction issue, must be the
// limiter: rocprof showed 6 SALU + 4 VALU per MFMA in the first version).
template <int MT, int NT, int KW>
__global__ __launch_bounds__(256, 2) void igemm_kernel(IgemmP p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int BM = 16 * MT, BN = 64 * NT;
  constexpr int BMpad = igemm_bmpad(MT);
  constexpr int BMp4 = BMpad / 4;
  const int kw = KW > 0 ? KW : p.kw;
  const int CC = p.CC, CG = p.CC >> 2;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, qd = lane >> 4;

  int bid = blockIdx.x;
  const int pt = bid % p.nPT; bid /= p.nPT;
  const int z = bid % p.Do;  bid /= p.Do;
  const int mt = bid % p.nMT; bid /= p.nMT;
  const int ks = bid % p.splitK;
  const int n = bid / p.splitK;

  const int m0 = mt * BM;
  const int q0 = pt * BN;
  const int qlast = min(q0 + BN, p.Q) - 1;
  const int r0 = q0 / p.Wo, c0 = q0 - r0 * p.Wo;
  const int rl = qlast / p.Wo, cl = qlast - rl * p.Wo;
  const int isY = (int)p.isY;
  const long span_lo = (long)r0 * p.isY + c0;
  const int L = (rl - r0) * isY + (cl - c0) + (p.kh - 1) * isY + p.kw;
  const int nJ = (L + 63) >> 6;
  const int nJ16 = (L + 255) >> 8;
  const int Lpad = p.Lpad;
  const int xFloats = CC * Lpad;

  int posoff[NT];
#pragma unroll
  for (int nb = 0; nb < NT; ++nb) {
    int q = min(q0 + wave * (16 * NT) + nb * 16 + l15, p.Q - 1);
    int r = q / p.Wo, c = q - r * p.Wo;
    posoff[nb] = (r - r0) * isY + (c - c0) + qd * Lpad;
  }

  f32x4 acc[MT][NT];
#pragma unroll
  for (int mb = 0; mb < MT; ++mb)
#pragma unroll
    for (int nb = 0; nb < NT; ++nb) acc[mb][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nChunks = p.kd * p.nChunkC;
  const int per = (nChunks + p.splitK - 1) / p.splitK;
  const int cb = ks * per, ce = min(cb + per, nChunks);
  const int aBase = qd * BMpad + l15;
  const int nRows = p.THW * CC;
  const int nPieces = (nRows * BMp4 + 63) >> 6;
  const float* in_n = p.in + (long)n * p.isN + (long)z * p.isZ + span_lo;

  auto stage = [&](int ch, int buf) {
    const int dz = ch / p.nChunkC;
    const int cgi0 = (ch - dz * p.nChunkC) * CG;
    float* xl = smem + buf * p.bufFloats;
    float* wl = xl + xFloats;
    const float* xb = in_n + (long)dz * p.isZ;
    for (int cc = wave; cc < CC; cc += 4) {
      const int ci = min(cgi0 * 4 + cc, p.Cin - 1);   // padded channels carry zero weights
      const float* src = xb + (long)ci * p.isC;
      float* dst = xl + cc * Lpad;
      // 16-byte pieces (256 floats per wave instruction); lanes past the span
      // are masked off, the straddling lane over-reads <= 12 bytes, which stays
      // inside the tensor except on its very last row: that row goes by dwords.
      const bool tail_row = (ci == p.Cin - 1) && (z + dz == p.Din - 1) && (n == p.N - 1);
      if (!tail_row) {
        for (int j = 0; j < nJ16; ++j) {
          const int u = 256 * j + 4 * lane;
          if (u < L) glds16(src + u, dst + 256 * j);
        }
      } else {
        for (int j = 0; j < nJ; ++j) glds4(src + min(64 * j + lane, L - 1), dst + 64 * j);
      }
    }
    // weights: rows [cg][ty][tx][qd] are contiguous in the packed image
    const float* wb = p.wp + ((long)(dz * (p.ciP >> 2) + cgi0) * p.THW * 4) * p.coP + m0;
    for (int pc = wave; pc < nPieces; pc += 4) {
      const int s = pc * 64 + lane;
      int row = s / BMp4;
      int c4 = s - row * BMp4;
      row = min(row, nRows - 1);
      c4 = min(c4, BM / 4 - 1);
      glds16(wb + (long)row * p.coP + 4 * c4, wl + pc * 256);
    }
  };

  if (cb < ce) stage(cb, 0);
  for (int ch = cb; ch < ce; ++ch) {
    const int cur = (ch - cb) & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (!(p.dbg & 4)) __syncthreads();
    if (ch + 1 < ce && !(p.dbg & 1)) stage(ch + 1, cur ^ 1);
    const float* xl = smem + cur * p.bufFloats;
    const float* wl = xl + xFloats + aBase;

    if (p.dbg & 2) continue;
    if constexpr (KW > 0) {
      // groups g = (cg, ty) flattened; A rows of consecutive groups are
      // contiguous, the B base moves by isY per ty and wraps per cg.
      const int nG = CG * p.kh;
      constexpr int ASTR = 4 * BMpad * 4;                  // bytes per tap of A rows
      GroupRegs<MT, NT, KW, ASTR> g0, g1;
      unsigned addrA = lds_addr(wl);
      unsigned addrB[NT];
      const unsigned xbase = lds_addr(xl);
#pragma unroll
      for (int nb = 0; nb < NT; ++nb) addrB[nb] = xbase + 4u * (unsigned)posoff[nb];
      const unsigned stepY = 4u * (unsigned)isY;
      const unsigned wrapCg = 4u * (unsigned)(4 * Lpad) - (unsigned)p.kh * stepY;
      int ty = 0;
#define E2_NEXT()                                                        \
      {                                                                  \
        addrA += KW * ASTR;                                              \
        ++ty;                                                            \
        const unsigned d = (ty == p.kh) ? (stepY + wrapCg) : stepY;      \
        ty = (ty == p.kh) ? 0 : ty;                                      \
        _Pragma("unroll") for (int nb = 0; nb < NT; ++nb) addrB[nb] += d; \
      }
#define E2_MFMA(G)                                                       \
      _Pragma("unroll") for (int tx = 0; tx < KW; ++tx)                  \
      _Pragma("unroll") for (int mb = 0; mb < MT; ++mb)                  \
      _Pragma("unroll") for (int nb = 0; nb < NT; ++nb)                  \
        acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(G.a[tx][mb], G.b[tx][nb], acc[mb][nb], 0, 0, 0);
#define E2_WAIT()                                                        \
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                 \
      __builtin_amdgcn_sched_barrier(0);
      g0.load(addrA, addrB);
      E2_WAIT()
      g0.touch();
      int g = 0;
      for (; g + 1 < nG; g += 2) {
        E2_NEXT()
#if E2_INTERLEAVE
        __builtin_amdgcn_sched_barrier(0);
        group_steps<MT, NT, KW, ASTR, 0>(g0, g1, acc, addrA, addrB);   // compute g, fetch g+1
        E2_WAIT()
        g1.touch();
        E2_NEXT()
        __builtin_amdgcn_sched_barrier(0);
        group_steps<MT, NT, KW, ASTR, 0>(g1, g0, acc, addrA, addrB);   // past the end: slack
        E2_WAIT()
        g0.touch();
#else
        g1.load(addrA, addrB);            // group g+1 in flight ...
        __builtin_amdgcn_sched_barrier(0);
        E2_MFMA(g0)                       // ... while group g computes
        __builtin_amdgcn_sched_barrier(0);
        E2_WAIT()
        g1.touch();
        E2_NEXT()
        g0.load(addrA, addrB);            // group g+2 (past the end: reads slack)
        __builtin_amdgcn_sched_barrier(0);
        E2_MFMA(g1)
        __builtin_amdgcn_sched_barrier(0);
        E2_WAIT()
        g0.touch();
#endif
      }
      if (g < nG) { E2_MFMA(g0) }
#undef E2_NEXT
#undef E2_MFMA
#undef E2_WAIT
    } else {
    for (int cg = 0; cg < CG; ++cg) {
      for (int ty = 0; ty < p.kh; ++ty) {
        const float* ap = wl + ((cg * p.kh + ty) * kw) * (4 * BMpad);
        const float* bq = xl + 4 * cg * Lpad + ty * isY;
        const float* bp[NT];
#pragma unroll
        for (int nb = 0; nb < NT; ++nb) bp[nb] = bq + posoff[nb];
        for (int tx = 0; tx < kw; ++tx) {
          float a0[MT], b0[NT];
#pragma unroll
          for (int mb = 0; mb < MT; ++mb) a0[mb] = ap[tx * (4 * BMpad) + mb * 16];
#pragma unroll
          for (int nb = 0; nb < NT; ++nb) b0[nb] = bp[nb][tx];
#pragma unroll
          for (int mb = 0; mb < MT; ++mb)
#pragma unroll
            for (int nb = 0; nb < NT; ++nb)
              acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[mb], b0[nb],
                                                                 acc[mb][nb], 0, 0, 0);
        }
      }
    }
    }
  }

  // ---- epilogue: D col = position (lane&15), row = channel 4*qd+reg -------
  const int R = p.upz * p.upy * p.upx;
#pragma unroll
  for (int nb = 0; nb < NT; ++nb) {
    const int q = q0 + wave * (16 * NT) + nb * 16 + l15;
    if (q >= p.Q) continue;
    const int r = q / p.Wo, c = q - r * p.Wo;
#pragma unroll
    for (int mb = 0; mb < MT; ++mb) {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int co = m0 + mb * 16 + 4 * qd + rr;
        if (co >= p.Cout) continue;
        float* dst;
        if (R == 1) {
          dst = p.out + (long)n * p.osN + (long)co * p.osC + (long)z * p.osZ +
                (long)r * p.osY + c;
        } else {
          const int cr = co / R, sub = co - cr * R;
          const int rz = sub / (p.upy * p.upx);
          const int rem = sub - rz * (p.upy * p.upx);
          const int ry = rem / p.upx, rx = rem - ry * p.upx;
          dst = p.out + (long)n * p.osN + (long)cr * p.osC +
                (long)(z * p.upz + rz) * p.osZ + (long)(r * p.upy + ry) * p.osY +
                (c * p.upx + rx);
        }
        const float v = acc[mb][nb][rr];
        if (p.atomic) unsafeAtomicAdd(dst, v);
        else *dst = v;
      }
    }
  }
}

// ---- weight packing ---------------------------------------------------------
// Wp[dz][cg = ic/4][t = ty*kw+tx][qd = ic%4][oc (coP)], zero padded: the K order
// the kernel consumes, so staging a chunk is a linear copy of rows.
__global__ void pack_weights_kernel(const float* __restrict__ w, float* __restrict__ wp,
                                    int Cout, int Cin, int kd, int THW, long wsO,
                                    long wsI, int flip, int ciP, int coP, int Rout,
                                    int Rin) {
  const long total = (long)kd * THW * ciP * coP;
  const int T = kd * THW;
  const int nCG = ciP >> 2;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int oc = (int)(i % coP);
    long r = i / coP;
    const int qd = (int)(r & 3); r >>= 2;
    const int t = (int)(r % THW); r /= THW;
    const int cg = (int)(r % nCG);
    const int dz = (int)(r / nCG);
    const int ic = cg * 4 + qd;
    float v = 0.f;
    if (oc < Cout && ic < Cin) {
      const int tl = dz * THW + t;
      const int tap = flip ? (T - 1 - tl) : tl;
      // Rout/Rin > 1: UpConv sub-position folded into the channel index
      v = w[(long)(oc / Rout) * wsO + (long)(ic / Rin) * wsI + tap + (oc % Rout) +
            (ic % Rin)];
    }
    wp[i] = v;
  }
}

// all layers' images in one launch (blockIdx.y = job): the repack of every conv
// weight tensor after an optimiser step costs one kernel instead of 2 per layer
struct PackJobDev {
  const float* w;
  float* wp;
  int Cout, Cin, kd, THW;
  long wsO, wsI;
  int flip, ciP, coP;
  long total;
};
__global__ void pack_multi_kernel(const PackJobDev* __restrict__ jobs) {
  const PackJobDev j = jobs[blockIdx.y];
  const int T = j.kd * j.THW;
  const int nCG = j.ciP >> 2;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < j.total;
       i += (long)gridDim.x * blockDim.x) {
    const int oc = (int)(i % j.coP);
    long r = i / j.coP;
    const int qd = (int)(r & 3); r >>= 2;
    const int t = (int)(r % j.THW); r /= j.THW;
    const int cg = (int)(r % nCG);
    const int dz = (int)(r / nCG);
    const int ic = cg * 4 + qd;
    float v = 0.f;
    if (oc < j.Cout && ic < j.Cin) {
      const int tl = dz * j.THW + t;
      v = j.w[(long)oc * j.wsO + (long)ic * j.wsI + (j.flip ? (T - 1 - tl) : tl)];
    }
    j.wp[i] = v;
  }
}

extern "C" size_t e2_pack_job_bytes(void) { return sizeof(PackJobDev); }

/* fill one host-side job record (to be copied into a device array) */
extern "C" int e2_pack_job_fill(void* rec, const float* w, void* wp, int cout, int cin, int kd,
                                int kh, int kw, int mode) {
  E2_REQUIRE(rec && w && wp, "pack_job_fill: null argument");
  PackJobDev* j = (PackJobDev*)rec;
  const int T = kd * kh * kw;
  j->w = w; j->wp = (float*)wp; j->kd = kd; j->THW = kh * kw;
  if (mode == 0) {
    j->Cout = cout; j->Cin = cin; j->wsO = (long)cin * T; j->wsI = T; j->flip = 1;
    e2i_pack_dims(cout, cin, &j->ciP, &j->coP);
  } else {
    j->Cout = cin; j->Cin = cout; j->wsO = T; j->wsI = (long)cin * T; j->flip = 0;
    e2i_pack_dims(cin, cout, &j->ciP, &j->coP);
  }
  j->total = (long)kd * kh * kw * j->ciP * j->coP;
  return 0;
}

extern "C" int e2_conv3d_pack_multi(e2_ctx* ctx, const void* jobs_dev, int njobs) {
  E2_REQUIRE(ctx && jobs_dev && njobs > 0 && njobs < 65536, "pack_multi: bad argument");
  hipLaunchKernelGGL(pack_multi_kernel, dim3(256, njobs), dim3(256), 0, ctx->stream,
                     (const PackJobDev*)jobs_dev);
  E2_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---- host side ----------------------------------------------------------------
struct IgemmCfg { int MT, NT, CC, SK; };

template <int MT, int NT, int KW>
static int launch_one(e2_ctx* ctx, const IgemmP& p, int grid, size_t lds) {
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(
        reinterpret_cast<const void*>(&igemm_kernel<MT, NT, KW>),
        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) { e2_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return 1; }
    attr_done = true;
  }
  hipLaunchKernelGGL((igemm_kernel<MT, NT, KW>), dim3(grid), dim3(256), lds, ctx->stream, p);
  E2_CHECK_HIP(hipGetLastError());
  return 0;
}

static const int kMTs[] = {1, 2, 3, 4, 5, 6, 7, 8, 10, 13};

template <int MT, int NT>
static int dispatch_kw(e2_ctx* ctx, const IgemmP& p, int grid, size_t lds) {
  switch (p.kw) {
    case 1: return launch_one<MT, NT, 1>(ctx, p, grid, lds);
    case 3: return launch_one<MT, NT, 3>(ctx, p, grid, lds);
    case 4: return launch_one<MT, NT, 4>(ctx, p, grid, lds);
    case 5: return launch_one<MT, NT, 5>(ctx, p, grid, lds);
    default: return launch_one<MT, NT, 0>(ctx, p, grid, lds);   // runtime tx loop
  }
}

static int dispatch(e2_ctx* ctx, const IgemmP& p, int MT, int NT, int grid, size_t lds) {
#define E2_CASE(M)                                                          \
  case M:                                                                   \
    if (NT == 1) return dispatch_kw<M, 1>(ctx, p, grid, lds);               \
    if (NT == 2) return dispatch_kw<M, 2>(ctx, p, grid, lds);               \
    break;
#define E2_CASE4(M)                                                         \
  case M:                                                                   \
    if (NT == 1) return dispatch_kw<M, 1>(ctx, p, grid, lds);               \
    if (NT == 2) return dispatch_kw<M, 2>(ctx, p, grid, lds);               \
    if (NT == 4) return dispatch_kw<M, 4>(ctx, p, grid, lds);               \
    break;
  switch (MT) {
    E2_CASE4(1) E2_CASE4(2) E2_CASE4(3) E2_CASE4(4) E2_CASE4(5) E2_CASE(6)
    E2_CASE(7) E2_CASE(8) E2_CASE(10) E2_CASE(13)
  }
#undef E2_CASE
#undef E2_CASE4
  e2_set_error("igemm: no instance MT=%d NT=%d", MT, NT);
  return 2;
}

static int pad16mod32(int v) {          // smallest s >= v with s % 32 == 16
  int s = ((v + 15) / 16) * 16;
  if ((s & 31) == 0) s += 16;
  return s;
}

static int span_rows(int BN, int Wo) { return (BN + Wo - 2) / Wo; }

// exact upper bound of the span length of a BN-position tile:
// L = (q_last - q_first) + rows_crossed*(isY - Wo) + (kh-1)*isY + kw
static int span_lmax(const IgemmArgs& a, int BN) {
  return (BN - 1) + span_rows(BN, a.Wo) * ((int)a.isY - a.Wo) + (a.kh - 1) * (int)a.isY + a.kw;
}
// LDS-DMA writes whole 256-float pieces (16 B per lane): rows hold a multiple of 256
static int span_lpad(const IgemmArgs& a, int BN) {
  return pad16mod32(((span_lmax(a, BN) + 255) / 256) * 256);
}
static size_t buf_floats(const IgemmArgs& a, int MT, int BN, int CC) {
  const int THW = a.kh * a.kw;
  return (size_t)CC * span_lpad(a, BN) + (size_t)THW * CC * igemm_bmpad(MT) + 256 +
         (size_t)(8 + 4 * 6) * igemm_bmpad(MT) + 4 * (size_t)span_lpad(a, BN) + 64;
  // DMA piece slack + one group of A/B prefetch past the end
}

// Pick the tiling.  Cost model (cycles): work-groups run 1 or 2 per CU
// (LDS-limited); a work-group's time is its MFMA count per wave at 32 cycles,
// stretched when two share the SIMDs, plus prologue/epilogue; split-K pays
// the chip-wide fp32 atomic rate (1.3 TB/s) and a memset.
static IgemmCfg choose_cfg(const e2_ctx* ctx, const IgemmArgs& a, int* ok) {
  const int mblocks = e2_cdiv(a.Cout, 16);
  const int THW = a.kh * a.kw;
  const long Q = (long)a.Ho * a.Wo;
  IgemmCfg best{0, 0, 0, 0};
  double bestCost = 1e300;
  const char* force = getenv("E2_IGEMM_FORCE");
  if (force) {
    IgemmCfg f{0, 0, 0, 0};
    if (sscanf(force, "%d,%d,%d,%d", &f.MT, &f.NT, &f.CC, &f.SK) == 4) { *ok = 1; return f; }
  }
  const double out_bytes = 4.0 * a.N * a.Cout * a.Do * (double)Q;
  for (int MT : kMTs) {
    if (MT > mblocks && MT != 1) continue;
    const int nMT = e2_cdiv(mblocks, MT);
    for (int NT = 1; NT <= 4; NT *= 2) {
      if (NT == 4 && MT > 5) continue;
      const int BN = 64 * NT;
      const int nPT = (int)((Q + BN - 1) / BN);
      const int cinP = ((a.Cin + 3) / 4) * 4;
      for (int CC = 4; CC <= 32 && CC <= cinP; CC += 4) {
        if (CC > 4 && e2_cdiv(a.Cin, CC) == e2_cdiv(a.Cin, CC - 4)) continue;  // no fewer chunks
        const size_t lds = 2 * buf_floats(a, MT, BN, CC) * 4;
        if (lds > 160 * 1024) break;
        const int perCU = lds <= 80 * 1024 ? 2 : 1;
        const int slots = ctx->num_cu * perCU;
        const int nChunkC = e2_cdiv(a.Cin, CC);
        const int nChunks = a.kd * nChunkC;
        const long wgs0 = (long)a.N * a.Do * nPT * nMT;
        const double chunk_bytes = 4.0 * (CC * (double)span_lmax(a, BN) + THW * CC * 16.0 * MT);
        for (int SK = 1; SK <= 8; SK *= 2) {
          if (SK > nChunks) break;
          const long wgs = wgs0 * SK;
          const int per = e2_cdiv(nChunks, SK);
          const double mfma = (double)MT * NT * THW * (CC / 4) * 32.0;      // per chunk
          const double issue = (MT + 2.0 * NT + 6) * THW * (CC / 4) * 5.0;  // non-MFMA issue
          double chunk = std::max(mfma, issue) * perCU * 1.05;
          // the next chunk's DMA must land while this one computes
          chunk = std::max(chunk, std::max(3500.0, chunk_bytes / 48.0 * perCU));
          const double wg_time = per * chunk + 4500.0 + MT * NT * 16 * 6.0;
          const double rounds = (double)((wgs + slots - 1) / slots);
          double cost = rounds * wg_time;
          if (SK > 1) cost += out_bytes * SK / 1.3e12 * 2.4e9 + out_bytes / 4e12 * 2.4e9 + 4000.0;
          if (cost < bestCost) { bestCost = cost; best = IgemmCfg{MT, NT, CC, SK}; }
        }
      }
    }
  }
  *ok = best.MT != 0;
  return best;
}

void e2i_pack_dims(int cout, int cin, int* ciP, int* coP) {
  *ciP = ((cin + 3) / 4) * 4 + 32;        // any channel-chunk size up to 32+
  *coP = ((cout + 15) / 16) * 16 + 16 * 13;   // room for any MT tiling
}

int e2i_pack_weights(e2_ctx* ctx, const float* w, float* wp, int Cout, int Cin,
                     int kd, int kh, int kw, int64_t wsO, int64_t wsI, int flip,
                     int ciP, int coP, int Rout, int Rin) {
  const long total = (long)kd * kh * kw * ciP * coP;
  int grid = (int)std::min<long>((total + 255) / 256, 4096);
  hipLaunchKernelGGL(pack_weights_kernel, dim3(grid), dim3(256), 0, ctx->stream, w, wp,
                     Cout, Cin, kd, kh * kw, (long)wsO, (long)wsI, flip, ciP, coP, Rout, Rin);
  E2_CHECK_HIP(hipGetLastError());
  return 0;
}

int e2i_igemm_conv(e2_ctx* ctx, const IgemmArgs& a) {
  E2_REQUIRE(a.Do > 0 && a.Ho > 0 && a.Wo > 0 && a.Cin > 0 && a.Cout > 0,
             "igemm: empty problem");
  E2_REQUIRE(a.isY < (1 << 20), "igemm: input row stride too large");
  int ok = 0;
  IgemmCfg c = choose_cfg(ctx, a, &ok);
  E2_REQUIRE(ok, "igemm: no tiling fits LDS (Cin=%d Cout=%d k=%dx%dx%d W=%d)", a.Cin,
             a.Cout, a.kd, a.kh, a.kw, a.Wo);
  E2_REQUIRE(c.CC >= 4 && c.CC % 4 == 0, "igemm: CC must be a multiple of 4");
  IgemmP p;
  p.in = a.in; p.wp = a.wp; p.out = a.out;
  p.Cin = a.Cin; p.Cout = a.Cout; p.kd = a.kd; p.kh = a.kh; p.kw = a.kw;
  p.THW = a.kh * a.kw;
  p.Do = a.Do; p.Ho = a.Ho; p.Wo = a.Wo; p.Q = a.Ho * a.Wo;
  p.isN = a.isN; p.isC = a.isC; p.isZ = a.isZ; p.isY = a.isY;
  p.osN = a.osN; p.osC = a.osC; p.osZ = a.osZ; p.osY = a.osY;
  p.ciP = a.ciP; p.coP = a.coP;
  const int BN = 64 * c.NT;
  p.Lpad = span_lpad(a, BN);
  p.CC = c.CC;
  p.Din = a.Do + a.kd - 1;
  p.dbg = getenv("E2_IGEMM_DBG") ? atoi(getenv("E2_IGEMM_DBG")) : 0;
  p.N = a.N;
  p.nPT = e2_cdiv(p.Q, BN);
  p.nMT = e2_cdiv(e2_cdiv(a.Cout, 16), c.MT);
  p.splitK = c.SK;
  p.nChunkC = e2_cdiv(a.Cin, c.CC);
  p.atomic = (c.SK > 1) ? 1 : 0;
  p.upz = a.upz; p.upy = a.upy; p.upx = a.upx;
  p.bufFloats = (int)buf_floats(a, c.MT, BN, c.CC);
  E2_REQUIRE(p.nMT * 16 * c.MT <= a.coP, "igemm: packed coP too small");
  E2_REQUIRE(p.nChunkC * c.CC <= a.ciP, "igemm: packed ciP too small");
  const size_t lds = 2 * (size_t)p.bufFloats * 4;
  E2_REQUIRE(lds <= 160 * 1024, "igemm: forced tiling needs %zu B of LDS", lds);
  const long grid = (long)a.N * p.splitK * p.nMT * p.Do * p.nPT;
  E2_REQUIRE(grid < (1L << 31), "igemm: grid too large");
  if (p.atomic) {
    // split-K accumulates with atomics: start from zero
    const int R = a.upz * a.upy * a.upx;
    const int oc = a.Cout / (R > 1 ? R : 1);
    const int od = a.Do * a.upz, oh = a.Ho * a.upy, ow = a.Wo * a.upx;
    if (a.osY == ow && a.osZ == (long)oh * ow && a.osC == (long)od * oh * ow &&
        (a.N == 1 || a.osN == (long)oc * od * oh * ow)) {
      E2_CHECK_HIP(hipMemsetAsync(a.out, 0, sizeof(float) * (size_t)a.N * oc * od * oh * ow,
                                  ctx->stream));
    } else {
      e2_tensor5 v{a.out, a.N, oc, od, oh, ow, a.osN, a.osC, a.osZ, a.osY};
      int rc = e2i_fill_view(ctx, &v, 0.f);
      if (rc) return rc;
    }
  }
  if (getenv("E2_VERBOSE"))
    fprintf(stderr, "[e2] igemm Cin=%d Cout=%d k=%d,%d,%d out=%d,%d,%d MT=%d NT=%d CC=%d SK=%d grid=%ld lds=%zu\n",
            a.Cin, a.Cout, a.kd, a.kh, a.kw, a.Do, a.Ho, a.Wo, c.MT, c.NT, c.CC, c.SK, grid, lds);
  return dispatch(ctx, p, c.MT, c.NT, (int)grid, lds);
}
