// igemm_core.hpp -- device side of the implicit-GEMM convolution (see conv_igemm.hip
// for the algorithm and the host side).  Included by the per-kernel-width
// translation units conv_igemm_k{1,3,4,5}.hip so that they compile in parallel.
#pragma once
#include "common.hpp"
#include <utility>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lds_vp;
typedef const __attribute__((address_space(1))) void* gbl_vp;

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
  int dbg;                // timing ablations only (E2_IGEMM_DBG): 1 = stage once, 2 = no MFMA
  int nPT, nMT, splitK, nChunkC;
  int atomic;
  int upz, upy, upx;
  int bufFloats;          // floats per LDS buffer
  unsigned long long* stamps;   // debug (E2_IGEMM_STAMPS): 8 s_memtime stamps per work-group
  const float* bias;      // fused bias (+ act) in the wide epilogue, or nullptr
  int act;
  int bf16;               // 1: operands rounded to bf16 for the matrix core (f32 sums)
  int wide;               // 1: epilogue transposes the tile through LDS and stores 16 B per lane
                          //    (no split-K, no UpConv scatter; rows of the output may be wider
                          //    than Wo -- the interior of a zero-padded buffer)
  // gradient-mask epilogue (data gradient fused with the activation backward of the layer
  // that PRODUCED this conv's input): out = acc * act'(gm_src), gm_dbias[co] += sum(out).
  // gm_src = that layer's activated output (dense rows, relu slopes read off its sign:
  // > 0 -> 1, +0.0 -> 0.5, -0.0 -> 0), nullptr for a linear activation.
  // gm_bias != nullptr: gm_src is that layer's PRE-activation y and the slope is taken from
  // y + gm_bias[co] (> 0 -> 1, == 0 -> 0.5, < 0 -> 0).
  int gm = 0;
  const float* gm_src = nullptr;
  long gsN = 0, gsC = 0, gsZ = 0;
  float* gm_dbias = nullptr;
  const float* gm_bias = nullptr;
  // data gradient: the first and last `zpad` z-planes of `in` are the zero border of the
  // padded gradient buffer.  A tap plane dz that reads only border for this tile's output
  // plane z is skipped (exact: it would add 0 * w); the K range of the tile shrinks to
  // dz in [max(0, zpad - z), min(kd - 1, Din - 1 - zpad - z)].
  int zpad = 0;
  // split-K WITHOUT atomics: split ks stores its partial sums (plain stores, any epilogue)
  // to out + ks * partStride; the consumer's pointwise kernel adds the parts up
  // (e2_pool_bias_act_fwd_parts / _bwd_parts).  No zero fill; a split whose K range is
  // empty (clipped border plane) stores zeros.
  int parts = 0;
  long partStride = 0;
};

// K range (in channel chunks) of output plane z: [lo, hi)
__device__ __forceinline__ void e2_chunk_range(const IgemmP& p, int z, int& lo, int& hi) {
  const int dzlo = max(0, p.zpad - z);
  const int dzhi = min(p.kd - 1, p.Din - 1 - p.zpad - z);
  lo = dzlo * p.nChunkC;
  hi = (dzhi + 1) * p.nChunkC;
}

// relu slope of the producing layer: o = its activated output (pre = false: the sign of a
// zero tells 0.5 from 0) or its pre-activation plus bias (pre = true)
__device__ __forceinline__ float e2_relu_slope(float o, bool pre) {
  return (o > 0.f) ? 1.f : (pre ? (o == 0.f ? 0.5f : 0.f) : (__builtin_signbit(o) ? 0.f : 0.5f));
}

// async global -> LDS copies (no VGPR destination); LDS address is
// wave-uniform base + lane*size, the global source is per lane.
__device__ __forceinline__ void glds4(const float* g, float* l) {
  __builtin_amdgcn_global_load_lds((gbl_vp)g, (lds_vp)l, 4, 0, 0);
}
__device__ __forceinline__ void glds16(const float* g, float* l) {
  __builtin_amdgcn_global_load_lds((gbl_vp)g, (lds_vp)l, 16, 0, 0);
}
__device__ __forceinline__ unsigned lds_addr(const float* p) {
  return (unsigned)(uintptr_t)(lds_vp)p;
}

// Operand loads are inline asm so that hipcc's waitcnt pass does not see them:
// the kernel retires them with its own s_waitcnt AFTER the MFMAs of the current
// group have been issued (cdna guide 5.7 form iii).
template <int OFF>
__device__ __forceinline__ float lds_ld(unsigned addr) {
  float v;
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF));
  return v;
}
// weights: scalar base + per-lane byte offset + immediate
template <int OFF>
__device__ __forceinline__ float gl_ld(const float* sbase, unsigned voff) {
  float v;
  asm volatile("global_load_dword %0, %1, %2 offset:%3"
               : "=v"(v) : "v"(voff), "s"(sbase), "i"(OFF));
  return v;
}

// operands of one group: TPG taps x (MT weight blocks + NT position blocks)
template <int MT, int NT, int TPG>
struct GRegs {
  float a[TPG][MT];
  float b[TPG][NT];
  // keep every asm-load destination allocated until the s_waitcnt that retires
  // the loads: a destination that is never read again could otherwise be handed
  // to another value while its data is still in flight.
  __device__ __forceinline__ void touch() {
#pragma unroll
    for (int t = 0; t < TPG; ++t) {
#pragma unroll
      for (int mb = 0; mb < MT; ++mb) asm volatile("" : "+v"(a[t][mb]));
#pragma unroll
      for (int nb = 0; nb < NT; ++nb) asm volatile("" : "+v"(b[t][nb]));
    }
  }
};

// addresses of a group's operands
template <int NT, int KW, int GU>
struct GAddr {
  const float* abase;           // SGPR pair: first weight row of the group (+ m0)
  unsigned voff[KW * GU];       // per-lane byte offset of tap j: 4*((4j+qd)*coP + l15)
  unsigned b[GU][NT];           // LDS byte address of tap 0 of channel group u
};

// read r of a group, in issue order: all weight (global) loads first -- they have
// the longer latency -- then the input (LDS) reads
template <int MT, int NT, int KW, int GU, int R>
__device__ __forceinline__ void group_read(GRegs<MT, NT, KW * GU>& g, const GAddr<NT, KW, GU>& ad) {
  constexpr int TPG = KW * GU;
  if constexpr (R < TPG * MT) {
    constexpr int j = R / MT, mb = R % MT;
    g.a[j][mb] = gl_ld<mb * 64>(ad.abase, ad.voff[j]);
  } else {
    constexpr int rb = R - TPG * MT;
    constexpr int j = rb / NT, nb = rb % NT;
    g.b[j][nb] = lds_ld<(j % KW) * 4>(ad.b[j / KW][nb]);
  }
}
template <int MT, int NT, int KW, int GU, int R0, int R1>
__device__ __forceinline__ void group_reads(GRegs<MT, NT, KW * GU>& g, const GAddr<NT, KW, GU>& ad) {
  if constexpr (R0 < R1) {
    group_read<MT, NT, KW, GU, R0>(g, ad);
    group_reads<MT, NT, KW, GU, R0 + 1, R1>(g, ad);
  }
}
// One group step, hand-scheduled: MFMA i of the CURRENT group, then reads
// [r0, r1) of the NEXT group, spread over the first ~3/4 of the MFMAs (a wave may
// have at most 15 LDS operations outstanding -- lgkmcnt is 4 bits -- so a burst
// would stall the wave and the matrix pipe behind it; 1-2 reads per 32-cycle
// MFMA slot are free).
template <int MT, int NT, int KW, int GU, int I>
__device__ __forceinline__ void group_steps(const GRegs<MT, NT, KW * GU>& cur,
                                            GRegs<MT, NT, KW * GU>& nxt, f32x4 (&acc)[MT][NT],
                                            const GAddr<NT, KW, GU>& ad) {
  constexpr int TPG = KW * GU;
  constexpr int M = TPG * MT * NT, R = TPG * (MT + NT);
  constexpr int j = I / (MT * NT), mb = (I / NT) % MT, nb = I % NT;
  acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.a[j][mb], cur.b[j][nb], acc[mb][nb],
                                                     0, 0, 0);
  constexpr int r0 = (I * R * 4) / (3 * M) < R ? (I * R * 4) / (3 * M) : R;
  constexpr int r1 = ((I + 1) * R * 4) / (3 * M) < R ? ((I + 1) * R * 4) / (3 * M) : R;
  group_reads<MT, NT, KW, GU, r0, r1>(nxt, ad);
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (I + 1 < M) group_steps<MT, NT, KW, GU, I + 1>(cur, nxt, acc, ad);
}
template <int MT, int NT, int TPG, int I>
__device__ __forceinline__ void group_mfma(const GRegs<MT, NT, TPG>& cur, f32x4 (&acc)[MT][NT]) {
  constexpr int j = I / (MT * NT), mb = (I / NT) % MT, nb = I % NT;
  acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.a[j][mb], cur.b[j][nb], acc[mb][nb],
                                                     0, 0, 0);
  if constexpr (I + 1 < TPG * MT * NT) group_mfma<MT, NT, TPG, I + 1>(cur, acc);
}

// ---- bf16 operand form (IgemmP::bf16) ---------------------------------
// Operands are rounded to bf16 (v_cvt_pk_bf16_f32, round to nearest even) on their way
// into the matrix core, sums stay f32.  Lane (l15, qd) of a group holds, for tap j of
// the group, the element of channel 4*cg + qd: with k = 4*qd + j (16x16x16, <= 4 taps)
// or k = 8*qd + j (16x16x32, <= 8 taps) ONE bf16 MFMA per (weight block, position
// block) covers what the f32 form spends TPG MFMAs on; unused k slots carry zeros.
// (a plain conversion, not inline asm: the compiler must see the VALU write to place the
// wait states an MFMA needs before it reads the register -- an asm cvt fed stale data)
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
  union { bf16x2 v; unsigned u; } r;
  r.v = (bf16x2){(__bf16)lo, (__bf16)hi};                    // one v_cvt_pk_bf16_f32
  return r.u;
}
template <int W>
__device__ __forceinline__ f32x4 bf_mfma(const unsigned (&a)[W], const unsigned (&b)[W], f32x4 c) {
  if constexpr (W == 2) {
    union { unsigned u[2]; s16x4 v; } A, B;
    A.u[0] = a[0]; A.u[1] = a[1]; B.u[0] = b[0]; B.u[1] = b[1];
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(A.v, B.v, c, 0, 0, 0);
  } else {
    union { unsigned u[4]; bf16x8 v; } A, B;
#pragma unroll
    for (int i = 0; i < 4; ++i) { A.u[i] = a[i]; B.u[i] = b[i]; }
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(A.v, B.v, c, 0, 0, 0);
  }
}
// all MFMAs of one group, bf16 form.  Every register index is a template constant, as
// in the f32 form: an operand array indexed by a loop variable is kept in scratch memory
// by the compiler, i.e. copied while the asm loads that fill it are still in flight.
template <int TPG, int H, int IDX, int N>
__device__ __forceinline__ unsigned bf_pair(const float (&arr)[TPG][N]) {
  if constexpr (2 * H >= TPG) return 0u;
  else if constexpr (2 * H + 1 < TPG) return cvt_pk_bf16(arr[2 * H][IDX], arr[2 * H + 1][IDX]);
  else return cvt_pk_bf16(arr[2 * H][IDX], 0.f);
}
template <int TPG, int IDX, int N, int W>
__device__ __forceinline__ void bf_frag(const float (&arr)[TPG][N], unsigned (&f)[W]) {
  f[0] = bf_pair<TPG, 0, IDX, N>(arr);
  f[1] = bf_pair<TPG, 1, IDX, N>(arr);
  if constexpr (W == 4) {
    f[2] = bf_pair<TPG, 2, IDX, N>(arr);
    f[3] = bf_pair<TPG, 3, IDX, N>(arr);
  }
}
template <int MT, int NT, int TPG, int W, int NB>
__device__ __forceinline__ void bf_pack_b(const GRegs<MT, NT, TPG>& cur, unsigned (&bu)[NT][W]) {
  bf_frag<TPG, NB, NT, W>(cur.b, bu[NB]);
  if constexpr (NB + 1 < NT) bf_pack_b<MT, NT, TPG, W, NB + 1>(cur, bu);
}
template <int MT, int NT, int W, int MB, int NB>
__device__ __forceinline__ void bf_row(const unsigned (&au)[W], const unsigned (&bu)[NT][W],
                                       f32x4 (&acc)[MT][NT]) {
  acc[MB][NB] = bf_mfma<W>(au, bu[NB], acc[MB][NB]);
  if constexpr (NB + 1 < NT) bf_row<MT, NT, W, MB, NB + 1>(au, bu, acc);
}
template <int MT, int NT, int TPG, int W, int MB>
__device__ __forceinline__ void bf_rows(const GRegs<MT, NT, TPG>& cur, const unsigned (&bu)[NT][W],
                                        f32x4 (&acc)[MT][NT]) {
  unsigned au[W];
  bf_frag<TPG, MB, MT, W>(cur.a, au);
  bf_row<MT, NT, W, MB, 0>(au, bu, acc);
  if constexpr (MB + 1 < MT) bf_rows<MT, NT, TPG, W, MB + 1>(cur, bu, acc);
}
template <int MT, int NT, int TPG>
__device__ __forceinline__ void group_mfma_bf(const GRegs<MT, NT, TPG>& cur, f32x4 (&acc)[MT][NT]) {
  static_assert(TPG <= 8, "a group holds at most 8 taps");
  constexpr int W = TPG <= 4 ? 2 : 4;        // dwords per fragment
  unsigned bu[NT][W];
  bf_pack_b<MT, NT, TPG, W, 0>(cur, bu);
  bf_rows<MT, NT, TPG, W, 0>(cur, bu, acc);
}
// One group step, bf16 form: the MFMA phase is a few hundred cycles, far less than an
// L2 round trip, so ALL reads of the next group are issued first and the arithmetic
// runs underneath them.
template <int MT, int NT, int KW, int GU>
__device__ __forceinline__ void group_steps_bf(const GRegs<MT, NT, KW * GU>& cur,
                                               GRegs<MT, NT, KW * GU>& nxt, f32x4 (&acc)[MT][NT],
                                               const GAddr<NT, KW, GU>& ad) {
  constexpr int TPG = KW * GU;
  group_reads<MT, NT, KW, GU, 0, TPG * (MT + NT)>(nxt, ad);
  __builtin_amdgcn_sched_barrier(0);
  group_mfma_bf<MT, NT, TPG>(cur, acc);
  __builtin_amdgcn_sched_barrier(0);
}

// ---------------------------------------------------------------------------
// The kernel.  Eight waves: 0-3 compute (side by side along the positions),
// 4-7 are PRODUCERS that only issue the LDS-DMA of the next chunk's input spans
// and wait for it.  (Measured on gfx950, tools/ubench/group_loop.hip: DMA bytes
// landing in LDS stall the ds_reads of the compute waves at ~32 B/clk whoever
// issues them, so the staged bytes per MFMA are what costs; the weights -- 5x the
// bytes of the input spans -- therefore bypass LDS: each lane fetches its A
// operands straight from the packed image in L2/L1 with global_load_dword.)
//
// Group = GU channel groups x KW taps of one tap row (GU > 1 only for 1x1 taps).
// Group g+1's operands are fetched while group g computes; the first group of a
// chunk is fetched in the open (its weights before the chunk's barrier), which
// costs ~1k cycles per chunk -- chunks are made as large as LDS allows.
template <int MT, int NT, int KW, int GU, bool BF>
__global__ __launch_bounds__(512, 1) void igemm_kernel(IgemmP p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int BM = 16 * MT, BN = 64 * NT;
  constexpr int TPG = KW * GU;
  const int CC = p.CC;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave = wave8 & 3;
  const bool producer = wave8 >= 4;
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
  const int Lpad = p.Lpad;

  int c_lo, c_hi;
  e2_chunk_range(p, z, c_lo, c_hi);
  const int per = (c_hi - c_lo + p.splitK - 1) / p.splitK;
  const int cb = c_lo + ks * per, ce = min(cb + per, c_hi);
  if (cb >= ce && !p.parts) return;   // (split-K over a clipped range: nothing left for this
                               //  split; every wave decides alike.  Atomic accumulation: nothing
                               //  to add; partial-sum stores: the epilogue stores zeros)
  const int nCG = (p.Cin + 3) >> 2;          // channel groups that carry data
  const int CG = CC >> 2;

  // channel groups of chunk ch, rounded up to whole groups of GU
  auto chunk_cgs = [&](int ch, int& dz, int& cgi0) {
    dz = ch / p.nChunkC;
    cgi0 = (ch - dz * p.nChunkC) * CG;
    const int cgs = min(CG, nCG - cgi0);
    return ((cgs + GU - 1) / GU) * GU;
  };

  if (producer) {
    // ---- producers: input spans of chunk ch -> LDS buffer (ch - cb) & 1 -----
    const int pw = wave8 - 4;
    const int nJ = (L + 63) >> 6;
    const int nJ16 = (L + 255) >> 8;
    const float* in_n = p.in + (long)n * p.isN + (long)z * p.isZ + span_lo;
    auto stage = [&](int ch, int buf) {
      int dz, cgi0;
      const int ccs = 4 * chunk_cgs(ch, dz, cgi0);
      float* xl = smem + buf * p.bufFloats;
      const float* xb = in_n + (long)dz * p.isZ;
      for (int cc = pw; cc < ccs; cc += 4) {
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
          for (int j = 0; j < nJ; ++j)
            if (64 * j + lane < L) glds4(src + 64 * j + lane, dst + 64 * j);
        }
      }
    };
    if (cb < ce) stage(cb, 0);
    for (int ch = cb; ch < ce; ++ch) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();                       // chunk ch is in LDS; buffer of ch-1 is free
      if (ch + 1 < ce && !(p.dbg & 1)) stage(ch + 1, ((ch - cb) & 1) ^ 1);
    }
    if (p.wide) __syncthreads();             // the compute waves reuse LDS for the epilogue
    return;
  }

  // ---- compute waves ----------------------------------------------------------
  unsigned long long* st = (p.stamps && tid == 0) ? p.stamps + 8L * blockIdx.x : nullptr;
  if (st) { st[0] = __builtin_amdgcn_s_memrealtime(); st[1] = __builtin_amdgcn_s_memtime(); }
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

  GAddr<NT, KW, GU> ad;
#pragma unroll
  for (int j = 0; j < TPG; ++j) ad.voff[j] = 4u * (unsigned)((4 * j + qd) * p.coP + l15);
  const long rowF = 4L * p.coP;                       // floats per tap (4 channel rows)
  const long grpF = (long)TPG * rowF;                 // floats per group
  auto chunk_abase = [&](int dz, int cgi0) {
    return p.wp + ((long)(dz * (p.ciP >> 2) + cgi0) * p.THW) * rowF + m0;
  };
  const unsigned stepY = 4u * (unsigned)isY;
  const unsigned stepCg = 16u * (unsigned)Lpad;       // bytes between channel groups in LDS

  GRegs<MT, NT, TPG> g0, g1, g2;            // g2: third set of the bf16 form's rotation
#define E2_WAIT()                                                         \
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");             \
  __builtin_amdgcn_sched_barrier(0);
#define E2_ADV()                                                          \
  {                                                                       \
    ad.abase += grpF;                                                     \
    ++ty;                                                                 \
    const bool wrap = (ty == p.kh);                                       \
    const unsigned d = wrap ? (GU * stepCg - (unsigned)(p.kh - 1) * stepY) : stepY; \
    ty = wrap ? 0 : ty;                                                   \
    _Pragma("unroll") for (int u = 0; u < GU; ++u)                        \
    _Pragma("unroll") for (int nb = 0; nb < NT; ++nb) ad.b[u][nb] += d;   \
  }
  for (int ch = cb; ch < ce; ++ch) {
    const int cur = (ch - cb) & 1;
    int dz, cgi0;
    const int nG = (chunk_cgs(ch, dz, cgi0) / GU) * p.kh;
    // weights of the first group: independent of the LDS contents, start them early
    ad.abase = chunk_abase(dz, cgi0);
    group_reads<MT, NT, KW, GU, 0, TPG * MT>(g0, ad);
    __syncthreads();                         // the producers saw their DMA land
    if (st && ch == cb) st[2] = __builtin_amdgcn_s_memtime();
    if (p.dbg & 2) continue;
    const unsigned xbase = lds_addr(smem + cur * p.bufFloats);
#pragma unroll
    for (int u = 0; u < GU; ++u)
#pragma unroll
      for (int nb = 0; nb < NT; ++nb)
        ad.b[u][nb] = xbase + 4u * (unsigned)posoff[nb] + (unsigned)u * stepCg;
    group_reads<MT, NT, KW, GU, TPG * MT, TPG * (MT + NT)>(g0, ad);
    E2_WAIT()
    g0.touch();
    int ty = 0;
    int g = 0;
    if constexpr (BF) {
      // bf16 form: a group's arithmetic (a few hundred cycles) is far shorter than a
      // round trip to L2, so TWO groups are kept in flight: while group g computes,
      // g+1 is landing and g+2 is being requested (three register sets in rotation).
      // The counted wait lets the newest group's weight loads stay outstanding; reads
      // past the last group of a chunk hit slack rows / LDS beyond the spans, unused.
      constexpr int NA = TPG * MT < 63 ? TPG * MT : 63;
#define E2_WAIT1()                                                        \
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NA) : "memory"); \
      __builtin_amdgcn_sched_barrier(0);
#define E2_BF_STEP(CUR, NXT, TGT)                                         \
      E2_ADV()                                                            \
      __builtin_amdgcn_sched_barrier(0);                                  \
      group_reads<MT, NT, KW, GU, 0, TPG*(MT + NT)>(TGT, ad);             \
      __builtin_amdgcn_sched_barrier(0);                                  \
      group_mfma_bf<MT, NT, TPG>(CUR, acc);                               \
      __builtin_amdgcn_sched_barrier(0);                                  \
      E2_WAIT1()                                                          \
      NXT.touch();
      // g0 is ready; request g1
      E2_ADV()
      __builtin_amdgcn_sched_barrier(0);
      group_reads<MT, NT, KW, GU, 0, TPG*(MT + NT)>(g1, ad);
      __builtin_amdgcn_sched_barrier(0);
      for (; g + 3 <= nG; g += 3) {
        E2_BF_STEP(g0, g1, g2)
        E2_BF_STEP(g1, g2, g0)
        E2_BF_STEP(g2, g0, g1)
      }
      // 0, 1 or 2 groups are left: g0 is ready, g1 in flight, nothing more to request
      if (g < nG) group_mfma_bf<MT, NT, TPG>(g0, acc);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      g0.touch(); g1.touch(); g2.touch();
      if (g + 1 < nG) group_mfma_bf<MT, NT, TPG>(g1, acc);
#undef E2_BF_STEP
#undef E2_WAIT1
    } else {
    for (; g + 1 < nG; g += 2) {
      E2_ADV()
      __builtin_amdgcn_sched_barrier(0);
      group_steps<MT, NT, KW, GU, 0>(g0, g1, acc, ad);   // compute g, fetch g+1
      E2_WAIT()
      g1.touch();
      E2_ADV()
      __builtin_amdgcn_sched_barrier(0);
      group_steps<MT, NT, KW, GU, 0>(g1, g0, acc, ad);   // past the end: slack rows
      E2_WAIT()
      g0.touch();
    }
    if (g < nG) group_mfma<MT, NT, TPG, 0>(g0, acc);
    }
  }
#undef E2_WAIT
#undef E2_ADV
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  g0.touch();
  g1.touch();
  if (st) st[3] = __builtin_amdgcn_s_memtime();

  // ---- epilogue, wide form: the wave's 16*MT x 16*NT tile goes through LDS so that
  // a lane stores 4 consecutive positions of one channel (16 B): 4x fewer store
  // instructions than the accumulator layout allows (the scalar form took ~10k
  // cycles per work-group, store-issue bound).
  if (p.wide) {
    constexpr int STR = 16 * NT + 4;           // row stride: 4*STR == 16 (mod 32) banks
    constexpr int LPR = 4 * NT;                // lanes per tile row
    constexpr int RPI = 64 / LPR;              // rows per store instruction
    __syncthreads();                           // every wave is done with the input spans
    float* tile = smem + wave * (16 * MT * STR);
#pragma unroll
    for (int mb = 0; mb < MT; ++mb)
#pragma unroll
      for (int nb = 0; nb < NT; ++nb)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
          tile[(mb * 16 + 4 * qd + rr) * STR + nb * 16 + l15] = acc[mb][nb][rr];
    // same wave reads back what it wrote: LDS operations of a wave complete in order
    const int rl_ = lane / LPR, c4 = lane - rl_ * LPR;
    const int qw = q0 + wave * (16 * NT) + 4 * c4;
    // rows of the output may be wider than Wo (interior of a zero-padded gradient
    // buffer): a piece of 4 positions can then straddle a row end (Wo >= 4: at most one)
    const bool dense = (p.osY == p.Wo);
    int ro = 0, cq = qw;
    if (!dense) { ro = qw / p.Wo; cq = qw - ro * p.Wo; }
    const long ooff = dense ? (long)qw : (long)ro * p.osY + cq;
    const bool whole = (qw + 3 < p.Q) && (dense || cq + 3 < p.Wo);
    float* ob = p.out + (long)ks * p.partStride + (long)n * p.osN + (long)z * p.osZ;
    const float* gb = (p.gm && p.gm_src) ? p.gm_src + (long)n * p.gsN + (long)z * p.gsZ + qw : nullptr;
    float* bsum = smem + 4 * (16 * MT * STR);      // [4 waves][16*MT rows], behind the tiles
#pragma unroll
    for (int it = 0; it < (16 * MT) / RPI; ++it) {
      const int row = it * RPI + rl_;
      f32x4 v = *reinterpret_cast<const f32x4*>(tile + row * STR + 4 * c4);
      const int co = m0 + row;
      if (co < p.Cout) {                       // (the LPR lanes of a row decide alike)
        if (p.bias) {
          const float bv = p.bias[co];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float t = v[e] + bv;
            if (p.act == E2_ACT_RELU) t = (t > 0.f) ? t : ((t == 0.f) ? 0.f : -0.f);
            v[e] = t;
          }
        }
        if (p.gm) {
          if (gb) {
            const float* gp = gb + (long)co * p.gsC;
            const bool pre = p.gm_bias != nullptr;
            const float gbv = pre ? p.gm_bias[co] : 0.f;
            if (qw + 3 < p.Q) {
              const f32x4 o = *reinterpret_cast<const f32x4*>(gp);
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] *= e2_relu_slope(pre ? o[e] + gbv : o[e], pre);   // (-0.0 + 0.0 would lose the sign)
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e)
                if (qw + e < p.Q) v[e] *= e2_relu_slope(pre ? gp[e] + gbv : gp[e], pre);
            }
          }
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (qw + e >= p.Q) v[e] = 0.f;
          float sb = (v[0] + v[1]) + (v[2] + v[3]);
#pragma unroll
          for (int off = 1; off < LPR; off <<= 1) sb += __shfl_xor(sb, off);
          if (c4 == 0) bsum[wave * (16 * MT) + row] = sb;
        }
        float* dst = ob + (long)co * p.osC;
        if (whole) {
          *reinterpret_cast<f32x4*>(dst + ooff) = v;     // 16 B, possibly unaligned
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (qw + e < p.Q) {
              if (dense) dst[qw + e] = v[e];
              else {
                int cc = cq + e, r2 = ro;
                if (cc >= p.Wo) { cc -= p.Wo; ++r2; }
                dst[(long)r2 * p.osY + cc] = v[e];
              }
            }
        }
      }
    }
    if (p.gm && p.gm_dbias) {
      __syncthreads();                         // (the producer waves have left)
      if (tid < 16 * MT && m0 + tid < p.Cout) {
        const float sb = (bsum[tid] + bsum[16 * MT + tid]) + (bsum[2 * 16 * MT + tid] + bsum[3 * 16 * MT + tid]);
        if (sb != 0.f) unsafeAtomicAdd(p.gm_dbias + m0 + tid, sb);
      }
    }
    if (st) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      st[4] = __builtin_amdgcn_s_memtime();
      st[5] = __builtin_amdgcn_s_memrealtime();
    }
    return;
  }
  // ---- epilogue: D col = position (lane&15), row = channel 4*qd+reg -------
  const int R = p.upz * p.upy * p.upx;
  float bs[MT][4];                             // gm: column sums of this lane's elements
#pragma unroll
  for (int mb = 0; mb < MT; ++mb)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) bs[mb][rr] = 0.f;
#pragma unroll
  for (int nb = 0; nb < NT; ++nb) {
    const int q = q0 + wave * (16 * NT) + nb * 16 + l15;
    if (q >= p.Q) continue;
    const int r = q / p.Wo, c = q - r * p.Wo;
    const float* gq = (p.gm && p.gm_src) ? p.gm_src + (long)n * p.gsN + (long)z * p.gsZ + q : nullptr;
#pragma unroll
    for (int mb = 0; mb < MT; ++mb) {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int co = m0 + mb * 16 + 4 * qd + rr;
        if (co >= p.Cout) continue;
        float* dst;
        if (R == 1) {
          dst = p.out + (long)ks * p.partStride + (long)n * p.osN + (long)co * p.osC +
                (long)z * p.osZ + (long)r * p.osY + c;
        } else {
          const int cr = co / R, sub = co - cr * R;
          const int rz = sub / (p.upy * p.upx);
          const int rem = sub - rz * (p.upy * p.upx);
          const int ry = rem / p.upx, rx = rem - ry * p.upx;
          dst = p.out + (long)n * p.osN + (long)cr * p.osC +
                (long)(z * p.upz + rz) * p.osZ + (long)(r * p.upy + ry) * p.osY +
                (c * p.upx + rx);
        }
        float v = acc[mb][nb][rr];
        if (p.gm) {                            // (linear in acc: split-K partial sums too)
          if (gq) {
            const bool pre = p.gm_bias != nullptr;
            const float o = gq[(long)co * p.gsC];
            v *= e2_relu_slope(pre ? o + p.gm_bias[co] : o, pre);
          }
          bs[mb][rr] += v;
        }
        if (p.atomic) unsafeAtomicAdd(dst, v);
        else *dst = v;
      }
    }
  }
  if (p.gm && p.gm_dbias) {
#pragma unroll
    for (int mb = 0; mb < MT; ++mb)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        float sb = bs[mb][rr];
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) sb += __shfl_xor(sb, off);
        const int co = m0 + mb * 16 + 4 * qd + rr;
        if (l15 == 0 && co < p.Cout && sb != 0.f) unsafeAtomicAdd(p.gm_dbias + co, sb);
      }
  }
  if (st) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    st[4] = __builtin_amdgcn_s_memtime();
    st[5] = __builtin_amdgcn_s_memrealtime();
  }
}

// ---- launch helpers ----------------------------------------------------------
// The bf16 form rotates three operand sets; tilings whose registers would not fit (the
// compiler would spill operands whose loads are still in flight) have no instance.
template <int MT, int NT, int KW, int GU>
constexpr bool igemm_bf_fits() {
  return 3 * KW * GU * (MT + NT) + 4 * MT * NT + 40 + (NT == 4 ? 20 : 0) <= 245;
}

template <int MT, int NT, int KW, int GU, bool BF>
static int igemm_launch(e2_ctx* ctx, const IgemmP& p, int grid, size_t lds) {
  if constexpr (BF && !igemm_bf_fits<MT, NT, KW, GU>()) {
    // this tiling runs in its f32 form (at least as exact; the tuner never prefers it)
    return igemm_launch<MT, NT, KW, GU, false>(ctx, p, grid, lds);
  } else {
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(
        reinterpret_cast<const void*>(&igemm_kernel<MT, NT, KW, GU, BF>),
        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) { e2_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return 1; }
    attr_done = true;
  }
  hipLaunchKernelGGL((igemm_kernel<MT, NT, KW, GU, BF>), dim3(grid), dim3(512), lds, ctx->stream, p);
  E2_CHECK_HIP(hipGetLastError());
  return 0;
  }
}

// all (MT, NT) instances of one kernel width
template <int KW, int GU, bool BF>
static int igemm_dispatch(e2_ctx* ctx, const IgemmP& p, int MT, int NT, int grid, size_t lds) {
#define E2_CASE(M)                                                              \
  case M:                                                                       \
    if (NT == 1) return igemm_launch<M, 1, KW, GU, BF>(ctx, p, grid, lds);          \
    if (NT == 2) return igemm_launch<M, 2, KW, GU, BF>(ctx, p, grid, lds);          \
    break;
#define E2_CASE4(M)                                                             \
  case M:                                                                       \
    if (NT == 1) return igemm_launch<M, 1, KW, GU, BF>(ctx, p, grid, lds);          \
    if (NT == 2) return igemm_launch<M, 2, KW, GU, BF>(ctx, p, grid, lds);          \
    if (NT == 4) return igemm_launch<M, 4, KW, GU, BF>(ctx, p, grid, lds);          \
    break;
  // (13 x 2 blocks: two operand sets + 104 accumulators do not fit 256 registers; the
  // compiler spills operand registers that inline-asm loads are still filling -- the build
  // refuses scratch in these kernels, csrc/check_scratch.py -- so 13 blocks come with NT = 1)
  switch (MT) {
    E2_CASE4(1) E2_CASE4(2) E2_CASE4(3) E2_CASE4(4) E2_CASE4(5) E2_CASE(6)
    E2_CASE(7) E2_CASE(8) E2_CASE(10)
    case 13:
      if (NT == 1) return igemm_launch<13, 1, KW, GU, BF>(ctx, p, grid, lds);
      break;
  }
#undef E2_CASE
#undef E2_CASE4
  e2_set_error("igemm: no instance MT=%d NT=%d", MT, NT);
  return 2;
}

// entry points of the per-width translation units
int e2i_igemm_launch_k1(e2_ctx*, const IgemmP&, int MT, int NT, int GU, int grid, size_t lds);
int e2i_igemm_launch_k3(e2_ctx*, const IgemmP&, int MT, int NT, int GU, int grid, size_t lds);
int e2i_igemm_launch_k4(e2_ctx*, const IgemmP&, int MT, int NT, int GU, int grid, size_t lds);
int e2i_igemm_launch_k5(e2_ctx*, const IgemmP&, int MT, int NT, int GU, int grid, size_t lds);
int e2i_igemm_launch_k1_bf(e2_ctx*, const IgemmP&, int MT, int NT, int GU, int grid, size_t lds);
int e2i_igemm_launch_k3_bf(e2_ctx*, const IgemmP&, int MT, int NT, int GU, int grid, size_t lds);
int e2i_igemm_launch_k4_bf(e2_ctx*, const IgemmP&, int MT, int NT, int GU, int grid, size_t lds);
int e2i_igemm_launch_k5_bf(e2_ctx*, const IgemmP&, int MT, int NT, int GU, int grid, size_t lds);
int e2i_igemm_launch_generic(e2_ctx*, const IgemmP&, int MT, int NT, int grid, size_t lds);
