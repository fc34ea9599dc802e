// conv_wgrad.hip -- weight gradient of the 3-D valid convolution as an implicit
// GEMM on the gfx950 fp32 matrix cores (replaces Theano's ConvGradW that
// T.grad creates at neuromancer/model.py:182).
//
//   G[co][(ci,dz,ty,tx)] = sum_{n,z,y,x} dy[n][co][z][y][x] *
//                                        x[n][ci][z+dz][y+ty][x+tx]
//   dw[co][ci][k] = G[co][ci][flip ? T-1-k : k]
//
// GEMM view: M = Cout (A operand = dy tile, [co][pos] in LDS), N = Cin*T with
// the tap index FLATTENED into N (so Cin = 1 or odd Cin waste nothing), K =
// output positions.  A work-group owns an (M-tile, N-tile) and a range of
// position tiles; it keeps the partial G tile in MFMA accumulators across all
// its position tiles and flushes once with fp32 atomics into the zeroed dw
// (the host keeps the number of position splits small: the chip-wide fp32
// atomic rate is ~1.3 TB/s).
//
// Position tile = BP consecutive positions of one z-plane.  As in the forward
// kernel the input the tile touches is one contiguous span per (ci, dz); the
// spans of every (ci, dz) the N-tile needs and the dy rows are brought in by
// LDS-DMA into one of two buffers while the other one is being consumed.
// B operand of lane (j = n-index, qd = pos & 3): x_l[lanebase(j) + inoff[pos]],
// lanebase = slot(ci,dz)*Lpad + ty*sY + tx (loop invariant), inoff[pos] = the
// position's offset inside the span (small LDS table, read 4 k-steps at a
// time with one ds_read_b128).
//
// Inner loop: "quads" of 4 k-steps.  All LDS reads of quad q+1 are issued (asm,
// invisible to hipcc's waitcnt pass) before the MFMAs of quad q; one
// s_waitcnt lgkmcnt(0) after the MFMAs.  Two register sets ping-pong.
#include "common.hpp"
#include <stdlib.h>
#include <algorithm>
#include <utility>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_vp;
typedef const __attribute__((address_space(1))) void* gbl_vp;

__device__ __forceinline__ void w_glds4(const float* g, float* l) {
  __builtin_amdgcn_global_load_lds((gbl_vp)g, (lds_vp)l, 4, 0, 0);
}
__device__ __forceinline__ void w_glds16(const float* g, float* l) {
  __builtin_amdgcn_global_load_lds((gbl_vp)g, (lds_vp)l, 16, 0, 0);
}
template <int OFF>
__device__ __forceinline__ float w_lds_ld(unsigned addr) {
  float v;
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF));
  return v;
}
__device__ __forceinline__ i32x4 w_lds_ld128(unsigned addr) {
  i32x4 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
  return v;
}
__device__ __forceinline__ unsigned w_lds_addr(const void* p) {
  return (unsigned)(uintptr_t)(lds_vp)p;
}

struct FastDivW { unsigned d, m, sh; };
static inline FastDivW mk_divw(unsigned d) {
  FastDivW f; f.d = d;
  if (d <= 1) { f.m = 0; f.sh = 0; return f; }
  unsigned l = 0;
  while ((1ull << l) < d) ++l;
  f.m = (unsigned)(((1ull << (31 + l)) + d - 1) / d);
  f.sh = l - 1;
  return f;
}
__device__ __forceinline__ unsigned fdivw(unsigned n, const FastDivW& f) {
  return f.d <= 1 ? n : (__umulhi(n, f.m) >> f.sh);
}

struct WgradP {
  const float* x;
  const float* dy;
  const float* zeros;     // >= 64 zero floats in global memory
  float* dw;
  int Cin, Cout, kd, kh, kw, T, THW;
  int Do, Ho, Wo, Q;
  long xsN, xsC, xsZ, xsY;
  long dsN, dsC, dsZ, dsY;
  int flip, upR;
  int NTOT;
  int Lpad;
  int nMT, nNT, nPS;
  int nPT, tilesTotal;
  int maxSpans;
  int bufFloats;
  int Din, N;
  int dbg;
  int bf16;               // operands rounded to bf16 into the matrix core (padded-gradient entry only)
  FastDivW divWo;
};

__device__ __forceinline__ f32x4 w_lds_ld128f(unsigned addr, int off_dummy = 0) {
  f32x4 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
  return v;
}
template <int OFF>
__device__ __forceinline__ f32x4 w_lds_ld128o(unsigned addr) {
  f32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF));
  return v;
}

// one quad = 16 positions = 4 k-steps.  K is permuted so that lane quarter qd
// owns positions 16q + 4qd + j (j = k-step): its A operands of the 4 steps are 4
// CONTIGUOUS floats of the dy row -> one ds_read_b128 per row block (4x fewer
// LDS instructions than b32; the matrix pipe does not care about the K order as
// long as A and B agree).  B: x_l[lanebase + inoff[16q + 4qd + j]], the 4
// offsets again one b128 of the (natural order) table.
template <int MT, int NT, int DLPAD>
struct QuadRegs {
  f32x4 a[MT];          // a[mb][j]
  float b[4][NT];
  i32x4 io;
  // The asm loads are invisible to the register allocator's liveness of IN-FLIGHT
  // data: a destination that is never read again (the prefetch past the end of a
  // tile) could be handed to another value while the LDS data is still on its
  // way.  touch() after the s_waitcnt keeps every destination allocated until
  // the data has landed (cdna guide 5.7: "may reuse it before the data lands").
  __device__ __forceinline__ void touch() {
#pragma unroll
    for (int mb = 0; mb < MT; ++mb) asm volatile("" : "+v"(a[mb]));
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int nb = 0; nb < NT; ++nb) asm volatile("" : "+v"(b[j][nb]));
    asm volatile("" : "+v"(io));
  }
  template <int... I>
  __device__ __forceinline__ void load_a(unsigned addr, std::integer_sequence<int, I...>) {
    ((a[I] = w_lds_ld128o<I * 16 * DLPAD * 4>(addr)), ...);
  }
  __device__ __forceinline__ void load(unsigned addrA, unsigned addrT, unsigned xbase,
                                       const int (&lanebase)[NT], const i32x4& cur_io) {
    load_a(addrA, std::make_integer_sequence<int, MT>{});
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int nb = 0; nb < NT; ++nb)
        b[j][nb] = w_lds_ld<0>(xbase + 4u * (unsigned)(lanebase[nb] + cur_io[j]));
    io = w_lds_ld128(addrT);
  }
};

// bf16 operand form (e2_set_mfma_dtype): the four k-steps a lane holds of a quad are
// exactly the k = 4*qd + j layout of v_mfma_f32_16x16x16_bf16 -- operands rounded to
// bf16 (nearest even) by plain casts (NOT inline asm: the compiler must see the VALU
// write to place the wait states in front of the MFMA), f32 sums.
typedef short w_s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 w_bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ w_s16x4 w_pack_bf16(float a, float b, float c, float d) {
  union { w_bf16x4 h; w_s16x4 v; } r;
  r.h = (w_bf16x4){(__bf16)a, (__bf16)b, (__bf16)c, (__bf16)d};
  return r.v;
}

template <int MT, int NT, int BP, int WK, bool BF>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(WgradP p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int WN = 4 / WK;
  constexpr int BM = 16 * MT;
  constexpr int BNn = 16 * NT * WN;
  constexpr int DLPAD = BP + 4;          // 16-byte aligned rows, 4 banks apart: conflict-free b128
  constexpr int NQ = BP / 16;            // quads per tile
  // buffer layout: [dy: BM*DLPAD][pad to 16B][table: BP ints + 16][x: maxSpans*Lpad]
  constexpr int DYF = ((BM * DLPAD + 3) / 4) * 4;
  constexpr int TBF = BP + 16 * 2 * WK;   // + the quads the pipeline prefetches past the end

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, qd = lane >> 4;
  const int wn = wave % WN, wk = wave / WN;

  int bid = blockIdx.x;
  const int nt = bid % p.nNT; bid /= p.nNT;
  const int mt = bid % p.nMT;
  const int ps = bid / p.nMT;

  const int m0 = mt * BM;
  const int n0 = nt * BNn;
  const int nEnd = min(n0 + BNn, p.NTOT) - 1;
  const int ciA = n0 / p.T;
  const int ciB = nEnd / p.T;
  const int nSpans = (ciB - ciA + 1) * p.kd;
  const int xsY = (int)p.xsY;
  const int Lpad = p.Lpad;

  int lanebase[NT];
#pragma unroll
  for (int nb = 0; nb < NT; ++nb) {
    const int jn = min(n0 + (wn * NT + nb) * 16 + l15, p.NTOT - 1);
    const int ci = jn / p.T;
    const int tap = jn - ci * p.T;
    const int dz = tap / p.THW;
    const int t2 = tap - dz * p.THW;
    const int ty = t2 / p.kw, tx = t2 - ty * p.kw;
    lanebase[nb] = ((ci - ciA) * p.kd + dz) * Lpad + ty * xsY + tx;
  }

  f32x4 acc[MT][NT];
#pragma unroll
  for (int mb = 0; mb < MT; ++mb)
#pragma unroll
    for (int nb = 0; nb < NT; ++nb) acc[mb][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int per = (p.tilesTotal + p.nPS - 1) / p.nPS;
  const int tb = ps * per, te = min(tb + per, p.tilesTotal);

  // ---- stage one position tile into buffer `buf` (async LDS-DMA) ----------
  auto stage = [&](int tt, int buf) {
    float* dyl = smem + buf * p.bufFloats;
    int* tbl = reinterpret_cast<int*>(dyl + DYF);
    float* xl = dyl + DYF + TBF;
    const int pt = tt % p.nPT;
    const int zz = tt / p.nPT;
    const int z = zz % p.Do;
    const int n = zz / p.Do;
    const int q0 = pt * BP;
    const int qlast = min(q0 + BP, p.Q) - 1;
    const int r0 = (int)fdivw(q0, p.divWo), c0 = q0 - r0 * p.Wo;
    const int rl = (int)fdivw(qlast, p.divWo), cl = qlast - rl * p.Wo;
    const long span_lo = (long)r0 * p.xsY + c0;
    const int L = (rl - r0) * xsY + (cl - c0) + (p.kh - 1) * xsY + p.kw;
    // dy rows: lane <-> position (64 positions per DMA instruction)
    const float* dyb = p.dy + (long)n * p.dsN + (long)z * p.dsZ;
#pragma unroll
    for (int j = 0; j < BP / 64; ++j) {
      const int pl = 64 * j + lane;
      const int q = q0 + pl;
      const bool valid = q <= qlast;
      const int qc = valid ? q : qlast;
      const int r = (int)fdivw(qc, p.divWo), c = qc - r * p.Wo;
      // span-offset table, [qd][BP/4] so a lane reads 4 consecutive k-steps at once
      if (wave == 0) tbl[pl] = (r - r0) * xsY + (c - c0);
      const float* src = dyb + (long)r * p.dsY + c;
      for (int co = wave; co < BM; co += 4) {
        float* dst = dyl + co * DLPAD + 64 * j;
        const int cg = min(m0 + co, p.Cout - 1);
        // invalid positions must contribute 0
        w_glds4(valid ? src + (long)cg * p.dsC : p.zeros, dst);
      }
    }
    // input spans, 16 B per lane; the tensor's very last row goes by dwords
    const float* xb = p.x + (long)n * p.xsN + (long)z * p.xsZ + span_lo;
    const int nJ = (L + 63) >> 6;
    for (int slot = wave; slot < nSpans; slot += 4) {
      const int cs = slot / p.kd;
      const int dz = slot - cs * p.kd;
      const int ci = ciA + cs;
      const float* src = xb + (long)ci * p.xsC + (long)dz * p.xsZ;
      float* dst = xl + slot * Lpad;
      for (int j = 0; j < nJ; ++j) w_glds4(src + min(64 * j + lane, L - 1), dst + 64 * j);
    }
  };

  // the quad pipeline reads one quad of offsets past the table: keep them in range
  for (int i = tid; i < 2 * (TBF - BP); i += 256) {
    const int bsel = i / (TBF - BP);
    int* t = reinterpret_cast<int*>(smem + bsel * p.bufFloats + DYF);
    t[BP + (i - bsel * (TBF - BP))] = 0;
  }
  if (tb < te) stage(tb, 0);
  for (int tt = tb; tt < te; ++tt) {
    const int cur = (tt - tb) & 1;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (tt + 1 < te && !(p.dbg & 1)) stage(tt + 1, cur ^ 1);
    if (p.dbg & 2) continue;
    const float* dyl = smem + cur * p.bufFloats;
    const unsigned xbase = w_lds_addr(dyl + DYF + TBF);
    // A: row (mb*16 + l15), columns 16*quad + 4*qd .. +3 ; table: natural order
    unsigned addrA = w_lds_addr(dyl + l15 * DLPAD + 4 * qd) + 64u * (unsigned)wk;
    unsigned addrT = w_lds_addr(dyl + DYF + 4 * qd) + 64u * (unsigned)wk;

    QuadRegs<MT, NT, DLPAD> g0, g1;
#define E2_MFMA(G)                                                       \
    if constexpr (BF) {                                                  \
      w_s16x4 bb_[NT];                                                   \
      _Pragma("unroll") for (int nb = 0; nb < NT; ++nb)                  \
        bb_[nb] = w_pack_bf16(G.b[0][nb], G.b[1][nb], G.b[2][nb], G.b[3][nb]); \
      _Pragma("unroll") for (int mb = 0; mb < MT; ++mb) {                \
        const w_s16x4 aa_ = w_pack_bf16(G.a[mb][0], G.a[mb][1], G.a[mb][2], G.a[mb][3]); \
        _Pragma("unroll") for (int nb = 0; nb < NT; ++nb)                \
          acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(aa_, bb_[nb], acc[mb][nb], 0, 0, 0); \
      }                                                                  \
    } else {                                                             \
    _Pragma("unroll") for (int j = 0; j < 4; ++j)                        \
    _Pragma("unroll") for (int mb = 0; mb < MT; ++mb)                    \
    _Pragma("unroll") for (int nb = 0; nb < NT; ++nb)                    \
      acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(G.a[mb][j], G.b[j][nb], acc[mb][nb], 0, 0, 0); \
    }
#define E2_WAIT()                                                        \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                   \
    __builtin_amdgcn_sched_barrier(0);
    // first quad of this wave: its offsets, then its operands
    i32x4 io0 = w_lds_ld128(addrT);
    E2_WAIT()
    asm volatile("" : "+v"(io0));
    addrT += 64u * WK;
    g0.load(addrA, addrT, xbase, lanebase, io0);   // also fetches the offsets of the next quad
    E2_WAIT()
    g0.touch();
    constexpr int NQW = NQ / WK;                   // quads per wave
    int q = 0;
    for (; q + 1 < NQW; q += 2) {
      addrA += 64u * WK; addrT += 64u * WK;
      g1.load(addrA, addrT, xbase, lanebase, g0.io);
      __builtin_amdgcn_sched_barrier(0);
      E2_MFMA(g0)
      __builtin_amdgcn_sched_barrier(0);
      E2_WAIT()
      g1.touch();
      addrA += 64u * WK; addrT += 64u * WK;
      g0.load(addrA, addrT, xbase, lanebase, g1.io);   // past the end: reads slack
      __builtin_amdgcn_sched_barrier(0);
      E2_MFMA(g1)
      __builtin_amdgcn_sched_barrier(0);
      E2_WAIT()
      g0.touch();
    }
    if (q < NQW) { E2_MFMA(g0) }
#undef E2_MFMA
#undef E2_WAIT
  }

  // ---- flush: row = co (4*qd+reg), col = n-index (lane&15) -----------------
#pragma unroll
  for (int nb = 0; nb < NT; ++nb) {
    const int jn = n0 + (wn * NT + nb) * 16 + l15;
    if (jn >= p.NTOT) continue;
    int col = jn;
    if (p.flip) {
      const int ci = jn / p.T;
      const int tap = jn - ci * p.T;
      col = ci * p.T + (p.T - 1 - tap);
    }
#pragma unroll
    for (int mb = 0; mb < MT; ++mb) {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int co = m0 + mb * 16 + 4 * qd + rr;
        if (co < p.Cout) {
          float* dst;
          if (p.upR > 1) {
            const int cr = co / p.upR;
            dst = p.dw + ((long)cr * p.NTOT + col) * p.upR + (co - cr * p.upR);
          } else {
            dst = p.dw + (long)co * p.NTOT + col;
          }
          unsafeAtomicAdd(dst, acc[mb][nb][rr]);
        }
      }
    }
  }
}

// ---- host side ----------------------------------------------------------------
template <int MT, int NT, int BP, int WK, bool BF>
static int launch_w2(e2_ctx* ctx, const WgradP& p, int grid, size_t lds) {
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(
        reinterpret_cast<const void*>(&wgrad_kernel<MT, NT, BP, WK, BF>),
        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) { e2_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return 1; }
    attr_done = true;
  }
  hipLaunchKernelGGL((wgrad_kernel<MT, NT, BP, WK, BF>), dim3(grid), dim3(256), lds, ctx->stream, p);
  E2_CHECK_HIP(hipGetLastError());
  return 0;
}
// the bf16 form only where the padded-gradient entry point asked for it (the plain
// e2_conv3d_wgrad / UpConv paths stay f32, include/e2hip.h)
template <int MT, int NT, int BP, int WK>
static int launch_w(e2_ctx* ctx, const WgradP& p, int grid, size_t lds) {
  if (p.bf16) return launch_w2<MT, NT, BP, WK, true>(ctx, p, grid, lds);
  return launch_w2<MT, NT, BP, WK, false>(ctx, p, grid, lds);
}

static const int kWMTs[] = {1, 2, 3, 4, 5, 7};

template <int MT>
static int dispatch_w2(e2_ctx* ctx, const WgradP& p, int NT, int BP, int WK, int grid,
                       size_t lds) {
  if (WK == 4) {
    if (BP == 64) return launch_w<MT, 1, 64, 4>(ctx, p, grid, lds);
    return launch_w<MT, 1, 128, 4>(ctx, p, grid, lds);
  }
  if (BP == 64) {
    if (NT == 1) return launch_w<MT, 1, 64, 1>(ctx, p, grid, lds);
    if (NT == 2) return launch_w<MT, 2, 64, 1>(ctx, p, grid, lds);
    if (NT == 4) return launch_w<MT, 4, 64, 1>(ctx, p, grid, lds);
  } else {
    if (NT == 1) return launch_w<MT, 1, 128, 1>(ctx, p, grid, lds);
    if (NT == 2) return launch_w<MT, 2, 128, 1>(ctx, p, grid, lds);
    if (NT == 4) return launch_w<MT, 4, 128, 1>(ctx, p, grid, lds);
  }
  e2_set_error("wgrad: no instance NT=%d BP=%d WK=%d", NT, BP, WK);
  return 2;
}

static int dispatch_w(e2_ctx* ctx, const WgradP& p, int MT, int NT, int BP, int WK, int grid,
                      size_t lds) {
  switch (MT) {
    case 1: return dispatch_w2<1>(ctx, p, NT, BP, WK, grid, lds);
    case 2: return dispatch_w2<2>(ctx, p, NT, BP, WK, grid, lds);
    case 3: return dispatch_w2<3>(ctx, p, NT, BP, WK, grid, lds);
    case 4: return dispatch_w2<4>(ctx, p, NT, BP, WK, grid, lds);
    case 5: return dispatch_w2<5>(ctx, p, NT, BP, WK, grid, lds);
    case 7: return dispatch_w2<7>(ctx, p, NT, BP, WK, grid, lds);
  }
  e2_set_error("wgrad: no instance MT=%d", MT);
  return 2;
}

// x-span row stride: whole 64-float (dword) DMA pieces, == 2 (mod 4) so rows of
// consecutive channels fall on different banks
static int w_lpad(int Lmax) {
  return ((Lmax + 63) / 64) * 64 + 2;    // whole 64-float DMA pieces, == 2 (mod 4)
}
static int w_lmax(const WgradArgs& a, int BP) {
  const int rows = (BP + a.Wo - 2) / a.Wo;      // rows crossed by BP consecutive positions
  return (BP - 1) + rows * ((int)a.xsY - a.Wo) + (a.kh - 1) * (int)a.xsY + a.kw;
}
static int w_maxspans(const WgradArgs& a, int BNn) {
  const int T = a.kd * a.kh * a.kw;
  int cis = (BNn - 1) / T + 2;
  if (cis > a.Cin) cis = a.Cin;
  return cis * a.kd;
}
static size_t w_buf_floats(const WgradArgs& a, int MT, int BNn, int BP) {
  const size_t dyf = (((size_t)16 * MT * (BP + 4) + 3) / 4) * 4;
  // + slack: the quad pipeline prefetches one quad past the end of the tile
  return dyf + (BP + 128) + (size_t)w_maxspans(a, BNn) * w_lpad(w_lmax(a, BP)) + 64;
}

struct WCfg { int MT, NT, WK, BP, PS; };

// Cost model (cycles): MFMA time of a work-group's waves (they run 1-2 per
// SIMD), DMA bytes at ~20 B/clk/CU, and the flush atomics against the chip-wide
// 1.3 TB/s rate; position splits only as needed to fill the CUs.
static WCfg choose_wcfg(const e2_ctx* ctx, const WgradArgs& a, int* ok, int* src) {
  const int mblocks = e2_cdiv(a.Cout, 16);
  const int T = a.kd * a.kh * a.kw;
  const long NTOT = (long)a.Cin * T;
  const int nblocks = (int)((NTOT + 15) / 16);
  const long Q = (long)a.Ho * a.Wo;
  WCfg best{0, 0, 0, 0, 0};
  double bestCost = 1e300;
  const double dw_bytes = 4.0 * a.Cout * (double)NTOT;
  const char* force = ctx->tiling[E2_TILING_WGRAD];
  *src = E2_SRC_MODEL;
  if (force[0]) {
    *src = E2_SRC_FALLBACK;           // ... unless one of the returns below honours the string
    WCfg f{0, 0, 0, 0, 0};
    if (sscanf(force, "%d,%d,%d,%d,%d", &f.MT, &f.NT, &f.WK, &f.BP, &f.PS) == 5) {
      // "MT,NT,7,0,S" names the K-contiguous 1x1x1 GEMM (conv_pw_wgrad.hip), which takes dense
      // channel planes and f32 mode only.  The tuning keys hold the row pitch, not the plane
      // pitch: a view that keeps the rows but not the planes (a crop along the second spatial
      // axis of the parent) matches a shipped entry it cannot run -- such a call takes the
      // cost model's choice instead of failing.
      const bool pw_ok = a.kd == 1 && a.kh == 1 && a.kw == 1 && a.xsY == a.Wo &&
                         a.xsZ == (int64_t)a.Ho * a.Wo && a.dsY == a.Wo &&
                         a.dsZ == (int64_t)a.Ho * a.Wo && !ctx->mfma_bf16;
      // "MT,NT,9,0,S": the same GEMM for kernels with taps; needs the padded gradient at the
      // input's row pitch and a zero border of >= 31 positions behind a plane
      // (and 32-bit byte offsets inside a sample: a larger problem takes the cost model's choice)
      const long spanK = (long)(a.Ho - 1) * a.dsY + a.Wo;
      const bool ks_fits = ((long)a.Cout * a.dsC + (long)a.Do * a.dsZ + spanK) * 4 + 256 < (1L << 32) &&
                           ((long)a.Cin * a.xsC + (long)(a.Do + a.kd - 1) * a.xsZ) * 4 + 256 < (1L << 32) &&
                           a.dsC >= 0 && a.xsC >= 0 && a.xsZ >= (int64_t)(a.Ho + a.kh - 1) * a.xsY &&
                           a.xsC >= (int64_t)(a.Do + a.kd - 1) * a.xsZ;
      const bool ks_ok = a.dy_padded && a.kd * a.kh * a.kw > 1 && a.upR <= 1 && a.dsY == a.xsY &&
                         (a.kh - 1) * a.xsY + (a.kw - 1) >= 31 && !ctx->mfma_bf16 &&
                         ctx->input_slack >= 128 && ks_fits;
      if (f.WK == 9) { if (ks_ok) { *ok = 1; *src = E2_SRC_FORCED; return f; } }
      else if ((f.WK != 7 && f.WK != 8) || pw_ok) { *ok = 1; *src = E2_SRC_FORCED; return f; }
    }
  }
  if (a.dy_padded && nblocks > 2) {
    // direct kernel (conv_wgrad_direct.hip): no dy staging; tiles of 128/256 span positions
    const long S = (long)(a.Ho - 1) * a.dsY + a.Wo;
    for (int MT : kWMTs) {
      if (MT > mblocks && MT != 1) continue;
      const int nMT = e2_cdiv(mblocks, MT);
      for (int NT = 1; NT <= 4; NT *= 2) {
        if (16 * NT * 4 > 16 * nblocks && NT > 1) continue;
        if (NT == 4 && MT * 4 >= 20) continue;      // no such instance (it would spill)
        const int nNT = e2_cdiv(nblocks, NT * 4);
        for (int BP = ctx->mfma_bf16 ? 256 : 128; BP <= 256; BP *= 2) {   // bf16 form: BP 256 only
          if (BP == 256 && S <= 128 && !ctx->mfma_bf16) continue;
          const size_t lds = 2 * e2i_wgrad_direct_buf_floats(a, NT, BP, 1) * 4;
          if (lds > 160 * 1024) continue;
          const int slots = ctx->num_cu;
          const int nPT = (int)((S + BP - 1) / BP);
          const long tiles = (long)a.N * a.Do * nPT;
          const long base = (long)nMT * nNT;
          for (int fill = 1; fill <= 2; ++fill) {
            long PS = std::max<long>(1, ((long)slots * fill + base - 1) / base);
            PS = std::min(PS, tiles);
            const int per = (int)((tiles + PS - 1) / PS);
            const double tile = (BP / 4.0) * MT * NT * 35.0 + (BP / 16.0) * 150.0 + 1200.0;
            const double flush = 16.0 * MT * 16.0 * NT * 4 * 0.6;
            const double wg_time = per * tile + flush + 5000.0;
            const long wgs = base * PS;
            const double rounds = (double)((wgs + slots - 1) / slots);
            const double atom = (double)PS * dw_bytes / 1.3e12 * 2.4e9 + dw_bytes / 4e12 * 2.4e9;
            const double cost = rounds * wg_time + atom;
            if (cost < bestCost) { bestCost = cost; best = WCfg{MT, NT, 1, BP, (int)PS}; }
          }
        }
      }
    }
    if (best.MT) { *ok = 1; return best; }
  }
  for (int MT : kWMTs) {
    if (MT > mblocks && MT != 1) continue;
    const int nMT = e2_cdiv(mblocks, MT);
    for (int v = 0; v < 4; ++v) {
      const int NT = (v == 3) ? 1 : (1 << v);       // 1,2,4 | WK variant
      const int WK = (v == 3) ? 4 : 1;
      const int WN = 4 / WK;
      const int BNn = 16 * NT * WN;
      if (16 * NT * WN > 16 * nblocks && NT > 1) continue;
      if (WK == 4 && nblocks > 2) continue;        // K-split waves only for tiny N
      const int nNT = e2_cdiv(nblocks, NT * WN);
      for (int BP = 64; BP <= 128; BP *= 2) {
        if (BP == 128 && Q <= 64) continue;
        const size_t bufF = w_buf_floats(a, MT, BNn, BP);
        const size_t lds = 2 * bufF * 4;
        if (lds > 160 * 1024) continue;
        const int perCU = lds <= 80 * 1024 ? 2 : 1;
        const int slots = ctx->num_cu * perCU;
        const int nPT = (int)((Q + BP - 1) / BP);
        const long tiles = (long)a.N * a.Do * nPT;
        const long base = (long)nMT * nNT;
        for (int fill = 1; fill <= 2; ++fill) {
          long PS = std::max<long>(1, ((long)slots * fill + base - 1) / base);
          PS = std::min(PS, tiles);
          const int per = (int)((tiles + PS - 1) / PS);
          const double mfma = (BP / 4.0 / WK) * MT * NT * 32.0;              // per tile
          const double bytes = 4.0 * (16.0 * MT * BP + (double)w_maxspans(a, BNn) * w_lmax(a, BP));
          const double issue = (BP / 4.0 / WK) * (MT + 2.0 * NT + 1) * 5.0;
          double tile = std::max(mfma, issue) * perCU * 1.08 + 400.0;
          tile = std::max(tile, bytes / 20.0 * perCU);
          const double flush = 16.0 * MT * 16.0 * NT * 4 * WK * 0.6;
          const double wg_time = per * tile + flush + 5000.0;
          const long wgs = base * PS;
          const double rounds = (double)((wgs + slots - 1) / slots);
          const double atom = (double)PS * dw_bytes * WK / 1.3e12 * 2.4e9 + dw_bytes / 4e12 * 2.4e9;
          const double cost = rounds * wg_time + atom;
          if (cost < bestCost) { bestCost = cost; best = WCfg{MT, NT, WK, BP, (int)PS}; }
        }
      }
    }
  }
  *ok = best.MT != 0;
  return best;
}

int e2i_wgrad_conv(e2_ctx* ctx, const WgradArgs& a) {
  E2_REQUIRE(a.Do > 0 && a.Ho > 0 && a.Wo > 0 && a.Cin > 0 && a.Cout > 0,
             "wgrad: empty problem");
  E2_REQUIRE(a.xsY < (1 << 20), "wgrad: input row stride too large");
  int ok = 0, src = E2_SRC_MODEL;
  WCfg c = choose_wcfg(ctx, a, &ok, &src);
  E2_REQUIRE(ok, "wgrad: no tiling fits LDS (Cin=%d Cout=%d k=%dx%dx%d)", a.Cin, a.Cout,
             a.kd, a.kh, a.kw);
  {
    // e2_last_launch: the kernel family this call is about to run, its tiling, and whether a
    // forced string was honoured ("fallback": a 7 / 8 / 9 form the problem's layout cannot run)
    const bool direct = a.dy_padded && (c.WK == 1 || c.WK == 14 || c.WK == 101 || c.WK == 114) &&
                        (c.BP == 128 || c.BP == 256);
    const char* fam = c.WK == 7 ? "pw_wgrad" : c.WK == 8 ? "pw_wgrad_ks" : c.WK == 9 ? "wgrad_ks"
                      : direct ? (ctx->mfma_bf16 ? "wgrad_direct_bf16r" : "wgrad_direct")
                      : "wgrad_lds";
    e2_note_launch(ctx, fam, src, "%d,%d,%d,%d,%d", c.MT, c.NT, c.WK, c.BP, c.PS);
  }
  // WK field: 1 = direct kernel when dy is padded (BP 128/256), else the LDS-staged
  // kernel; 14 = direct kernel, waves split the quads of a tile (NT 2/4);
  // 0 = LDS-staged kernel forced; 4 = LDS-staged kernel, waves split K
  // WK 101 / 114: the direct kernel with the XCD-grouped block order (nMT * PS % 8 == 0)
  // WK 7 ("MT,NT,7,0,S"): 1x1x1 kernels -- the GEMM with K-contiguous operands (conv_pw_wgrad.hip)
  // WK 8 ("MT,NT,8,0,S"): the same GEMM, one 16 MT x 16 NT tile per work-group whose four waves split the positions
  if (c.WK == 7) return e2i_pw_wgrad(ctx, a, c.MT, c.NT, c.PS);
  if (c.WK == 8) return e2i_pw_wgrad_ks(ctx, a, c.MT, c.NT, c.PS);
  // WK 9 ("MT,NT,9,0,S"): that GEMM for kernels WITH taps -- every (input channel, tap) column of dW is
  // a K-contiguous row of x at the tap's shift (needs the padded gradient at the input's row pitch)
  if (c.WK == 9) return e2i_wgrad_ks(ctx, a, c.MT, c.NT, c.PS);
  if (a.dy_padded && (c.WK == 1 || c.WK == 14 || c.WK == 101 || c.WK == 114) && (c.BP == 128 || c.BP == 256))
    return e2i_wgrad_direct(ctx, a, c.MT, c.NT, c.BP, c.PS, (c.WK % 100) == 14 ? 4 : 1, c.WK >= 100);
  E2_REQUIRE(c.WK != 14 && c.WK < 100, "wgrad: WK=14/101/114 need the padded-gradient entry point and BP 128/256");
  if (c.WK == 0) c.WK = 1;
  E2_REQUIRE(c.BP == 64 || c.BP == 128, "wgrad: BP must be 64 or 128");
  E2_REQUIRE(c.WK == 1 || (c.WK == 4 && c.NT == 1), "wgrad: WK=4 needs NT=1");
  WgradP p;
  p.x = a.x; p.dy = a.dy; p.dw = a.dw;
  p.zeros = ctx->zeros;
  p.Cin = a.Cin; p.Cout = a.Cout; p.kd = a.kd; p.kh = a.kh; p.kw = a.kw;
  p.THW = a.kh * a.kw; p.T = a.kd * p.THW;
  p.Do = a.Do; p.Ho = a.Ho; p.Wo = a.Wo; p.Q = a.Ho * a.Wo;
  p.xsN = a.xsN; p.xsC = a.xsC; p.xsZ = a.xsZ; p.xsY = a.xsY;
  p.dsN = a.dsN; p.dsC = a.dsC; p.dsZ = a.dsZ; p.dsY = a.dsY;
  p.flip = a.flip;
  p.upR = a.upR > 1 ? a.upR : 1;
  const long NTOT = (long)a.Cin * p.T;
  E2_REQUIRE(NTOT < (1L << 30), "wgrad: Cin*T too large");
  p.NTOT = (int)NTOT;
  const int WN = 4 / c.WK;
  const int BNn = 16 * c.NT * WN;
  p.Lpad = w_lpad(w_lmax(a, c.BP));
  p.maxSpans = w_maxspans(a, BNn);
  p.nMT = e2_cdiv(e2_cdiv(a.Cout, 16), c.MT);
  p.nNT = e2_cdiv(e2_cdiv(p.NTOT, 16), c.NT * WN);
  p.nPT = e2_cdiv(p.Q, c.BP);
  p.tilesTotal = a.N * a.Do * p.nPT;
  p.nPS = std::max(1, std::min(c.PS, p.tilesTotal));
  p.bufFloats = (int)w_buf_floats(a, c.MT, BNn, c.BP);
  p.Din = a.Do + a.kd - 1;
  p.N = a.N;
  p.divWo = mk_divw((unsigned)a.Wo);
  p.dbg = e2_dbg_env_int("E2_WGRAD_DBG");
  p.bf16 = (ctx->mfma_bf16 && a.dy_padded) ? 1 : 0;
  size_t lds = 2 * (size_t)p.bufFloats * 4;
  lds += (size_t)e2_dbg_env_int("E2_WGRAD_LDSPAD");
  E2_REQUIRE(lds <= 160 * 1024, "wgrad: tiling needs %zu B of LDS", lds);
  const long grid = (long)p.nMT * p.nNT * p.nPS;
  E2_REQUIRE(grid < (1L << 31), "wgrad: grid too large");
  if (!a.accumulate)
    if (int rc = e2i_fill_flat(ctx, a.dw, (size_t)a.Cout * p.NTOT, 0.f)) return rc;
  if (e2_dbg_env("E2_VERBOSE"))
    fprintf(stderr, "[e2] wgrad%s Cin=%d Cout=%d k=%d,%d,%d out=%d,%d,%d MT=%d NT=%d WK=%d BP=%d PS=%d grid=%ld lds=%zu\n",
            p.bf16 ? "(bf16)" : "", a.Cin, a.Cout, a.kd, a.kh, a.kw, a.Do, a.Ho, a.Wo, c.MT, c.NT, c.WK, c.BP, p.nPS, grid, lds);
  return dispatch_w(ctx, p, c.MT, c.NT, c.BP, c.WK, (int)grid, lds);
}
