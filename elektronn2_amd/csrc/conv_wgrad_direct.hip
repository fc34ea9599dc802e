// conv_wgrad_direct.hip -- weight gradient, "direct" variant: the dy operand comes
// straight from global memory, only the input spans go through LDS.
//
//   G[co][(ci,dz,ty,tx)] = sum_{n,z,s} dy_pad[n][co][z][s] * x[n][ci][z+dz][xoff(s)+ty*sY+tx]
//
// Same GEMM view as conv_wgrad.hip (M = Cout, N = Cin*T with the taps flattened
// into N, K = positions, partial tile kept in MFMA accumulators over a range of
// position tiles, one atomic flush), with three differences that follow from the
// measurement in tools/ubench/group_loop.hip (bytes landing in LDS by DMA stall
// the ds_reads of the compute waves at ~32 B/clk):
//   * K runs over the MEMORY span s of the zero-padded gradient plane
//     (s = r*dsY + c, c in [0, dsY)), not over the output positions: the padding
//     columns between two rows hold zeros (the buffer is the one dgrad reads), so
//     they add nothing, and 4 consecutive k are 4 consecutive floats.  Lane
//     (l15, qd) fetches A[co = 16*mb + l15][s = 16q + 4qd .. +3] with ONE
//     global_load_dwordx4 per row block and quad -- dy never touches LDS.
//   * a small LDS table maps s -> offset of the position inside the staged input
//     span (gap columns are clamped to the row's last position: their dy is 0).
//   * four PRODUCER waves build the table and issue the LDS-DMA of the input spans
//     of the next tile; the four compute waves only load operands and issue MFMAs.
// Positions past the end of a plane's span (the last one or two quads of a plane) are
// masked out of A with v_cndmask after the load; the loads themselves may run up to
// 31 floats past the span, hence the 128 B of slack the entry point asks for.
#include "common.hpp"
#include <stdlib.h>
#include <algorithm>
#include <vector>
#include <utility>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_vp;
typedef const __attribute__((address_space(1))) void* gbl_vp;

namespace {

__device__ __forceinline__ void d_glds4(const float* g, float* l) {
  __builtin_amdgcn_global_load_lds((gbl_vp)g, (lds_vp)l, 4, 0, 0);
}
__device__ __forceinline__ void d_glds16(const float* g, float* l) {
  __builtin_amdgcn_global_load_lds((gbl_vp)g, (lds_vp)l, 16, 0, 0);
}
__device__ __forceinline__ float d_lds_ld(unsigned addr) {
  float v;
  asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(addr));
  return v;
}
__device__ __forceinline__ i32x4 d_lds_ld128(unsigned addr) {
  i32x4 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
  return v;
}
__device__ __forceinline__ f32x4 d_gl_ld128(const float* sbase, unsigned voff) {
  f32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v) : "v"(voff), "s"(sbase));
  return v;
}
__device__ __forceinline__ unsigned d_lds_addr(const void* p) {
  return (unsigned)(uintptr_t)(lds_vp)p;
}

struct FastDivD { unsigned d, m, sh; };
static inline FastDivD mk_divd(unsigned d) {
  FastDivD f; f.d = d;
  if (d <= 1) { f.m = 0; f.sh = 0; return f; }
  unsigned l = 0;
  while ((1ull << l) < d) ++l;
  f.m = (unsigned)(((1ull << (31 + l)) + d - 1) / d);
  f.sh = l - 1;
  return f;
}
__device__ __forceinline__ unsigned fdivd(unsigned n, const FastDivD& f) {
  return f.d <= 1 ? n : (__umulhi(n, f.m) >> f.sh);
}

struct WdP {
  const float* x;
  const float* dy;        // interior origin of the zero-padded gradient buffer
  float* dw;
  int Cin, Cout, kd, kh, kw, T, THW;
  int Do, Ho, Wo, S;      // S = (Ho-1)*dsY + Wo: span of one gradient plane
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
  int xcd;                // 1: work-groups that share gradient rows run on ONE XCD (see the kernel)
  FastDivD divDsY;
  unsigned long long* stamps;   // debug (E2_WGRAD_STAMPS): s_memtime stamps per work-group
};

// one quad = 16 span positions = 4 k-steps; lane quarter qd owns 16q + 4qd + j
template <int MT, int NT>
struct DQuad {
  f32x4 a[MT];          // a[mb][j]
  float b[4][NT];
  i32x4 io;             // span offsets of the NEXT quad
  __device__ __forceinline__ void touch() {
#pragma unroll
    for (int mb = 0; mb < MT; ++mb) asm volatile("" : "+v"(a[mb]));
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int nb = 0; nb < NT; ++nb) asm volatile("" : "+v"(b[j][nb]));
    asm volatile("" : "+v"(io));
  }
  __device__ __forceinline__ void load_a(const float* abase, const unsigned (&voff)[MT]) {
#pragma unroll
    for (int mb = 0; mb < MT; ++mb) a[mb] = d_gl_ld128(abase, voff[mb]);
  }
  __device__ __forceinline__ void load_b(unsigned addrT, unsigned xbase, const int (&lanebase)[NT],
                                         const i32x4& cur_io) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int nb = 0; nb < NT; ++nb)
        b[j][nb] = d_lds_ld(xbase + 4u * (unsigned)(lanebase[nb] + cur_io[j]));
    io = d_lds_ld128(addrT);
  }
  // zero the k-steps whose span position lies past `lim` (relative to the quad's lane)
  __device__ __forceinline__ void mask(int lim) {
#pragma unroll
    for (int mb = 0; mb < MT; ++mb)
#pragma unroll
      for (int j = 0; j < 4; ++j) a[mb][j] = (j <= lim) ? a[mb][j] : 0.f;
  }
};

// addresses the next quad's loads need
template <int MT, int NT>
struct DAddr {
  const float* abase;
  unsigned voff[MT];
  unsigned addrT;
  unsigned xbase;
  int lanebase[NT];
};
// load r of the next quad, in issue order: dy rows (global, longest latency), the
// offsets of the quad after it, then the gathered inputs (LDS)
template <int MT, int NT, int R>
__device__ __forceinline__ void dquad_read(DQuad<MT, NT>& g, const DAddr<MT, NT>& ad,
                                           const i32x4& cur_io) {
  if constexpr (R < MT) {
    g.a[R] = d_gl_ld128(ad.abase, ad.voff[R]);
  } else if constexpr (R == MT) {
    g.io = d_lds_ld128(ad.addrT);
  } else {
    constexpr int rb = R - MT - 1;
    constexpr int j = rb / NT, nb = rb % NT;
    g.b[j][nb] = d_lds_ld(ad.xbase + 4u * (unsigned)(ad.lanebase[nb] + cur_io[j]));
  }
}
template <int MT, int NT, int R0, int R1>
__device__ __forceinline__ void dquad_reads(DQuad<MT, NT>& g, const DAddr<MT, NT>& ad,
                                            const i32x4& cur_io) {
  if constexpr (R0 < R1) {
    dquad_read<MT, NT, R0>(g, ad, cur_io);
    dquad_reads<MT, NT, R0 + 1, R1>(g, ad, cur_io);
  }
}
// MFMA i of the current quad, then its share of the next quad's loads (spread over
// the first 3/4 of the MFMAs; PF = false: no next quad)
template <int MT, int NT, bool PF, int I>
__device__ __forceinline__ void dquad_steps(const DQuad<MT, NT>& cur, DQuad<MT, NT>& nxt,
                                            f32x4 (&acc)[MT][NT], const DAddr<MT, NT>& ad,
                                            const i32x4& nio) {
  constexpr int M = 4 * MT * NT, R = MT + 1 + 4 * NT;
  constexpr int j = I / (MT * NT), mb = (I / NT) % MT, nb = I % NT;
  acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.a[mb][j], cur.b[j][nb], acc[mb][nb], 0, 0, 0);
  if constexpr (PF) {
    constexpr int r0 = (I * R * 4) / (3 * M) < R ? (I * R * 4) / (3 * M) : R;
    constexpr int r1 = ((I + 1) * R * 4) / (3 * M) < R ? ((I + 1) * R * 4) / (3 * M) : R;
    dquad_reads<MT, NT, r0, r1>(nxt, ad, nio);
  }
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (I + 1 < M) dquad_steps<MT, NT, PF, I + 1>(cur, nxt, acc, ad, nio);
}

// ---- bf16 operand form (e2_set_mfma_dtype) --------------------------
// A quad is 16 span positions; lane (l15, qd) holds positions 4*qd + j, j = 0..3, of its
// dy row and of its input column -- exactly the k = 4*qd + j layout of
// v_mfma_f32_16x16x16_bf16.  Operands are rounded to bf16 (round to nearest even) on
// the way into the matrix core, sums stay f32: ONE MFMA per (row block, column block)
// and quad instead of four.  The MFMA phase is then far shorter than a round trip to
// L2, so all loads of the next quad are issued first.
// (plain conversions -> v_cvt_pk_bf16_f32; not inline asm, so that the compiler places the
// wait states between the VALU write and the MFMA that reads it)
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ s16x4 d_pack_bf16(float a, float b, float c, float d) {
  union { bf16x4 h; s16x4 v; } r;
  r.h = (bf16x4){(__bf16)a, (__bf16)b, (__bf16)c, (__bf16)d};   // two v_cvt_pk_bf16_f32
  return r.v;
}
template <int MT, int NT, bool PF>
__device__ __forceinline__ void dquad_steps_bf(const DQuad<MT, NT>& cur, DQuad<MT, NT>& nxt,
                                               f32x4 (&acc)[MT][NT], const DAddr<MT, NT>& ad,
                                               const i32x4& nio) {
  if constexpr (PF) dquad_reads<MT, NT, 0, MT + 1 + 4 * NT>(nxt, ad, nio);
  __builtin_amdgcn_sched_barrier(0);
  s16x4 bb[NT];
#pragma unroll
  for (int nb = 0; nb < NT; ++nb)
    bb[nb] = d_pack_bf16(cur.b[0][nb], cur.b[1][nb], cur.b[2][nb], cur.b[3][nb]);
#pragma unroll
  for (int mb = 0; mb < MT; ++mb) {
    const s16x4 aa = d_pack_bf16(cur.a[mb][0], cur.a[mb][1], cur.a[mb][2], cur.a[mb][3]);
#pragma unroll
    for (int nb = 0; nb < NT; ++nb)
      acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(aa, bb[nb], acc[mb][nb], 0, 0, 0);
  }
  __builtin_amdgcn_sched_barrier(0);
}
template <int MT, int NT, bool PF, bool BF>
__device__ __forceinline__ void dquad_go(const DQuad<MT, NT>& cur, DQuad<MT, NT>& nxt,
                                         f32x4 (&acc)[MT][NT], const DAddr<MT, NT>& ad,
                                         const i32x4& nio) {
  if constexpr (BF) dquad_steps_bf<MT, NT, PF>(cur, nxt, acc, ad, nio);
  else dquad_steps<MT, NT, PF, 0>(cur, nxt, acc, ad, nio);
}

// WK = 4: the four compute waves share ONE 16*NT-wide n-tile and split the quads of
// every position tile among them (wave w takes quads w, w+4, ...): for small or
// awkward Cin*T (L1: 540) this keeps NT large -- enough MFMAs per pipeline step to
// hide the operand loads -- without padding N up to 64*NT; the staged spans serve all
// four waves.  Each wave flushes its own partial sums (4x the atomics of a small dw).
template <int MT, int NT, int BP, int WK, bool BF>
__global__ __launch_bounds__(512, 1) void wgrad_direct_kernel(WdP p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int BM = 16 * MT;
  constexpr int BNn = 16 * NT * (4 / WK);
  constexpr int TBF = BP + 256;          // + the offsets fetched past the end of a tile (zeros)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool producer = wave8 >= 4;
  const int wn = (WK == 4) ? 0 : (wave8 & 3);   // n-block column of the wave
  const int wk = (WK == 4) ? (wave8 & 3) : 0;   // its share of the quads
  const int l15 = lane & 15, qd = lane >> 4;

  // The nNT work-groups of one (row tile, position split) read the SAME dy rows.  Blocks
  // are dealt round-robin over the 8 XCDs (b and b + 8 share one, MI355X_MICROARCH.md
  // "Workgroup dispatch"), each with its own L2: with the plain order every XCD fetched
  // every dy line (FETCH_SIZE 5-11x the operand bytes, profiles/r01_e_pmc_traffic.csv).
  // xcd mode: the (group, n-tile) pairs, group-major, are cut into 8 equal contiguous ranges
  // and range x is run by the blocks of residue class x mod 8 -- the work-groups of one
  // group sit on ONE XCD (two, where a range boundary cuts through the group), so its dy
  // lines are fetched into one or two L2s instead of eight, and every XCD gets the same
  // number of work-groups whatever the group count is.  (Round 2 dealt whole groups to the
  // XCDs: 50 groups = 7 + 6 per XCD made the long work-groups of the 7-group XCDs the
  // kernel's duration, and the tuner kept the plain order for such launches.)
  // Placement only changes speed / traffic, never results.
  int nt, mt, ps;
  if (p.xcd) {
    const int b = blockIdx.x;
    const int pairs = p.nMT * p.nPS * p.nNT;
    const int perX = (pairs + 7) >> 3;
    const int j = b >> 3;
    const int pair = (b & 7) * perX + j;
    if (pair >= pairs) return;              // (j < perX by the grid size)
    const int g = pair / p.nNT;
    nt = pair - g * p.nNT;
    mt = g % p.nMT;
    ps = g / p.nMT;
  } else {
    int bid = blockIdx.x;
    nt = bid % p.nNT; bid /= p.nNT;
    mt = bid % p.nMT;
    ps = bid / p.nMT;
  }

  const int m0 = mt * BM;
  const int n0 = nt * BNn;
  const int nEnd = min(n0 + BNn, p.NTOT) - 1;
  const int ciA = n0 / p.T;
  const int ciB = nEnd / p.T;
  const int nSpans = (ciB - ciA + 1) * p.kd;
  const int xsY = (int)p.xsY, dsY = (int)p.dsY;
  const int Lpad = p.Lpad;

  const int per = (p.tilesTotal + p.nPS - 1) / p.nPS;
  const int tb = ps * per, te = min(tb + per, p.tilesTotal);

  if (producer) {
    const int pw = wave8 - 4;
    auto stage = [&](int tt, int buf) {
      int* tbl = reinterpret_cast<int*>(smem + buf * p.bufFloats);
      float* xl = smem + buf * p.bufFloats + TBF;
      const int pt = tt % p.nPT;
      const int zz = tt / p.nPT;
      const int z = zz % p.Do;
      const int n = zz / p.Do;
      const int s0 = pt * BP;
      const int sLast = min(s0 + BP, p.S) - 1;
      const int r0 = (int)fdivd(s0, p.divDsY), c0 = min(s0 - r0 * dsY, p.Wo - 1);
      const int rl = (int)fdivd(sLast, p.divDsY), cl = min(sLast - rl * dsY, p.Wo - 1);
      const int span_lo = r0 * xsY + c0;
      const int L = rl * xsY + cl - span_lo + (p.kh - 1) * xsY + p.kw;
      // span position -> offset inside the staged input span
      for (int i = pw * 64 + lane; i < BP; i += 256) {
        const int s = min(s0 + i, sLast);
        const int r = (int)fdivd(s, p.divDsY);
        const int c = min(s - r * dsY, p.Wo - 1);
        tbl[i] = r * xsY + c - span_lo;
      }
      // input spans, 16 B per lane; lanes past the span are masked off, the
      // straddling lane over-reads <= 12 bytes, which stays inside the tensor
      // except on its very last row: that row goes by dwords.
      const float* xb = p.x + (long)n * p.xsN + (long)z * p.xsZ + span_lo;
      const int nJ = (L + 63) >> 6;
      const int nJ16 = (L + 255) >> 8;
      for (int slot = pw; slot < nSpans; slot += 4) {
        const int cs = slot / p.kd;
        const int dz = slot - cs * p.kd;
        const int ci = ciA + cs;
        const float* src = xb + (long)ci * p.xsC + (long)dz * p.xsZ;
        float* dst = xl + slot * Lpad;
        const bool tail_row = (ci == p.Cin - 1) && (z + dz == p.Din - 1) && (n == p.N - 1);
        if (!tail_row) {
          for (int j = 0; j < nJ16; ++j) {
            const int u = 256 * j + 4 * lane;
            if (u < L) d_glds16(src + u, dst + 256 * j);
          }
        } else {
          for (int j = 0; j < nJ; ++j)
            if (64 * j + lane < L) d_glds4(src + 64 * j + lane, dst + 64 * j);
        }
      }
    };
    // the quad pipeline reads one quad of offsets past the table: keep them in range
    if (pw < 2) {
      int* t = reinterpret_cast<int*>(smem + pw * p.bufFloats);
#pragma unroll
      for (int i = 0; i < 4; ++i) t[BP + 64 * i + lane] = 0;
    }
    if (tb < te) stage(tb, 0);
    for (int tt = tb; tt < te; ++tt) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __syncthreads();
      if (tt + 1 < te && !(p.dbg & 1)) stage(tt + 1, ((tt - tb) & 1) ^ 1);
    }
    return;
  }

  // ---- compute waves ----------------------------------------------------------
  unsigned long long* st = (p.stamps && tid == 0) ? p.stamps + 8L * blockIdx.x : nullptr;
  if (st) { st[0] = __builtin_amdgcn_s_memrealtime(); st[1] = __builtin_amdgcn_s_memtime(); }
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
  DAddr<MT, NT> ad;
#pragma unroll
  for (int nb = 0; nb < NT; ++nb) ad.lanebase[nb] = lanebase[nb];
#pragma unroll
  for (int mb = 0; mb < MT; ++mb) {
    const int co = min(m0 + mb * 16 + l15, p.Cout - 1);      // padded rows: discarded at the flush
    ad.voff[mb] = 4u * (unsigned)((long)co * p.dsC + 4 * qd);
  }
  f32x4 acc[MT][NT];
#pragma unroll
  for (int mb = 0; mb < MT; ++mb)
#pragma unroll
    for (int nb = 0; nb < NT; ++nb) acc[mb][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

  DQuad<MT, NT> g0, g1, g2;      // g2: third set of the bf16 form's rotation
#define E2_WAIT()                                                        \
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");            \
  __builtin_amdgcn_sched_barrier(0);

  // geometry of a tile.  Every wave runs the same EVEN number of quads per tile (the
  // two register sets ping-pong); quads that lie (partly) past the span are masked to
  // zero and fetch from the last valid quad's address.
  int nQ = 0, len = 0, aoff = 0, amax = 0;
  const float* tbase = nullptr;
  auto tile_geom = [&](int tt) {
    const int pt = tt % p.nPT;
    const int zz = tt / p.nPT;
    const int z = zz % p.Do;
    const int n = zz / p.Do;
    const int s0 = pt * BP;
    len = min(s0 + BP, p.S) - s0;                    // valid span positions of the tile
    const int nQreal = (len + 15) >> 4;
    nQ = 2 * ((nQreal + 2 * WK - 1) / (2 * WK));       // quads per wave
    amax = 16 * (nQreal - 1);
    tbase = p.dy + (long)n * p.dsN + (long)z * p.dsZ + s0;
    aoff = 16 * wk;
    ad.abase = tbase + min(aoff, amax);
  };
#define E2_ADV()                                                         \
  aoff += 16 * WK; ad.abase = tbase + min(aoff, amax); ad.addrT += 64u * WK;
  // zero the k-steps of this wave's local quad j that lie past the tile's span
#define E2_MASK(G, j)                                                    \
  { const int qi_ = wk + WK * (j);                                       \
    if (16 * (qi_ + 1) > len) G.mask(len - 16 * qi_ - 1 - 4 * qd); }
  if constexpr (BF) {
    // bf16 form: a quad's arithmetic (MT*NT short MFMAs) is far shorter than a round trip
    // to L2, so TWO quads are kept in flight -- three register sets in rotation; the
    // offsets a quad's gathers need are fetched two quads ahead (with quad q: those of
    // q+2).  Every tile starts from an empty pipeline (one barrier per tile, as the
    // producers expect); the quads per wave are padded to 3m+2 with masked quads, whose
    // offsets come from the zeroed slack of the table.
    constexpr int NA = MT;                             // dy loads per quad
    if (st) st[2] = __builtin_amdgcn_s_memtime();
    for (int tt = tb; tt < te; ++tt) {
      tile_geom(tt);
      {
        const int nQreal = (len + 15) >> 4;
        const int per = (nQreal + WK - 1) / WK;
        nQ = per <= 2 ? 2 : 3 * ((per - 2 + 2) / 3) + 2;
      }
      __syncthreads();                                 // tile tt landed; the other buffer is free
      const float* bufp = smem + ((tt - tb) & 1) * p.bufFloats;
      ad.xbase = d_lds_addr(bufp + TBF);
      ad.addrT = d_lds_addr(bufp + 4 * qd) + 64u * (unsigned)wk;
      i32x4 io0 = d_lds_ld128(ad.addrT);
      i32x4 io1 = d_lds_ld128(ad.addrT + 64u * WK);
      ad.addrT += 128u * WK;                           // offsets of quad 2
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("" : "+v"(io0));
      asm volatile("" : "+v"(io1));
#define E2_ISSUE(G, IO)                                                  \
      ad.abase = tbase + min(aoff, amax);                                \
      __builtin_amdgcn_sched_barrier(0);                                 \
      dquad_reads<MT, NT, 0, MT + 1 + 4 * NT>(G, ad, IO);                \
      __builtin_amdgcn_sched_barrier(0);                                 \
      aoff += 16 * WK; ad.addrT += 64u * WK;
#define E2_WAITN()                                                       \
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NA) : "memory"); \
      __builtin_amdgcn_sched_barrier(0);
#define E2_BF_STEP(CUR, NXT, TGT, J)                                     \
      E2_ISSUE(TGT, CUR.io)                                              \
      dquad_steps_bf<MT, NT, false>(CUR, TGT, acc, ad, CUR.io);          \
      E2_WAITN()                                                         \
      NXT.touch();                                                       \
      E2_MASK(NXT, (J) + 1)
      E2_ISSUE(g0, io0)
      E2_ISSUE(g1, io1)
      E2_WAITN()
      g0.touch();
      E2_MASK(g0, 0)
      int q = 0;
      for (; q + 2 < nQ; q += 3) {
        E2_BF_STEP(g0, g1, g2, q)
        E2_BF_STEP(g1, g2, g0, q + 1)
        E2_BF_STEP(g2, g0, g1, q + 2)
      }
      // quads nQ-2 (g0, ready) and nQ-1 (g1, in flight)
      dquad_steps_bf<MT, NT, false>(g0, g2, acc, ad, g0.io);
      E2_WAIT()
      g1.touch();
      E2_MASK(g1, q + 1)
      dquad_steps_bf<MT, NT, false>(g1, g2, acc, ad, g1.io);
#undef E2_BF_STEP
#undef E2_WAITN
#undef E2_ISSUE
    }
  } else {
  if (tb < te) {
    // first quad of the first tile, fetched in the open
    tile_geom(tb);
    g0.load_a(ad.abase, ad.voff);                     // dy does not depend on the LDS contents
    __syncthreads();                                  // the producers' table + spans landed
    ad.xbase = d_lds_addr(smem + TBF);
    ad.addrT = d_lds_addr(smem + 4 * qd) + 64u * (unsigned)wk;
    i32x4 io0 = d_lds_ld128(ad.addrT);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" : "+v"(io0));
    ad.addrT += 64u * WK;
    g0.load_b(ad.addrT, ad.xbase, lanebase, io0);     // + the offsets of the next quad
    E2_WAIT()
    g0.touch();
    E2_MASK(g0, 0)
  }
  if (st) st[2] = __builtin_amdgcn_s_memtime();
  for (int tt = tb; tt < te; ++tt) {
    // g0 holds this wave's quad 0 of tile tt; nQ is even
    int q = 0;
    for (; q + 2 < nQ; q += 2) {
      E2_ADV()
      __builtin_amdgcn_sched_barrier(0);
      dquad_go<MT, NT, true, BF>(g0, g1, acc, ad, g0.io);   // compute q, fetch q+1
      E2_WAIT()
      g1.touch();
      E2_MASK(g1, q + 1)
      E2_ADV()
      __builtin_amdgcn_sched_barrier(0);
      dquad_go<MT, NT, true, BF>(g1, g0, acc, ad, g1.io);   // compute q+1, fetch q+2
      E2_WAIT()
      g0.touch();
      E2_MASK(g0, q + 2)
    }
    E2_ADV()
    __builtin_amdgcn_sched_barrier(0);
    dquad_go<MT, NT, true, BF>(g0, g1, acc, ad, g0.io);     // quad nQ-2, fetch the last one
    E2_WAIT()
    g1.touch();
    E2_MASK(g1, q + 1)
    // While the tile's last quad (g1) computes, fetch the first quad of the NEXT
    // tile: its barrier, its first offsets, then the loads.
    if (tt + 1 < te) {
      tile_geom(tt + 1);
      __syncthreads();                                // tile tt+1 landed; nobody reads tile tt's buffer any more
      const float* bufp = smem + ((tt + 1 - tb) & 1) * p.bufFloats;
      ad.xbase = d_lds_addr(bufp + TBF);
      ad.addrT = d_lds_addr(bufp + 4 * qd) + 64u * (unsigned)wk;
      i32x4 io0 = d_lds_ld128(ad.addrT);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("" : "+v"(io0));
      ad.addrT += 64u * WK;
      dquad_go<MT, NT, true, BF>(g1, g0, acc, ad, io0);
      E2_WAIT()
      g0.touch();
      E2_MASK(g0, 0)
    } else {
      dquad_go<MT, NT, false, BF>(g1, g0, acc, ad, g1.io);
    }
  }
  }
#undef E2_ADV
#undef E2_MASK
#undef E2_WAIT
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  g0.touch();
  g1.touch();
  if (st) st[3] = __builtin_amdgcn_s_memtime();

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
  if (st) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    st[4] = __builtin_amdgcn_s_memtime();
    st[5] = __builtin_amdgcn_s_memrealtime();
  }
}

template <int MT, int NT, int BP, int WK, bool BF>
static int launch_d2(e2_ctx* ctx, const WdP& p, int grid, size_t lds) {
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(
        reinterpret_cast<const void*>(&wgrad_direct_kernel<MT, NT, BP, WK, BF>),
        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) { e2_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return 1; }
    attr_done = true;
  }
  hipLaunchKernelGGL((wgrad_direct_kernel<MT, NT, BP, WK, BF>), dim3(grid), dim3(512), lds, ctx->stream, p);
  E2_CHECK_HIP(hipGetLastError());
  return 0;
}

// the bf16 form exists for the larger position tile only (BP = 256)
template <int MT, int NT, int BP, int WK>
static int launch_d(e2_ctx* ctx, const WdP& p, int grid, size_t lds) {
  if (ctx->mfma_bf16) {
    // (7 x 4 blocks: the three operand sets of the bf16 form do not fit the registers)
    // (otherwise the tiling runs in its f32 form: at least as exact, never preferred)
    if constexpr (BP == 256 && MT * NT < 28) return launch_d2<MT, NT, BP, WK, true>(ctx, p, grid, lds);
  }
  return launch_d2<MT, NT, BP, WK, false>(ctx, p, grid, lds);
}

template <int MT>
static int dispatch_d2(e2_ctx* ctx, const WdP& p, int NT, int BP, int WK, int grid, size_t lds) {
  // (MT x 4 tiles of 20+ blocks do not fit 256 registers: the compiler spills operand
  // registers of the hand-scheduled loop to scratch -- no such instance, check_scratch.py)
  if (WK == 4) {
    if (BP == 128 && NT == 2) return launch_d<MT, 2, 128, 4>(ctx, p, grid, lds);
    if (BP == 256 && NT == 2) return launch_d<MT, 2, 256, 4>(ctx, p, grid, lds);
    if constexpr (MT * 4 < 20) {
      if (BP == 128 && NT == 4) return launch_d<MT, 4, 128, 4>(ctx, p, grid, lds);
      if (BP == 256 && NT == 4) return launch_d<MT, 4, 256, 4>(ctx, p, grid, lds);
    }
  } else if (BP == 128) {
    if (NT == 1) return launch_d<MT, 1, 128, 1>(ctx, p, grid, lds);
    if (NT == 2) return launch_d<MT, 2, 128, 1>(ctx, p, grid, lds);
    if constexpr (MT * 4 < 20) if (NT == 4) return launch_d<MT, 4, 128, 1>(ctx, p, grid, lds);
  } else if (BP == 256) {
    if (NT == 1) return launch_d<MT, 1, 256, 1>(ctx, p, grid, lds);
    if (NT == 2) return launch_d<MT, 2, 256, 1>(ctx, p, grid, lds);
    if constexpr (MT * 4 < 20) if (NT == 4) return launch_d<MT, 4, 256, 1>(ctx, p, grid, lds);
  }
  e2_set_error("wgrad(direct): no instance NT=%d BP=%d WK=%d", NT, BP, WK);
  return 2;
}

}  // namespace

// geometry helpers shared with the host-side tiling choice (conv_wgrad.hip)
int e2i_wgrad_direct_lpad(const WgradArgs& a, int BP) {
  const int rows = (BP + (int)a.dsY - 2) / (int)a.dsY;      // rows crossed by BP span positions
  const int lmax = rows * (int)a.xsY + (a.Wo - 1) + (a.kh - 1) * (int)a.xsY + a.kw;
  int lp = ((lmax + 3 + 3) / 4) * 4;                        // + the straddling DMA lane
  if ((lp & 31) == 0) lp += 4;
  return lp;
}
static int d_maxspans(const WgradArgs& a, int BNn) {
  const int T = a.kd * a.kh * a.kw;
  int cis = (BNn - 1) / T + 2;
  if (cis > a.Cin) cis = a.Cin;
  return cis * a.kd;
}
size_t e2i_wgrad_direct_buf_floats(const WgradArgs& a, int NT, int BP, int WK) {
  return (size_t)(BP + 256) + (size_t)d_maxspans(a, 16 * NT * (4 / WK)) * e2i_wgrad_direct_lpad(a, BP) + 64;
}

int e2i_wgrad_direct(e2_ctx* ctx, const WgradArgs& a, int MT, int NT, int BP, int PS, int WK, int xcd) {
  E2_REQUIRE(BP == 128 || BP == 256, "wgrad(direct): BP must be 128 or 256");
  E2_REQUIRE(WK == 1 || (WK == 4 && (NT == 2 || NT == 4)), "wgrad(direct): WK=4 needs NT 2 or 4");
  E2_REQUIRE(a.xsY < (1 << 20) && a.dsY < (1 << 20), "wgrad: row stride too large");
  E2_REQUIRE((long)a.Cout * a.dsC < (1L << 29), "wgrad(direct): gradient sample too large");
  WdP p;
  p.x = a.x; p.dy = a.dy; p.dw = a.dw;
  p.Cin = a.Cin; p.Cout = a.Cout; p.kd = a.kd; p.kh = a.kh; p.kw = a.kw;
  p.THW = a.kh * a.kw; p.T = a.kd * p.THW;
  p.Do = a.Do; p.Ho = a.Ho; p.Wo = a.Wo;
  p.S = (a.Ho - 1) * (int)a.dsY + a.Wo;
  p.xsN = a.xsN; p.xsC = a.xsC; p.xsZ = a.xsZ; p.xsY = a.xsY;
  p.dsN = a.dsN; p.dsC = a.dsC; p.dsZ = a.dsZ; p.dsY = a.dsY;
  p.flip = a.flip;
  p.upR = a.upR > 1 ? a.upR : 1;
  const long NTOT = (long)a.Cin * p.T;
  E2_REQUIRE(NTOT < (1L << 30), "wgrad: Cin*T too large");
  p.NTOT = (int)NTOT;
  const int BNn = 16 * NT * (4 / WK);
  p.Lpad = e2i_wgrad_direct_lpad(a, BP);
  p.maxSpans = d_maxspans(a, BNn);
  p.nMT = e2_cdiv(e2_cdiv(a.Cout, 16), MT);
  p.nNT = e2_cdiv(e2_cdiv(p.NTOT, 16), NT * (4 / WK));
  p.nPT = e2_cdiv(p.S, BP);
  p.tilesTotal = a.N * a.Do * p.nPT;
  p.nPS = std::max(1, std::min(PS, p.tilesTotal));
  p.bufFloats = (int)e2i_wgrad_direct_buf_floats(a, NT, BP, WK);
  p.Din = a.Do + a.kd - 1;
  p.N = a.N;
  p.divDsY = mk_divd((unsigned)a.dsY);
  p.dbg = e2_dbg_env_int("E2_WGRAD_DBG");
  p.xcd = xcd ? 1 : 0;
  const size_t lds = 2 * (size_t)p.bufFloats * 4;
  E2_REQUIRE(lds <= 160 * 1024, "wgrad(direct): tiling needs %zu B of LDS", lds);
  const long pairs = (long)p.nMT * p.nPS * p.nNT;
  const long grid = p.xcd ? ((pairs + 7) / 8) * 8 : pairs;
  p.stamps = nullptr;
  static unsigned long long* stamp_buf = nullptr;
  const bool want_stamps = e2_dbg_env("E2_WGRAD_STAMPS") != nullptr && !ctx->capturing && grid <= 65536;
  if (want_stamps) {
    if (!stamp_buf) E2_CHECK_HIP(hipMalloc(&stamp_buf, 8 * sizeof(unsigned long long) * 65536));
    E2_CHECK_HIP(hipMemsetAsync(stamp_buf, 0, 8 * sizeof(unsigned long long) * grid, ctx->stream));
    p.stamps = stamp_buf;
  }
  E2_REQUIRE(grid < (1L << 31), "wgrad: grid too large");
  if (!a.accumulate)
    if (int rc = e2i_fill_flat(ctx, a.dw, (size_t)a.Cout * p.NTOT, 0.f)) return rc;
  if (e2_dbg_env("E2_VERBOSE"))
    fprintf(stderr, "[e2] wgrad(direct%s) Cin=%d Cout=%d k=%d,%d,%d out=%d,%d,%d MT=%d NT=%d BP=%d WK=%d PS=%d grid=%ld lds=%zu\n",
            ctx->mfma_bf16 ? ", bf16" : "", a.Cin, a.Cout, a.kd, a.kh, a.kw, a.Do, a.Ho, a.Wo, MT, NT, BP, WK, p.nPS, grid, lds);
  int rc = 2;
  switch (MT) {
    case 1: rc = dispatch_d2<1>(ctx, p, NT, BP, WK, (int)grid, lds); break;
    case 2: rc = dispatch_d2<2>(ctx, p, NT, BP, WK, (int)grid, lds); break;
    case 3: rc = dispatch_d2<3>(ctx, p, NT, BP, WK, (int)grid, lds); break;
    case 4: rc = dispatch_d2<4>(ctx, p, NT, BP, WK, (int)grid, lds); break;
    case 5: rc = dispatch_d2<5>(ctx, p, NT, BP, WK, (int)grid, lds); break;
    case 7: rc = dispatch_d2<7>(ctx, p, NT, BP, WK, (int)grid, lds); break;
    default: e2_set_error("wgrad(direct): no instance MT=%d", MT);
  }
  if (rc == 0 && want_stamps) {
    std::vector<unsigned long long> h(8 * grid);
    E2_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    E2_CHECK_HIP(hipMemcpy(h.data(), stamp_buf, 8 * sizeof(unsigned long long) * grid, hipMemcpyDeviceToHost));
    unsigned long long r0 = ~0ull, r1 = 0;
    double s_first = 0, s_loop = 0, s_epi = 0, s_tot = 0, max_tot = 0, last_start = 0;
    for (long b = 0; b < grid; ++b) {
      const unsigned long long* s = &h[8 * b];
      r0 = std::min(r0, s[0]); r1 = std::max(r1, s[5]);
      s_first += (double)(s[2] - s[1]); s_loop += (double)(s[3] - s[2]);
      s_epi += (double)(s[4] - s[3]); s_tot += (double)(s[4] - s[1]);
      max_tot = std::max(max_tot, (double)(s[4] - s[1]));
    }
    for (long b = 0; b < grid; ++b) last_start = std::max(last_start, (double)(h[8 * b] - r0));
    const int per = (p.tilesTotal + p.nPS - 1) / p.nPS;
    fprintf(stderr, "[e2 wgrad stamps] grid=%ld tiles/wg=%d  span=%.2f us  last start +%.2f us | per wg (cycles, mean): "
            "to first quad %.0f, main loop %.0f (ideal %.0f), flush %.0f, total %.0f (max %.0f)\n",
            grid, per, (double)(r1 - r0) * 0.01, last_start * 0.01, s_first / grid, s_loop / grid,
            (double)per * (BP / 4 / WK) * MT * NT * 32.0, s_epi / grid, s_tot / grid, max_tot);
  }
  return rc;
}
