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
// its position tiles and flushes once with fp32 atomics into the zeroed dw.
//
// Position tile = BP consecutive positions of one z-plane.  As in the forward
// kernel the input the tile touches is one contiguous span per (ci, dz); the
// spans of every (ci, dz) the N-tile needs are staged in LDS.  The B operand of
// lane (j = n-index, qd = pos & 3) is x_l[lanebase(j) + inoff[pos]] where
// lanebase = slot(ci,dz)*Lpad + ty*sY + tx is loop invariant and inoff[pos]
// (the position's offset inside the span) comes from a small LDS table.
//
// LDS row strides are == 2 (mod 4) so that 16 channel rows x 2 position
// quarters of a 32-lane ds_read_b32 group fall on 32 distinct banks.
#include "common.hpp"
#include <stdlib.h>
#include <algorithm>

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct WgradP {
  const float* x;
  const float* dy;
  float* dw;
  int Cin, Cout, kd, kh, kw, T, THW;
  int Do, Ho, Wo, Q;
  long xsN, xsC, xsZ, xsY;
  long dsN, dsC, dsZ, dsY;
  int flip, upR;
  int NTOT;
  int BP, log2BP;
  int DLpad, Lpad;
  int nMT, nNT, nPS;
  int nPT, tilesTotal;
  int maxSpans;
};

template <int MT, int NT, int WN, int WK>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(WgradP p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int BM = 16 * MT;
  constexpr int BNn = 16 * NT * WN;
  float* dyl = smem;                                  // BM * DLpad
  float* xl = dyl + BM * p.DLpad;                     // maxSpans * Lpad
  int* inoff = reinterpret_cast<int*>(xl + p.maxSpans * p.Lpad);   // BP

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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

  int lanebase[NT];
#pragma unroll
  for (int nb = 0; nb < NT; ++nb) {
    const int jn = min(n0 + (wn * NT + nb) * 16 + l15, p.NTOT - 1);
    const int ci = jn / p.T;
    const int tap = jn - ci * p.T;
    const int dz = tap / p.THW;
    const int t2 = tap - dz * p.THW;
    const int ty = t2 / p.kw, tx = t2 - ty * p.kw;
    lanebase[nb] = ((ci - ciA) * p.kd + dz) * p.Lpad + ty * xsY + tx;
  }

  f32x4 acc[MT][NT];
#pragma unroll
  for (int mb = 0; mb < MT; ++mb)
#pragma unroll
    for (int nb = 0; nb < NT; ++nb) acc[mb][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int per = (p.tilesTotal + p.nPS - 1) / p.nPS;
  const int tb = ps * per, te = min(tb + per, p.tilesTotal);
  const int BP = p.BP;
  const int pl = tid & (BP - 1);          // this thread's position in the tile
  const int coStep = 256 >> p.log2BP;
  const int co0 = tid >> p.log2BP;

  for (int tt = tb; tt < te; ++tt) {
    const int pt = tt % p.nPT;
    const int zz = tt / p.nPT;
    const int z = zz % p.Do;
    const int n = zz / p.Do;
    const int q0 = pt * BP;
    const int qlast = min(q0 + BP, p.Q) - 1;
    const int r0 = q0 / p.Wo, c0 = q0 - r0 * p.Wo;
    const int rl = qlast / p.Wo, cl = qlast - rl * p.Wo;
    const long span_lo = (long)r0 * p.xsY + c0;
    const int L = (rl - r0) * xsY + (cl - c0) + (p.kh - 1) * xsY + p.kw;

    // ---- position table + dy tile (thread <-> fixed position) ------------
    {
      const int q = q0 + pl;
      const bool valid = q <= qlast;
      const int qc = valid ? q : qlast;
      const int r = qc / p.Wo, c = qc - r * p.Wo;
      if (tid < BP) inoff[pl] = (r - r0) * xsY + (c - c0);
      const float* src = p.dy + (long)n * p.dsN + (long)z * p.dsZ + (long)r * p.dsY + c;
      for (int co = co0; co < BM; co += coStep) {
        float v = 0.f;
        if (valid && (m0 + co) < p.Cout) v = src[(long)(m0 + co) * p.dsC];
        dyl[co * p.DLpad + pl] = v;
      }
    }
    // ---- input spans --------------------------------------------------------
    {
      const float* xb = p.x + (long)n * p.xsN + (long)z * p.xsZ + span_lo;
      for (int slot = wave; slot < nSpans; slot += 4) {
        const int ci = ciA + slot / p.kd;
        const int dz = slot - (slot / p.kd) * p.kd;
        const float* src = xb + (long)ci * p.xsC + (long)dz * p.xsZ;
        float* dst = xl + slot * p.Lpad;
        for (int u = lane; u < L; u += 64) dst[u] = src[u];
      }
    }
    __syncthreads();
    // ---- MFMA: K = positions, 4 per step ----------------------------------
    const int nsteps = (qlast - q0 + 4) >> 2;
    for (int s = wk; s < nsteps; s += WK) {
      const int pk = 4 * s + qd;
      const int io = inoff[pk];
      float a[MT], b[NT];
#pragma unroll
      for (int mb = 0; mb < MT; ++mb) a[mb] = dyl[(mb * 16 + l15) * p.DLpad + pk];
#pragma unroll
      for (int nb = 0; nb < NT; ++nb) b[nb] = xl[lanebase[nb] + io];
#pragma unroll
      for (int mb = 0; mb < MT; ++mb)
#pragma unroll
        for (int nb = 0; nb < NT; ++nb)
          acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mb], b[nb],
                                                             acc[mb][nb], 0, 0, 0);
    }
    __syncthreads();
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
template <int MT, int NT, int WN, int WK>
static int launch_w(e2_ctx* ctx, const WgradP& p, int grid, size_t lds) {
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(
        reinterpret_cast<const void*>(&wgrad_kernel<MT, NT, WN, WK>),
        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) { e2_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return 1; }
    attr_done = true;
  }
  hipLaunchKernelGGL((wgrad_kernel<MT, NT, WN, WK>), dim3(grid), dim3(256), lds,
                     ctx->stream, p);
  E2_CHECK_HIP(hipGetLastError());
  return 0;
}

static const int kWMTs[] = {1, 2, 3, 4, 5, 7};

static int dispatch_w(e2_ctx* ctx, const WgradP& p, int MT, int NT, int WK, int grid,
                      size_t lds) {
#define E2_W(M)                                                                  \
  case M:                                                                        \
    if (WK == 4) return launch_w<M, 1, 1, 4>(ctx, p, grid, lds);                 \
    if (NT == 1) return launch_w<M, 1, 4, 1>(ctx, p, grid, lds);                 \
    if (NT == 2) return launch_w<M, 2, 4, 1>(ctx, p, grid, lds);                 \
    if (NT == 4) return launch_w<M, 4, 4, 1>(ctx, p, grid, lds);                 \
    break;
  switch (MT) { E2_W(1) E2_W(2) E2_W(3) E2_W(4) E2_W(5) E2_W(7) }
#undef E2_W
  e2_set_error("wgrad: no instance MT=%d NT=%d WK=%d", MT, NT, WK);
  return 2;
}

static int pad2mod4(int v) {             // smallest s >= v with s % 4 == 2
  int s = v;
  while ((s & 3) != 2) ++s;
  return s;
}

struct WCfg { int MT, NT, WK, BP, PS; };

static WCfg choose_wcfg(const e2_ctx* ctx, const WgradArgs& a, int* ok) {
  const int mblocks = e2_cdiv(a.Cout, 16);
  const int T = a.kd * a.kh * a.kw;
  const long NTOT = (long)a.Cin * T;
  const int nblocks = (int)((NTOT + 15) / 16);
  const long Q = (long)a.Ho * a.Wo;
  const int slots = ctx->num_cu * 2;
  WCfg best{0, 0, 0, 0, 0};
  double bestCost = 1e300;
  const char* force = getenv("E2_WGRAD_FORCE");
  if (force) {
    WCfg f{0, 0, 0, 0, 0};
    if (sscanf(force, "%d,%d,%d,%d,%d", &f.MT, &f.NT, &f.WK, &f.BP, &f.PS) == 5) { *ok = 1; return f; }
  }
  for (int MT : kWMTs) {
    if (MT > mblocks && MT != 1) continue;
    const int nMT = e2_cdiv(mblocks, MT);
    for (int v = 0; v < 4; ++v) {
      const int NT = (v == 3) ? 1 : (1 << v);       // 1,2,4 | WK variant
      const int WK = (v == 3) ? 4 : 1;
      const int WN = (v == 3) ? 1 : 4;
      const int BNn = 16 * NT * WN;
      if (BNn > 16 * nblocks && !(NT == 1)) continue;
      const int nNT = e2_cdiv(nblocks, NT * WN);
      for (int BP = 64; BP <= 256; BP *= 2) {
        if (BP > 64 && Q <= BP / 2) continue;
        const int rows = (BP + a.Wo - 2) / a.Wo;
        const int Lmax = (rows + a.kh - 1) * (int)a.xsY + a.kw + a.Wo;
        const int Lpad = pad2mod4(Lmax);
        const int DLpad = pad2mod4(BP);
        const int maxSpans = ((BNn - 1) / T + 2) * a.kd;
        const int spansEff = std::min<long>(maxSpans, (long)a.Cin * a.kd);
        const size_t lds = ((size_t)16 * MT * DLpad + (size_t)maxSpans * Lpad + BP) * 4;
        if (lds > 72 * 1024) continue;
        const int nPT = (int)((Q + BP - 1) / BP);
        const long tiles = (long)a.N * a.Do * nPT;
        const long base = (long)nMT * nNT;
        // position splits: fill the machine, but bound the atomic volume
        long PS = std::max<long>(1, (slots + base - 1) / base);
        PS = std::min(PS, tiles);
        const int per = (int)((tiles + PS - 1) / PS);
        const double steps = BP / 4.0 / WK;
        const double mfma = per * steps * MT * NT * 32.0;
        const double reads = per * steps * (MT + NT + 1) * 6.0;
        const double stage = per * ((16.0 * MT * BP + (double)spansEff * Lmax) / 256.0 * 6.0 + 700.0);
        const double flush = 16.0 * MT * 16.0 * NT * 4 * 0.5 * WK;
        const double wg_time = std::max(mfma, reads) + stage + flush + 1500.0;
        const long wgs = base * PS;
        const double rounds = (double)((wgs + slots - 1) / slots);
        // chip-wide atomic floor: bytes / 1.3 TB/s in cycles @2.4GHz
        const double atom = (double)wgs * 16 * MT * BNn * 4.0 * WK / 1.3e12 * 2.4e9;
        const double cost = std::max(rounds * wg_time * 2.0, atom);
        if (cost < bestCost) { bestCost = cost; best = WCfg{MT, NT, WK, BP, (int)PS}; }
      }
    }
  }
  *ok = best.MT != 0;
  return best;
}

int e2i_wgrad_conv(e2_ctx* ctx, const WgradArgs& a) {
  E2_REQUIRE(a.Do > 0 && a.Ho > 0 && a.Wo > 0 && a.Cin > 0 && a.Cout > 0,
             "wgrad: empty problem");
  int ok = 0;
  WCfg c = choose_wcfg(ctx, a, &ok);
  E2_REQUIRE(ok, "wgrad: no tiling fits LDS (Cin=%d Cout=%d k=%dx%dx%d)", a.Cin, a.Cout,
             a.kd, a.kh, a.kw);
  WgradP p;
  p.x = a.x; p.dy = a.dy; p.dw = a.dw;
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
  p.BP = c.BP;
  p.log2BP = (c.BP == 64) ? 6 : (c.BP == 128 ? 7 : (c.BP == 256 ? 8 : 5));
  E2_REQUIRE((1 << p.log2BP) == c.BP, "wgrad: BP must be 32..256 pow2");
  const int WN = (c.WK == 4) ? 1 : 4;
  const int BNn = 16 * c.NT * WN;
  const int rows = (c.BP + a.Wo - 2) / a.Wo;
  const int Lmax = (rows + a.kh - 1) * (int)a.xsY + a.kw + a.Wo;
  p.Lpad = pad2mod4(Lmax);
  p.DLpad = pad2mod4(c.BP);
  p.maxSpans = ((BNn - 1) / p.T + 2) * a.kd;
  p.nMT = e2_cdiv(e2_cdiv(a.Cout, 16), c.MT);
  p.nNT = e2_cdiv(e2_cdiv(p.NTOT, 16), c.NT * WN);
  p.nPT = e2_cdiv(p.Q, c.BP);
  p.tilesTotal = a.N * a.Do * p.nPT;
  p.nPS = std::min(c.PS, p.tilesTotal);
  E2_REQUIRE(a.xsY < (1 << 20), "wgrad: input row stride too large");
  const size_t lds = ((size_t)16 * c.MT * p.DLpad + (size_t)p.maxSpans * p.Lpad + c.BP) * 4;
  const long grid = (long)p.nMT * p.nNT * p.nPS;
  E2_REQUIRE(grid < (1L << 31), "wgrad: grid too large");
  E2_CHECK_HIP(hipMemsetAsync(a.dw, 0, sizeof(float) * (size_t)a.Cout * p.NTOT, ctx->stream));
  if (getenv("E2_VERBOSE"))
    fprintf(stderr, "[e2] wgrad Cin=%d Cout=%d k=%d,%d,%d out=%d,%d,%d MT=%d NT=%d WK=%d BP=%d PS=%d grid=%ld lds=%zu\n",
            a.Cin, a.Cout, a.kd, a.kh, a.kw, a.Do, a.Ho, a.Wo, c.MT, c.NT, c.WK, c.BP, p.nPS, grid, lds);
  return dispatch_w(ctx, p, c.MT, c.NT, c.WK, (int)grid, lds);
}
