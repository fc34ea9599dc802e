// wgrad_bf16.hip -- conv weight gradient with bf16 operands IN MEMORY, on
// v_mfma_f32_32x32x16_bf16; the companion of conv_bf16.hip for the third GEMM of a step.
//
//   dw[oc][ic][tap] (+)= sum_pos dy[oc][pos] * x[ic][pos + shift(tap)]       (T.grad's ConvGradW)
//
// GEMM view: M = out channels, N = (tap, ic) columns, K = the positions of a plane.  The
// two operands want opposite memory orders:
//   * dy is never shifted: converted to bf16 CHANNEL-major planes dyc[n][oc][z][planeD]
//     written at the INPUT's row pitch (zeros in the kw-1 extra columns and behind the plane),
//     K is contiguous and 16-byte aligned: staged into LDS row by row, read with ds_read_b128;
//   * x is read at a tap's shift, which in a K-contiguous image would be 2 bytes off any
//     LDS vector read (DESIGN.md finding 20); it is converted to CHANNELS-LAST pixels
//     xcl[n][z][kg][pixel][8] (16 bytes = 8 channels of one pixel, as conv_bf16.hip), where a
//     shift is a whole number of pixels, and read with the TRANSPOSED LDS read of gfx950,
//     ds_read_b64_tr_b16: a 16-lane group fetches 4 pixels x 16 channels and every lane
//     receives 4 consecutive positions of ITS channel (tools/ubench/tr_read_check.hip).
// With dy at the input's pitch the flattened position q of a gradient plane pairs with
// q + sy*Win + sx of input plane z + sz for the whole plane: no row bookkeeping in the K loop.
// A work-group (8 waves: 2 along the out channels x 4 along the input-channel blocks; a wave
// owns MB x NB blocks of 32 x 32, its NB column blocks being NB taps of one row of the kernel
// plane for one block of 32 input channels) walks a range of 64-position units, staging both
// tiles of the NEXT unit by LDS-DMA while it computes the current one, reads its operands one
// step ahead, and flushes with f32 atomics into dw.
// Same rounding as the operand-rounding form: the bounds of tests/test_bf16_gpu.py hold.
#include "common.hpp"
#include <algorithm>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_vp;
typedef const __attribute__((address_space(1))) void* gbl_vp;

namespace {

constexpr int kKU = 64;       // positions per unit (4 steps of 16)
constexpr int kKGW = 16;      // channel groups (of 8) per work-group: 4 wave columns x 32 channels
                              // (row form: 4 -- the wave columns are kernel rows)

struct WcP {                  // the conversion pass
  const float* x; long xsN, xsC, xsZ, xsY;
  const float* dy; long dsN, dsC, dsZ, dsY;
  int N, Cin, Cout, Din, Hin, Win, Do, Ho, Wo;
  int KG, planeD, planePix;
  int cx, cd, nbx, nbd, nbs;  // blocks per x row / dy row; blocks of the x part, dy part, slack
  long xPieces, xSlack;
  __bf16* xcl; __bf16* dyc;
  float* dwt; long nT4;       // the f32 sums to zero (float4 units)
};
// one launch, four kinds of blocks (32-bit index arithmetic, uniform per block):
//   x  -> channels-last pixels [n][z][kg][pixel][8]       (256 pixels of one (n, z, kg) per block)
//   dy -> channel-major planes at the input's row pitch    (2048 positions of one (n, c, z))
//   zero pixels in the slack behind the last plane; zeros in the f32 sums
__global__ __launch_bounds__(256) void wgrad_bf16_cvt_kernel(WcP p) {
  int b = blockIdx.x;
  const int tid = threadIdx.x;
  if (b < p.nbx) {
    const int row = b / p.cx, pix = (b - row * p.cx) * 256 + tid;
    if (pix >= p.planePix) return;
    const int kg = row % p.KG, nz = row / p.KG;
    const int z = nz % p.Din, n = nz / p.Din;
    const int y = pix / p.Win, xx = pix - y * p.Win;
    const float* src = p.x + (long)n * p.xsN + (long)z * p.xsZ + (long)y * p.xsY + xx;
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = kg * 8 + j;
      v[j] = (__bf16)(c < p.Cin ? src[(long)c * p.xsC] : 0.f);
    }
    *reinterpret_cast<bf16x8*>(p.xcl + ((long)row * p.planePix + pix) * 8) = v;
    return;
  }
  b -= p.nbx;
  if (b < p.nbd) {
    const int row = b / p.cd, q0 = ((b - row * p.cd) * 256 + tid) * 8;
    if (q0 >= p.planeD) return;                  // (planeD is a multiple of 64)
    const int z = row % p.Do, nc = row / p.Do;
    const int c = nc % p.Cout, n = nc / p.Cout;
    const float* src = p.dy + (long)n * p.dsN + (long)c * p.dsC + (long)z * p.dsZ;
    int y = q0 / p.Win, xx = q0 - y * p.Win;
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      v[j] = (__bf16)((y < p.Ho && xx < p.Wo) ? src[(long)y * p.dsY + xx] : 0.f);
      if (++xx == p.Win) { xx = 0; ++y; }
    }
    *reinterpret_cast<bf16x8*>(p.dyc + (long)row * p.planeD + q0) = v;
    return;
  }
  b -= p.nbd;
  if (b < p.nbs) {
    const long i = (long)b * 256 + tid;
    if (i < p.xSlack) {
      bf16x8 v;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (__bf16)0.f;
      *reinterpret_cast<bf16x8*>(p.xcl + (p.xPieces + i) * 8) = v;
    }
    return;
  }
  b -= p.nbs;
  const long i = (long)b * 256 + tid;
  if (i < p.nT4) reinterpret_cast<float4*>(p.dwt)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

struct WbP {
  const __bf16* xcl;
  const __bf16* dyc;
  float* dwt;                         // [Cout][T][CinP] f32 sums (atomics), zeroed by the conversion pass
  int N, Cin, Cout, Din, Do, KG, CinP;
  int KGs;                            // channel groups per plane IN MEMORY (>= KG: the forward's copy pads to an even count)
  int kd, kh, kw, T;
  int Win, planeD;
  long planePix;
  int unitsPerPlane, units, per;      // K units of kKU positions; per work-group
  int nMT, nIT, nTG;                  // tiles: out channels, input-channel blocks, tap groups per dz
  int NS;                             // stages of the LDS ring (2..4)
  int Lpix, Lwin, nJ, TB;             // x window: LDS pixels per channel group, pixels read, 64-pixel
                                      // DMA pieces, DMA instructions per wave and unit
  int nWG, perXcd;                    // work-groups; per XCD (the grid is 8 * perXcd)
};

constexpr int kTBmax = 4;             // DMA instructions of the x window per wave and unit, at most

// ROWS = false: the four wave columns are four blocks of 32 input channels (128 per work-group)
// and the work-group owns NB taps of ONE kernel row.  ROWS = true (layers with few input
// channels): the wave columns are four ROWS of the kernel plane, the work-group owns 32 input
// channels and one window that spans the rows -- 40 channels pad to 64 instead of 128.
template <int MB, int NB, bool ROWS>
__global__ __launch_bounds__(512) void wgrad_bf16_kernel(WbP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int c32 = lane & 31, kh8 = lane >> 5;
  constexpr int KGW = ROWS ? 4 : kKGW;
  const int CNT = MB + p.TB;                     // LDS-DMA instructions per wave and unit

  // Work-groups are numbered split-major (all tiles of a split are neighbours) and XCD x
  // (blockIdx % 8) takes the x-th eighth of that order: the tiles of a split run on one XCD
  // at the same time and share their slice of dy and x in that XCD's L2
  int b = (blockIdx.x & 7) * p.perXcd + (blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= p.perXcd || b >= p.nWG) return;
  const int tg = b % p.nTG; b /= p.nTG;          // group of NB taps in one kernel-plane row
  const int dz = b % p.kd;  b /= p.kd;
  const int it = b % p.nIT; b /= p.nIT;          // 128 (32) input channels
  const int mt = b % p.nMT;                      // 64*MB out channels
  const int sp = b / p.nMT;
  const int u0 = sp * p.per, u1 = min(u0 + p.per, p.units);
  if (u0 >= u1) return;                          // (whole work-group)
  const int tgPerRow = (p.kw + NB - 1) / NB;
  const int tyg = tg / tgPerRow, tx0 = (tg - tyg * tgPerRow) * NB;
  const int ty0 = ROWS ? 4 * tyg : tyg;          // first kernel row of the work-group
  const int tyw = ROWS ? ty0 + wn : ty0;         // this wave's kernel row (>= kh: an idle wave column)
  // true convolution: weight tap (dz, ty, tx) meets the input at the FLIPPED shift
  const int sz = p.kd - 1 - dz, sy = p.kh - 1 - min(tyw, p.kh - 1);
  const int syLo = p.kh - 1 - min(ROWS ? ty0 + 3 : ty0, p.kh - 1);
  const int shiftLo = syLo * p.Win + (p.kw - 1 - min(tx0 + NB - 1, p.kw - 1));   // smallest shift of the group
  const int M0 = mt * 64 * MB;                   // first out channel of the work-group
  const int kg0 = it * KGW;                      // first channel group of the work-group
  const int kgs = min(KGW, p.KG - kg0);
  // a stage: [64*MB dy rows][128 B, 16-byte pieces XOR-swizzled by the row] + [KGW kg][Lpix][16 B];
  // Lpix = 12 mod 16 so that the four groups of a transposed read fall into different banks
  constexpr unsigned bytesA = 64 * MB * kKU * 2;
  const unsigned bufBytes = bytesA + (unsigned)KGW * p.Lpix * 16;

  // ---- per-lane read addresses ------------------------------------------------------
  // A (ds_read_b128): row = out channel of the wave's block, piece (2 st + lane/32) of the
  // row's 8, stored at piece ^ (row & 7): 8 consecutive lanes hit 8 different 16-byte columns
  unsigned aaddr[MB][kKU / 16];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int st = 0; st < kKU / 16; ++st) {
      const int row = (wm * MB + mb) * 32 + c32;
      aaddr[mb][st] = (unsigned)(uintptr_t)(lds_vp)(lds + row * (kKU * 2) + (((2 * st + kh8) ^ (row & 7)) << 4));
    }
  // B (2 x ds_read_b64_tr_b16 per block): the 16-lane group (lane >> 4) takes channels
  // 16*(g & 1) .. +15 of the wave's 32-channel block and positions 8*(g >> 1) .. +7 of the
  // step; lane 4q+p of the group supplies pixel q, channels 4p..4p+3
  const int g = lane >> 4, i16 = lane & 15, q4 = i16 >> 2, p4 = i16 & 3;
  const int kgl = (ROWS ? 0 : wn * 4) + 2 * (g & 1) + (p4 >> 1);          // channel group inside the window
  unsigned baddr[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int tx = min(tx0 + nb, p.kw - 1);
    const int shift = sy * p.Win + (p.kw - 1 - tx) - shiftLo;  // pixels, relative to the window
    baddr[nb] = (unsigned)(uintptr_t)(lds_vp)(lds + bytesA + (kgl * p.Lpix + shift + 8 * (g >> 1) + q4) * 16 +
                                              (p4 & 1) * 8);
  }
  f32x16 acc[MB][NB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mb][nb][i] = 0.f;

  // ---- staging: per-lane parts of the global addresses (elements), LDS destinations --------
  // dy: wave w moves rows 8 (w + 8 c) .. +7 for c < MB with one DMA each (lane = 8 r + piece:
  // the DMA writes lane-linear, the swizzle is applied to the global side)
  long aoff[MB];
  unsigned adst[MB];
#pragma unroll
  for (int c = 0; c < MB; ++c) {
    const int row = 8 * (wave + 8 * c) + (lane >> 3);
    const int oc = min(M0 + row, p.Cout - 1);
    aoff[c] = (long)oc * p.Do * p.planeD + (((lane & 7) ^ (row & 7)) << 3);
    adst[c] = (unsigned)(8 * (wave + 8 * c)) * (kKU * 2);
  }
  // x window: KGW channel groups x nJ pieces of 64 pixels, TB DMAs per wave (item = wave + 8 t;
  // items past the last one repeat it, so that EVERY wave issues CNT DMAs per unit -- the
  // counted waits below rely on it); channel groups past the last one re-read it (their
  // columns are never flushed)
  long boff[kTBmax];
  unsigned bdst[kTBmax];
  bool bact[kTBmax];
#pragma unroll
  for (int t = 0; t < kTBmax; ++t) {
    const int i = min(wave + 8 * t, KGW * p.nJ - 1);
    const int kg = i / p.nJ, pj = i - kg * p.nJ;
    boff[t] = ((long)min(kg, kgs - 1) * p.planePix + 64 * pj + lane) * 8;
    bdst[t] = bytesA + (unsigned)(kg * p.Lpix + 64 * pj) * 16;
    bact[t] = 64 * pj + lane < p.Lwin;           // (the last piece of a window is never empty)
  }
  // unit u -> plane (n, z), unit of the plane
  int su = u0, spl = u0 / p.unitsPerPlane, squ = u0 - spl * p.unitsPerPlane;
  unsigned sbo = 0, cbo = 0;                     // byte offsets of the staging / computing ring slot
  const unsigned ringBytes = (unsigned)p.NS * bufBytes;
  auto stage = [&]() {                           // stages unit su into the next ring slot
    const int n = spl / p.Do, z = spl - n * p.Do;
    unsigned char* lb = lds + sbo;
    sbo += bufBytes; if (sbo == ringBytes) sbo = 0;
    const __bf16* ap = p.dyc + ((((long)n * p.Cout) * p.Do + z) * p.planeD + (long)squ * kKU);
#pragma unroll
    for (int c = 0; c < MB; ++c)
      __builtin_amdgcn_global_load_lds((gbl_vp)(ap + aoff[c]), (lds_vp)(lb + adst[c]), 16, 0, 0);
    const __bf16* xp = p.xcl + ((((long)n * p.Din + z + sz) * p.KGs + kg0) * p.planePix +
                                (long)squ * kKU + shiftLo) * 8;
#pragma unroll
    for (int t = 0; t < kTBmax; ++t)
      if (t < p.TB && bact[t])
        __builtin_amdgcn_global_load_lds((gbl_vp)(xp + boff[t]), (lds_vp)(lb + bdst[t]), 16, 0, 0);
    ++su;
    if (++squ == p.unitsPerPlane) { squ = 0; ++spl; }
  };
  // operand reads of step st of the stage at byte offset bo (inline asm: the kernel counts
  // its own LDS reads in flight, as conv_bf16.hip does for its filter rows)
  auto readA = [&](unsigned bo, int st, bf16x8 (&Ar)[MB]) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
      asm volatile("ds_read_b128 %0, %1" : "=v"(Ar[mb]) : "v"(aaddr[mb][st] + bo));
  };
  auto readB = [&](unsigned bo, int st, u32x2 (&Bl)[NB], u32x2 (&Bh)[NB]) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const unsigned a = baddr[nb] + bo + (unsigned)st * 256u;
      asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(Bl[nb]) : "v"(a));
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:64" : "=v"(Bh[nb]) : "v"(a));
    }
  };
  auto fma = [&](bf16x8 (&Ar)[MB], u32x2 (&Bl)[NB], u32x2 (&Bh)[NB]) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) asm volatile("" : "+v"(Ar[mb]));
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) { asm volatile("" : "+v"(Bl[nb])); asm volatile("" : "+v"(Bh[nb])); }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      union { u32x2 h[2]; bf16x8 v; } cvt;
      cvt.h[0] = Bl[nb]; cvt.h[1] = Bh[nb];
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ar[mb], cvt.v, acc[mb][nb], 0, 0, 0);
    }
  };
  constexpr int NST = kKU / 16;                  // 4 steps per unit
  bf16x8 A0[MB], A1[MB];
  u32x2 Bl0[NB], Bh0[NB], Bl1[NB], Bh1[NB];
  for (int s = 0; s < p.NS - 1 && su < u1; ++s) stage();
  for (int u = u0; u < u1; ++u) {
    // unit u has landed when only the stages issued after it are outstanding (the ring is
    // full in the steady state; the tail simply drains)
    switch ((su - u - 1) * CNT) {                // (MB + TB <= 6 instructions, <= 3 units ahead)
#define E2W(n) case n: asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); break;
      E2W(2) E2W(3) E2W(4) E2W(5) E2W(6) E2W(8) E2W(9) E2W(10) E2W(12) E2W(15) E2W(18)
#undef E2W
      default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();                // everyone's part of unit u; slot of unit u-1 is free
    asm volatile("" ::: "memory");               // (not __syncthreads: its fence drains the whole DMA ring)
    if (su < u1) stage();
    __builtin_amdgcn_sched_barrier(0);
    const unsigned bo = cbo;
    cbo += bufBytes; if (cbo == ringBytes) cbo = 0;
    readA(bo, 0, A0); readB(bo, 0, Bl0, Bh0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int st = 0; st < NST; st += 2) {
      // step st + 1 is read while step st computes, and so on
      readA(bo, st + 1, A1); readB(bo, st + 1, Bl1, Bh1);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(MB + 2 * NB) : "memory");
      __builtin_amdgcn_sched_barrier(0);
      fma(A0, Bl0, Bh0);
      __builtin_amdgcn_sched_barrier(0);
      if (st + 2 < NST) { readA(bo, st + 2, A0); readB(bo, st + 2, Bl0, Bh0); }
      __builtin_amdgcn_sched_barrier(0);
      if (st + 2 < NST) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(MB + 2 * NB) : "memory");
      else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      fma(A1, Bl1, Bh1);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- flush: D[row][col], col = lane % 32 (input channel), rows 8*(i/4) + 4*(lane/32) + i%4,
  // added into the one zeroed copy [oc][tap][ic] (32 lanes = 128 contiguous bytes) with f32
  // atomics.  (Measured against plain stores of one copy per split that the output pass adds:
  // the same 51 us on 200 -> 200 (1,3,3), 86 against 100 us on 40 -> 150 (2,4,4).)
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int tx = tx0 + nb;
    const int ic = kg0 * 8 + (ROWS ? 0 : wn * 32) + c32;
    if (tx >= p.kw || tyw >= p.kh || ic >= p.Cin) continue;
    const int tap = (dz * p.kh + tyw) * p.kw + tx;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int oc = M0 + (wm * MB + mb) * 32 + 8 * (i >> 2) + 4 * kh8 + (i & 3);
        if (oc < p.Cout) unsafeAtomicAdd(p.dwt + ((long)oc * p.T + tap) * p.CinP + ic, acc[mb][nb][i]);
      }
  }
}

// dw[oc][ic][tap] (+)= dwt[oc][tap][ic], through an LDS tile so that both sides are coalesced
__global__ __launch_bounds__(256) void wgrad_bf16_out_kernel(float* __restrict__ dwt, float* __restrict__ dw,
                                                             int Cin, int CinP, int T, int accumulate, int rezero) {
  extern __shared__ float tile[];                // [T][33]
  const int oc = blockIdx.y, ic0 = blockIdx.x * 32;
  const int nic = min(32, Cin - ic0);
  for (int i = threadIdx.x; i < T * 32; i += 256) {
    const int t = i >> 5, c = i & 31;
    tile[t * 33 + c] = dwt[((long)oc * T + t) * CinP + ic0 + c];
    if (rezero) dwt[((long)oc * T + t) * CinP + ic0 + c] = 0.f;   // (a caller-owned sum buffer stays zero between calls)
  }
  __syncthreads();
  float* o = dw + ((long)oc * Cin + ic0) * T;
  for (int i = threadIdx.x; i < nic * T; i += 256) {
    const int c = i / T, t = i - c * T;
    const float v = tile[t * 33 + c];
    o[i] = accumulate ? o[i] + v : v;
  }
}

static int view_ok(const e2_tensor5* t, const char* name) {
  E2_REQUIRE(t && t->ptr, "%s: null tensor", name);
  E2_REQUIRE(t->n > 0 && t->c > 0 && t->d > 0 && t->h > 0 && t->w > 0,
             "%s: empty tensor (%d,%d,%d,%d,%d)", name, t->n, t->c, t->d, t->h, t->w);
  return 0;
}

struct Geo { long planePix, xPieces, xSlack, dTotal; int planeD, KG, CinP; size_t xbytes, dbytes, tbytes; };
static Geo geo(int n, int cin, int d, int h, int w, int cout, int kd, int kh, int kw) {
  Geo g;
  const int Do = d - kd + 1, Ho = h - kh + 1;
  g.KG = (cin + 7) / 8;
  g.planePix = (long)h * w;
  g.planeD = (Ho * w + kKU - 1) / kKU * kKU;
  g.xPieces = (long)n * d * g.KG * g.planePix;
  g.xSlack = g.planeD + (long)(kh - 1) * w + kw + 128;      // window reads past the last plane
  g.dTotal = (long)n * cout * Do * g.planeD;
  g.CinP = (cin + 31) / 32 * 32;
  g.xbytes = (((size_t)(g.xPieces + g.xSlack) * 16 + 255) / 256) * 256;
  g.dbytes = (((size_t)g.dTotal * 2 + 2048 + 255) / 256) * 256;     // (+ the swizzled row reads of the last unit)
  g.tbytes = (size_t)cout * kd * kh * kw * g.CinP * 4;
  return g;
}

template <int MB, int NB, bool ROWS>
static int launch(e2_ctx* ctx, const WbP& p, long grid, size_t ldsb) {
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_bf16_kernel<MB, NB, ROWS>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) { e2_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return 1; }
    attr_done = true;
  }
  hipLaunchKernelGGL((wgrad_bf16_kernel<MB, NB, ROWS>), dim3((unsigned)grid), dim3(512), ldsb, ctx->stream, p);
  E2_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace

extern "C" size_t e2_conv3d_wgrad_bf16_workspace_bytes(int n, int cin, int d, int h, int w, int cout,
                                                       int kd, int kh, int kw) {
  const Geo g = geo(n, cin, d, h, w, cout, kd, kh, kw);
  return g.xbytes + g.dbytes + g.tbytes + 256;
}

// xcl_ext != nullptr: the channels-last bf16 copy of x already exists (written by the layer's
// forward pass, e2_conv3d_fwd_bf16_keep: the same layout with kgs_ext channel groups per
// plane and zero pixels behind the last plane) -- x is used for its shape only
// dyc_ext != nullptr: the channel-major bf16 planes of dy already exist (written by the kernel that
// produced dy, e2_pool_bias_act_bwd_bf16: [n][oc][z][planeD] at the input's row pitch, zeros in
// the gaps and 2 KB behind the end) -- dy is used for its shape only.  dwt_ext: the f32 sums
// live in the caller's buffer (e2_conv3d_wgrad_bf16_sums_bytes), ZERO on entry and zero again on
// return.  With all three the conversion pass disappears and ws may be NULL.
static int wgrad_bf16(e2_ctx* ctx, const e2_tensor5* x, const void* xcl_ext, int kgs_ext,
                      const e2_tensor5* dy, float* dw, int kd, int kh, int kw, int accumulate,
                      void* ws, size_t ws_bytes, const void* dyc_ext = nullptr, float* dwt_ext = nullptr) {
  E2_REQUIRE(ctx && dw, "conv3d_wgrad_bf16: null argument");
  if (xcl_ext) {
    E2_REQUIRE(x && x->n > 0 && x->c > 0 && x->d > 0 && x->h > 0 && x->w > 0, "conv3d_wgrad_bf16_xcl: bad x shape");
    E2_REQUIRE(kgs_ext >= (x->c + 7) / 8 && ((uintptr_t)xcl_ext & 15) == 0,
               "conv3d_wgrad_bf16_xcl: %d channel groups per plane for %d channels", kgs_ext, x->c);
  } else if (int rc = view_ok(x, "conv3d_wgrad_bf16 x")) return rc;
  if (dyc_ext) {
    E2_REQUIRE(dy && dy->n > 0 && dy->c > 0 && dy->d > 0 && dy->h > 0 && dy->w > 0 && ((uintptr_t)dyc_ext & 15) == 0,
               "conv3d_wgrad_bf16: bad dy shape / unaligned planes");
  } else if (int rc = view_ok(dy, "conv3d_wgrad_bf16 dy")) return rc;
  E2_REQUIRE(dy->n == x->n && dy->d == x->d - kd + 1 && dy->h == x->h - kh + 1 &&
                 dy->w == x->w - kw + 1,
             "conv3d_wgrad_bf16: dy spatial (%d,%d,%d) != x (%d,%d,%d) - k + 1", dy->d, dy->h,
             dy->w, x->d, x->h, x->w);
  const int Cin = x->c, Cout = dy->c, T = kd * kh * kw;
  const Geo g = geo(x->n, Cin, x->d, x->h, x->w, Cout, kd, kh, kw);
  const bool all_ext = xcl_ext && dyc_ext && dwt_ext;
  const size_t need = all_ext ? 0 : g.xbytes + g.dbytes + g.tbytes;
  E2_REQUIRE(all_ext || (ws && ws_bytes >= need), "conv3d_wgrad_bf16: workspace of %zu bytes needed, %zu given",
             need, ws_bytes);
  E2_REQUIRE(all_ext || ((uintptr_t)ws & 15) == 0, "conv3d_wgrad_bf16: workspace must be 16-byte aligned");
  E2_REQUIRE(((uintptr_t)dwt_ext & 15) == 0, "conv3d_wgrad_bf16: the sum buffer must be 16-byte aligned");
  int MB = 2, NB = 0, S = 0, rows = Cin <= 64 ? 1 : 0;
  {
    int v[5];
    if (sscanf(ctx->tiling[E2_TILING_WGRAD], "%d,%d,%d,%d,%d", &v[0], &v[1], &v[2], &v[3], &v[4]) == 5 &&
        v[0] == 32) { MB = v[1]; NB = v[2]; rows = v[3] != 0; S = v[4]; }
  }
  if (NB <= 0) NB = kw >= 4 ? 4 : (kw == 3 ? 3 : kw);        // taps of a kernel row per wave
  E2_REQUIRE((MB == 1 || MB == 2) && NB >= 1 && NB <= 4, "conv3d_wgrad_bf16: MB 1..2, NB 1..4");
  __bf16* xcl = xcl_ext ? const_cast<__bf16*>(reinterpret_cast<const __bf16*>(xcl_ext))
                        : reinterpret_cast<__bf16*>(ws);
  __bf16* dyc = dyc_ext ? const_cast<__bf16*>(reinterpret_cast<const __bf16*>(dyc_ext))
                        : reinterpret_cast<__bf16*>((char*)ws + g.xbytes);
  float* dwt = dwt_ext ? dwt_ext : reinterpret_cast<float*>((char*)ws + g.xbytes + g.dbytes);
  WcP c;
  c.x = x->ptr; c.xsN = x->sn; c.xsC = x->sc; c.xsZ = x->sd; c.xsY = x->sh;
  c.dy = dy->ptr; c.dsN = dy->sn; c.dsC = dy->sc; c.dsZ = dy->sd; c.dsY = dy->sh;
  c.N = x->n; c.Cin = Cin; c.Cout = Cout; c.Din = x->d; c.Hin = x->h; c.Win = x->w;
  c.Do = dy->d; c.Ho = dy->h; c.Wo = dy->w; c.KG = g.KG; c.planeD = g.planeD;
  c.planePix = (int)g.planePix; c.xPieces = g.xPieces; c.xSlack = g.xSlack;
  c.xcl = xcl; c.dyc = dyc; c.dwt = dwt; c.nT4 = (long)(g.tbytes / 16);
  c.cx = (int)((g.planePix + 255) / 256);
  c.cd = (g.planeD + 2047) / 2048;
  const long nbx = xcl_ext ? 0 : (long)x->n * x->d * g.KG * c.cx;
  const long nbd = dyc_ext ? 0 : (long)x->n * Cout * dy->d * c.cd;
  const long nbs = xcl_ext ? 0 : (g.xSlack + 255) / 256, nbt = dwt_ext ? 0 : (c.nT4 + 255) / 256;
  E2_REQUIRE(g.planePix < (1L << 30) && nbx + nbd + nbs + nbt < (1L << 31), "conv3d_wgrad_bf16: volume too large");
  c.nbx = (int)nbx; c.nbd = (int)nbd; c.nbs = (int)nbs;
  if (nbx + nbd + nbs + nbt > 0) {
    hipLaunchKernelGGL(wgrad_bf16_cvt_kernel, dim3((unsigned)(nbx + nbd + nbs + nbt)), dim3(256), 0, ctx->stream, c);
    E2_CHECK_HIP(hipGetLastError());
  }
  WbP p;
  p.xcl = xcl; p.dyc = dyc; p.dwt = dwt; p.CinP = g.CinP;
  p.N = x->n; p.Cin = Cin; p.Cout = Cout; p.Din = x->d; p.Do = dy->d; p.KG = g.KG;
  p.KGs = xcl_ext ? kgs_ext : g.KG;
  p.kd = kd; p.kh = kh; p.kw = kw; p.T = T;
  p.Win = x->w; p.planeD = g.planeD; p.planePix = g.planePix;
  p.unitsPerPlane = g.planeD / kKU;
  p.units = x->n * dy->d * p.unitsPerPlane;
  p.nMT = (Cout + 64 * MB - 1) / (64 * MB);
  const int KGW = rows ? 4 : kKGW;
  p.nIT = (g.KG + KGW - 1) / KGW;
  p.nTG = (rows ? (kh + 3) / 4 : kh) * ((kw + NB - 1) / NB);
  p.Lwin = kKU + (rows ? std::min(kh - 1, 3) * x->w : 0) + NB - 1;
  p.Lpix = (p.Lwin + 3) / 16 * 16 + 12;                  // smallest >= Lwin that is 12 mod 16
  p.nJ = (p.Lwin + 63) / 64;
  p.TB = (KGW * p.nJ + 7) / 8;
  E2_REQUIRE(p.TB <= kTBmax, "conv3d_wgrad_bf16: rows of %d pixels are too long for the row form", x->w);
  const size_t bufb = (size_t)64 * MB * kKU * 2 + (size_t)KGW * p.Lpix * 16;
  p.NS = (int)std::min<size_t>(4, (160 * 1024) / bufb);
  E2_REQUIRE(p.NS >= 2, "conv3d_wgrad_bf16: a stage of %zu B does not fit LDS twice", bufb);
  const size_t ldsb = p.NS * bufb;
  const long tiles = (long)p.nMT * p.nIT * kd * p.nTG;
  if (S <= 0) S = (int)std::max<long>(1, ctx->num_cu / tiles);      // at most one work-group per CU
  S = std::min(S, p.units);
  p.per = (p.units + S - 1) / S;
  S = (p.units + p.per - 1) / p.per;
  E2_REQUIRE(tiles * S < (1L << 30), "conv3d_wgrad_bf16: grid too large");
  p.nWG = (int)(tiles * S);
  p.perXcd = (p.nWG + 7) / 8;
  const long grid = 8L * p.perXcd;                                 // (work-groups past the last one exit)
  E2_REQUIRE(grid < (1L << 31), "conv3d_wgrad_bf16: grid too large");
  E2_REQUIRE((size_t)T * 33 * 4 <= 64 * 1024, "conv3d_wgrad_bf16: %d taps exceed the output pass's LDS tile", T);
  {
    int v[5];
    const bool forced = sscanf(ctx->tiling[E2_TILING_WGRAD], "%d,%d,%d,%d,%d", &v[0], &v[1], &v[2], &v[3], &v[4]) == 5 && v[0] == 32;
    e2_note_launch(ctx, "wgrad_bf16", forced ? E2_SRC_FORCED : (ctx->tiling[E2_TILING_WGRAD][0] ? E2_SRC_FALLBACK : E2_SRC_MODEL),
                   "32,%d,%d,%d,%d", MB, NB, rows ? 1 : 0, S);
  }
  int rc = 2;
#define E2_L(M, N_) if (MB == M && NB == N_) rc = rows ? launch<M, N_, true>(ctx, p, grid, ldsb) : launch<M, N_, false>(ctx, p, grid, ldsb);
  E2_L(1, 1) E2_L(1, 2) E2_L(1, 3) E2_L(1, 4) E2_L(2, 1) E2_L(2, 2) E2_L(2, 3) E2_L(2, 4)
#undef E2_L
  if (rc) return rc;
  hipLaunchKernelGGL(wgrad_bf16_out_kernel, dim3((unsigned)(g.CinP / 32), (unsigned)Cout), dim3(256),
                     (size_t)T * 33 * 4, ctx->stream, dwt, dw, Cin, g.CinP, T, accumulate, dwt_ext ? 1 : 0);
  E2_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int e2_conv3d_wgrad_bf16(e2_ctx* ctx, const e2_tensor5* x, const e2_tensor5* dy, float* dw,
                                    int kd, int kh, int kw, int accumulate, void* ws,
                                    size_t ws_bytes) {
  return wgrad_bf16(ctx, x, nullptr, 0, dy, dw, kd, kh, kw, accumulate, ws, ws_bytes);
}

extern "C" int e2_conv3d_wgrad_bf16_xcl(e2_ctx* ctx, const e2_tensor5* x_shape, const void* xcl,
                                        int kg_per_plane, const e2_tensor5* dy, float* dw, int kd,
                                        int kh, int kw, int accumulate, void* ws, size_t ws_bytes) {
  E2_REQUIRE(xcl, "conv3d_wgrad_bf16_xcl: null copy");
  return wgrad_bf16(ctx, x_shape, xcl, kg_per_plane, dy, dw, kd, kh, kw, accumulate, ws, ws_bytes);
}

/* geometry of the ready-made operands of e2_conv3d_wgrad_bf16_ex */
extern "C" int e2_conv3d_wgrad_bf16_geometry(int n, int cin, int d, int h, int w, int cout, int kd,
                                             int kh, int kw, int64_t* plane_d, size_t* dyc_bytes,
                                             size_t* sums_bytes) {
  const Geo g = geo(n, cin, d, h, w, cout, kd, kh, kw);
  if (plane_d) *plane_d = g.planeD;
  if (dyc_bytes) *dyc_bytes = g.dbytes;
  if (sums_bytes) *sums_bytes = (g.tbytes + 255) / 256 * 256;
  return 0;
}

extern "C" int e2_conv3d_wgrad_bf16_ex(e2_ctx* ctx, const e2_tensor5* x_shape, const void* xcl,
                                       int kg_per_plane, const e2_tensor5* dy, const void* dyc,
                                       float* sums, float* dw, int kd, int kh, int kw,
                                       int accumulate, void* ws, size_t ws_bytes) {
  return wgrad_bf16(ctx, x_shape, xcl, kg_per_plane, dy, dw, kd, kh, kw, accumulate, ws, ws_bytes, dyc, sums);
}
