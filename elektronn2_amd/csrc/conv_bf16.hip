// conv_bf16.hip -- conv forward / data gradient with bf16 operands IN MEMORY and LDS-staged
// input windows, built for the 32-cycle v_mfma_f32_32x32x16_bf16.
//
// Two conversion passes per call write
//   Xb[n][z][kg][y][x][8]   the input as bf16: for one z-plane and one group of 8 channels the
//                           pixels are consecutive 16-byte pieces (a "pixel" below)
//   Wb[dz][slot][kg][oc][8] the filter rows as bf16; the taps of one dz fill the first
//                           kh*kw*KG/2 "steps", the slab is padded with ZERO steps to a
//                           multiple of the operand ring depth
// and the GEMM kernel (4 waves = 2 along the channels x 2 along the positions, each with
// MB x NB blocks of 32 x 32):
//   * per dz, the work-group's input window -- ONE contiguous span of pixels per channel
//     group, as in igemm_core.hpp, with 16-byte elements -- is copied into LDS by LDS-DMA
//     (double buffered over dz); every tap reads it at a shifted pixel offset
//     (ds_read_b128, 512 consecutive bytes per half-wave), so the input leaves L2 once per
//     work-group instead of once per wave and tap (DESIGN.md finding 19: the LDS-free
//     form was bound by exactly that traffic);
//   * the filter rows come straight from L2, 16 bytes per lane, PD steps ahead in a ring
//     of registers whose steady-state loop has no branch (the compiler then counts the
//     loads in flight instead of draining them).
// Arithmetic: operands rounded to bf16 (nearest even) as in the operand-rounding form of
// igemm_core.hpp, f32 products and sums: the bounds of tests/test_bf16_gpu.py hold
// unchanged.  Output: f32, any strided view, optional fused bias + activation.
// Limits: all channels of one dz must fit the LDS buffer (K <= ~400 channels for the
// neuro3d shapes); the caller falls back to the operand-rounding form otherwise.
#include "common.hpp"
#include <algorithm>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lds_vp;
typedef const __attribute__((address_space(1))) void* gbl_vp;

namespace {

// a pointer the compiler should keep in SGPRs (wave-uniform by construction)
__device__ __forceinline__ const void* e2b_uniform(const void* q) {
  const unsigned long long v = (unsigned long long)(uintptr_t)q;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (const void*)(uintptr_t)(((unsigned long long)hi << 32) | lo);
}

constexpr int kPD = 4;        // filter-row steps in flight per wave

struct CbP {
  const __bf16* xb;       // [N][Din][KG][Hin][Win][8]
  const __bf16* wb;       // [kd][stepsP][2][ocP][8]   (step = (tap, channel-group pair))
  float* out;
  long osN, osC, osZ, osY;
  const float* bias;      // fused bias (+ act) or nullptr
  int act;
  int N, Cout, KG, ocP;
  int kd, kh, kw;
  int Do, Ho, Wo, Q;
  int Din, Hin, Win;
  int nPT, nMT;
  int Lpad;               // pixels per channel group in an LDS buffer
  // a kernel plane's channel groups are staged in chunks of KGC (the last one kgsLast); the
  // steps of a chunk -- (tap, pair of its channel groups) -- are padded to FL / LL (multiples
  // of the ring depth) with zero filter rows
  int KGC, nck, kgsLast, FL, LL;
  // the epilogue's second output (forward with fused bias + act): the channels-last bf16 copy
  // of `out` that the NEXT layer's kernels read, [N][Do][nxKG][Ho * Wo][8] -- or nullptr
  __bf16* nxb;
  int nxKG;
};

// ---- f32 (strided NCDHW view) -> Xb[n][z][kg][y][x][8] --------------------------------
struct CvP {
  const float* x;
  long sN, sC, sZ, sY;
  int N, C, D, H, W, KG;
  __bf16* xb;
};
// thread = (channel group, pixel): 8 coalesced row reads, one 16-byte store
__device__ __forceinline__ void cvt_bf16_part(const CvP& p, int blk, int nblk) {
  const long total = (long)p.N * p.D * p.KG * p.H * p.W;
  for (long i = blk * 256L + threadIdx.x; i < total; i += (long)nblk * 256) {
    const int x = (int)(i % p.W);
    long r = i / p.W;
    const int y = (int)(r % p.H);   r /= p.H;
    const int kg = (int)(r % p.KG); r /= p.KG;
    const int z = (int)(r % p.D);
    const int n = (int)(r / p.D);
    const float* src = p.x + (long)n * p.sN + (long)z * p.sZ + (long)y * p.sY + x;
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = kg * 8 + j;
      v[j] = (__bf16)(c < p.C ? src[(long)c * p.sC] : 0.f);
    }
    *reinterpret_cast<bf16x8*>(p.xb + i * 8) = v;
  }
}

// ---- canonical f32 weights (n_f, n_in, kd, kh, kw) -> Wb[dz][step][half][rows][8] ----------
// step = tap2d * KC + kc (kc = pair of channel groups), half = which group of the pair;
// steps >= T2 * KC are zero.  mode 0 (forward): rows = out channels, k = in channels, taps
// flipped (true convolution); mode 1 (data gradient): rows = in channels, k = out
// channels, taps as they are.
struct PwP {
  const float* w;
  __bf16* wb;
  int nf, nin, kd, T2, rowsP, mode, KGC, nck, kgsLast, FL, LL;
};
__device__ __forceinline__ void pack_w_bf16_part(const PwP& q, int blk, int nblk) {
  const float* __restrict__ w = q.w;
  __bf16* __restrict__ wb = q.wb;
  const int nf = q.nf, nin = q.nin, kd = q.kd, T2 = q.T2, rowsP = q.rowsP, mode = q.mode;
  const int KGC = q.KGC, nck = q.nck, kgsLast = q.kgsLast, FL = q.FL, LL = q.LL;
  const int DL = (nck - 1) * FL + LL;            // steps of one kernel plane
  // (+ kPD zero steps behind the last plane: the ring's prefetches past the end)
  const long total = ((long)kd * DL + kPD) * 2 * rowsP * 8;
  const int rows = mode ? nin : nf, kk = mode ? nf : nin;
  const int T = kd * T2;
  // a thread writes one 16-byte piece (8 consecutive k of one row): the index arithmetic -- a
  // dozen integer divisions -- once per piece, not per element (the per-element form made the
  // plan's one-launch pack of all layers ALU-bound: 14 / 22 us for neuro3d_lite / neuro3d)
  for (long i8 = blk * 256L + threadIdx.x; i8 < (total >> 3); i8 += (long)nblk * 256) {
    long r1 = i8;
    const int r = (int)(r1 % rowsP); r1 /= rowsP;
    const int half = (int)(r1 & 1);   r1 >>= 1;
    const int dz = (int)(r1 / DL);
    const int rem = (int)(r1 - (long)dz * DL);
    const int c = min(rem / FL, nck - 1);
    const int step = rem - c * FL;
    const int KCc = (c == nck - 1 ? kgsLast : KGC) >> 1;
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (__bf16)0.f;
    if (dz < kd && step < T2 * KCc && r < rows) {
      const int tap2 = step / KCc, kc = step - tap2 * KCc;
      const int k0 = (c * KGC + 2 * kc + half) * 8;
      const int tap = dz * T2 + tap2;
      const int tsrc = mode ? tap : (T - 1 - tap);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = k0 + j;
        if (k < kk) {
          const int oc = mode ? k : r, ic = mode ? r : k;
          v[j] = (__bf16)w[((long)oc * nin + ic) * T + tsrc];
        }
      }
    }
    *reinterpret_cast<bf16x8*>(wb + i8 * 8) = v;
  }
}

// both conversion passes in ONE launch (they are tiny; a launch apiece cost more than the work)
__global__ __launch_bounds__(256) void prep_bf16_kernel(CvP c, PwP q, int cblocks) {
  if ((int)blockIdx.x < cblocks) cvt_bf16_part(c, blockIdx.x, cblocks);
  else pack_w_bf16_part(q, blockIdx.x - cblocks, gridDim.x - cblocks);
}

// the filter rows of MANY launches in one (the plan packs every layer's images at the start of
// the step: e2_conv3d_bf16_pack_w_multi); blockIdx.y = job
__global__ __launch_bounds__(256) void pack_w_bf16_multi_kernel(const PwP* __restrict__ jobs) {
  const PwP q = jobs[blockIdx.y];
  pack_w_bf16_part(q, blockIdx.x, gridDim.x);
}

// ---- the GEMM -----------------------------------------------------------------------
template <int MB, int NB>
__global__ __launch_bounds__(256) void conv_bf16_kernel(CbP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int PD = kPD;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int c32 = lane & 31, kh8 = lane >> 5;

  // all channel tiles of one (n, z, position tile) unit on ONE XCD (blocks are dealt
  // round-robin over the 8 XCDs): its input window is fetched into one L2
  const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
  const int mt = jb % p.nMT;
  int u = (jb / p.nMT) * 8 + xcd;
  if (u >= p.N * p.Do * p.nPT) return;
  const int pt = u % p.nPT; u /= p.nPT;
  const int z = u % p.Do;
  const int n = u / p.Do;
  const int m0 = (mt * 2 + wm) * (32 * MB);
  const bool idle = m0 >= p.ocP;                // (a wave beyond the padded channels still stages)

  // the work-group's window: one span of pixels per channel group
  const int Q0 = pt * (64 * NB);
  const int qlast = min(Q0 + 64 * NB, p.Q) - 1;
  const int r0 = Q0 / p.Wo, c0 = Q0 - r0 * p.Wo;
  const int rl = qlast / p.Wo, cl = qlast - rl * p.Wo;
  const int L = (rl - r0) * p.Win + (cl - c0) + (p.kh - 1) * p.Win + p.kw;     // pixels
  const long planePix = (long)p.Hin * p.Win;
  const long span_lo = (long)r0 * p.Win + c0;
  const int Lpad = p.Lpad;
  const unsigned bufBytes = (unsigned)p.KGC * Lpad * 16;

  // stage chunk ci = (dz, range of channel groups) of the window into buffer buf: (channel
  // group, 64-pixel piece) pairs dealt over the 4 waves; LDS-DMA, 16 bytes per lane, lanes
  // past the span masked
  const int nJ = (L + 63) >> 6;
  const int nChunks = p.kd * p.nck;
  auto stage = [&](int ci, int buf) {
    const int dz = ci / p.nck, c = ci - dz * p.nck;
    const int kgs = (c == p.nck - 1) ? p.kgsLast : p.KGC;
    const __bf16* xp = p.xb + ((((long)n * p.Din + z + dz) * p.KG + c * p.KGC) * planePix + span_lo) * 8;
    unsigned char* lb = lds + buf * bufBytes;
    const int pieces = kgs * nJ;
    for (int i = wave; i < pieces; i += 4) {
      const int kg = i / nJ, j = i - kg * nJ;
      const int v = 64 * j + lane;
      if (v < L)
        __builtin_amdgcn_global_load_lds((gbl_vp)(xp + ((long)kg * planePix + v) * 8),
                                         (lds_vp)(lb + ((unsigned)kg * Lpad + 64 * j) * 16), 16, 0, 0);
    }
  };

  // per-lane offsets: filter rows (elements inside a step), window pixels (inside a buffer)
  unsigned aoff[MB], boff[NB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
    aoff[mb] = (unsigned)((kh8 * p.ocP + min(m0 + mb * 32 + c32, p.ocP - 1)) * 16);   // bytes
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int q = min(Q0 + (wn * NB + nb) * 32 + c32, p.Q - 1);
    const int y = q / p.Wo, x = q - y * p.Wo;
    boff[nb] = (unsigned)(((y - r0) * p.Win + (x - c0)) + kh8 * Lpad) * 16u;
  }
  f32x16 acc[MB][NB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mb][nb][i] = 0.f;

  const long stepAb = 2L * p.ocP * 16;          // bytes of one step's filter rows
  bf16x8 A[PD][MB];
  // filter-row loads are inline asm: hipcc's wait-count pass drained ALL loads at the loop
  // header (s_waitcnt vmcnt(0)) whatever the source order, so the ring was never in flight;
  // the kernel retires them with its own counted waits (cdna guide 5.7 form iii), and keeps
  // every destination allocated until then (touch).  Scalar base (one 64-bit add per step)
  // + per-lane byte offset: no per-step vector address arithmetic.  Wb carries PD zero
  // steps behind the last dz, so the prefetches past the end need no clamp.
  const char* abase = reinterpret_cast<const char*>(p.wb);
  auto loadA = [&](bf16x8 (&Ar)[MB]) {
    const char* sb = reinterpret_cast<const char*>(e2b_uniform(abase));
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
      asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(Ar[mb]) : "v"(aoff[mb]), "s"(sb));
    abase += stepAb;
  };
  auto touchA = [&](bf16x8 (&Ar)[MB]) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) asm volatile("" : "+v"(Ar[mb]));
  };
  // the oldest MB loads have landed when at most (PD - 1) * MB are outstanding
#define E2B_WAIT_OLDEST()                                                        \
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PD - 1) * MB) : "memory");           \
  __builtin_amdgcn_sched_barrier(0);
  // the window at a step's tap shift and channel-group pair.  The byte shift advances
  // incrementally (a few scalar adds / selects per step): decoding (tap, channel pair) from
  // the step index cost two integer divisions -- more issue slots than a step's MFMAs
  // leave -- and a shift table costs a scalar-load round trip per step (both measured).
  // Zero steps past the last real one read the first pixels again (finite values times
  // zero weights).
  const unsigned dK = 2u * (unsigned)Lpad * 16u;                 // next channel-group pair
  const unsigned dY = (unsigned)(p.Win - p.kw) * 16u;            // next tap row
  unsigned sh = 0, dX = 0;
  int cK = 0, cX = 0, cS = 0, KCc = 1, realSteps = 0;            // state of the NEXT read
  auto resetB = [&](int kgs) {
    KCc = kgs >> 1;
    realSteps = p.kh * p.kw * KCc;
    dX = 16u - (unsigned)KCc * dK;                                // next tap in the row
    sh = 0; cK = 0; cX = 0; cS = 0;
  };
  auto readB = [&](unsigned bufb, bf16x8 (&Br)[NB]) {
    const unsigned a = bufb + (cS < realSteps ? sh : 0u);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
      Br[nb] = *reinterpret_cast<const bf16x8*>(lds + a + boff[nb]);
    ++cS;
    sh += dK;
    if (++cK == KCc) {
      cK = 0; sh += dX;
      if (++cX == p.kw) { cX = 0; sh += dY; }
    }
  };
  auto fma = [&](const bf16x8 (&Ar)[MB], const bf16x8 (&Br)[NB]) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ar[mb], Br[nb], acc[mb][nb], 0, 0, 0);
  };

  stage(0, 0);
#pragma unroll
  for (int j = 0; j < PD; ++j) loadA(A[j]);
  bf16x8 B0[NB], B1[NB];
  for (int ci = 0; ci < nChunks; ++ci) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                            // chunk ci landed; the other buffer is free
    if (ci + 1 < nChunks) stage(ci + 1, (ci + 1) & 1);
    if (idle) continue;
    const unsigned bufb = (ci & 1) * bufBytes;
    const bool lastc = (ci % p.nck) == p.nck - 1;
    const int nsteps = lastc ? p.LL : p.FL;
    resetB(lastc ? p.kgsLast : p.KGC);
    readB(bufb, B0);
    // (nsteps is a multiple of PD, PD is even: no branch inside; the order below is pinned
    // -- next step's window read, this step's MFMAs, the filter rows PD steps ahead -- so
    // that the compiler's waits count the loads in flight instead of draining them)
    for (int st = 0; st < nsteps; st += PD) {
#pragma unroll
      for (int j = 0; j < PD; j += 2) {
        readB(bufb, B1);
        __builtin_amdgcn_sched_barrier(0);
        E2B_WAIT_OLDEST()
        touchA(A[j]);
        fma(A[j], B0);
        __builtin_amdgcn_sched_barrier(0);
        loadA(A[j]);
        __builtin_amdgcn_sched_barrier(0);
        readB(bufb, B0);
        __builtin_amdgcn_sched_barrier(0);
        E2B_WAIT_OLDEST()
        touchA(A[j + 1]);
        fma(A[j + 1], B1);
        __builtin_amdgcn_sched_barrier(0);
        loadA(A[j + 1]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (prefetches past the last step)
#pragma unroll
  for (int j = 0; j < PD; ++j) touchA(A[j]);
#undef E2B_WAIT_OLDEST
  if (idle) return;

  // ---- epilogue: D[row][col], col = lane % 32 (position), rows 8*(i/4) + 4*(lane/32) + i%4
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int q = Q0 + (wn * NB + nb) * 32 + c32;
    if (q >= p.Q) continue;
    const int y = q / p.Wo, x = q - y * p.Wo;
    float* ob = p.out + (long)n * p.osN + (long)z * p.osZ + (long)y * p.osY + x;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int co = m0 + mb * 32 + 8 * (i >> 2) + 4 * kh8 + (i & 3);
        float v = acc[mb][nb][i];
        if (co < p.Cout) {
          if (p.bias) {
            v += p.bias[co];
            if (p.act == E2_ACT_RELU) v = (v > 0.f) ? v : ((v == 0.f) ? 0.f : -0.f);
          }
          ob[(long)co * p.osC] = v;
        } else {
          v = 0.f;                                  // (padding channels of the next layer's copy)
        }
        acc[mb][nb][i] = v;
      }
    if (p.nxb) {
      // lanes (c32, kh8 = 0 / 1) hold channels 8g + 0..3 / 8g + 4..7 of position q: the two
      // 8-byte halves of one pixel piece
      __bf16* nb0 = p.nxb + ((((long)n * p.Do + z) * p.nxKG) * p.Q + q) * 8 + 4 * kh8;
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int kg = ((m0 + mb * 32) >> 3) + g;
          if (kg >= p.nxKG) continue;
          bf16x4 h;
#pragma unroll
          for (int e = 0; e < 4; ++e) h[e] = (__bf16)acc[mb][nb][4 * g + e];
          *reinterpret_cast<bf16x4*>(nb0 + (long)kg * p.Q * 8) = h;
        }
    }
  }
}

static int view_ok(const e2_tensor5* t, const char* name) {
  E2_REQUIRE(t && t->ptr, "%s: null tensor", name);
  E2_REQUIRE(t->n > 0 && t->c > 0 && t->d > 0 && t->h > 0 && t->w > 0,
             "%s: empty tensor (%d,%d,%d,%d,%d)", name, t->n, t->c, t->d, t->h, t->w);
  return 0;
}
static int pad16(int v) { return (v + 15) / 16 * 16; }

template <int MB, int NB>
static int launch(e2_ctx* ctx, const CbP& p, int grid, size_t ldsb) {
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16_kernel<MB, NB>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) { e2_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return 1; }
    attr_done = true;
  }
  hipLaunchKernelGGL((conv_bf16_kernel<MB, NB>), dim3(grid), dim3(256), ldsb, ctx->stream, p);
  E2_CHECK_HIP(hipGetLastError());
  return 0;
}

// upper bound of the window span (pixels) of a tile of BN positions
static int span_pixels(int BN, int Wo, int Win, int kh, int kw) {
  const int rows = (BN - 1 + Wo - 1) / Wo + 1;                 // image rows a tile can touch
  return (BN - 1) + rows * (Win - Wo) + (kh - 1) * Win + kw;
}

// the geometry of one launch: a function of the GEMM's shape and tile only (the plan packs the
// filter rows ahead of the launch with the same numbers, e2_bf16_wjob_fill)
struct CbGeo {
  int Cp, KG, ocP, T2, Lpad, KGC, nck, kgsLast, FL, LL, DL;
  size_t bufBytes, ldsb, wb_bytes;
  bool ok;
};
static CbGeo cb_geo(int rows, int kk, int kd, int kh, int kw, int in_w, int out_w, int MB, int NB) {
  CbGeo g;
  g.ok = false;
  g.Cp = pad16(kk); g.KG = g.Cp / 8;
  g.ocP = (rows + 32 * MB - 1) / (32 * MB) * (32 * MB);          // whole wave tiles
  g.T2 = kh * kw;
  g.Lpad = (span_pixels(64 * NB, out_w, in_w, kh, kw) + 3) / 4 * 4;
  // channel groups per LDS chunk: two buffers within ~72 KB, so that two work-groups share
  // a CU; an even count, as equal as possible over the chunks
  g.KGC = (int)std::min<long>(g.KG, (36 * 1024) / ((long)g.Lpad * 16) / 2 * 2);
  if (g.KGC < 2) return g;
  g.nck = (g.KG + g.KGC - 1) / g.KGC;
  g.KGC = ((g.KG / 2 + g.nck - 1) / g.nck) * 2;                  // balance the chunks
  g.nck = (g.KG + g.KGC - 1) / g.KGC;
  g.kgsLast = g.KG - (g.nck - 1) * g.KGC;
  g.FL = (g.T2 * (g.KGC / 2) + kPD - 1) / kPD * kPD;
  g.LL = (g.T2 * (g.kgsLast / 2) + kPD - 1) / kPD * kPD;
  g.DL = (g.nck - 1) * g.FL + g.LL;
  g.bufBytes = (size_t)g.KGC * g.Lpad * 16;
  g.ldsb = (kd * g.nck > 1 ? 2 : 1) * g.bufBytes;
  g.wb_bytes = ((size_t)kd * g.DL + kPD) * 2 * g.ocP * 8 * 2;
  g.ok = true;
  return g;
}

// what a caller may hand over ready-made (all optional)
struct CbExt {
  const void* xb = nullptr;      // the channels-last bf16 copy of `in` (its producer wrote it)
  const void* wb = nullptr;      // the filter rows (e2_conv3d_bf16_pack_w_multi, same tile)
  void* nxb = nullptr;           // forward: receives the channels-last bf16 copy of `out`
  int nxKG = 0;
};

// in: input view (forward: x; data gradient: the zero-padded dy); rows = output channels of
// the GEMM (forward: n_f; data gradient: n_in), kk = its reduction channels
static int conv_bf16(e2_ctx* ctx, const e2_tensor5* in, const float* w, int nf, int nin, int kd,
                     int kh, int kw, int mode, const float* bias, int act, const e2_tensor5* out,
                     void* ws, size_t ws_bytes, int MB, int NB, void* xkeep = nullptr,
                     size_t xkeep_bytes = 0, const CbExt& ext = CbExt()) {
  const int rows = mode ? nin : nf, kk = mode ? nf : nin;
  E2_REQUIRE(in->c == kk && out->c == rows, "conv_bf16: channel mismatch");
  E2_REQUIRE(out->n == in->n && out->d == in->d - kd + 1 && out->h == in->h - kh + 1 &&
                 out->w == in->w - kw + 1, "conv_bf16: out shape does not match in - k + 1");
  const CbGeo g = cb_geo(rows, kk, kd, kh, kw, in->w, out->w, MB, NB);
  E2_REQUIRE(g.ok, "conv_bf16: the window of one channel-group pair (%d pixels) does not fit LDS", g.Lpad);
  const int Cp = g.Cp, KG = g.KG, ocP = g.ocP, T2 = g.T2, Lpad = g.Lpad, KGC = g.KGC, nck = g.nck;
  const int kgsLast = g.kgsLast, FL = g.FL, LL = g.LL, DL = g.DL;
  const size_t ldsb = g.ldsb;
  E2_REQUIRE(ldsb <= 160 * 1024, "conv_bf16: LDS window of %zu B", ldsb);
  const size_t xb_bytes = (size_t)in->n * in->d * in->h * in->w * Cp * 2;
  const size_t wb_bytes = g.wb_bytes;
  const bool need_x = !ext.xb, need_w = !ext.wb;
  const size_t need = ((need_x && !xkeep) ? ((xb_bytes + 255) / 256) * 256 : 0) + (need_w ? wb_bytes + 256 : 0);
  E2_REQUIRE(need == 0 || (ws && ws_bytes >= need), "conv_bf16: workspace of %zu bytes needed, %zu given", need, ws_bytes);
  E2_REQUIRE(need == 0 || ((uintptr_t)ws & 15) == 0, "conv_bf16: workspace must be 16-byte aligned");
  E2_REQUIRE(((uintptr_t)ext.xb & 15) == 0 && ((uintptr_t)ext.wb & 15) == 0 && ((uintptr_t)ext.nxb & 15) == 0,
             "conv_bf16: ready-made operands must be 16-byte aligned");
  E2_REQUIRE((long)ocP * 32 < (1L << 31), "conv_bf16: too many channels");
  E2_REQUIRE(!xkeep || (xkeep_bytes >= xb_bytes && ((uintptr_t)xkeep & 15) == 0),
             "conv_bf16: the kept input copy needs %zu bytes, 16-byte aligned (%zu given)", xb_bytes, xkeep_bytes);
  E2_REQUIRE(!ext.nxb || (mode == 0 && ext.nxKG * 8 >= rows && ext.nxKG * 8 <= ocP),
             "conv_bf16: the next layer's copy holds %d channel groups for %d channels (%d rows computed)",
             ext.nxKG, rows, ocP);
  // (xkeep: the channels-last bf16 copy of the input goes to the caller's buffer instead of
  // the workspace -- the layer's weight gradient reads it again, e2_conv3d_wgrad_bf16_xcl)
  const __bf16* xb;
  char* wsp = reinterpret_cast<char*>(ws);
  if (ext.xb) xb = reinterpret_cast<const __bf16*>(ext.xb);
  else if (xkeep) xb = reinterpret_cast<const __bf16*>(xkeep);
  else { xb = reinterpret_cast<const __bf16*>(wsp); wsp += ((xb_bytes + 255) / 256) * 256; }
  const __bf16* wb = ext.wb ? reinterpret_cast<const __bf16*>(ext.wb) : reinterpret_cast<const __bf16*>(wsp);
  if (need_x || need_w) {
    CvP c{in->ptr, in->sn, in->sc, in->sd, in->sh, in->n, in->c, in->d, in->h, in->w, KG, const_cast<__bf16*>(xb)};
    const long ctot = (long)in->n * in->d * in->h * KG * in->w;
    const long wtot = ((long)kd * DL + kPD) * 2 * ocP * 8;
    const int cblocks = need_x ? (int)std::min<long>((ctot + 255) / 256, 8192) : 0;
    const int wblocks = need_w ? (int)std::min<long>((wtot / 8 + 255) / 256, 2048) : 0;   // (a thread per 16-byte piece)
    PwP q{w, const_cast<__bf16*>(wb), nf, nin, kd, T2, ocP, mode, KGC, nck, kgsLast, FL, LL};
    E2_REQUIRE(!need_w || w, "conv_bf16: null weights");
    hipLaunchKernelGGL(prep_bf16_kernel, dim3((unsigned)(cblocks + wblocks)), dim3(256), 0, ctx->stream,
                       c, q, cblocks);
  }
  CbP p;
  p.xb = xb; p.wb = wb; p.out = out->ptr;
  p.osN = out->sn; p.osC = out->sc; p.osZ = out->sd; p.osY = out->sh;
  p.bias = bias; p.act = act;
  p.N = in->n; p.Cout = rows; p.KG = KG; p.ocP = ocP;
  p.kd = kd; p.kh = kh; p.kw = kw;
  p.Do = out->d; p.Ho = out->h; p.Wo = out->w; p.Q = out->h * out->w;
  p.Din = in->d; p.Hin = in->h; p.Win = in->w;
  p.nPT = (p.Q + 64 * NB - 1) / (64 * NB);
  p.nMT = (ocP + 64 * MB - 1) / (64 * MB);
  p.Lpad = Lpad; p.KGC = KGC; p.nck = nck; p.kgsLast = kgsLast; p.FL = FL; p.LL = LL;
  p.nxb = reinterpret_cast<__bf16*>(ext.nxb); p.nxKG = ext.nxKG;
  const long units = (long)p.N * p.Do * p.nPT;
  const long grid = (units + 7) / 8 * 8 * p.nMT;
  E2_REQUIRE(grid < (1L << 31), "conv_bf16: grid too large");
  {
    int v[3];
    const bool forced = sscanf(ctx->tiling[E2_TILING_IGEMM], "%d,%d,%d", &v[0], &v[1], &v[2]) == 3 && v[0] == 32;
    e2_note_launch(ctx, "conv_bf16", forced ? E2_SRC_FORCED : (ctx->tiling[E2_TILING_IGEMM][0] ? E2_SRC_FALLBACK : E2_SRC_MODEL),
                   "32,%d,%d", MB, NB);
  }
  if (MB == 1 && NB == 1) return launch<1, 1>(ctx, p, (int)grid, ldsb);
  if (MB == 1 && NB == 2) return launch<1, 2>(ctx, p, (int)grid, ldsb);
  if (MB == 2 && NB == 1) return launch<2, 1>(ctx, p, (int)grid, ldsb);
  if (MB == 2 && NB == 2) return launch<2, 2>(ctx, p, (int)grid, ldsb);
  if (MB == 4 && NB == 1) return launch<4, 1>(ctx, p, (int)grid, ldsb);
  if (MB == 4 && NB == 2) return launch<4, 2>(ctx, p, (int)grid, ldsb);
  if (MB == 2 && NB == 4) return launch<2, 4>(ctx, p, (int)grid, ldsb);
  if (MB == 1 && NB == 4) return launch<1, 4>(ctx, p, (int)grid, ldsb);
  e2_set_error("conv_bf16: no instance MB=%d NB=%d", MB, NB);
  return 2;
}

static void tile_from_ctx(const e2_ctx* ctx, int* MB, int* NB) {
  *MB = 2; *NB = 2;
  int v[3];
  if (sscanf(ctx->tiling[E2_TILING_IGEMM], "%d,%d,%d", &v[0], &v[1], &v[2]) == 3 && v[0] == 32) {
    *MB = v[1]; *NB = v[2];
  }
}

}  // namespace

extern "C" size_t e2_conv3d_bf16_workspace_bytes(int n, int cin, int d, int h, int w, int cout,
                                                 int kd, int kh, int kw) {
  // both directions: the forward converts x (cin channels), the data gradient the padded dy
  const int cpF = pad16(cin), cpD = pad16(cout);
  const size_t xF = (size_t)n * d * h * w * cpF * 2;
  const size_t xD = (size_t)n * (d + kd - 1) * (h + kh - 1) * (w + kw - 1) * cpD * 2;  // (bound)
  // (steps of one plane: kh*kw per channel-group pair, + the ring padding of every chunk)
  const size_t sF = (size_t)(kh * kw + kPD) * (cpF / 16) + kPD, sD = (size_t)(kh * kw + kPD) * (cpD / 16) + kPD;
  const size_t wF = ((size_t)kd * sF + kPD) * 2 * ((cout + 127) / 128 * 128) * 16;   // (any MB)
  const size_t wD = ((size_t)kd * sD + kPD) * 2 * ((cin + 127) / 128 * 128) * 16;
  return std::max(xF, xD) + std::max(wF, wD) + 2048;
}

extern "C" int e2_conv3d_fwd_bf16(e2_ctx* ctx, const e2_tensor5* x, const float* w, int cout,
                                  int kd, int kh, int kw, const float* bias, int act,
                                  const e2_tensor5* out, void* ws, size_t ws_bytes) {
  E2_REQUIRE(ctx && w, "conv3d_fwd_bf16: null argument");
  if (int rc = view_ok(x, "conv3d_fwd_bf16 x")) return rc;
  if (int rc = view_ok(out, "conv3d_fwd_bf16 out")) return rc;
  E2_REQUIRE(!bias || act == E2_ACT_LIN || act == E2_ACT_RELU, "conv3d_fwd_bf16: bad act %d", act);
  int MB, NB;
  tile_from_ctx(ctx, &MB, &NB);
  return conv_bf16(ctx, x, w, cout, x->c, kd, kh, kw, 0, bias, act, out, ws, ws_bytes, MB, NB);
}

extern "C" size_t e2_conv3d_bf16_xkeep_bytes(int n, int cin, int d, int h, int w, int kh, int kw) {
  // the forward's copy [n][z][kg][pixel][8] (kg = channel groups of 8, an even count) + the
  // zero pixels behind the last plane that the weight gradient's windows run into
  const size_t pieces = (size_t)n * d * (pad16(cin) / 8) * h * w;
  const size_t slack = (size_t)(((h - kh + 1) * w + 63) / 64 * 64) + (size_t)(kh - 1) * w + kw + 128;
  return ((pieces + slack) * 16 + 255) / 256 * 256;
}

extern "C" int e2_conv3d_fwd_bf16_keep(e2_ctx* ctx, const e2_tensor5* x, const float* w, int cout,
                                       int kd, int kh, int kw, const float* bias, int act,
                                       const e2_tensor5* out, void* ws, size_t ws_bytes,
                                       void* xkeep, size_t xkeep_bytes) {
  E2_REQUIRE(ctx && w && xkeep, "conv3d_fwd_bf16_keep: null argument");
  if (int rc = view_ok(x, "conv3d_fwd_bf16_keep x")) return rc;
  if (int rc = view_ok(out, "conv3d_fwd_bf16_keep out")) return rc;
  E2_REQUIRE(!bias || act == E2_ACT_LIN || act == E2_ACT_RELU, "conv3d_fwd_bf16_keep: bad act %d", act);
  int MB, NB;
  tile_from_ctx(ctx, &MB, &NB);
  return conv_bf16(ctx, x, w, cout, x->c, kd, kh, kw, 0, bias, act, out, ws, ws_bytes, MB, NB,
                   xkeep, xkeep_bytes);
}

extern "C" int e2_conv3d_dgrad_bf16(e2_ctx* ctx, const e2_tensor5* dy_pad, const float* w, int cin,
                                    int kd, int kh, int kw, const e2_tensor5* dx, void* ws,
                                    size_t ws_bytes) {
  E2_REQUIRE(ctx && w, "conv3d_dgrad_bf16: null argument");
  if (int rc = view_ok(dy_pad, "conv3d_dgrad_bf16 dy_pad")) return rc;
  if (int rc = view_ok(dx, "conv3d_dgrad_bf16 dx")) return rc;
  int MB, NB;
  tile_from_ctx(ctx, &MB, &NB);
  return conv_bf16(ctx, dy_pad, w, dy_pad->c, cin, kd, kh, kw, 1, nullptr, 0, dx, ws, ws_bytes, MB, NB);
}

// ---- operands made ahead of the launch (the producers' epilogues) ----------
extern "C" size_t e2_conv3d_bf16_wb_bytes(int rows, int kk, int kd, int kh, int kw, int in_w,
                                          int out_w, int mb, int nb) {
  const CbGeo g = cb_geo(rows, kk, kd, kh, kw, in_w, out_w, mb, nb);
  return g.ok ? g.wb_bytes + 256 : 0;
}

extern "C" size_t e2_bf16_wjob_bytes(void) { return sizeof(PwP); }

extern "C" int e2_bf16_wjob_fill(void* rec, const float* w, int nf, int nin, int kd, int kh, int kw,
                                 int mode, int in_w, int out_w, int mb, int nb, void* wb,
                                 size_t wb_bytes) {
  E2_REQUIRE(rec && w && wb && ((uintptr_t)wb & 15) == 0, "bf16_wjob_fill: null / unaligned argument");
  const int rows = mode ? nin : nf, kk = mode ? nf : nin;
  const CbGeo g = cb_geo(rows, kk, kd, kh, kw, in_w, out_w, mb, nb);
  E2_REQUIRE(g.ok && wb_bytes >= g.wb_bytes, "bf16_wjob_fill: %zu bytes given, %zu needed", wb_bytes, g.wb_bytes);
  PwP q{w, reinterpret_cast<__bf16*>(wb), nf, nin, kd, g.T2, g.ocP, mode, g.KGC, g.nck, g.kgsLast, g.FL, g.LL};
  *reinterpret_cast<PwP*>(rec) = q;
  return 0;
}

extern "C" int e2_conv3d_bf16_pack_w_multi(e2_ctx* ctx, const void* jobs_dev, int njobs) {
  E2_REQUIRE(ctx && jobs_dev && njobs > 0, "conv3d_bf16_pack_w_multi: bad argument");
  hipLaunchKernelGGL(pack_w_bf16_multi_kernel, dim3(256, (unsigned)njobs), dim3(256), 0, ctx->stream,
                     reinterpret_cast<const PwP*>(jobs_dev));
  E2_CHECK_HIP(hipGetLastError());
  return 0;
}

// forward / data gradient with any of {input copy, filter rows} ready-made, and (forward) the
// next layer's input copy as a second output; the tile must be the one the filter rows were
// packed for (the forced "32,MB,NB")
extern "C" int e2_conv3d_fwd_bf16_ex(e2_ctx* ctx, const e2_tensor5* x, const float* w, int cout,
                                     int kd, int kh, int kw, const float* bias, int act,
                                     const e2_tensor5* out, void* ws, size_t ws_bytes,
                                     void* xkeep, size_t xkeep_bytes, int x_ready, const void* wb,
                                     void* next_xb, int next_kg) {
  E2_REQUIRE(ctx && (w || wb), "conv3d_fwd_bf16_ex: null argument");
  E2_REQUIRE(x && x->n > 0 && x->c > 0 && x->d > 0 && x->h > 0 && x->w > 0 && (x->ptr || x_ready),
             "conv3d_fwd_bf16_ex: bad x");
  if (int rc = view_ok(out, "conv3d_fwd_bf16_ex out")) return rc;
  E2_REQUIRE(!bias || act == E2_ACT_LIN || act == E2_ACT_RELU, "conv3d_fwd_bf16_ex: bad act %d", act);
  E2_REQUIRE(!x_ready || xkeep, "conv3d_fwd_bf16_ex: x_ready names the copy in xkeep");
  int MB, NB;
  tile_from_ctx(ctx, &MB, &NB);
  CbExt e;
  e.xb = x_ready ? xkeep : nullptr; e.wb = wb; e.nxb = next_xb; e.nxKG = next_kg;
  return conv_bf16(ctx, x, w, cout, x->c, kd, kh, kw, 0, bias, act, out, ws, ws_bytes, MB, NB,
                   xkeep, xkeep_bytes, e);
}

extern "C" int e2_conv3d_dgrad_bf16_ex(e2_ctx* ctx, const e2_tensor5* dy_pad, const float* w, int cin,
                                       int kd, int kh, int kw, const e2_tensor5* dx, void* ws,
                                       size_t ws_bytes, const void* dy_cl, const void* wb) {
  E2_REQUIRE(ctx && (w || wb), "conv3d_dgrad_bf16_ex: null argument");
  E2_REQUIRE(dy_pad && dy_pad->n > 0 && dy_pad->c > 0 && dy_pad->d > 0 && dy_pad->h > 0 && dy_pad->w > 0 &&
                 (dy_pad->ptr || dy_cl), "conv3d_dgrad_bf16_ex: bad dy_pad");
  if (int rc = view_ok(dx, "conv3d_dgrad_bf16_ex dx")) return rc;
  int MB, NB;
  tile_from_ctx(ctx, &MB, &NB);
  CbExt e;
  e.xb = dy_cl; e.wb = wb;
  return conv_bf16(ctx, dy_pad, w, dy_pad->c, cin, kd, kh, kw, 1, nullptr, 0, dx, ws, ws_bytes, MB, NB,
                   nullptr, 0, e);
}
