// conv_pw_wgrad.hip -- the weight gradient of a 1x1x1 conv (and of UpConv, whose GEMMs are
// 1x1x1: api.hip upconv_bwd) as a plain GEMM of its own:
//
//   dw[m][n] (+)= sum over samples and positions k of  dy[m][k] * x[n][k]
//
// (T.grad's ConvGradW for the reference's "tensordot" branch, computations.py:330-335,
// 377-384; model.py:182-186.)  Both operands are K-CONTIGUOUS -- a channel's positions lie
// next to each other in memory -- which is the one shape the direct weight-gradient kernel
// (conv_wgrad_direct.hip) is not built for: it stages input SPANS per (channel, tap) in LDS and
// gathers from them, machinery a single tap does not need, and with NT = 1 it re-reads every
// dy row once per 16 input channels (DESIGN.md known loss 5: 31 us for 7 us of MFMA on
// neuro3d_lite's 200 -> 200 layer, 56-61 us for the U-Nets' 1.4-2.4 GF UpConv gradients).
//
// Here: 16x16x4 f32 MFMA with operands straight from memory.  For 16 consecutive positions
// lane (i = l & 15, q = l >> 4) loads ONE float4 -- positions 4q .. 4q+3 of channel row i --
// per 16-row block; the four MFMAs of the step take component j as their k index, i.e. k-step
// j covers positions {4q + j}: A and B use the same permutation of the 16 positions, so the
// sum is unchanged and no lane ever shuffles.  A wave owns MT x NT blocks, a work-group 2 x 2
// waves (a tile of 32 MT x 32 NT outputs; the two waves of a row share their dy rows through
// L1), S work-groups split the positions of a tile and add their partial tiles to dw with
// f32 atomics (dw zeroed first unless accumulate).  The operand sets of 2-5 steps are in flight
// (a ring of 3-6 register sets, deeper for the smaller tiles).
// Tiling "MT,NT,7,0,S" through e2_set_tiling(E2_TILING_WGRAD, ...); never chosen untuned.
// Requires dense channel planes (row stride = W, plane stride = H * W) for x and dy.
#include "common.hpp"
#include <algorithm>
#include <type_traits>
#include <vector>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifdef E2_DEBUG_ENV
#define KS_STAMP(i) do { if (p.stamps && threadIdx.x == 0) p.stamps[8L * blockIdx.x + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define KS_STAMP(i) do {} while (0)
#endif

namespace {

struct PgP {
  const float* a; const float* b; float* c;        // dy, x, dw
  long asN, asC, bsN, bsC;
  int N, M, Ncol, K;                               // K = positions of one sample
  int R;                                           // UpConv: row m = co * R + r
  int nMT, nNT, S;
  int stepsPerSample, steps, per;                  // 16-position steps: per sample, total, per split
  int rem;                                         // (position-split waves) K % 32
  // (position-split waves, kernels with taps) a unit belongs to plane z of sample n; column j of
  // the GEMM is (input channel j / T, tap j % T) and reads the channel's plane at the tap's shift
  int T, kh, kw, flip, planes;                     // taps, kernel rows / cols, planes per sample
  long asZ, bsZ, bsY;                              // plane strides of dy / x, row stride of x
  unsigned long long* stamps;                      // debug build (E2_PWKS_STAMPS): 8 s_memtime stamps per work-group
};

// lane's float4 of a 16-position step: positions k0 + 4q .. + 3 of one channel row; past the end
// of the sample's K the values are zero (the row pointer itself is always valid)
__device__ __forceinline__ f32x4 pg_load(const float* row, int k, int K) {
  if (k + 3 < K) {
    f32x4 v;
    __builtin_memcpy(&v, row + k, 16);             // (4-byte aligned only: rows need not be 16-byte aligned)
    return v;
  }
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (k + j < K) v[j] = row[k + j];
  return v;
}

template <int MT, int NT>
__global__ __launch_bounds__(256) void pw_wgrad_kernel(PgP p) {
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, q = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  int b = blockIdx.x;
  const int nt = b % p.nNT; b /= p.nNT;
  const int mt = b % p.nMT;
  const int sp = b / p.nMT;
  const int m0 = (mt * 2 + wm) * 16 * MT;
  const int n0 = (nt * 2 + wn) * 16 * NT;
  const int s0 = sp * p.per, s1 = min(s0 + p.per, p.steps);

  // row offsets (clamped: rows past the end are computed and never flushed)
  long aoff[MT], boff[NT];
#pragma unroll
  for (int mb = 0; mb < MT; ++mb) aoff[mb] = (long)min(m0 + 16 * mb + l15, p.M - 1) * p.asC;
#pragma unroll
  for (int nb = 0; nb < NT; ++nb) boff[nb] = (long)min(n0 + 16 * nb + l15, p.Ncol - 1) * p.bsC;

  f32x4 acc[MT][NT];
#pragma unroll
  for (int mb = 0; mb < MT; ++mb)
#pragma unroll
    for (int nb = 0; nb < NT; ++nb) acc[mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto load = [&](int s, f32x4 (&A)[MT], f32x4 (&B)[NT]) {
    const int sc = min(s, s1 - 1);                   // (a load past the range repeats the last step: unused)
    const int n = sc / p.stepsPerSample;
    const int k = (sc - n * p.stepsPerSample) * 16 + 4 * q;
    const float* ap = p.a + (long)n * p.asN;
    const float* bp = p.b + (long)n * p.bsN;
#pragma unroll
    for (int mb = 0; mb < MT; ++mb) A[mb] = pg_load(ap + aoff[mb], k, p.K);
#pragma unroll
    for (int nb = 0; nb < NT; ++nb) B[nb] = pg_load(bp + boff[nb], k, p.K);
  };
  auto fma = [&](const f32x4 (&A)[MT], const f32x4 (&B)[NT]) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int mb = 0; mb < MT; ++mb)
#pragma unroll
        for (int nb = 0; nb < NT; ++nb)
          acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[mb][j], B[nb][j], acc[mb][nb], 0, 0, 0);
  };

  // a ring of D operand sets: D - 1 steps are in flight while one is multiplied (a step of a
  // 2 x 2 tile is 16 MFMAs = 0.2 us, a round trip to L2 / HBM five to ten times that)
  constexpr int D = (MT + NT) <= 4 ? 6 : ((MT + NT) <= 6 ? 4 : 3);
  if (s0 < s1) {
    f32x4 A[D][MT], B[D][NT];
#pragma unroll
    for (int d = 0; d < D - 1; ++d) load(s0 + d, A[d], B[d]);
    int s = s0;
    for (; s + D <= s1; s += D) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        load(s + d + D - 1, A[(d + D - 1) % D], B[(d + D - 1) % D]);   // into the set the last fma freed
        fma(A[d], B[d]);
      }
    }
    // fewer than D steps left: sets 0 .. D-2 hold steps s .. s + D - 2
#pragma unroll
    for (int d = 0; d < D - 1; ++d)
      if (s + d < s1) fma(A[d], B[d]);
  }

  // ---- flush: D row = 4 q + r, col = l15 -----------------------------------------------------
  const bool single = p.S == 1;
#pragma unroll
  for (int mb = 0; mb < MT; ++mb) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + 16 * mb + 4 * q + r;
      if (m >= p.M) continue;
      long rbase;
      int cstride;
      if (p.R > 1) { const int co = m / p.R; rbase = (long)co * p.Ncol * p.R + (m - co * p.R); cstride = p.R; }
      else { rbase = (long)m * p.Ncol; cstride = 1; }
#pragma unroll
      for (int nb = 0; nb < NT; ++nb) {
        const int n = n0 + 16 * nb + l15;
        if (n >= p.Ncol) continue;
        float* dst = p.c + rbase + (long)n * cstride;
        if (single) *dst += acc[mb][nb][r];          // (dw was zeroed or holds what to add to)
        else unsafeAtomicAdd(dst, acc[mb][nb][r]);
      }
    }
  }
}

// ---- the same GEMM with the four waves of a work-group splitting the POSITIONS ---------------
// ("MT,NT,8,0,S").  The tiles above give every wave its own output blocks, so a work-group's
// tile is 32 MT x 32 NT outputs: 200 channels fill 224 x 256 at best (1.43 x the MFMAs), and
// the small tiles that would fit ask L2 for ~20 TB/s.  Here ONE wave holds the whole
// 16 MT x 16 NT tile (13 x 2 blocks = 208 x 32 for 200 channels: 1.16 x), the four waves take
// every fourth 32-position unit of the work-group's range, their partial tiles are summed
// through LDS and flushed once.  Operand registers: A is single-buffered -- block mb's two
// float4 of the NEXT unit are requested right after block mb's MFMAs of this unit, so every
// load has a whole unit (13 x 2 x 8 MFMAs = 2.8 us) to arrive -- B double-buffered.  Units are
// WHOLE; the K % 32 last positions of the samples are a masked step of the last range's work-groups.
template <int MT, int NT, bool TAPS>
__global__ __launch_bounds__(256) void pw_wgrad_ks_kernel(PgP p) {
  extern __shared__ float red[];
  constexpr int RW = 16 * NT + 4;                    // padded row of a wave's partial tile
  constexpr int RS = 16 * MT * RW;
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, q = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // work-groups b, b + 8, b + 16 ... share an XCD: the tiles of one position range are
  // consecutive THERE, so an operand row comes out of HBM once per range, not once per tile
  int L = blockIdx.x;
  if ((gridDim.x & 7) == 0) L = (L & 7) * (gridDim.x >> 3) + (L >> 3);
  const int nt = L % p.nNT; L /= p.nNT;
  const int mt = L % p.nMT;
  const int sp = L / p.nMT;
  const int m0 = mt * 16 * MT, n0 = nt * 16 * NT;
  const bool single = p.S == 1;
  auto dst_of = [&](int m, int n) -> float* {
    if (p.R > 1) { const int co = m / p.R; return p.c + (long)co * p.Ncol * p.R + (m - co * p.R) + (long)n * p.R; }
    return p.c + (long)m * p.Ncol + n;
  };

  if (sp >= p.S) return;                             // (grid padded to a multiple of 8)
  const int u0 = sp * p.per + wave, u1 = min(sp * p.per + p.per, p.steps);
  const int cnt = u0 < u1 ? (u1 - u0 + 3) >> 2 : 0;
  KS_STAMP(0);

  // BYTE offsets of the lane's rows inside a sample (32 bits: checked by the host), so that a
  // load is "scalar sample base + vector offset"
  unsigned aoff[MT], boff[NT];
#pragma unroll
  for (int mb = 0; mb < MT; ++mb) aoff[mb] = (unsigned)min(m0 + 16 * mb + l15, p.M - 1) * (unsigned)p.asC * 4u + 16u * q;
#pragma unroll
  for (int nb = 0; nb < NT; ++nb) {
    const int j = min(n0 + 16 * nb + l15, p.Ncol - 1);
    if (TAPS) {
      // column j = (input channel, tap as stored); the tap's shift inside the channel's volume
      const int ci = j / p.T, ts = j - ci * p.T;
      const int t = p.flip ? p.T - 1 - ts : ts;
      const int tz = t / (p.kh * p.kw), r2 = t - tz * (p.kh * p.kw);
      const int ty = r2 / p.kw, tx = r2 - ty * p.kw;
      const unsigned row = (unsigned)ci * (unsigned)p.bsC * 4u;
      boff[nb] = row + (unsigned)(tz * p.bsZ + ty * p.bsY + tx) * 4u + 16u * q;
    } else {
      boff[nb] = (unsigned)j * (unsigned)p.bsC * 4u + 16u * q;
    }
  }

  f32x4 acc[MT][NT];
#pragma unroll
  for (int mb = 0; mb < MT; ++mb)
#pragma unroll
    for (int nb = 0; nb < NT; ++nb) acc[mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};

  // unit i of this wave (WHOLE 32-position units only): sample bases and byte offset of the unit
  // (TAPS: units of a PLANE; zb = byte offset of the plane inside the sample's x.  A unit may run
  // past the end of its plane: dy holds zeros there -- its border -- and x whatever follows, up to
  // 124 bytes behind the tensor for the last plane of the last channel: e2_set_input_slack)
  auto where = [&](int i, const char*& ap, const char*& bp, unsigned& kb, unsigned& zb) {
    const int u = u0 + 4 * i;
    const int pl = u / p.stepsPerSample;
    kb = (unsigned)(u - pl * p.stepsPerSample) * 128u;
    if (TAPS) {
      const int n = pl / p.planes, z = pl - n * p.planes;
      ap = reinterpret_cast<const char*>(p.a + (long)n * p.asN + (long)z * p.asZ);
      bp = reinterpret_cast<const char*>(p.b + (long)n * p.bsN);
      zb = (unsigned)(z * p.bsZ) * 4u;
    } else {
      ap = reinterpret_cast<const char*>(p.a + (long)pl * p.asN);
      bp = reinterpret_cast<const char*>(p.b + (long)pl * p.bsN);
      zb = 0;
    }
  };
  auto bo = [&](int nb, unsigned kb, unsigned zb) -> unsigned {
    return TAPS ? boff[nb] + zb + kb : boff[nb] + kb;
  };
  auto load2 = [&](f32x4 (&d)[2], const char* base, unsigned off) {
    const char* r = base + off;                      // (4-byte aligned only: rows need not be 16-byte aligned)
    __builtin_memcpy(&d[0], r, 16);
    __builtin_memcpy(&d[1], r + 64, 16);
  };

  if (cnt > 0) {
    f32x4 A[MT][2], Bc[NT][2], Bn[NT][2];
    const char *ap, *bp;
    unsigned kb, zb;
    where(0, ap, bp, kb, zb);
#pragma unroll
    for (int nb = 0; nb < NT; ++nb) load2(Bc[nb], bp, bo(nb, kb, zb));
#pragma unroll
    for (int mb = 0; mb < MT; ++mb) load2(A[mb], ap, aoff[mb] + kb);
    KS_STAMP(1);
    // ONE loop body (accumulators stay where they are): the last trip requests its own unit
    // again -- L1 / L2 hits, nothing waits for them
#pragma unroll 1
    for (int i = 0; i < cnt; ++i) {
      where(min(i + 1, cnt - 1), ap, bp, kb, zb);
#pragma unroll
      for (int nb = 0; nb < NT; ++nb) load2(Bn[nb], bp, bo(nb, kb, zb));
#pragma unroll
      for (int mb = 0; mb < MT; ++mb) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int nb = 0; nb < NT; ++nb)
              acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[mb][h][j], Bc[nb][h][j], acc[mb][nb], 0, 0, 0);
        load2(A[mb], ap, aoff[mb] + kb);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int nb = 0; nb < NT; ++nb) { Bc[nb][0] = Bn[nb][0]; Bc[nb][1] = Bn[nb][1]; }
#ifdef E2_DEBUG_ENV
      if (i == 0) KS_STAMP(2);
#endif
    }
  }
  KS_STAMP(3);
  // ---- the K % 32 last positions of every sample: one masked step of the work-groups of the LAST
  // position range (the shortest one: per is rounded up).  Wave w takes the row blocks w, w + 4,
  // ...; every load is a clamped dword + select (no branches: all of them are in flight at once)
  if (!TAPS && sp == p.S - 1 && p.rem > 0) {
    constexpr int MBW = (MT + 3) / 4;
    const int k = p.K - p.rem + 4 * q;
#pragma unroll 1
    for (int n = 0; n < p.N; ++n) {
      float av[MBW][8], bv[NT][8];
      auto mload = [&](float (&d)[8], const float* row) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int pos = k + (e >> 2) * 16 + (e & 3);
          const float v = row[min(pos, p.K - 1)];
          d[e] = pos < p.K ? v : 0.f;
        }
      };
#pragma unroll
      for (int nb = 0; nb < NT; ++nb)
        mload(bv[nb], p.b + (long)n * p.bsN + (long)min(n0 + 16 * nb + l15, p.Ncol - 1) * p.bsC);
#pragma unroll
      for (int i = 0; i < MBW; ++i)
        mload(av[i], p.a + (long)n * p.asN + (long)min(m0 + 16 * min(wave + 4 * i, MT - 1) + l15, p.M - 1) * p.asC);
#pragma unroll
      for (int mb = 0; mb < MT; ++mb)
        if ((mb & 3) == wave) {
#pragma unroll
          for (int e = 0; e < 8; ++e)
#pragma unroll
            for (int nb = 0; nb < NT; ++nb)
              acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mb >> 2][e], bv[nb][e], acc[mb][nb], 0, 0, 0);
        }
    }
  }

  // ---- the four partial tiles through LDS: D row = 4 q + r, col = l15 ------------------------
  float* mine = red + wave * RS;
#pragma unroll
  for (int mb = 0; mb < MT; ++mb)
#pragma unroll
    for (int nb = 0; nb < NT; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) mine[(16 * mb + 4 * q + r) * RW + 16 * nb + l15] = acc[mb][nb][r];
  __syncthreads();
  KS_STAMP(4);
#pragma unroll 2
  for (int e = tid; e < 16 * MT * 16 * NT; e += 256) {
    const int ml = e / (16 * NT), nl = e - ml * (16 * NT);
    const int m = m0 + ml, n = n0 + nl;
    const float* s = red + ml * RW + nl;
    const float v = (s[0] + s[RS]) + (s[2 * RS] + s[3 * RS]);
    if (m >= p.M || n >= p.Ncol) continue;
    float* dst = dst_of(m, n);
    if (single) *dst += v;                           // (dw was zeroed or holds what to add to)
    else unsafeAtomicAdd(dst, v);
  }
  KS_STAMP(5);
#ifdef E2_DEBUG_ENV
  if (p.stamps) { __builtin_amdgcn_s_waitcnt(0); KS_STAMP(6); }
#endif
}

template <int MT, int NT, bool TAPS = false>
int launch_ks(e2_ctx* ctx, const PgP& p, long grid) {
  const int lds = 4 * 16 * MT * (16 * NT + 4) * (int)sizeof(float);
  static bool raised = false;                        // (one device per process: plan.py get_ctx)
  if (!raised) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pw_wgrad_ks_kernel<MT, NT, TAPS>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) { e2_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return 1; }
    raised = true;
  }
#ifdef E2_DEBUG_ENV
  // in-kernel timeline (debug build): mean s_memtime ticks between the stamps over the work-groups
  PgP ps = p;
  const bool stamps = e2_dbg_env("E2_PWKS_STAMPS") != nullptr && !ctx->capturing;
  if (stamps) {
    E2_CHECK_HIP(hipMalloc(&ps.stamps, sizeof(unsigned long long) * 8 * grid));
    E2_CHECK_HIP(hipMemsetAsync(ps.stamps, 0, sizeof(unsigned long long) * 8 * grid, ctx->stream));
  }
  hipLaunchKernelGGL((pw_wgrad_ks_kernel<MT, NT, TAPS>), dim3((unsigned)grid), dim3(256), lds, ctx->stream, ps);
  E2_CHECK_HIP(hipGetLastError());
  if (stamps) {
    E2_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    std::vector<unsigned long long> h(8 * grid);
    E2_CHECK_HIP(hipMemcpy(h.data(), ps.stamps, sizeof(unsigned long long) * 8 * grid, hipMemcpyDeviceToHost));
    static const char* names[6] = {"prologue issue", "first unit", "other units", "LDS + barrier", "flush issue", "flush done"};
    double sum[6] = {0};
    unsigned long long t0 = ~0ull, t1 = 0, s1 = 0;
    long nb = 0;
    for (long b = 0; b < grid; ++b) {
      if (!h[8 * b + 6]) continue;                   // (pad / tail work-groups)
      ++nb;
      for (int i = 0; i < 6; ++i) sum[i] += (double)(h[8 * b + i + 1] - h[8 * b + i]);
      t0 = std::min(t0, h[8 * b]); t1 = std::max(t1, h[8 * b + 6]); s1 = std::max(s1, h[8 * b]);
    }
    fprintf(stderr, "[e2] pw_wgrad_ks<%d,%d> grid %ld (%ld main), S %d, per %d: first start -> last end %llu ticks, last start %llu\n",
            MT, NT, grid, nb, p.S, p.per, t1 - t0, s1 - t0);
    for (int i = 0; i < 6; ++i) fprintf(stderr, "   %-14s %9.0f ticks\n", names[i], sum[i] / std::max(1L, nb));
    (void)hipFree(ps.stamps);
  }
#else
  hipLaunchKernelGGL((pw_wgrad_ks_kernel<MT, NT, TAPS>), dim3((unsigned)grid), dim3(256), lds, ctx->stream, p);
  E2_CHECK_HIP(hipGetLastError());
#endif
  return 0;
}

template <int MT, int NT>
int launch(e2_ctx* ctx, const PgP& p, long grid) {
  hipLaunchKernelGGL((pw_wgrad_kernel<MT, NT>), dim3((unsigned)grid), dim3(256), 0, ctx->stream, p);
  E2_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace

int e2i_pw_wgrad(e2_ctx* ctx, const WgradArgs& a, int MT, int NT, int S) {
  E2_REQUIRE(a.kd == 1 && a.kh == 1 && a.kw == 1, "pointwise wgrad: kernel %dx%dx%d is not 1x1x1", a.kd, a.kh, a.kw);
  E2_REQUIRE(a.xsY == a.Wo && a.xsZ == (int64_t)a.Ho * a.Wo && a.dsY == a.Wo && a.dsZ == (int64_t)a.Ho * a.Wo,
             "pointwise wgrad: x and dy need dense channel planes (row stride = W, plane stride = H * W)");
  E2_REQUIRE(!ctx->mfma_bf16, "pointwise wgrad: an f32 kernel, not offered in bf16 mode");
  PgP p;
  p.a = a.dy; p.b = a.x; p.c = a.dw;
  p.asN = a.dsN; p.asC = a.dsC; p.bsN = a.xsN; p.bsC = a.xsC;
  p.N = a.N; p.M = a.Cout; p.Ncol = a.Cin;
  const long K = (long)a.Do * a.Ho * a.Wo;
  E2_REQUIRE(K < (1L << 30), "pointwise wgrad: sample too large");
  p.K = (int)K;
  p.R = a.upR > 1 ? a.upR : 1;
  E2_REQUIRE(a.Cout % p.R == 0, "pointwise wgrad: %d rows for %d sub-positions", a.Cout, p.R);
  p.nMT = e2_cdiv(a.Cout, 32 * MT);
  p.nNT = e2_cdiv(a.Cin, 32 * NT);
  p.rem = 0; p.stamps = nullptr;
  p.T = 1; p.kh = p.kw = 1; p.flip = 0; p.planes = 1; p.asZ = p.bsZ = p.bsY = 0;
  p.stepsPerSample = (int)((K + 15) / 16);
  const long steps = (long)a.N * p.stepsPerSample;
  E2_REQUIRE(steps < (1L << 30), "pointwise wgrad: too many positions");
  p.steps = (int)steps;
  S = (int)std::max<long>(1, std::min<long>(S, steps));
  p.per = (int)((steps + S - 1) / S);
  p.S = (int)((steps + p.per - 1) / p.per);
  const long grid = (long)p.nMT * p.nNT * p.S;
  E2_REQUIRE(grid < (1L << 31), "pointwise wgrad: grid too large");
  if (!a.accumulate)
    if (int rc = e2i_fill_flat(ctx, a.dw, (size_t)a.Cout * a.Cin, 0.f)) return rc;
#define E2_L(M, N_) if (MT == M && NT == N_) return launch<M, N_>(ctx, p, grid);
  E2_L(2, 2) E2_L(4, 2) E2_L(2, 4) E2_L(4, 4) E2_L(4, 3) E2_L(3, 4) E2_L(7, 2) E2_L(2, 7) E2_L(7, 4) E2_L(4, 7)
#undef E2_L
  e2_set_error("pointwise wgrad: no instance MT=%d NT=%d", MT, NT);
  return 2;
}

// "MT,NT,8,0,S": the waves of a work-group split the positions (pw_wgrad_ks_kernel)
int e2i_pw_wgrad_ks(e2_ctx* ctx, const WgradArgs& a, int MT, int NT, int S) {
  E2_REQUIRE(a.kd == 1 && a.kh == 1 && a.kw == 1, "pointwise wgrad: kernel %dx%dx%d is not 1x1x1", a.kd, a.kh, a.kw);
  E2_REQUIRE(a.xsY == a.Wo && a.xsZ == (int64_t)a.Ho * a.Wo && a.dsY == a.Wo && a.dsZ == (int64_t)a.Ho * a.Wo,
             "pointwise wgrad: x and dy need dense channel planes (row stride = W, plane stride = H * W)");
  E2_REQUIRE(!ctx->mfma_bf16, "pointwise wgrad: an f32 kernel, not offered in bf16 mode");
  PgP p;
  p.a = a.dy; p.b = a.x; p.c = a.dw;
  p.asN = a.dsN; p.asC = a.dsC; p.bsN = a.xsN; p.bsC = a.xsC;
  p.N = a.N; p.M = a.Cout; p.Ncol = a.Cin;
  const long K = (long)a.Do * a.Ho * a.Wo;
  E2_REQUIRE(K >= 4 && K < (1L << 30), "pointwise wgrad: sample of %ld positions", K);
  // 32-bit element offsets of a row inside its sample (+ the 32 positions of a unit)
  E2_REQUIRE(((long)a.Cout * a.dsC + K) * 4 < (1L << 32) && ((long)a.Cin * a.xsC + K) * 4 < (1L << 32) && a.dsC >= 0 && a.xsC >= 0,
             "pointwise wgrad: sample too large for 32-bit byte offsets");
  p.K = (int)K;
  p.R = a.upR > 1 ? a.upR : 1;
  E2_REQUIRE(a.Cout % p.R == 0, "pointwise wgrad: %d rows for %d sub-positions", a.Cout, p.R);
  p.nMT = e2_cdiv(a.Cout, 16 * MT);
  p.nNT = e2_cdiv(a.Cin, 16 * NT);
  p.stepsPerSample = (int)(K / 32);                  // (here: WHOLE 32-position units)
  p.rem = (int)(K % 32); p.stamps = nullptr;
  p.T = 1; p.kh = p.kw = 1; p.flip = 0; p.planes = 1; p.asZ = p.bsZ = p.bsY = 0;
  const long units = (long)a.N * p.stepsPerSample;
  E2_REQUIRE(units < (1L << 29), "pointwise wgrad: too many positions");
  p.steps = (int)units;
  S = (int)std::max<long>(1, std::min<long>(S, std::max<long>(1, (units + 3) / 4)));
  p.per = (int)std::max<long>(1, (units + S - 1) / S);
  p.per = (p.per + 3) & ~3;                          // whole rounds of the four waves
  p.S = (int)std::max<long>(1, (units + p.per - 1) / p.per);
  long grid = (long)p.nMT * p.nNT * p.S;
  grid = (grid + 7) & ~7L;                           // XCD-grouped order (the pad returns at once)
  E2_REQUIRE(grid < (1L << 31), "pointwise wgrad: grid too large");
  if (!a.accumulate)
    if (int rc = e2i_fill_flat(ctx, a.dw, (size_t)a.Cout * a.Cin, 0.f)) return rc;
#define E2_L(M, N_) if (MT == M && NT == N_) return launch_ks<M, N_>(ctx, p, grid);
  E2_L(13, 2) E2_L(7, 2) E2_L(7, 4) E2_L(4, 4) E2_L(10, 2)
#undef E2_L
  e2_set_error("pointwise wgrad (position-split waves): no instance MT=%d NT=%d", MT, NT);
  return 2;
}

// "MT,NT,9,0,S": the SAME kernel for a conv with taps (T = kd * kh * kw > 1).  dy lives in the
// interior of its zero-padded buffer at the INPUT's row pitch (WgradArgs.dy_padded), so over the
// memory span of a plane -- K = (Ho - 1) * pitch + Wo positions, zeros in the kw - 1 gap columns
// -- tap (tz, ty, tx) of input channel ci is the channel's plane read at the constant shift
// tz * plane + ty * pitch + tx: every column (ci, tap) of dW (Cout x Cin * T, the weight tensor's
// own layout) is a K-contiguous row, and the 1x1x1 GEMM above applies with per-lane row offsets.
// Units are whole: one that runs past the end of its plane multiplies dy's zero border ((kh - 1)
// rows >= 31 positions: required) with whatever x holds there -- the next plane / channel / sample,
// or up to 124 bytes behind the tensor, which the caller vouches for (e2_set_input_slack).
int e2i_wgrad_ks(e2_ctx* ctx, const WgradArgs& a, int MT, int NT, int S) {
  const int T = a.kd * a.kh * a.kw;
  E2_REQUIRE(T > 1 && a.upR <= 1, "wgrad (position-split GEMM): a conv kernel with taps");
  E2_REQUIRE(a.dy_padded, "wgrad (position-split GEMM): dy must be the interior of its zero-padded buffer");
  E2_REQUIRE(a.dsY == a.xsY, "wgrad (position-split GEMM): dy rows at the input's pitch (%ld vs %ld)", (long)a.dsY, (long)a.xsY);
  E2_REQUIRE((a.kh - 1) * a.xsY + (a.kw - 1) >= 31, "wgrad (position-split GEMM): %d x %d kernel rows of %ld leave fewer than 31 zeros behind a plane",
             a.kh, a.kw, (long)a.xsY);
  E2_REQUIRE(!ctx->mfma_bf16, "wgrad (position-split GEMM): an f32 kernel, not offered in bf16 mode");
  const int Din = a.Do + a.kd - 1, Hin = a.Ho + a.kh - 1;
  E2_REQUIRE(a.xsZ >= (int64_t)Hin * a.xsY && a.xsC >= (int64_t)Din * a.xsZ, "wgrad (position-split GEMM): x planes / channels overlap");
  PgP p;
  p.a = a.dy; p.b = a.x; p.c = a.dw;
  p.asN = a.dsN; p.asC = a.dsC; p.bsN = a.xsN; p.bsC = a.xsC;
  p.N = a.N; p.M = a.Cout; p.Ncol = a.Cin * T;
  const long K = (long)(a.Ho - 1) * a.dsY + a.Wo;             // memory span of a gradient plane
  E2_REQUIRE(K >= 4 && K < (1L << 28), "wgrad (position-split GEMM): plane of %ld positions", K);
  E2_REQUIRE(((long)a.Cout * a.dsC + (long)a.Do * a.dsZ + K) * 4 + 256 < (1L << 32) &&
                 ((long)a.Cin * a.xsC + (long)Din * a.xsZ) * 4 + 256 < (1L << 32) && a.dsC >= 0 && a.xsC >= 0,
             "wgrad (position-split GEMM): sample too large for 32-bit byte offsets");
  p.K = (int)K; p.R = 1; p.rem = 0; p.stamps = nullptr;
  p.T = T; p.kh = a.kh; p.kw = a.kw; p.flip = a.flip; p.planes = a.Do;
  p.asZ = a.dsZ; p.bsZ = a.xsZ; p.bsY = a.xsY;
  E2_REQUIRE(ctx->input_slack >= 128, "wgrad (position-split GEMM): needs e2_set_input_slack(ctx, >= 128)");
#ifdef E2_DEBUG_ENV
  // debug build (make DEBUG_ENV=1): CHECK the caller's promise -- the 32 floats behind the last
  // element of x must be finite (they meet the gradient's zero border: 0 x Inf / NaN = NaN in dW
  // where the reference produces none).  Synchronises: never during capture.
  if (!ctx->capturing) {
    const float* end = a.x + (int64_t)(a.N - 1) * a.xsN + (int64_t)(a.Cin - 1) * a.xsC +
                       (int64_t)(Din - 1) * a.xsZ + (int64_t)(Hin - 1) * a.xsY + (a.Wo + a.kw - 1);
    float h[32];
    E2_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    E2_CHECK_HIP(hipMemcpy(h, end, sizeof(h), hipMemcpyDeviceToHost));
    for (int i = 0; i < 32; ++i)
      E2_REQUIRE(h[i] == h[i] && h[i] - h[i] == 0.f, "wgrad (position-split GEMM): e2_set_input_slack promised finite "
                 "values behind x, but float %d behind its last element is %g", i, (double)h[i]);
  }
#endif
  p.nMT = e2_cdiv(a.Cout, 16 * MT);
  p.nNT = e2_cdiv(p.Ncol, 16 * NT);
  p.stepsPerSample = (int)((K + 31) / 32);                    // (whole units of a PLANE)
  const long units = (long)a.N * a.Do * p.stepsPerSample;
  E2_REQUIRE(units < (1L << 29), "wgrad (position-split GEMM): too many positions");
  p.steps = (int)units;
  S = (int)std::max<long>(1, std::min<long>(S, std::max<long>(1, (units + 3) / 4)));
  p.per = (int)std::max<long>(1, (units + S - 1) / S);
  p.per = (p.per + 3) & ~3;
  p.S = (int)std::max<long>(1, (units + p.per - 1) / p.per);
  long grid = (long)p.nMT * p.nNT * p.S;
  grid = (grid + 7) & ~7L;
  E2_REQUIRE(grid < (1L << 31), "wgrad (position-split GEMM): grid too large");
  if (!a.accumulate)
    if (int rc = e2i_fill_flat(ctx, a.dw, (size_t)a.Cout * p.Ncol, 0.f)) return rc;
#define E2_L(M, N_) if (MT == M && NT == N_) return launch_ks<M, N_, true>(ctx, p, grid);
  E2_L(13, 2) E2_L(7, 2) E2_L(7, 4) E2_L(4, 4) E2_L(10, 2) E2_L(5, 4) E2_L(3, 4) E2_L(2, 4) E2_L(8, 2) E2_L(8, 4) E2_L(6, 4)
#undef E2_L
  e2_set_error("wgrad (position-split GEMM): no instance MT=%d NT=%d", MT, NT);
  return 2;
}
