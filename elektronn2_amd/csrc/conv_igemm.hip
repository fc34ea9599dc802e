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
#include <stdlib.h>
#include <algorithm>

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
  int CC, log2CC;
  int Lpad, BMpad;
  int nPT, nMT, splitK, nChunkC;
  int atomic;
  int upz, upy, upx;
};

template <int MT, int NT>
__global__ __launch_bounds__(256, 2) void igemm_kernel(IgemmP p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xl = smem;
  float* wl = smem + p.CC * p.Lpad;
  constexpr int BM = 16 * MT, BN = 64 * NT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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

  int posoff[NT];
#pragma unroll
  for (int nb = 0; nb < NT; ++nb) {
    int q = min(q0 + wave * (16 * NT) + nb * 16 + l15, p.Q - 1);
    int r = q / p.Wo, c = q - r * p.Wo;
    posoff[nb] = (r - r0) * isY + (c - c0) + qd * p.Lpad;
  }

  f32x4 acc[MT][NT];
#pragma unroll
  for (int mb = 0; mb < MT; ++mb)
#pragma unroll
    for (int nb = 0; nb < NT; ++nb) acc[mb][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nChunks = p.kd * p.nChunkC;
  const int per = (nChunks + p.splitK - 1) / p.splitK;
  const int cb = ks * per, ce = min(cb + per, nChunks);
  const int CC = p.CC;
  const int aBase = qd * p.BMpad + l15;

  for (int ch = cb; ch < ce; ++ch) {
    const int dz = ch / p.nChunkC;
    const int cc0 = (ch - dz * p.nChunkC) * CC;
    // ---- stage the input spans: CC rows of L floats, coalesced ----------
    const float* xb = p.in + (long)n * p.isN + (long)(z + dz) * p.isZ + span_lo;
    for (int cc = wave; cc < CC; cc += 4) {
      const int ci = cc0 + cc;
      const float* src = xb + (long)ci * p.isC;
      float* dst = xl + cc * p.Lpad;
      if (ci < p.Cin) {
        for (int u = lane; u < L; u += 64) dst[u] = src[u];
      } else {
        for (int u = lane; u < L; u += 64) dst[u] = 0.f;
      }
    }
    // ---- stage the packed weights: THW*CC rows of BM floats (float4) -----
    {
      const int total4 = p.THW * CC * (4 * MT);
      for (int i = tid; i < total4; i += 256) {
        const int rr = i / (4 * MT);
        const int c4 = i - rr * (4 * MT);
        const int t = rr >> p.log2CC;
        const int cc = rr & (CC - 1);
        const float* src = p.wp +
            (((long)(dz * p.THW + t) * p.ciP + (cc0 + cc)) * p.coP + m0 + 4 * c4);
        const float4 v = *reinterpret_cast<const float4*>(src);
        *reinterpret_cast<float4*>(wl + rr * p.BMpad + 4 * c4) = v;
      }
    }
    __syncthreads();
    // ---- MFMA over the chunk: K = THW taps x CC channels ------------------
    int ty = 0, tx = 0;
    for (int t = 0; t < p.THW; ++t) {
      const int tapoff = ty * isY + tx;
      for (int cg = 0; cg < (CC >> 2); ++cg) {
        const float* ap = wl + (t * CC + 4 * cg) * p.BMpad + aBase;
        const float* bp = xl + 4 * cg * p.Lpad + tapoff;
        float a[MT], b[NT];
#pragma unroll
        for (int mb = 0; mb < MT; ++mb) a[mb] = ap[mb * 16];
#pragma unroll
        for (int nb = 0; nb < NT; ++nb) b[nb] = bp[posoff[nb]];
#pragma unroll
        for (int mb = 0; mb < MT; ++mb)
#pragma unroll
          for (int nb = 0; nb < NT; ++nb)
            acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                a[mb], b[nb], acc[mb][nb], 0, 0, 0);
      }
      if (++tx == p.kw) { tx = 0; ++ty; }
    }
    __syncthreads();
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
// Wp[dz][t][ic(ciP)][oc(coP)], zero padded.  One thread per packed element.
__global__ void pack_weights_kernel(const float* __restrict__ w, float* __restrict__ wp,
                                    int Cout, int Cin, int kd, int THW, long wsO,
                                    long wsI, int flip, int ciP, int coP, int Rout,
                                    int Rin) {
  const long total = (long)kd * THW * ciP * coP;
  const int T = kd * THW;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int oc = (int)(i % coP);
    long r = i / coP;
    const int ic = (int)(r % ciP);
    r /= ciP;                       // r = dz*THW + t
    float v = 0.f;
    if (oc < Cout && ic < Cin) {
      const int tap = flip ? (T - 1 - (int)r) : (int)r;
      // Rout/Rin > 1: UpConv sub-position folded into the channel index
      v = w[(long)(oc / Rout) * wsO + (long)(ic / Rin) * wsI + tap + (oc % Rout) +
            (ic % Rin)];
    }
    wp[i] = v;
  }
}

// ---- host side ----------------------------------------------------------------
struct IgemmCfg { int MT, NT, CC, SK; };

template <int MT, int NT>
static int launch_one(e2_ctx* ctx, const IgemmP& p, int grid, size_t lds) {
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(
        reinterpret_cast<const void*>(&igemm_kernel<MT, NT>),
        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) { e2_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return 1; }
    attr_done = true;
  }
  hipLaunchKernelGGL((igemm_kernel<MT, NT>), dim3(grid), dim3(256), lds, ctx->stream, p);
  E2_CHECK_HIP(hipGetLastError());
  return 0;
}

static const int kMTs[] = {1, 2, 3, 4, 5, 6, 7, 8, 10, 13};

static int dispatch(e2_ctx* ctx, const IgemmP& p, int MT, int NT, int grid, size_t lds) {
#define E2_CASE(M)                                                          \
  case M:                                                                   \
    return NT == 1 ? launch_one<M, 1>(ctx, p, grid, lds)                    \
                   : launch_one<M, 2>(ctx, p, grid, lds);
  switch (MT) {
    E2_CASE(1) E2_CASE(2) E2_CASE(3) E2_CASE(4) E2_CASE(5) E2_CASE(6)
    E2_CASE(7) E2_CASE(8) E2_CASE(10) E2_CASE(13)
  }
#undef E2_CASE
  e2_set_error("igemm: no instance MT=%d NT=%d", MT, NT);
  return 2;
}

static int pad16mod32(int v) {          // smallest s >= v with s % 32 == 16
  int s = ((v + 15) / 16) * 16;
  if ((s & 31) == 0) s += 16;
  return s;
}

static int span_rows(int BN, int Wo) { return (BN + Wo - 2) / Wo; }

// Pick the tiling.  Cost model: work-groups run 2 per CU; a work-group's time
// is its MFMA count per wave (32 cycles each) plus a staging term.
static IgemmCfg choose_cfg(const e2_ctx* ctx, const IgemmArgs& a, int* ok) {
  const int mblocks = e2_cdiv(a.Cout, 16);
  const int THW = a.kh * a.kw;
  const long Q = (long)a.Ho * a.Wo;
  const int slots = ctx->num_cu * 2;
  IgemmCfg best{0, 0, 0, 0};
  double bestCost = 1e300;
  const char* force = getenv("E2_IGEMM_FORCE");
  if (force) {
    IgemmCfg f{0, 0, 0, 0};
    if (sscanf(force, "%d,%d,%d,%d", &f.MT, &f.NT, &f.CC, &f.SK) == 4) { *ok = 1; return f; }
  }
  for (int MT : kMTs) {
    if (MT > mblocks && MT != 1) continue;
    const int nMT = e2_cdiv(mblocks, MT);
    for (int NT = 1; NT <= 2; ++NT) {
      const int BN = 64 * NT;
      const int nPT = (int)((Q + BN - 1) / BN);
      for (int CC = 8; CC >= 4; CC -= 4) {
        if (CC == 8 && a.Cin <= 4) continue;
        const int Lmax = (span_rows(BN, a.Wo) + a.kh - 1) * (int)a.isY + a.kw + a.Wo;
        const int Lpad = pad16mod32(Lmax);
        const int BMpad = pad16mod32(16 * MT);
        const size_t lds = ((size_t)CC * Lpad + (size_t)THW * CC * BMpad) * 4;
        if (lds > 72 * 1024) continue;
        const int nChunkC = e2_cdiv(a.Cin, CC);
        const int nChunks = a.kd * nChunkC;
        const long wgs0 = (long)a.N * a.Do * nPT * nMT;
        for (int SK = 1; SK <= 8; SK *= 2) {
          if (SK > nChunks) break;
          const long wgs = wgs0 * SK;
          const int per = e2_cdiv(nChunks, SK);
          const double mfma = (double)MT * NT * THW * (CC / 4) * per * 32.0;
          const double stage = per * (CC * (double)Lmax / 256.0 * 6.0 +
                                      THW * CC * 4.0 * MT / 256.0 * 8.0 + 600.0);
          const double wg_time = mfma + stage + 1500.0 + (SK > 1 ? 400.0 : 0.0);
          const double rounds = (double)((wgs + slots - 1) / slots);
          // two co-resident work-groups share the 4 SIMDs
          const double cost = rounds * wg_time * 2.0;
          if (cost < bestCost) { bestCost = cost; best = IgemmCfg{MT, NT, CC, SK}; }
        }
      }
    }
  }
  *ok = best.MT != 0;
  return best;
}

void e2i_pack_dims(int cout, int cin, int* ciP, int* coP) {
  *ciP = ((cin + 7) / 8) * 8;
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
  int ok = 0;
  IgemmCfg c = choose_cfg(ctx, a, &ok);
  E2_REQUIRE(ok, "igemm: no tiling fits LDS (Cin=%d Cout=%d k=%dx%dx%d W=%d)", a.Cin,
             a.Cout, a.kd, a.kh, a.kw, a.Wo);
  IgemmP p;
  p.in = a.in; p.wp = a.wp; p.out = a.out;
  p.Cin = a.Cin; p.Cout = a.Cout; p.kd = a.kd; p.kh = a.kh; p.kw = a.kw;
  p.THW = a.kh * a.kw;
  p.Do = a.Do; p.Ho = a.Ho; p.Wo = a.Wo; p.Q = a.Ho * a.Wo;
  p.isN = a.isN; p.isC = a.isC; p.isZ = a.isZ; p.isY = a.isY;
  p.osN = a.osN; p.osC = a.osC; p.osZ = a.osZ; p.osY = a.osY;
  p.ciP = a.ciP; p.coP = a.coP;
  p.CC = c.CC; p.log2CC = (c.CC == 8) ? 3 : 2;
  const int BN = 64 * c.NT;
  const int Lmax = (span_rows(BN, a.Wo) + a.kh - 1) * (int)a.isY + a.kw + a.Wo;
  p.Lpad = pad16mod32(Lmax);
  p.BMpad = pad16mod32(16 * c.MT);
  p.nPT = e2_cdiv(p.Q, BN);
  p.nMT = e2_cdiv(e2_cdiv(a.Cout, 16), c.MT);
  p.splitK = c.SK;
  p.nChunkC = e2_cdiv(a.Cin, c.CC);
  p.atomic = (c.SK > 1) ? 1 : 0;
  p.upz = a.upz; p.upy = a.upy; p.upx = a.upx;
  E2_REQUIRE(p.nMT * 16 * c.MT <= a.coP, "igemm: packed coP too small");
  E2_REQUIRE(p.nChunkC * c.CC <= a.ciP, "igemm: packed ciP too small");
  E2_REQUIRE(a.isY < (1 << 20), "igemm: input row stride too large");
  const size_t lds = ((size_t)p.CC * p.Lpad + (size_t)p.THW * p.CC * p.BMpad) * 4;
  const long grid = (long)a.N * p.splitK * p.nMT * p.Do * p.nPT;
  E2_REQUIRE(grid < (1L << 31), "igemm: grid too large");
  if (p.atomic) {
    // split-K accumulates: the caller-visible result must start from zero.
    // Output views of this library are dense in (d,h,w) per channel in the
    // fwd/dgrad use; zero row by row to stay correct for strided views.
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
