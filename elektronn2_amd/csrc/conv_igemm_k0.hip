// conv_igemm_k0.hip -- implicit-GEMM convolution, GENERIC kernel width (runtime tap
// loop).  Used for widths without a specialised instance (kw not in {1,3,4,5}).
// Both operands go through LDS (weights rows + input spans, LDS-DMA double
// buffer, four waves); see conv_igemm.hip / igemm_core.hpp for the fast path.
#include "igemm_core.hpp"

constexpr int igemm_bmpad(int MT) {          // row stride == 16 (mod 32)
  return ((16 * MT) & 31) == 16 ? 16 * MT : 16 * MT + 16;
}

template <int MT, int NT>
__global__ __launch_bounds__(256, 2) void igemm_generic_kernel(IgemmP p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int BM = 16 * MT, BN = 64 * NT;
  constexpr int BMpad = igemm_bmpad(MT);
  constexpr int BMp4 = BMpad / 4;
  const int kw = p.kw;
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

  int c_lo, c_hi;
  e2_chunk_range(p, z, c_lo, c_hi);             // (data gradient: border-only tap planes skipped)
  const int per = (c_hi - c_lo + p.splitK - 1) / p.splitK;
  const int cb = c_lo + ks * per, ce = min(cb + per, c_hi);
  if (cb >= ce && !p.parts) return;
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
    __syncthreads();
    if (ch + 1 < ce) stage(ch + 1, cur ^ 1);
    const float* xl = smem + cur * p.bufFloats;
    const float* wl = xl + xFloats + aBase;
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
          dst = p.out + (long)ks * p.partStride + (long)n * p.osN + (long)co * p.osC + (long)z * p.osZ +
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

template <int MT, int NT>
static int launch_generic(e2_ctx* ctx, const IgemmP& p, int grid, size_t lds) {
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(
        reinterpret_cast<const void*>(&igemm_generic_kernel<MT, NT>),
        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) { e2_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return 1; }
    attr_done = true;
  }
  hipLaunchKernelGGL((igemm_generic_kernel<MT, NT>), dim3(grid), dim3(256), lds, ctx->stream, p);
  E2_CHECK_HIP(hipGetLastError());
  return 0;
}

int e2i_igemm_launch_generic(e2_ctx* ctx, const IgemmP& p, int MT, int NT, int grid, size_t lds) {
#define E2_CASE(M)                                                      \
  case M:                                                               \
    if (NT == 1) return launch_generic<M, 1>(ctx, p, grid, lds);        \
    if (NT == 2) return launch_generic<M, 2>(ctx, p, grid, lds);        \
    break;
#define E2_CASE4(M)                                                     \
  case M:                                                               \
    if (NT == 1) return launch_generic<M, 1>(ctx, p, grid, lds);        \
    if (NT == 2) return launch_generic<M, 2>(ctx, p, grid, lds);        \
    if (NT == 4) return launch_generic<M, 4>(ctx, p, grid, lds);        \
    break;
  switch (MT) {
    E2_CASE4(1) E2_CASE4(2) E2_CASE4(3) E2_CASE4(4) E2_CASE4(5) E2_CASE(6)
    E2_CASE(7) E2_CASE(8) E2_CASE(10) E2_CASE(13)
  }
#undef E2_CASE
#undef E2_CASE4
  e2_set_error("igemm: no generic instance MT=%d NT=%d", MT, NT);
  return 2;
}
