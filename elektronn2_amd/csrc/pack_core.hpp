// pack_core.hpp -- device side of the weight repack (conv_igemm.hip holds the host side and
// the format): shared with conv_first_mfma.hip, whose first-layer forward kernel carries the
// repack of ALL layers in the same launch (the two are independent, both latency-bound).
#pragma once
#include "common.hpp"

// all layers' images in one launch (blockIdx.y = job): the repack of every conv
// weight tensor after an optimiser step costs one kernel instead of 2 per layer
struct PackDiv { unsigned d, m, sh; };       // n / d == umulhi(n, m) >> sh  (n < 2^31)
static inline PackDiv mk_pack_div(unsigned d) {
  PackDiv f; f.d = d;
  if (d <= 1) { f.m = 0; f.sh = 0; return f; }
  unsigned l = 0;
  while ((1ull << l) < d) ++l;
  f.m = (unsigned)(((1ull << (31 + l)) + d - 1) / d);
  f.sh = l - 1;
  return f;
}
__device__ __forceinline__ int pdiv(int n, const PackDiv& f) {
  return f.d <= 1 ? n : (int)(__umulhi((unsigned)n, f.m) >> f.sh);
}
struct PackJobDev {
  const float* w;
  float* wp;
  int Cout, Cin, kd, THW;
  long wsO, wsI;
  int flip, ciP, coP;
  long total;
  PackDiv dT, dKT, d32T, dTHW;       // divisors of the tiled repack (T, ICT*T, 32*T, THW)
  int up, Rout, Rin;                 // up: an UpConv image (e2_pack_job_fill modes 2 / 3) -- the
                                     // sub-position r of w[co][ci][r] folded into the row / k index
};
// The same through LDS tiles (32 output channels x a few input channels x all taps):
// the weight tensor is read along its contiguous axis and the image is written along
// ITS contiguous axis (oc).  The plain gather above reads one 4-byte element per cache
// line -- 16-32x the bytes through L2 -- and took 134 us per step for unet3d_lite.
constexpr int kPackTileFloats = 8192;
// input channels per tile: ~32 (ic, tap) pairs x 32 output channels.  (Round 3 tried ~128
// pairs per tile -- a quarter of the tiles, four times the work between two barriers:
// neuro3d's repack went from 37 to 41 us, so the tile count is not what bounds it.)
__host__ __device__ inline int e2_pack_ict(int T) {
  const int ict = (32 + T - 1) / T;
  return ict < 1 ? 1 : ict;
}
// block b of nb of ONE job (256 threads; `tile`: kPackTileFloats floats of LDS)
__device__ __forceinline__ void e2_pack_job_tiles(const PackJobDev& j, float* tile, int b, int nb) {
  if (j.up) {
    // UpConv (one tap, Wp[cg][qd][oc']): the whole image, as pack_weights_kernel writes it
    for (long i = b * 256L + threadIdx.x; i < j.total; i += (long)nb * 256) {
      const int oc = (int)(i % j.coP);
      const int ic = (int)(i / j.coP);
      float v = 0.f;
      if (oc < j.Cout && ic < j.Cin)
        v = j.w[(long)(oc / j.Rout) * j.wsO + (long)(ic / j.Rin) * j.wsI + (oc % j.Rout) + (ic % j.Rin)];
      j.wp[i] = v;
    }
    return;
  }
  const int T = j.kd * j.THW;
  const int nCG = j.ciP >> 2;
  const int nCGw = min(nCG, ((j.Cin + 3) >> 2) + 4);
  const int coW = min(j.coP, ((j.Cout + 15) / 16) * 16 + 96);
  const int icW = 4 * nCGw;
  const int ICT = e2_pack_ict(T);                     // input channels per tile
  const int KT = ICT * T;                             // (ic, tap) pairs per tile
  const int nOT = (coW + 31) >> 5, nIT = (icW + ICT - 1) / ICT;
  const bool oc_major = (j.wsI == T);                 // forward image: w[oc][ic][tap] contiguous in (ic, tap)
  const int tid = threadIdx.x;
  for (int tl_ = b; tl_ < nOT * nIT; tl_ += nb) {
    const int ot = tl_ % nOT, it = tl_ / nOT;
    const int oc0 = ot * 32, ic0 = it * ICT;
    // ---- read: consecutive threads walk the tensor's contiguous axis -----------------
    for (int e = tid; e < 32 * KT; e += 256) {
      int ol, k;                                      // local oc, local (ic, source tap)
      if (oc_major) { ol = pdiv(e, j.dKT); k = e - ol * KT; }
      else { const int il = pdiv(e, j.d32T); const int r = e - il * (32 * T);
             ol = pdiv(r, j.dT); k = il * T + (r - ol * T); }
      const int il = pdiv(k, j.dT), ts = k - il * T;
      const int oc = oc0 + ol, ic = ic0 + il;
      float v = 0.f;
      if (oc < j.Cout && ic < j.Cin) v = j.w[(long)oc * j.wsO + (long)ic * j.wsI + ts];
      tile[k * 33 + ol] = v;                          // (k stride 33: conflict-free both ways)
    }
    __syncthreads();
    // ---- write: 32 consecutive output channels per (ic, tap) -------------------------
    for (int f = tid; f < 32 * KT; f += 256) {
      const int ol = f & 31, kk = f >> 5;
      const int il = pdiv(kk, j.dT), tl = kk - il * T;   // image tap index
      const int ts = j.flip ? (T - 1 - tl) : tl;      // ... comes from this tensor tap
      const int oc = oc0 + ol, ic = ic0 + il;
      if (oc < coW && ic < icW) {
        const int dz = pdiv(tl, j.dTHW), t = tl - dz * j.THW;
        const int cg = ic >> 2, qd = ic & 3;
        j.wp[((((long)dz * nCG + cg) * j.THW + t) * 4 + qd) * j.coP + oc] = tile[(il * T + ts) * 33 + ol];
      }
    }
    __syncthreads();
  }
}

