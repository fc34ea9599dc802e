// conv_first.hip -- the FIRST layer of the neuro3d nets: Cin = 1, kd = 1,
// (kh,kw) taps, pooling (1,py,px), bias, activation -- fused.
//
// Reference ops: Conv._make_output conv -> pool -> +b -> act (neural.py:662-712)
// on the raw input, and T.grad of it wrt w and b (model.py:182).  With one
// input channel the layer is 0.5-1 GF but its conv output is the largest tensor
// of the net (59.6 MB fp32 for C-lite@183): as an implicit GEMM it is all
// padding (K = 16 taps) and all HBM traffic.  Here:
//   forward : one lane per POOLED output; the (py+kh-1)x(px+kw-1) input window
//             lives in registers, the flipped taps come in as scalar (SGPR)
//             operands, the py*px conv values are maxed, biased and activated
//             in registers; only the pooled output (15 MB) is written.
//   backward: the conv values are RECOMPUTED from the same window (no 59.6 MB
//             read), ties handled like Theano's MaxPoolGrad (every element equal
//             to the window max gets the gradient, relu'(0) = 0.5); the lane's
//             dw[tap] contributions are reduced wave-shuffle -> LDS -> one global
//             atomic per (co,tap) per work-group; work-groups are persistent so
//             the atomic count stays ~256 per address.
// VALU kernels: bounded by HBM (read 3 MB, write 15 MB / read 3+15 MB).
#include "common.hpp"
#include <algorithm>

struct First {
  const float* x;      // (n,1,d,h,w) view
  const float* w;      // [cout][1][1][kh][kw] dense
  const float* bias;
  float* out;          // forward: pooled output; backward: unused
  const float* dout;   // backward
  float* dw;           // backward, accumulated atomically (pre-zeroed)
  float* dbias;
  int N, Cout, D, Ho, Wo;        // pooled output dims
  long xsN, xsD, xsH;
  long osN, osC, osD, osH;       // strides of out / dout
  int act;
  int tilesX, tilesY;            // tiles of 32 x 8 pooled outputs
};

__device__ __forceinline__ float ff_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// conv value at window offset (a, b):  sum_t w[T-1-t] * win[a+ty][b+tx]   (F1: flipped)
template <int KH, int KW, int PY, int PX>
__device__ __forceinline__ float first_conv(const float (&win)[PY + KH - 1][PX + KW - 1],
                                            const float* __restrict__ wc, int a, int b) {
  float s = 0.f;
#pragma unroll
  for (int ty = 0; ty < KH; ++ty)
#pragma unroll
    for (int tx = 0; tx < KW; ++tx)
      s = fmaf(wc[KH * KW - 1 - (ty * KW + tx)], win[a + ty][b + tx], s);
  return s;
}

template <int KH, int KW, int PY, int PX>
__device__ __forceinline__ bool first_load_window(const First& p, int tile, int lane_x, int row,
                                                  float (&win)[PY + KH - 1][PX + KW - 1],
                                                  int& n, int& z, int& yo, int& xo) {
  const int tx_ = tile % p.tilesX;
  int r = tile / p.tilesX;
  const int ty_ = r % p.tilesY; r /= p.tilesY;
  z = r % p.D;
  n = r / p.D;
  xo = tx_ * 32 + lane_x;
  yo = ty_ * 8 + row;
  const bool valid = xo < p.Wo && yo < p.Ho;
  const int xc = valid ? xo : 0, yc = valid ? yo : 0;
  const float* src = p.x + (long)n * p.xsN + (long)z * p.xsD + (long)(yc * PY) * p.xsH + xc * PX;
#pragma unroll
  for (int i = 0; i < PY + KH - 1; ++i)
#pragma unroll
    for (int j = 0; j < PX + KW - 1; ++j) win[i][j] = src[(long)i * p.xsH + j];
  return valid;
}

template <int KH, int KW, int PY, int PX>
__global__ __launch_bounds__(256) void first_fwd_kernel(First p, int nTiles) {
  const int lane_x = threadIdx.x & 31, row = threadIdx.x >> 5;
  for (int tile = blockIdx.x; tile < nTiles; tile += gridDim.x) {
    float win[PY + KH - 1][PX + KW - 1];
    int n, z, yo, xo;
    const bool valid = first_load_window<KH, KW, PY, PX>(p, tile, lane_x, row, win, n, z, yo, xo);
    float* ob = p.out + (long)n * p.osN + (long)z * p.osD + (long)yo * p.osH + xo;
    for (int co = 0; co < p.Cout; ++co) {
      const float* wc = p.w + co * (KH * KW);
      float m = -INFINITY;
#pragma unroll
      for (int a = 0; a < PY; ++a)
#pragma unroll
        for (int b = 0; b < PX; ++b) m = fmaxf(m, first_conv<KH, KW, PY, PX>(win, wc, a, b));
      float v = m + p.bias[co];
      if (p.act == E2_ACT_RELU) v = fmaxf(v, 0.f);
      if (valid) ob[(long)co * p.osC] = v;
    }
  }
}

template <int KH, int KW, int PY, int PX>
__global__ __launch_bounds__(256) void first_bwd_kernel(First p, int nTiles) {
  constexpr int T = KH * KW;
  __shared__ float red[4][T + 1];
  const int lane_x = threadIdx.x & 31, row = threadIdx.x >> 5;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // channel-outer: the lane keeps its dw[tap] partial sums of ONE output channel
  // in registers over all of the work-group's tiles (windows are re-read from
  // L1/L2), so the cross-lane reduction runs once per channel, not per tile.
  for (int co = 0; co < p.Cout; ++co) {
    const float* wc = p.w + co * T;
    const float bv = p.bias[co];
    float dwa[KH][KW];
#pragma unroll
    for (int ty = 0; ty < KH; ++ty)
#pragma unroll
      for (int tx = 0; tx < KW; ++tx) dwa[ty][tx] = 0.f;
    float gsum = 0.f;
    for (int tile = blockIdx.x; tile < nTiles; tile += gridDim.x) {
      float win[PY + KH - 1][PX + KW - 1];
      int n, z, yo, xo;
      const bool valid = first_load_window<KH, KW, PY, PX>(p, tile, lane_x, row, win, n, z, yo, xo);
      float c[PY][PX];
      float m = -INFINITY;
#pragma unroll
      for (int a = 0; a < PY; ++a)
#pragma unroll
        for (int b = 0; b < PX; ++b) {
          c[a][b] = first_conv<KH, KW, PY, PX>(win, wc, a, b);
          m = fmaxf(m, c[a][b]);
        }
      float g = valid ? p.dout[(long)n * p.osN + (long)co * p.osC + (long)z * p.osD +
                               (long)yo * p.osH + xo] : 0.f;
      if (p.act == E2_ACT_RELU) {
        const float pre = m + bv;
        g *= (pre > 0.f) ? 1.f : ((pre == 0.f) ? 0.5f : 0.f);
      }
      gsum += g;
#pragma unroll
      for (int a = 0; a < PY; ++a)
#pragma unroll
        for (int b = 0; b < PX; ++b) {
          const float gm = (c[a][b] == m) ? g : 0.f;     // every tied maximum gets the gradient
#pragma unroll
          for (int ty = 0; ty < KH; ++ty)
#pragma unroll
            for (int tx = 0; tx < KW; ++tx) dwa[ty][tx] = fmaf(gm, win[a + ty][b + tx], dwa[ty][tx]);
        }
    }
    // wave shuffle -> LDS -> one global atomic per (co, tap) per work-group
#pragma unroll
    for (int ty = 0; ty < KH; ++ty)
#pragma unroll
      for (int tx = 0; tx < KW; ++tx) {
        const float s = ff_wave_sum(dwa[ty][tx]);
        if (lane == 0) red[wave][ty * KW + tx] = s;
      }
    const float gs = ff_wave_sum(gsum);
    if (lane == 0) red[wave][T] = gs;
    __syncthreads();
    if (threadIdx.x <= T) {
      const float v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] +
                      red[3][threadIdx.x];
      if (v != 0.f) {
        // tap position t  <->  weight index T-1-t (flip, F1)
        if (threadIdx.x < T) unsafeAtomicAdd(p.dw + co * T + (T - 1 - threadIdx.x), v);
        else unsafeAtomicAdd(p.dbias + co, v);
      }
    }
    __syncthreads();
  }
}

// ---- host -------------------------------------------------------------------------
static int first_supported(int kd, int kh, int kw, int pz, int py, int px) {
  if (kd != 1 || pz != 1) return 0;
  if (kh == 4 && kw == 4 && py == 2 && px == 2) return 1;
  if (kh == 6 && kw == 6 && py == 2 && px == 2) return 2;
  return 0;
}

extern "C" int e2_conv1_supported(int cin, int kd, int kh, int kw, int pz, int py, int px) {
  return cin == 1 && first_supported(kd, kh, kw, pz, py, px) != 0;
}

static int first_fill(First& p, const e2_tensor5* x, const e2_tensor5* o, int cout, int kh,
                      int kw, int py, int px, const char* name) {
  E2_REQUIRE(x && x->ptr && o && o->ptr, "%s: null tensor", name);
  E2_REQUIRE(x->c == 1, "%s: needs exactly one input channel", name);
  E2_REQUIRE(o->n == x->n && o->c == cout && o->d == x->d &&
                 o->h == (x->h - kh + 1) / py && o->w == (x->w - kw + 1) / px &&
                 (x->h - kh + 1) % py == 0 && (x->w - kw + 1) % px == 0,
             "%s: pooled output is (%d,%d,%d,%d,%d), input (%d,1,%d,%d,%d), kernel %dx%d pool %dx%d",
             name, o->n, o->c, o->d, o->h, o->w, x->n, x->d, x->h, x->w, kh, kw, py, px);
  p.x = x->ptr;
  p.N = x->n; p.Cout = cout; p.D = x->d; p.Ho = o->h; p.Wo = o->w;
  p.xsN = x->sn; p.xsD = x->sd; p.xsH = x->sh;
  p.osN = o->sn; p.osC = o->sc; p.osD = o->sd; p.osH = o->sh;
  p.tilesX = e2_cdiv(o->w, 32);
  p.tilesY = e2_cdiv(o->h, 8);
  return 0;
}

extern "C" int e2_conv1_pool_act_fwd(e2_ctx* ctx, const e2_tensor5* x, const float* w,
                                     const float* bias, int cout, int kh, int kw, int py,
                                     int px, int act, const e2_tensor5* out) {
  E2_REQUIRE(ctx && w && bias, "conv1_pool_act_fwd: null argument");
  const int v = first_supported(1, kh, kw, 1, py, px);
  E2_REQUIRE(v, "conv1_pool_act_fwd: unsupported kernel/pool %dx%d / %dx%d", kh, kw, py, px);
  First p{};
  if (int rc = first_fill(p, x, out, cout, kh, kw, py, px, "conv1_pool_act_fwd")) return rc;
  p.w = w; p.bias = bias; p.out = out->ptr; p.act = act;
  const long nTiles = (long)p.N * p.D * p.tilesY * p.tilesX;
  const int grid = (int)std::min<long>(nTiles, ctx->num_cu * 8);
  if (v == 1)
    hipLaunchKernelGGL((first_fwd_kernel<4, 4, 2, 2>), dim3(grid), dim3(256), 0, ctx->stream, p, (int)nTiles);
  else
    hipLaunchKernelGGL((first_fwd_kernel<6, 6, 2, 2>), dim3(grid), dim3(256), 0, ctx->stream, p, (int)nTiles);
  E2_CHECK_HIP(hipGetLastError());
  return 0;
}

/* dw (cout*kh*kw) and dbias (cout) are ACCUMULATED into: zero them first. */
extern "C" int e2_conv1_pool_act_bwd(e2_ctx* ctx, const e2_tensor5* x, const float* w,
                                     const float* bias, const e2_tensor5* dout, int kh, int kw,
                                     int py, int px, int act, float* dw, float* dbias) {
  E2_REQUIRE(ctx && w && bias && dw && dbias, "conv1_pool_act_bwd: null argument");
  const int v = first_supported(1, kh, kw, 1, py, px);
  E2_REQUIRE(v, "conv1_pool_act_bwd: unsupported kernel/pool %dx%d / %dx%d", kh, kw, py, px);
  First p{};
  if (int rc = first_fill(p, x, dout, dout ? dout->c : 0, kh, kw, py, px, "conv1_pool_act_bwd"))
    return rc;
  p.w = w; p.bias = bias; p.dout = dout->ptr; p.dw = dw; p.dbias = dbias; p.act = act;
  const long nTiles = (long)p.N * p.D * p.tilesY * p.tilesX;
  const int grid = (int)std::min<long>(nTiles, ctx->num_cu * 2);
  const size_t lds = 0;
  if (v == 1)
    hipLaunchKernelGGL((first_bwd_kernel<4, 4, 2, 2>), dim3(grid), dim3(256), lds, ctx->stream, p, (int)nTiles);
  else
    hipLaunchKernelGGL((first_bwd_kernel<6, 6, 2, 2>), dim3(grid), dim3(256), lds, ctx->stream, p, (int)nTiles);
  E2_CHECK_HIP(hipGetLastError());
  return 0;
}
