// conv_first.hip -- the FIRST layer of the neuro3d nets / the U-Net stem: Cin = 1, kd = 1,
// (kh,kw) taps, pooling (1,py,px) (or none), bias, activation -- fused.
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
//             dw[tap] contributions are reduced with a transposing butterfly
//             (15 cross-lane moves per 16 taps) -> LDS -> 17 sums per (tile,
//             channel) in a workspace; a second tiny kernel adds the tiles up
//             (contended same-line atomics were the cost of the first version).
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

// Transposing butterfly: every lane holds T = 16 partial sums v[0..15] (one per
// tap); after 4 exchange steps lane l holds in v[0] the sum over its 16-lane group
// of tap bitrev4(l & 15).  15 cross-lane moves instead of 16 x 4.
template <int MASK, int N>
__device__ __forceinline__ void ff_bfly_step(float (&v)[16], int lane) {
  const bool up = (lane & MASK) != 0;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const float keep = up ? v[i + N] : v[i];
    const float send = up ? v[i] : v[i + N];
    v[i] = keep + __shfl_xor(send, MASK, 64);
  }
}
__device__ __forceinline__ void ff_butterfly16(float (&v)[16], int lane) {
  ff_bfly_step<1, 8>(v, lane);
  ff_bfly_step<2, 4>(v, lane);
  ff_bfly_step<4, 2>(v, lane);
  ff_bfly_step<8, 1>(v, lane);
}

// One work-group = one tile of 32 x 8 pooled outputs, all channels: the window is
// loaded once, the lane's dw[tap] contributions of a channel are reduced with the
// butterfly -> LDS -> 17 sums per (tile, channel) stored to the workspace
// part[tile][co][T+1]; first_bwd_reduce_kernel adds the tiles up.
template <int KH, int KW, int PY, int PX>
__global__ __launch_bounds__(256) void first_bwd_kernel(First p, float* __restrict__ part) {
  constexpr int T = KH * KW;
  static_assert(T == 9 || T == 16 || T == 36, "tap count");
  __shared__ float red[2][16][T + 1];
  const int lane_x = threadIdx.x & 31, row = threadIdx.x >> 5;
  const int lane = threadIdx.x & 63;
  const int grp = threadIdx.x >> 4;                 // 16 groups of 16 lanes
  const int tile = blockIdx.x;
  float win[PY + KH - 1][PX + KW - 1];
  int n, z, yo, xo;
  const bool valid = first_load_window<KH, KW, PY, PX>(p, tile, lane_x, row, win, n, z, yo, xo);
  const float* gp = p.dout + (long)n * p.osN + (long)z * p.osD + (long)yo * p.osH + xo;
  for (int co = 0; co < p.Cout; ++co) {
    const float* wc = p.w + co * T;
    const float bv = p.bias[co];
    float c[PY][PX];
    float m = -INFINITY;
#pragma unroll
    for (int a = 0; a < PY; ++a)
#pragma unroll
      for (int b = 0; b < PX; ++b) {
        c[a][b] = first_conv<KH, KW, PY, PX>(win, wc, a, b);
        m = fmaxf(m, c[a][b]);
      }
    float g = valid ? gp[(long)co * p.osC] : 0.f;
    if (p.act == E2_ACT_RELU) {
      const float pre = m + bv;
      g *= (pre > 0.f) ? 1.f : ((pre == 0.f) ? 0.5f : 0.f);
    }
    float (&rb)[16][T + 1] = red[co & 1];
    // taps in chunks of 16: dw partials of the lane, then the transposing butterfly
#pragma unroll
    for (int t0 = 0; t0 < T; t0 += 16) {
      float v[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int t = t0 + i;
        float acc = 0.f;
        if (t < T) {
          const int ty = t / KW, tx = t % KW;
#pragma unroll
          for (int a = 0; a < PY; ++a)
#pragma unroll
            for (int b = 0; b < PX; ++b)      // every tied maximum gets the gradient
              acc = fmaf((c[a][b] == m) ? g : 0.f, win[a + ty][b + tx], acc);
        }
        v[i] = acc;
      }
      ff_butterfly16(v, lane);
      const int l15 = lane & 15;
      const int tap = t0 + ((l15 & 1) << 3 | (l15 & 2) << 1 | (l15 & 4) >> 1 | (l15 & 8) >> 3);
      if (tap < T) rb[grp][tap] = v[0];
    }
    float gs = g;
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) gs += __shfl_xor(gs, o, 64);
    if ((lane & 15) == 0) rb[grp][T] = gs;
    __syncthreads();
    if (threadIdx.x <= T) {
      float sum = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) sum += rb[k][threadIdx.x];
      part[((long)tile * p.Cout + co) * (T + 1) + threadIdx.x] = sum;
    }
    // red[] is double buffered: the next channel writes the other half, and the
    // barrier of that channel orders this read before the half is written again
  }
}

// dw[co][T-1-t] += sum_tiles part[tile][co][t];  dbias[co] += sum_tiles part[tile][co][T]
__global__ __launch_bounds__(256) void first_bwd_reduce_kernel(const float* __restrict__ part,
                                                               int nTiles, int Cout, int T,
                                                               float* dw, float* dbias) {
  const int per = (nTiles + gridDim.y - 1) / gridDim.y;
  const int t0 = blockIdx.y * per, t1 = min(t0 + per, nTiles);
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int total = Cout * (T + 1);
  if (idx >= total) return;
  float s = 0.f;
  for (int t = t0; t < t1; ++t) s += part[(long)t * total + idx];
  const int co = idx / (T + 1), k = idx - co * (T + 1);
  if (s != 0.f) {
    if (k < T) unsafeAtomicAdd(dw + co * T + (T - 1 - k), s);    // tap t <-> weight index T-1-t (flip, F1)
    else unsafeAtomicAdd(dbias + co, s);
  }
}

// the same for slot-major partial sums part[co][T + 1][slot] (conv_first_mfma.hip): one wave per
// element, its slots one contiguous run, ONE writer per element -- no atomics, fixed order
// (the strided form above: 12.8 us for neuro3d's 740 x 768 sums)
__global__ __launch_bounds__(256) void first_bwd_reduce_sm_kernel(const float* __restrict__ part,
                                                                  int nSlots, int Cout, int T,
                                                                  float* dw, float* dbias) {
  const int idx = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (idx >= Cout * (T + 1)) return;
  const float* row = part + (long)idx * nSlots;
  float s = 0.f;
#pragma unroll 4
  for (int b = lane; b < nSlots; b += 64) s += row[b];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane != 0) return;
  const int co = idx / (T + 1), k = idx - co * (T + 1);
  if (k < T) dw[co * T + (T - 1 - k)] += s;            // tap t <-> weight index T-1-t (flip, F1)
  else dbias[co] += s;
}

// ---- host -------------------------------------------------------------------------
// conv_first_mfma.hip: the same layer on the matrix cores (Cout <= 32)
int e2i_firstm_mg(int cout);
size_t e2i_firstm_ws_floats(long nTiles, int cout, int T);
int e2i_firstm_fwd(e2_ctx*, int v, const e2_tensor5* x, const float* w, const float* bias, int cout,
                   int py, int px, int act, const e2_tensor5* out, void* next_xb, int next_kg);
int e2i_firstm_bwd(e2_ctx*, int v, const e2_tensor5* x, const float* w, const float* bias,
                   const e2_tensor5* dout, int py, int px, int act, float* part, int* nslots);

static int first_supported(int kd, int kh, int kw, int pz, int py, int px) {
  if (kd != 1 || pz != 1) return 0;
  if (kh == 4 && kw == 4 && py == 2 && px == 2) return 1;
  if (kh == 6 && kw == 6 && py == 2 && px == 2) return 2;
  if (kh == 3 && kw == 3 && py == 1 && px == 1) return 3;     // U-Net stem: no pooling
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

static int conv1_fwd(e2_ctx* ctx, const e2_tensor5* x, const float* w, const float* bias, int cout,
                     int kh, int kw, int py, int px, int act, const e2_tensor5* out, void* next_xb,
                     int next_kg);

extern "C" int e2_conv1_pool_act_fwd(e2_ctx* ctx, const e2_tensor5* x, const float* w,
                                     const float* bias, int cout, int kh, int kw, int py,
                                     int px, int act, const e2_tensor5* out) {
  return conv1_fwd(ctx, x, w, bias, cout, kh, kw, py, px, act, out, nullptr, 0);
}

// ... and the channels-last bf16 copy of out for the next conv layer (bf16 mode, SURVEY.md 8f-3;
// include/e2hip.h "the producers' epilogues"): matrix-core form only (cout <= 32)
extern "C" int e2_conv1_pool_act_fwd_bf16(e2_ctx* ctx, const e2_tensor5* x, const float* w,
                                          const float* bias, int cout, int kh, int kw, int py,
                                          int px, int act, const e2_tensor5* out, void* next_xb,
                                          int next_kg) {
  E2_REQUIRE(next_xb, "conv1_pool_act_fwd_bf16: null copy");
  return conv1_fwd(ctx, x, w, bias, cout, kh, kw, py, px, act, out, next_xb, next_kg);
}

static int conv1_fwd(e2_ctx* ctx, const e2_tensor5* x, const float* w, const float* bias, int cout,
                     int kh, int kw, int py, int px, int act, const e2_tensor5* out, void* next_xb,
                     int next_kg) {
  E2_REQUIRE(ctx && w && bias, "conv1_pool_act_fwd: null argument");
  const int v = first_supported(1, kh, kw, 1, py, px);
  E2_REQUIRE(v, "conv1_pool_act_fwd: unsupported kernel/pool %dx%d / %dx%d", kh, kw, py, px);
  First p{};
  if (int rc = first_fill(p, x, out, cout, kh, kw, py, px, "conv1_pool_act_fwd")) return rc;
  p.w = w; p.bias = bias; p.out = out->ptr; p.act = act;
  if (e2i_firstm_mg(cout) && (next_xb || !e2_dbg_env("E2_FIRST_VALU")))
    return e2i_firstm_fwd(ctx, v, x, w, bias, cout, py, px, act, out, next_xb, next_kg);
  E2_REQUIRE(!next_xb, "conv1_pool_act_fwd_bf16: %d output channels (matrix-core form: <= 32)", cout);
  const long nTiles = (long)p.N * p.D * p.tilesY * p.tilesX;
  const int grid = (int)std::min<long>(nTiles, ctx->num_cu * 8);
  if (v == 1)
    hipLaunchKernelGGL((first_fwd_kernel<4, 4, 2, 2>), dim3(grid), dim3(256), 0, ctx->stream, p, (int)nTiles);
  else if (v == 2)
    hipLaunchKernelGGL((first_fwd_kernel<6, 6, 2, 2>), dim3(grid), dim3(256), 0, ctx->stream, p, (int)nTiles);
  else
    hipLaunchKernelGGL((first_fwd_kernel<3, 3, 1, 1>), dim3(grid), dim3(256), 0, ctx->stream, p, (int)nTiles);
  E2_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" size_t e2_conv1_bwd_workspace_bytes(int n, int cout, int d, int ho, int wo, int kh,
                                               int kw) {
  const long tiles = (long)n * d * e2_cdiv(ho, 8) * e2_cdiv(wo, 32);
  const long tilesM = (long)n * d * e2_cdiv(ho, 4) * e2_cdiv(wo * 2, 64);     // (px <= 2)
  const size_t valu = (size_t)tiles * cout * (kh * kw + 1);
  return sizeof(float) * std::max(valu, e2i_firstm_ws_floats(tilesM, cout, kh * kw));
}

/* dw (cout*kh*kw) and dbias (cout) are ACCUMULATED into: zero them first.
 * ws: e2_conv1_bwd_workspace_bytes(n, cout, d, ho, wo, kh, kw) bytes, (ho, wo) = pooled dims. */
extern "C" int e2_conv1_pool_act_bwd(e2_ctx* ctx, const e2_tensor5* x, const float* w,
                                     const float* bias, const e2_tensor5* dout, int kh, int kw,
                                     int py, int px, int act, float* dw, float* dbias,
                                     void* ws, size_t ws_bytes) {
  E2_REQUIRE(ctx && w && bias && dw && dbias && ws, "conv1_pool_act_bwd: null argument");
  const int v = first_supported(1, kh, kw, 1, py, px);
  E2_REQUIRE(v, "conv1_pool_act_bwd: unsupported kernel/pool %dx%d / %dx%d", kh, kw, py, px);
  First p{};
  if (int rc = first_fill(p, x, dout, dout ? dout->c : 0, kh, kw, py, px, "conv1_pool_act_bwd"))
    return rc;
  p.w = w; p.bias = bias; p.dout = dout->ptr; p.dw = dw; p.dbias = dbias; p.act = act;
  const long nTiles = (long)p.N * p.D * p.tilesY * p.tilesX;
  E2_REQUIRE(nTiles < (1L << 31), "conv1_pool_act_bwd: too many tiles");
  E2_REQUIRE(ws_bytes >= e2_conv1_bwd_workspace_bytes(p.N, p.Cout, p.D, p.Ho, p.Wo, kh, kw),
             "conv1_pool_act_bwd: workspace too small");
  float* part = (float*)ws;
  if (e2i_firstm_mg(p.Cout) && !e2_dbg_env("E2_FIRST_VALU")) {
    int nslots = 0;
    if (int rc = e2i_firstm_bwd(ctx, v, x, w, bias, dout, py, px, act, part, &nslots)) return rc;
    const int T = kh * kw, total = p.Cout * (T + 1);
    hipLaunchKernelGGL(first_bwd_reduce_sm_kernel, dim3(e2_cdiv(total, 4)), dim3(256), 0, ctx->stream,
                       part, nslots, p.Cout, T, dw, dbias);
    E2_CHECK_HIP(hipGetLastError());
    return 0;
  }
  if (v == 1)
    hipLaunchKernelGGL((first_bwd_kernel<4, 4, 2, 2>), dim3((int)nTiles), dim3(256), 0, ctx->stream, p, part);
  else if (v == 2)
    hipLaunchKernelGGL((first_bwd_kernel<6, 6, 2, 2>), dim3((int)nTiles), dim3(256), 0, ctx->stream, p, part);
  else
    hipLaunchKernelGGL((first_bwd_kernel<3, 3, 1, 1>), dim3((int)nTiles), dim3(256), 0, ctx->stream, p, part);
  E2_CHECK_HIP(hipGetLastError());
  const int T = kh * kw, total = p.Cout * (T + 1);
  const int slices = (int)std::min<long>(nTiles, 64);
  hipLaunchKernelGGL(first_bwd_reduce_kernel, dim3(e2_cdiv(total, 256), slices), dim3(256), 0,
                     ctx->stream, part, (int)nTiles, p.Cout, T, dw, dbias);
  E2_CHECK_HIP(hipGetLastError());
  return 0;
}
