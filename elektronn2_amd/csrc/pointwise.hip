// pointwise.hip -- the HBM-bound kernels of the step: max-pool + bias +
// activation (forward and backward), stand-alone pool, view fill / copy,
// NCDHW<->NDHWC transposes through LDS, depth-to-space helpers for UpConv,
// channel softmax + MultinoulliNLL, and the reference's Adam / SGD updates.
// All are one-element-per-lane, W-contiguous (coalesced) streaming kernels;
// reductions go wave-shuffle -> LDS -> one atomic per work-group.
#include "common.hpp"
#include <algorithm>

#define E2_EPS_NLL 1e-5f
#define E2_EPS_ADAM 1e-5f

struct View5 {
  float* p;
  int n, c, d, h, w;
  long sn, sc, sd, sh;
};
static inline View5 mk(const e2_tensor5* t) {
  return View5{t->ptr, t->n, t->c, t->d, t->h, t->w, (long)t->sn, (long)t->sc,
               (long)t->sd, (long)t->sh};
}
__device__ __forceinline__ long vidx(const View5& v, int n, int c, int z, int y, int x) {
  return (long)n * v.sn + (long)c * v.sc + (long)z * v.sd + (long)y * v.sh + x;
}

// exact unsigned division of n < 2^31 by a runtime constant (host-made magic):
// l = ceil(log2 d), m = ceil(2^(31+l) / d);  n / d == umulhi(n, m) >> (l - 1)
struct FastDiv {
  unsigned d, m, sh;
};
static inline FastDiv mk_div(unsigned d) {
  FastDiv f;
  f.d = d;
  if (d <= 1) { f.m = 0; f.sh = 0; return f; }
  unsigned l = 0;
  while ((1ull << l) < d) ++l;
  const unsigned long long num = 1ull << (31 + l);
  f.m = (unsigned)((num + d - 1) / d);
  f.sh = l - 1;
  return f;
}
__device__ __forceinline__ unsigned fdiv(unsigned n, const FastDiv& f) {
  return f.d <= 1 ? n : (__umulhi(n, f.m) >> f.sh);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}
// sum over a 256-thread block; result valid in thread 0
__device__ __forceinline__ float block_sum256(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float r = 0.f;
  if (threadIdx.x == 0) r = red[0] + red[1] + red[2] + red[3];
  __syncthreads();
  return r;
}

// ---------------------------------------------------------------------------
// fill / copy
// ---------------------------------------------------------------------------
__global__ void fill_flat_kernel(float* p, size_t n, float v) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x)
    p[i] = v;
}

// (32-bit index math with magic-number division: 64-bit % and / made these ALU-bound)
__global__ void fill_view_kernel(View5 v, float val, FastDiv dw, FastDiv dh) {
  // grid: (ceil(d*h*w/256), c, n)
  const unsigned S = (unsigned)v.d * v.h * v.w;
  const unsigned s = blockIdx.x * 256u + threadIdx.x;
  if (s >= S) return;
  const unsigned t = fdiv(s, dw);
  const unsigned x = s - t * v.w;
  const unsigned z = fdiv(t, dh);
  const unsigned y = t - z * v.h;
  v.p[vidx(v, blockIdx.z, blockIdx.y, (int)z, (int)y, (int)x)] = val;
}

__global__ void copy_view_kernel(View5 src, View5 dst, int accumulate, FastDiv dw, FastDiv dh) {
  const unsigned S = (unsigned)src.d * src.h * src.w;
  const unsigned s = blockIdx.x * 256u + threadIdx.x;
  if (s >= S) return;
  const unsigned t = fdiv(s, dw);
  const unsigned x = s - t * src.w;
  const unsigned z = fdiv(t, dh);
  const unsigned y = t - z * src.h;
  const float v = src.p[vidx(src, blockIdx.z, blockIdx.y, (int)z, (int)y, (int)x)];
  float* d = dst.p + vidx(dst, blockIdx.z, blockIdx.y, (int)z, (int)y, (int)x);
  *d = accumulate ? (*d + v) : v;
}

// ---------------------------------------------------------------------------
// max-pool (+bias +act) forward:  out = act(max_window(y) + b[c])
// grid: (chunks of the (z,y,x) output range, c, n); 32-bit index math with
// magic-number division; each thread walks its chunk with stride 256.
// ---------------------------------------------------------------------------

template <bool HAS_BIAS>
__global__ __launch_bounds__(256) void pool_fwd_kernel(View5 y, const float* __restrict__ bias,
                                                       int pz, int py, int px, int act,
                                                       View5 out, FastDiv dw, FastDiv dh,
                                                       unsigned chunk) {
  const unsigned S = (unsigned)out.d * out.h * out.w;
  const unsigned s0 = blockIdx.x * chunk;
  const unsigned s1 = min(s0 + chunk, S);
  const int c = blockIdx.y, n = blockIdx.z;
  const float bv = HAS_BIAS ? bias[c] : 0.f;
  const float* ybase = y.p + (long)n * y.sn + (long)c * y.sc;
  float* obase = out.p + (long)n * out.sn + (long)c * out.sc;
  for (unsigned s = s0 + threadIdx.x; s < s1; s += 256) {
    const unsigned t = fdiv(s, dw);
    const unsigned xo = s - t * out.w;
    const unsigned zo = fdiv(t, dh);
    const unsigned yo = t - zo * out.h;
    const float* src = ybase + (long)(zo * pz) * y.sd + (long)(yo * py) * y.sh + xo * px;
    float m = src[0];
    for (int a = 0; a < pz; ++a)
      for (int b = 0; b < py; ++b) {
        const float* row = src + a * y.sd + b * y.sh;
        for (int e = 0; e < px; ++e) m = fmaxf(m, row[e]);
      }
    float v = m + bv;
    if (act == E2_ACT_RELU) v = fmaxf(v, 0.f);
    obase[(long)zo * out.sd + (long)yo * out.sh + xo] = v;
  }
}

// backward: dy[window] = (y == max) ? dout * act'(max + b) : 0 ; dbias += sum
template <bool HAS_BIAS>
__global__ __launch_bounds__(256) void pool_bwd_kernel(View5 dout, View5 y,
                                                       const float* __restrict__ bias, int pz,
                                                       int py, int px, int act, View5 dy,
                                                       float* __restrict__ dbias,
                                                       int accumulate, FastDiv dw,
                                                       FastDiv dh, unsigned chunk) {
  __shared__ float red[4];
  const unsigned S = (unsigned)dout.d * dout.h * dout.w;
  const unsigned s0 = blockIdx.x * chunk;
  const unsigned s1 = min(s0 + chunk, S);
  const int c = blockIdx.y, n = blockIdx.z;
  const float bv = HAS_BIAS ? bias[c] : 0.f;
  const float* ybase = y.p + (long)n * y.sn + (long)c * y.sc;
  const float* gbase = dout.p + (long)n * dout.sn + (long)c * dout.sc;
  float* dbase = dy.p + (long)n * dy.sn + (long)c * dy.sc;
  float gsum = 0.f;
  for (unsigned s = s0 + threadIdx.x; s < s1; s += 256) {
    const unsigned t = fdiv(s, dw);
    const unsigned xo = s - t * dout.w;
    const unsigned zo = fdiv(t, dh);
    const unsigned yo = t - zo * dout.h;
    const float* src = ybase + (long)(zo * pz) * y.sd + (long)(yo * py) * y.sh + xo * px;
    float m = src[0];
    for (int a = 0; a < pz; ++a)
      for (int b = 0; b < py; ++b) {
        const float* row = src + a * y.sd + b * y.sh;
        for (int e = 0; e < px; ++e) m = fmaxf(m, row[e]);
      }
    float g = gbase[(long)zo * dout.sd + (long)yo * dout.sh + xo];
    if (act == E2_ACT_RELU) {
      const float pre = m + bv;
      g *= (pre > 0.f) ? 1.f : ((pre == 0.f) ? 0.5f : 0.f);
    }
    gsum += g;
    float* dst = dbase + (long)(zo * pz) * dy.sd + (long)(yo * py) * dy.sh + xo * px;
    for (int a = 0; a < pz; ++a)
      for (int b = 0; b < py; ++b) {
        const float* row = src + a * y.sd + b * y.sh;
        float* drow = dst + a * dy.sd + b * dy.sh;
        for (int e = 0; e < px; ++e) {
          const float v = (row[e] == m) ? g : 0.f;
          drow[e] = accumulate ? (drow[e] + v) : v;
        }
      }
  }
  if (dbias != nullptr) {
    const float tot = block_sum256(gsum, red);
    if (threadIdx.x == 0 && tot != 0.f) unsafeAtomicAdd(dbias + c, tot);
  }
}

// Fixed-window forms of the two kernels above for the pool shapes of the BASELINE nets
// ((1,2,2), (2,1,1), (2,2,2), and (1,1,1) = bias + activation of a layer that does not
// pool): a thread owns FOUR consecutive input x (16-byte row accesses, any alignment) =
// 4 / PX pooled outputs; the window lives in registers (read once), the loops are
// compile-time.  (The one-output-per-thread forms with 4- and 8-byte accesses ran at
// 2.2-3.1 TB/s.)
typedef float pw_f4 __attribute__((ext_vector_type(4), aligned(4)));
typedef float pw_f2 __attribute__((ext_vector_type(2), aligned(4)));
// nv valid elements of a row piece (the others read as `pad`)
__device__ __forceinline__ void pw_load4(const float* row, int nv, float pad, float (&w)[4]) {
  if (nv == 4) {
    const pw_f4 v = *reinterpret_cast<const pw_f4*>(row);
    w[0] = v[0]; w[1] = v[1]; w[2] = v[2]; w[3] = v[3];
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) w[e] = e < nv ? row[e] : pad;
  }
}
__device__ __forceinline__ void pw_store4(float* row, int nv, const float (&v)[4]) {
  if (nv == 4) {
    const pw_f4 o = {v[0], v[1], v[2], v[3]};
    *reinterpret_cast<pw_f4*>(row) = o;
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) if (e < nv) row[e] = v[e];
  }
}
template <int NO>
__device__ __forceinline__ void pw_loadN(const float* p, int nv, float (&g)[NO]) {
  if (nv == NO) {
    if constexpr (NO == 4) {
      const pw_f4 v = *reinterpret_cast<const pw_f4*>(p);
      g[0] = v[0]; g[1] = v[1]; g[2] = v[2]; g[3] = v[3];
    } else {
      const pw_f2 v = *reinterpret_cast<const pw_f2*>(p);
      g[0] = v[0]; g[1] = v[1];
    }
  } else {
#pragma unroll
    for (int e = 0; e < NO; ++e) g[e] = e < nv ? p[e] : 0.f;
  }
}
template <int NO>
__device__ __forceinline__ void pw_storeN(float* p, int nv, const float (&g)[NO]) {
  if (nv == NO) {
    if constexpr (NO == 4) {
      const pw_f4 o = {g[0], g[1], g[2], g[3]};
      *reinterpret_cast<pw_f4*>(p) = o;
    } else {
      const pw_f2 o = {g[0], g[1]};
      *reinterpret_cast<pw_f2*>(p) = o;
    }
  } else {
#pragma unroll
    for (int e = 0; e < NO; ++e) if (e < nv) p[e] = g[e];
  }
}
constexpr int kMaxParts = 8;      // split-K partial-sum slabs a consumer adds up (e2hip.h)
template <int PZ, int PY, int PX, bool HAS_BIAS>
__global__ __launch_bounds__(256) void pool_fwd_fixed_kernel(View5 y, const float* __restrict__ bias,
                                                             int act, View5 out, FastDiv dvw,
                                                             FastDiv dh, unsigned chunk,
                                                             int nparts, long pstride) {
  static_assert(PX == 1 || PX == 2, "x windows of 1 or 2");
  constexpr int NO = 4 / PX;                                  // pooled outputs per thread
  const unsigned VW = ((unsigned)out.w + NO - 1) / NO;        // pieces per output row
  const unsigned S = (unsigned)out.d * out.h * VW;
  const unsigned s0 = blockIdx.x * chunk;
  const unsigned s1 = min(s0 + chunk, S);
  const int c = blockIdx.y, n = blockIdx.z;
  const float bv = HAS_BIAS ? bias[c] : 0.f;
  float* ybase = y.p + (long)n * y.sn + (long)c * y.sc;
  float* __restrict__ obase = out.p + (long)n * out.sn + (long)c * out.sc;
#pragma unroll 2
  for (unsigned s = s0 + threadIdx.x; s < s1; s += 256) {
    const unsigned t = fdiv(s, dvw);
    const unsigned xo = (s - t * VW) * NO;
    const unsigned zo = fdiv(t, dh);
    const unsigned yo = t - zo * out.h;
    const int nvo = min(NO, out.w - (int)xo);
    float* src = ybase + (long)(zo * PZ) * y.sd + (long)(yo * PY) * y.sh + xo * PX;
    float m[NO];
#pragma unroll
    for (int j = 0; j < NO; ++j) m[j] = -INFINITY;
#pragma unroll
    for (int a = 0; a < PZ; ++a)
#pragma unroll
      for (int b = 0; b < PY; ++b) {
        float w[4];
        float* row = src + a * y.sd + b * y.sh;
        pw_load4(row, nvo * PX, 0.f, w);
        if (nparts > 1) {      // split-K partial sums: add them up, leave the sum in part 0
          // (all parts requested before the first add: a load per loop trip cost the
          // launches of neuro3d's 0.5-2 MB layers 8 dependent round trips, 5.8 us each)
          float u[kMaxParts - 1][4];
#pragma unroll
          for (int q = 1; q < kMaxParts; ++q)
            if (q < nparts) pw_load4(row + q * pstride, nvo * PX, 0.f, u[q - 1]);
#pragma unroll
          for (int q = 1; q < kMaxParts; ++q)
            if (q < nparts) {
#pragma unroll
              for (int e = 0; e < 4; ++e) w[e] += u[q - 1][e];
            }
          pw_store4(row, nvo * PX, w);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) m[e / PX] = fmaxf(m[e / PX], w[e]);
      }
    float v[NO];
#pragma unroll
    for (int j = 0; j < NO; ++j) {
      v[j] = m[j] + bv;
      if (act == E2_ACT_RELU) v[j] = fmaxf(v[j], 0.f);
    }
    pw_storeN<NO>(obase + (long)zo * out.sd + (long)yo * out.sh + xo, nvo, v);
  }
}
template <int PZ, int PY, int PX, bool HAS_BIAS>
__global__ __launch_bounds__(256) void pool_bwd_fixed_kernel(View5 dout, View5 y,
                                                             const float* __restrict__ bias,
                                                             int act, View5 dy,
                                                             float* __restrict__ dbias,
                                                             int accumulate, FastDiv dvw,
                                                             FastDiv dh, unsigned chunk,
                                                             int gparts, long gstride) {
  static_assert(PX == 1 || PX == 2, "x windows of 1 or 2");
  constexpr int NO = 4 / PX;
  __shared__ float red[4];
  const unsigned VW = ((unsigned)dout.w + NO - 1) / NO;
  const unsigned S = (unsigned)dout.d * dout.h * VW;
  const unsigned s0 = blockIdx.x * chunk;
  const unsigned s1 = min(s0 + chunk, S);
  const int c = blockIdx.y, n = blockIdx.z;
  const float bv = HAS_BIAS ? bias[c] : 0.f;
  const float* __restrict__ ybase = y.p + (long)n * y.sn + (long)c * y.sc;
  const float* __restrict__ gbase = dout.p + (long)n * dout.sn + (long)c * dout.sc;
  float* __restrict__ dbase = dy.p + (long)n * dy.sn + (long)c * dy.sc;
  float gsum = 0.f;
#pragma unroll 2
  for (unsigned s = s0 + threadIdx.x; s < s1; s += 256) {
    const unsigned t = fdiv(s, dvw);
    const unsigned xo = (s - t * VW) * NO;
    const unsigned zo = fdiv(t, dh);
    const unsigned yo = t - zo * dout.h;
    const int nvo = min(NO, dout.w - (int)xo);
    const int nvi = nvo * PX;
    const float* src = ybase + (long)(zo * PZ) * y.sd + (long)(yo * PY) * y.sh + xo * PX;
    float w[PZ][PY][4];
    float m[NO];
#pragma unroll
    for (int j = 0; j < NO; ++j) m[j] = -INFINITY;
#pragma unroll
    for (int a = 0; a < PZ; ++a)
#pragma unroll
      for (int b = 0; b < PY; ++b) {
        pw_load4(src + a * y.sd + b * y.sh, nvi, 0.f, w[a][b]);
#pragma unroll
        for (int e = 0; e < 4; ++e) m[e / PX] = fmaxf(m[e / PX], w[a][b][e]);
      }
    float g[NO];
    pw_loadN<NO>(gbase + (long)zo * dout.sd + (long)yo * dout.sh + xo, nvo, g);
    if (gparts > 1) {                       // dout arrives as split-K partial sums
      float u[kMaxParts - 1][NO];
#pragma unroll
      for (int q = 1; q < kMaxParts; ++q)
        if (q < gparts)
          pw_loadN<NO>(gbase + q * gstride + (long)zo * dout.sd + (long)yo * dout.sh + xo, nvo, u[q - 1]);
#pragma unroll
      for (int q = 1; q < kMaxParts; ++q)
        if (q < gparts) {
#pragma unroll
          for (int j = 0; j < NO; ++j) g[j] += u[q - 1][j];
        }
    }
#pragma unroll
    for (int j = 0; j < NO; ++j) {
      if (act == E2_ACT_RELU) {
        const float pre = m[j] + bv;
        g[j] *= (pre > 0.f) ? 1.f : ((pre == 0.f) ? 0.5f : 0.f);
      }
      if (j < nvo) gsum += g[j];
    }
    float* dst = dbase + (long)(zo * PZ) * dy.sd + (long)(yo * PY) * dy.sh + xo * PX;
#pragma unroll
    for (int a = 0; a < PZ; ++a)
#pragma unroll
      for (int b = 0; b < PY; ++b) {
        float* drow = dst + a * dy.sd + b * dy.sh;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (w[a][b][e] == m[e / PX]) ? g[e / PX] : 0.f;
        if (accumulate) {
          float old[4];
          pw_load4(drow, nvi, 0.f, old);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += old[e];
        }
        pw_store4(drow, nvi, v);
      }
  }
  if (dbias != nullptr) {
    const float tot = block_sum256(gsum, red);
    if (threadIdx.x == 0 && tot != 0.f) unsafeAtomicAdd(dbias + c, tot);
  }
}

// backward of the FUSED conv+bias+act forward (no pooling): dy = dout * act'(out),
// relu' read off the activated output: > 0 -> 1, +0.0 -> 0.5 (pre-activation was
// exactly 0), -0.0 -> 0 (it was negative; see e2_conv3d_fwd_packed_act); dbias += sum
// (four consecutive x per thread, 16-byte accesses -- the scalar form ran at 2.9 TB/s)
__global__ __launch_bounds__(256) void act_bwd_out_kernel(View5 dout, View5 out, int act,
                                                          View5 dy, float* __restrict__ dbias,
                                                          FastDiv dvw, FastDiv dh,
                                                          unsigned chunk, int gparts,
                                                          long gstride) {
  __shared__ float red[4];
  const unsigned VW = (unsigned)(dout.w + 3) >> 2;            // 4-element pieces per row
  const unsigned S = (unsigned)dout.d * dout.h * VW;
  const unsigned s0 = blockIdx.x * chunk;
  const unsigned s1 = min(s0 + chunk, S);
  const int c = blockIdx.y, n = blockIdx.z;
  const float* obase = out.p + (long)n * out.sn + (long)c * out.sc;
  const float* gbase = dout.p + (long)n * dout.sn + (long)c * dout.sc;
  float* dbase = dy.p + (long)n * dy.sn + (long)c * dy.sc;
  float gsum = 0.f;
  for (unsigned s = s0 + threadIdx.x; s < s1; s += 256) {
    const unsigned t = fdiv(s, dvw);
    const unsigned xo = (s - t * VW) << 2;
    const unsigned zo = fdiv(t, dh);
    const unsigned yo = t - zo * dout.h;
    const float* gp = gbase + (long)zo * dout.sd + (long)yo * dout.sh + xo;
    const float* op = obase + (long)zo * out.sd + (long)yo * out.sh + xo;
    float* dp = dbase + (long)zo * dy.sd + (long)yo * dy.sh + xo;
    const int nv = min(4, dout.w - (int)xo);
    float g[4], o[4];
    if (nv == 4) {
      const pw_f4 gv = *reinterpret_cast<const pw_f4*>(gp);
      const pw_f4 ov = *reinterpret_cast<const pw_f4*>(op);
#pragma unroll
      for (int e = 0; e < 4; ++e) { g[e] = gv[e]; o[e] = ov[e]; }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) { g[e] = e < nv ? gp[e] : 0.f; o[e] = e < nv ? op[e] : 1.f; }
    }
    if (gparts > 1) {                       // dout arrives as split-K partial sums
      float u[kMaxParts - 1][4];
#pragma unroll
      for (int q = 1; q < kMaxParts; ++q)
        if (q < gparts) pw_load4(gp + q * gstride, nv, 0.f, u[q - 1]);
#pragma unroll
      for (int q = 1; q < kMaxParts; ++q)
        if (q < gparts) {
#pragma unroll
          for (int e = 0; e < 4; ++e) g[e] += u[q - 1][e];
        }
    }
    if (act == E2_ACT_RELU) {
#pragma unroll
      for (int e = 0; e < 4; ++e) g[e] *= (o[e] > 0.f) ? 1.f : (__builtin_signbit(o[e]) ? 0.f : 0.5f);
    }
    gsum += (g[0] + g[1]) + (g[2] + g[3]);
    if (nv == 4) {
      pw_f4 v = {g[0], g[1], g[2], g[3]};
      *reinterpret_cast<pw_f4*>(dp) = v;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) if (e < nv) dp[e] = g[e];
    }
  }
  if (dbias != nullptr) {
    const float tot = block_sum256(gsum, red);
    if (threadIdx.x == 0 && tot != 0.f) unsafeAtomicAdd(dbias + c, tot);
  }
}

// ---------------------------------------------------------------------------
// UpConv helpers: dpre in space-to-depth layout
//   s2d[n][co*R + r][z][y][x] = dout[n][co][pz*z+rz][py*y+ry][px*x+rx] * act'(yout)
// one thread per dout element (reads coalesced), dbias[co] += sum
// ---------------------------------------------------------------------------
__global__ void upconv_dpre_s2d_kernel(View5 dout, View5 yout, int pz, int py, int px,
                                       int act, float* __restrict__ s2d,
                                       float* __restrict__ dbias, FastDiv dw, FastDiv dh,
                                       FastDiv dpz, FastDiv dpy, FastDiv dpx, unsigned chunk) {
  // 32-bit index math with magic-number division (the 64-bit % and / of the first
  // version made this kernel ALU-bound at ~0.7 TB/s), a chunk of positions per work-group
  __shared__ float red[4];
  const unsigned S = (unsigned)dout.d * dout.h * dout.w;
  const unsigned s0 = blockIdx.x * chunk;
  const unsigned s1 = min(s0 + chunk, S);
  const int c = blockIdx.y, n = blockIdx.z;
  const unsigned R = pz * py * px;
  const unsigned di = dout.d / pz, hi = dout.h / py, wi = dout.w / px;
  const float* gb = dout.p + (long)n * dout.sn + (long)c * dout.sc;
  const float* ob = yout.p + (long)n * yout.sn + (long)c * yout.sc;
  float* sb = s2d + ((long)n * dout.c + c) * (long)R * di * hi * wi;
  float gsum = 0.f;
  for (unsigned s = s0 + threadIdx.x; s < s1; s += 256) {
    const unsigned t = fdiv(s, dw);
    const unsigned x = s - t * dout.w;
    const unsigned z = fdiv(t, dh);
    const unsigned y = t - z * dout.h;
    float g = gb[(long)z * dout.sd + (long)y * dout.sh + x];
    if (act == E2_ACT_RELU) {
      const float o = ob[(long)z * yout.sd + (long)y * yout.sh + x];
      g = (o > 0.f) ? g : 0.f;
    }
    const unsigned zi = fdiv(z, dpz), rz = z - zi * pz;
    const unsigned yi = fdiv(y, dpy), ry = y - yi * py;
    const unsigned xi = fdiv(x, dpx), rx = x - xi * px;
    const unsigned r = (rz * py + ry) * px + rx;
    sb[((long)r * di + zi) * (hi * wi) + yi * wi + xi] = g;
    gsum += g;
  }
  if (dbias != nullptr) {
    const float tot = block_sum256(gsum, red);
    if (threadIdx.x == 0 && tot != 0.f) unsafeAtomicAdd(dbias + c, tot);
  }
}

// ---------------------------------------------------------------------------
// transposes through a 32x33 LDS tile:  [C][S] <-> [S][C]
// ---------------------------------------------------------------------------
__global__ void ncdhw_to_ndhwc_kernel(View5 src, float* __restrict__ dst) {
  __shared__ float tile[32][33];
  const long S = (long)src.d * src.h * src.w;
  const int n = blockIdx.z;
  const long s0 = blockIdx.x * 32L;
  const int c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i;
    const long s = s0 + tx;
    float v = 0.f;
    if (c < src.c && s < S) {
      const int x = (int)(s % src.w);
      const long t = s / src.w;
      v = src.p[vidx(src, n, c, (int)(t / src.h), (int)(t % src.h), x)];
    }
    tile[i][tx] = v;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const long s = s0 + i;
    const int c = c0 + tx;
    if (c < src.c && s < S) dst[((long)n * S + s) * src.c + c] = tile[tx][i];
  }
}

__global__ void ndhwc_to_ncdhw_kernel(const float* __restrict__ src, View5 dst) {
  __shared__ float tile[32][33];
  const long S = (long)dst.d * dst.h * dst.w;
  const int n = blockIdx.z;
  const long s0 = blockIdx.x * 32L;
  const int c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8) {
    const long s = s0 + i;
    const int c = c0 + tx;
    float v = 0.f;
    if (c < dst.c && s < S) v = src[((long)n * S + s) * dst.c + c];
    tile[i][tx] = v;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i;
    const long s = s0 + tx;
    if (c < dst.c && s < S) {
      const int x = (int)(s % dst.w);
      const long t = s / dst.w;
      dst.p[vidx(dst, n, c, (int)(t / dst.h), (int)(t % dst.h), x)] = tile[tx][i];
    }
  }
}

// ---------------------------------------------------------------------------
// channel softmax + MultinoulliNLL (sparse target), thread per position
// ---------------------------------------------------------------------------
__global__ void softmax_nll_fwd_kernel(View5 lg, View5 tg, View5 pr,
                                       float* __restrict__ stats) {
  __shared__ float red[4];
  const long S = (long)lg.d * lg.h * lg.w;
  const long s = blockIdx.x * 256L + threadIdx.x;
  const int n = blockIdx.z;
  float lsum = 0.f, nlab = 0.f;
  if (s < S) {
    const int x = (int)(s % lg.w);
    const long t = s / lg.w;
    const int y = (int)(t % lg.h), z = (int)(t / lg.h);
    const float* lp = lg.p + vidx(lg, n, 0, z, y, x);
    float m = lp[0];
    for (int c = 1; c < lg.c; ++c) m = fmaxf(m, lp[c * lg.sc]);
    float den = 0.f;
    for (int c = 0; c < lg.c; ++c) den += expf(lp[c * lg.sc] - m);
    const float tv = tg.p[vidx(tg, n, 0, z, y, x)];
    float* pp = pr.p + vidx(pr, n, 0, z, y, x);
    for (int c = 0; c < lg.c; ++c) {
      const float pc = expf(lp[c * lg.sc] - m) / den;
      pp[c * pr.sc] = pc;
      if (tv == (float)c) { lsum -= logf(pc + E2_EPS_NLL); nlab += 1.f; }
    }
  }
  const float a = block_sum256(lsum, red);
  const float b = block_sum256(nlab, red);
  if (threadIdx.x == 0) {
    if (a != 0.f) unsafeAtomicAdd(stats + 0, a);
    if (b != 0.f) unsafeAtomicAdd(stats + 1, b);
  }
}

__global__ void softmax_nll_bwd_kernel(View5 pr, View5 tg, const float* __restrict__ stats,
                                       View5 dl, float* __restrict__ loss_out, int sum_mode,
                                       float* __restrict__ count_out) {
  const long S = (long)pr.d * pr.h * pr.w;
  const long s = blockIdx.x * 256L + threadIdx.x;
  const int n = blockIdx.z;
  float inv = 1.f / (stats[1] + E2_EPS_NLL);
  if (blockIdx.x == 0 && blockIdx.z == 0 && threadIdx.x == 0) {
    if (loss_out) loss_out[0] = stats[0] * inv;
    if (count_out) count_out[0] = stats[1];
  }
  if (sum_mode) inv = 1.f;                  // (e2_set_loss_grad_mode: unnormalised gradients)
  if (s >= S) return;
  const int x = (int)(s % pr.w);
  const long t = s / pr.w;
  const int y = (int)(t % pr.h), z = (int)(t / pr.h);
  const float tv = tg.p[vidx(tg, n, 0, z, y, x)];
  const float* pp = pr.p + vidx(pr, n, 0, z, y, x);
  float* dp = dl.p + vidx(dl, n, 0, z, y, x);
  float pt = 0.f;
  int tc = -1;
  for (int c = 0; c < pr.c; ++c)
    if (tv == (float)c) { tc = c; pt = pp[c * pr.sc]; }
  // dL/dp_t = -inv/(p_t+eps);  dlogit_c = p_c*(dp_c - sum_k dp_k p_k)
  const float gpt = (tc >= 0) ? (-inv / (pt + E2_EPS_NLL)) * pt : 0.f;
  for (int c = 0; c < pr.c; ++c) {
    const float pc = pp[c * pr.sc];
    dp[c * dl.sc] = gpt * ((c == tc ? 1.f : 0.f) - pc);
  }
}

// ---------------------------------------------------------------------------
// MALIS NLL (loss.py:560-690): probs (1, 2E, d,h,w) holds E independent 2-class
// softmaxes (channel 2e = "disconnected", 2e+1 = affinity).  With the MALIS counts
// P (pairs this edge should connect) and N (pairs it should keep apart):
//   loss = -sum(P log(p1+eps) + N log(p0+eps)) * norm[0],  norm[0] = 1/(n_tot+eps)
// and, the counts being constants (malisop.py:114-120: zero gradient),
//   dlogit_c = p_c (g_c - (p0 g0 + p1 g1)),  g1 = -P norm/(p1+eps), g0 = -N norm/(p0+eps)
// thread per (edge, position); loss_sum accumulates the normalised loss.
// ---------------------------------------------------------------------------
__global__ void malis_nll_kernel(View5 pr, const float* __restrict__ pos,
                                 const float* __restrict__ neg,
                                 const float* __restrict__ norm, View5 dl, int want_grad,
                                 float* __restrict__ loss_sum) {
  __shared__ float red[4];
  const long S = (long)pr.d * pr.h * pr.w;
  const long s = blockIdx.x * 256L + threadIdx.x;
  const int e = blockIdx.y;
  const float inv = norm[0];
  float l = 0.f;
  if (s < S) {
    const int x = (int)(s % pr.w);
    const long t = s / pr.w;
    const int y = (int)(t % pr.h), z = (int)(t / pr.h);
    const float* pp = pr.p + vidx(pr, 0, 2 * e, z, y, x);
    const float p0 = pp[0], p1 = pp[pr.sc];
    const float P = pos[(long)e * S + s], N = neg[(long)e * S + s];
    // xlogy0 (loss.py:26-28): 0 where the count is 0, whatever the logarithm
    if (P != 0.f) l -= P * logf(p1 + E2_EPS_NLL);
    if (N != 0.f) l -= N * logf(p0 + E2_EPS_NLL);
    if (want_grad) {
      const float g1 = -P * inv / (p1 + E2_EPS_NLL), g0 = -N * inv / (p0 + E2_EPS_NLL);
      const float m = p0 * g0 + p1 * g1;
      float* dp = dl.p + vidx(dl, 0, 2 * e, z, y, x);
      dp[0] = p0 * (g0 - m);
      dp[dl.sc] = p1 * (g1 - m);
    }
  }
  const float a = block_sum256(l * inv, red);
  if (threadIdx.x == 0 && a != 0.f) unsafeAtomicAdd(loss_sum, a);
}

// ---------------------------------------------------------------------------
// optimisers on a flat arena (hyper = {lr, mom, beta2, wd, t, factor, -, arrivals})
// ---------------------------------------------------------------------------
// Every tensor of the arena starts on a 16-byte boundary and is padded to a multiple of
// four elements (model.py ensure_arena), so a float4 never straddles two tensors: one
// weight-decay multiplier per float4, found by a binary search over the segment table in
// LDS.  Adam's step counter t lives on the device (hipGraph replay cannot pass a new
// value): every work-group reads t_old, uses t = t_old + 1, and the LAST work-group to
// finish -- an arrival counter, at most 256 arrivals -- publishes t for the next step.
// No work-group can still need t_old then.  (A separate one-thread "tick" kernel took 5 us
// per step; one arrival per work-group of a 4096-group grid serialised for 46 us.)
constexpr int kOptMaxSeg = 1024;
__device__ __forceinline__ float seg_mult_lds(const long* so, const float* sr, int n_seg, size_t i) {
  int lo = 0, hi = n_seg - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if ((size_t)so[mid] <= i) lo = mid; else hi = mid - 1;
  }
  return sr[lo];
}

// gdiv / gmul: the gradient that enters the update is g * gmul / (gdiv ? gdiv[0] + 1e-5 : 1) --
// the data-parallel step sums UNNORMALISED gradients over the ranks and divides by the summed
// labelled-voxel count here (one device scalar behind the arena) instead of in elementwise
// launches of its own.  zero_g: the arena is left ZERO for the next backward pass, which then
// needs no fill launch (e2_adam_step_ex).
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ s, size_t n,
                                                   const int64_t* __restrict__ seg_off,
                                                   const float* __restrict__ seg_reg, int n_seg,
                                                   float* __restrict__ hyper,
                                                   const float* __restrict__ gdiv, float gmul, int zero_g) {
  __shared__ long so[kOptMaxSeg];
  __shared__ float sr[kOptMaxSeg];
  for (int i = threadIdx.x; i < n_seg; i += blockDim.x) { so[i] = seg_off[i]; sr[i] = seg_reg[i]; }
  const float lr = hyper[0], mom = hyper[1], b2 = hyper[2], wd = hyper[3];
  const float t = hyper[4] + 1.f;
  const float fac = sqrtf(1.f - powf(b2, t)) / (1.f - powf(mom, t));
  const float gs = gdiv ? gmul / (gdiv[0] + 1e-5f) : gmul;
  __syncthreads();
  const size_t n4 = n >> 2;
  float4* p4 = reinterpret_cast<float4*>(p);
  float4* g4 = reinterpret_cast<float4*>(g);
  float4* m4 = reinterpret_cast<float4*>(m);
  float4* s4 = reinterpret_cast<float4*>(s);
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4;
       i += (size_t)gridDim.x * blockDim.x) {
    float4 gv = g4[i];
    const float4 mv = m4[i], sv = s4[i];
    float4 pv = p4[i];
    if (gs != 1.f) { gv.x *= gs; gv.y *= gs; gv.z *= gs; gv.w *= gs; }
    if (zero_g) g4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float mult = seg_mult_lds(so, sr, n_seg, 4 * i) * wd;
    float4 nm, ns;
#define E2_ADAM1(c)                                                      \
    nm.c = mom * mv.c + (1.f - mom) * gv.c;                              \
    ns.c = b2 * sv.c + (1.f - b2) * gv.c * gv.c;                         \
    {                                                                    \
      float dir = fac * nm.c / sqrtf(ns.c + E2_EPS_ADAM);                \
      if (mult != 0.f) dir += mult * pv.c;                               \
      pv.c = pv.c - lr * dir;                                            \
    }
    E2_ADAM1(x) E2_ADAM1(y) E2_ADAM1(z) E2_ADAM1(w)
#undef E2_ADAM1
    m4[i] = nm; s4[i] = ns; p4[i] = pv;
  }
  if (blockIdx.x == 0)                                 // (n is a multiple of 4 in the plan's arena)
    for (size_t i = 4 * n4 + threadIdx.x; i < n; i += blockDim.x) {
      const float gi = g[i] * gs;
      if (zero_g) g[i] = 0.f;
      const float nm = mom * m[i] + (1.f - mom) * gi;
      const float ns = b2 * s[i] + (1.f - b2) * gi * gi;
      float dir = fac * nm / sqrtf(ns + E2_EPS_ADAM);
      const float mult = seg_mult_lds(so, sr, n_seg, i) * wd;
      const float pi = p[i];
      if (mult != 0.f) dir += mult * pi;
      m[i] = nm; s[i] = ns; p[i] = pi - lr * dir;
    }
  // publish t once every work-group has read the old value
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned* arrivals = reinterpret_cast<unsigned*>(hyper + 7);
    const unsigned prev = __hip_atomic_fetch_add(arrivals, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (prev == gridDim.x - 1) {
      __hip_atomic_store(arrivals, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      hyper[4] = t;
      hyper[5] = fac;
    }
  }
}

__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, float* __restrict__ g,
                                                  float* __restrict__ d, size_t n,
                                                  const int64_t* __restrict__ seg_off,
                                                  const float* __restrict__ seg_reg, int n_seg,
                                                  const float* __restrict__ hyper,
                                                  const float* __restrict__ gdiv, float gmul, int zero_g) {
  __shared__ long so[kOptMaxSeg];
  __shared__ float sr[kOptMaxSeg];
  for (int i = threadIdx.x; i < n_seg; i += blockDim.x) { so[i] = seg_off[i]; sr[i] = seg_reg[i]; }
  const float lr = hyper[0], mom = hyper[1], wd = hyper[3];
  const float gs = gdiv ? gmul / (gdiv[0] + 1e-5f) : gmul;
  __syncthreads();
  const size_t n4 = n >> 2;
  float4* p4 = reinterpret_cast<float4*>(p);
  float4* g4 = reinterpret_cast<float4*>(g);
  float4* d4 = reinterpret_cast<float4*>(d);
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4;
       i += (size_t)gridDim.x * blockDim.x) {
    float4 gv = g4[i];
    const float4 dv = d4[i];
    float4 pv = p4[i];
    if (gs != 1.f) { gv.x *= gs; gv.y *= gs; gv.z *= gs; gv.w *= gs; }
    if (zero_g) g4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float mult = seg_mult_lds(so, sr, n_seg, 4 * i) * wd;
    float4 nd;
#define E2_SGD1(c)                                                       \
    nd.c = gv.c + mom * dv.c;                                            \
    pv.c = pv.c - lr * (mult != 0.f ? nd.c + mult * pv.c : nd.c);
    E2_SGD1(x) E2_SGD1(y) E2_SGD1(z) E2_SGD1(w)
#undef E2_SGD1
    d4[i] = nd; p4[i] = pv;
  }
  if (blockIdx.x == 0)
    for (size_t i = 4 * n4 + threadIdx.x; i < n; i += blockDim.x) {
      const float nd = g[i] * gs + mom * d[i];
      if (zero_g) g[i] = 0.f;
      const float mult = seg_mult_lds(so, sr, n_seg, i) * wd;
      const float pi = p[i];
      d[i] = nd;
      p[i] = pi - lr * (mult != 0.f ? nd + mult * pi : nd);
    }
}

// ---------------------------------------------------------------------------
// host wrappers
// ---------------------------------------------------------------------------
static int check_view(const e2_tensor5* t, const char* name) {
  E2_REQUIRE(t && t->ptr, "%s: null tensor", name);
  E2_REQUIRE(t->n > 0 && t->c > 0 && t->d > 0 && t->h > 0 && t->w > 0,
             "%s: empty tensor (%d,%d,%d,%d,%d)", name, t->n, t->c, t->d, t->h, t->w);
  E2_REQUIRE(t->c < 65536 && t->n < 65536, "%s: n/c too large for grid", name);
  return 0;
}
static dim3 grid_for(const View5& v) {
  const long S = (long)v.d * v.h * v.w;
  return dim3((unsigned)((S + 255) / 256), (unsigned)v.c, (unsigned)v.n);
}
// outputs per work-group: enough groups to fill the chip (>= ~2048), at most
// 4096 per group so that per-channel reductions need few atomics
static unsigned pw_chunk(const View5& v) {
  const long S = (long)v.d * v.h * v.w;
  long per = (S * v.c * v.n) / 2048;
  per = ((per + 255) / 256) * 256;
  if (per < 256) per = 256;
  if (per > 4096) per = 4096;
  return (unsigned)per;
}
static dim3 grid_chunked(const View5& v, unsigned chunk) {
  const long S = (long)v.d * v.h * v.w;
  return dim3((unsigned)((S + chunk - 1) / chunk), (unsigned)v.c, (unsigned)v.n);
}

// the grid of the fixed-window kernels counts pieces of 4 input x (4 / PX outputs)
template <int PZ, int PY, int PX>
static void launch_pool_fwd_fixed(e2_ctx* ctx, const View5& vy, const float* bias, int act,
                                  const View5& vo, int nparts = 1, long pstride = 0) {
  View5 vq = vo; vq.w = (vo.w + 4 / PX - 1) / (4 / PX);
  const unsigned chunk = pw_chunk(vq);
  const dim3 g = grid_chunked(vq, chunk);
  const FastDiv dvw = mk_div(vq.w), dh = mk_div(vo.h);
  if (bias)
    hipLaunchKernelGGL((pool_fwd_fixed_kernel<PZ, PY, PX, true>), g, dim3(256), 0, ctx->stream,
                       vy, bias, act, vo, dvw, dh, chunk, nparts, pstride);
  else
    hipLaunchKernelGGL((pool_fwd_fixed_kernel<PZ, PY, PX, false>), g, dim3(256), 0, ctx->stream,
                       vy, bias, act, vo, dvw, dh, chunk, nparts, pstride);
}
template <int PZ, int PY, int PX>
static void launch_pool_bwd_fixed(e2_ctx* ctx, const View5& vd, const View5& vy, const float* bias,
                                  int act, const View5& vdy, float* dbias, int accumulate,
                                  int gparts = 1, long gstride = 0) {
  View5 vq = vd; vq.w = (vd.w + 4 / PX - 1) / (4 / PX);
  const unsigned chunk = pw_chunk(vq);
  const dim3 g = grid_chunked(vq, chunk);
  const FastDiv dvw = mk_div(vq.w), dh = mk_div(vd.h);
  if (bias)
    hipLaunchKernelGGL((pool_bwd_fixed_kernel<PZ, PY, PX, true>), g, dim3(256), 0, ctx->stream,
                       vd, vy, bias, act, vdy, dbias, accumulate, dvw, dh, chunk, gparts, gstride);
  else
    hipLaunchKernelGGL((pool_bwd_fixed_kernel<PZ, PY, PX, false>), g, dim3(256), 0, ctx->stream,
                       vd, vy, bias, act, vdy, dbias, accumulate, dvw, dh, chunk, gparts, gstride);
}

int e2i_fill_view(e2_ctx* ctx, const e2_tensor5* t, float value) {
  if (int rc = check_view(t, "fill_view")) return rc;
  View5 v = mk(t);
  E2_REQUIRE((long)v.d * v.h * v.w < (1L << 31), "fill_view: channel too large");
  hipLaunchKernelGGL(fill_view_kernel, grid_for(v), dim3(256), 0, ctx->stream, v, value,
                     mk_div(v.w), mk_div(v.h));
  E2_CHECK_HIP(hipGetLastError());
  return 0;
}

// Flat fill as a KERNEL, also for zeros: hipMemsetAsync nodes inside a captured
// hipGraph were observed to be re-ordered against the kernel that follows them on
// replay (unet3d_lite: the split-K accumulation of a conv started from stale data in
// ~1 of 3 processes); kernel nodes keep their order.
int e2i_fill_flat(e2_ctx* ctx, float* ptr, size_t n, float value) {
  if (n == 0) return 0;
  int grid = (int)std::min<size_t>((n + 255) / 256, 8192);
  hipLaunchKernelGGL(fill_flat_kernel, dim3(grid), dim3(256), 0, ctx->stream, ptr, n, value);
  E2_CHECK_HIP(hipGetLastError());
  return 0;
}

// many flat fills in one launch (blockIdx.y = region)
__global__ void fill_multi_kernel(float* const* __restrict__ ptrs,
                                  const unsigned long long* __restrict__ counts, float v) {
  float* p = ptrs[blockIdx.y];
  const size_t n = (size_t)counts[blockIdx.y];
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x)
    p[i] = v;
}

extern "C" int e2_fill_multi(e2_ctx* ctx, const void* ptrs_dev, const void* counts_dev,
                             int nregions, float value) {
  E2_REQUIRE(ctx && ptrs_dev && counts_dev && nregions > 0 && nregions < 65536,
             "e2_fill_multi: bad argument");
  hipLaunchKernelGGL(fill_multi_kernel, dim3(512, nregions), dim3(256), 0, ctx->stream,
                     (float* const*)ptrs_dev, (const unsigned long long*)counts_dev, value);
  E2_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---- several steps in one graph: the batches come out of a ring in HBM, the losses go into one ----
// A captured step is the same launches every time; what changes from step to step are the batch
// and the loss.  Both become position-independent through a launch COUNT kept in device memory:
// launch number L of the prologue reads slot L % n of the batch ring and stores the loss the
// PREVIOUS step left behind in slot (L - 1) % n of the history.  The count is read with plain
// loads (it was written by the previous launch) and advanced by the work-group that arrives last
// at the end of the launch -- no load of the copy waits for an atomic.  So the SAME captured step
// can stand k times in one graph (DESIGN finding 55).
struct PrologueP {
  const float* ring; int nSlots; long slotFloats; float* dst;
  const float* src; int nVals; float* hist; int histSlots;
  unsigned long long* state;           // [0] launches so far, [1] arrivals of the running launch
};
__global__ __launch_bounds__(256) void step_prologue_kernel(PrologueP p) {
  const unsigned long long L = *(volatile unsigned long long*)p.state;
  if (p.hist && L > 0 && blockIdx.x == 0) {
    float* d = p.hist + (long)((L - 1) % (unsigned long long)p.histSlots) * p.nVals;
    for (int i = threadIdx.x; i < p.nVals; i += 256) d[i] = p.src[i];
  }
  if (p.ring) {
    const float* src = p.ring + (long)(L % (unsigned long long)p.nSlots) * p.slotFloats;
    const long n4 = p.slotFloats >> 2;
    const float4* s4 = reinterpret_cast<const float4*>(src);
    float4* d4 = reinterpret_cast<float4*>(p.dst);
    const long stride = gridDim.x * 256L;
    long i = blockIdx.x * 256L + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {       // four 16-byte loads in flight per lane
      const float4 a = s4[i], b = s4[i + stride], c = s4[i + 2 * stride], d = s4[i + 3 * stride];
      d4[i] = a; d4[i + stride] = b; d4[i + 2 * stride] = c; d4[i + 3 * stride] = d;
    }
    for (; i < n4; i += stride) d4[i] = s4[i];
    if (blockIdx.x == 0 && threadIdx.x < (p.slotFloats & 3)) p.dst[4 * n4 + threadIdx.x] = src[4 * n4 + threadIdx.x];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (atomicAdd(p.state + 1, 1ull) == (unsigned long long)gridDim.x - 1) {
      p.state[1] = 0;                  // (every work-group of this launch has read state[0] by now)
      p.state[0] = L + 1;
    }
  }
}

/* The first launch of a captured step that may stand several times in one graph:
 *   dst[0 .. slot_floats) = ring[L % n_slots]              (ring != NULL)
 *   hist[(L - 1) % hist_slots][0 .. n_vals) = src[..]       (hist != NULL and L > 0: what the step
 *                                                            BEFORE this one left in src, its loss)
 * with L = the number of prologue launches on `state` so far; state: two zeroed 64-bit words in
 * device memory owned by the caller ([0] = L, readable by the host after a synchronisation).
 * ring, dst 16-byte aligned, slot_floats a multiple of 4.  Launches on one `state` must be
 * ordered (one stream). */
extern "C" int e2_step_prologue(e2_ctx* ctx, const float* ring, int n_slots, size_t slot_floats,
                                float* dst, const float* src, int n_vals, float* hist,
                                int hist_slots, void* state) {
  E2_REQUIRE(ctx && state && ((uintptr_t)state & 7) == 0, "e2_step_prologue: null / misaligned state");
  E2_REQUIRE(!ring || (dst && n_slots > 0 && slot_floats > 0 &&
                       (((uintptr_t)ring | (uintptr_t)dst) & 15) == 0 && ((slot_floats * 4) & 15) == 0),
             "e2_step_prologue: ring / dst must be 16-byte aligned and slots a multiple of 16 bytes");
  E2_REQUIRE(!hist || (src && n_vals > 0 && hist_slots > 0), "e2_step_prologue: bad history arguments");
  PrologueP p;
  p.ring = ring; p.nSlots = n_slots; p.slotFloats = (long)slot_floats; p.dst = dst;
  p.src = src; p.nVals = n_vals; p.hist = hist; p.histSlots = hist_slots;
  p.state = (unsigned long long*)state;
  const int grid = ring ? (int)std::min<size_t>(std::max<size_t>((slot_floats / 4 + 1023) / 1024, 1),
                                                (size_t)ctx->num_cu) : 1;
  hipLaunchKernelGGL(step_prologue_kernel, dim3(grid), dim3(256), 0, ctx->stream, p);
  E2_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int e2_set_skip_zero_fill(e2_ctx* ctx, int on) {
  E2_REQUIRE(ctx, "e2_set_skip_zero_fill: null context");
  ctx->skip_zero_fill = on ? 1 : 0;
  return 0;
}

extern "C" int e2_conv_last_zero_fill(const e2_ctx* ctx, void** ptr, size_t* n) {
  E2_REQUIRE(ctx && ptr && n, "e2_conv_last_zero_fill: null argument");
  *ptr = ctx->last_fill_ptr;
  *n = ctx->last_fill_n;
  return 0;
}

extern "C" int e2_fill(e2_ctx* ctx, float* ptr, size_t n, float value) {
  E2_REQUIRE(ctx && ptr, "e2_fill: null argument");
  return e2i_fill_flat(ctx, ptr, n, value);
}

extern "C" int e2_copy5(e2_ctx* ctx, const e2_tensor5* src, const e2_tensor5* dst,
                        int accumulate) {
  E2_REQUIRE(ctx, "e2_copy5: null ctx");
  if (int rc = check_view(src, "copy5 src")) return rc;
  if (int rc = check_view(dst, "copy5 dst")) return rc;
  E2_REQUIRE(src->n == dst->n && src->c == dst->c && src->d == dst->d &&
                 src->h == dst->h && src->w == dst->w,
             "e2_copy5: size mismatch");
  View5 s = mk(src), d = mk(dst);
  E2_REQUIRE((long)s.d * s.h * s.w < (1L << 31), "e2_copy5: channel too large");
  hipLaunchKernelGGL(copy_view_kernel, grid_for(s), dim3(256), 0, ctx->stream, s, d,
                     accumulate, mk_div(s.w), mk_div(s.h));
  E2_CHECK_HIP(hipGetLastError());
  return 0;
}

static int pool_shapes_ok(const e2_tensor5* big, const e2_tensor5* small, int pz, int py,
                          int px, const char* name) {
  E2_REQUIRE(pz >= 1 && py >= 1 && px >= 1, "%s: pool factors must be >= 1", name);
  E2_REQUIRE(big->n == small->n && big->c == small->c, "%s: n/c mismatch", name);
  E2_REQUIRE(big->d / pz == small->d && big->h / py == small->h && big->w / px == small->w,
             "%s: pooled shape mismatch: (%d,%d,%d)/(%d,%d,%d) != (%d,%d,%d)", name,
             big->d, big->h, big->w, pz, py, px, small->d, small->h, small->w);
  return 0;
}

static int pool_fwd_impl(e2_ctx* ctx, const e2_tensor5* y, const float* bias, int pz, int py,
                         int px, int act, const e2_tensor5* out, int nparts, int64_t pstride);
extern "C" int e2_pool_bias_act_fwd(e2_ctx* ctx, const e2_tensor5* y, const float* bias,
                                    int pz, int py, int px, int act,
                                    const e2_tensor5* out) {
  return pool_fwd_impl(ctx, y, bias, pz, py, px, act, out, 1, 0);
}
extern "C" int e2_pool_bias_act_fwd_parts(e2_ctx* ctx, const e2_tensor5* y, int64_t part_stride,
                                          int nparts, const float* bias, int pz, int py, int px,
                                          int act, const e2_tensor5* out) {
  E2_REQUIRE(nparts >= 1 && nparts <= kMaxParts && (nparts == 1 || part_stride > 0),
             "pool_bias_act_fwd_parts: bad parts (1 .. %d)", kMaxParts);
  return pool_fwd_impl(ctx, y, bias, pz, py, px, act, out, nparts, part_stride);
}
static int pool_fwd_impl(e2_ctx* ctx, const e2_tensor5* y, const float* bias, int pz, int py,
                         int px, int act, const e2_tensor5* out, int nparts, int64_t pstride) {
  E2_REQUIRE(ctx, "pool_bias_act_fwd: null ctx");
  if (int rc = check_view(y, "pool_bias_act_fwd y")) return rc;
  if (int rc = check_view(out, "pool_bias_act_fwd out")) return rc;
  if (int rc = pool_shapes_ok(y, out, pz, py, px, "pool_bias_act_fwd")) return rc;
  E2_REQUIRE(act == E2_ACT_LIN || act == E2_ACT_RELU, "pool_bias_act_fwd: bad act %d", act);
  View5 vy = mk(y), vo = mk(out);
  E2_REQUIRE((long)vo.d * vo.h * vo.w < (1L << 31), "pool_bias_act_fwd: channel too large");
  const FastDiv dw = mk_div(vo.w), dh = mk_div(vo.h);
  const unsigned chunk = pw_chunk(vo);
  const int pcode = pz * 100 + py * 10 + px;
  E2_REQUIRE(nparts == 1 || pcode == 122 || pcode == 211 || pcode == 222 || pcode == 111,
             "pool_bias_act_fwd_parts: partial sums are added up by the fixed-window kernels "
             "((1,1,1), (1,2,2), (2,1,1), (2,2,2)) only");
  if (pcode == 122) launch_pool_fwd_fixed<1, 2, 2>(ctx, vy, bias, act, vo, nparts, (long)pstride);
  else if (pcode == 211) launch_pool_fwd_fixed<2, 1, 1>(ctx, vy, bias, act, vo, nparts, (long)pstride);
  else if (pcode == 222) launch_pool_fwd_fixed<2, 2, 2>(ctx, vy, bias, act, vo, nparts, (long)pstride);
  else if (pcode == 111) launch_pool_fwd_fixed<1, 1, 1>(ctx, vy, bias, act, vo, nparts, (long)pstride);
  else if (bias)
    hipLaunchKernelGGL((pool_fwd_kernel<true>), grid_chunked(vo, chunk), dim3(256), 0,
                       ctx->stream, vy, bias, pz, py, px, act, vo, dw, dh, chunk);
  else
    hipLaunchKernelGGL((pool_fwd_kernel<false>), grid_chunked(vo, chunk), dim3(256), 0,
                       ctx->stream, vy, bias, pz, py, px, act, vo, dw, dh, chunk);
  E2_CHECK_HIP(hipGetLastError());
  return 0;
}

static int pool_bwd_common(e2_ctx* ctx, const e2_tensor5* dout, const e2_tensor5* y,
                           const float* bias, int pz, int py, int px, int act,
                           const e2_tensor5* dy, float* dbias, int accumulate,
                           int gparts = 1, int64_t gstride = 0) {
  if (int rc = check_view(dout, "pool_bwd dout")) return rc;
  if (int rc = check_view(y, "pool_bwd y")) return rc;
  if (int rc = check_view(dy, "pool_bwd dy")) return rc;
  if (int rc = pool_shapes_ok(y, dout, pz, py, px, "pool_bwd")) return rc;
  E2_REQUIRE(dy->n == y->n && dy->c == y->c && dy->d == y->d && dy->h == y->h &&
                 dy->w == y->w, "pool_bwd: dy/y shape mismatch");
  // floor semantics: rows/cols beyond the pooled extent receive no gradient
  if (!accumulate && (y->d % pz || y->h % py || y->w % px)) {
    if (int rc = e2i_fill_view(ctx, dy, 0.f)) return rc;
  }
  View5 vd = mk(dout), vy = mk(y), vdy = mk(dy);
  E2_REQUIRE((long)vd.d * vd.h * vd.w < (1L << 31), "pool_bwd: channel too large");
  const FastDiv dw = mk_div(vd.w), dh = mk_div(vd.h);
  const unsigned chunk = pw_chunk(vd);
  const int pcode = pz * 100 + py * 10 + px;
  E2_REQUIRE(gparts == 1 || pcode == 122 || pcode == 211 || pcode == 222 || pcode == 111,
             "pool_bias_act_bwd_parts: partial sums are added up by the fixed-window kernels only");
  if (pcode == 122)
    launch_pool_bwd_fixed<1, 2, 2>(ctx, vd, vy, bias, act, vdy, dbias, accumulate, gparts, (long)gstride);
  else if (pcode == 211)
    launch_pool_bwd_fixed<2, 1, 1>(ctx, vd, vy, bias, act, vdy, dbias, accumulate, gparts, (long)gstride);
  else if (pcode == 222)
    launch_pool_bwd_fixed<2, 2, 2>(ctx, vd, vy, bias, act, vdy, dbias, accumulate, gparts, (long)gstride);
  else if (pcode == 111)
    launch_pool_bwd_fixed<1, 1, 1>(ctx, vd, vy, bias, act, vdy, dbias, accumulate, gparts, (long)gstride);
  else if (bias)
    hipLaunchKernelGGL((pool_bwd_kernel<true>), grid_chunked(vd, chunk), dim3(256), 0,
                       ctx->stream, vd, vy, bias, pz, py, px, act, vdy, dbias, accumulate, dw,
                       dh, chunk);
  else
    hipLaunchKernelGGL((pool_bwd_kernel<false>), grid_chunked(vd, chunk), dim3(256), 0,
                       ctx->stream, vd, vy, bias, pz, py, px, act, vdy, dbias, accumulate, dw,
                       dh, chunk);
  E2_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int e2_pool_bias_act_bwd(e2_ctx* ctx, const e2_tensor5* dout,
                                    const e2_tensor5* y, const float* bias, int pz, int py,
                                    int px, int act, const e2_tensor5* dy, float* dbias) {
  E2_REQUIRE(ctx, "pool_bias_act_bwd: null ctx");
  E2_REQUIRE(act == E2_ACT_LIN || act == E2_ACT_RELU, "pool_bias_act_bwd: bad act %d", act);
  return pool_bwd_common(ctx, dout, y, bias, pz, py, px, act, dy, dbias, 0);
}

extern "C" int e2_pool_bias_act_bwd_parts(e2_ctx* ctx, const e2_tensor5* dout,
                                          int64_t dout_part_stride, int dout_parts,
                                          const e2_tensor5* y, const float* bias, int pz, int py,
                                          int px, int act, const e2_tensor5* dy, float* dbias) {
  E2_REQUIRE(ctx, "pool_bias_act_bwd_parts: null ctx");
  E2_REQUIRE(act == E2_ACT_LIN || act == E2_ACT_RELU, "pool_bias_act_bwd_parts: bad act %d", act);
  E2_REQUIRE(dout_parts >= 1 && dout_parts <= kMaxParts && (dout_parts == 1 || dout_part_stride > 0),
             "pool_bias_act_bwd_parts: bad parts (1 .. %d)", kMaxParts);
  return pool_bwd_common(ctx, dout, y, bias, pz, py, px, act, dy, dbias, 0, dout_parts,
                         dout_part_stride);
}

static int act_bwd_out_impl(e2_ctx* ctx, const e2_tensor5* dout, const e2_tensor5* out, int act,
                            const e2_tensor5* dy, float* dbias, int gparts, int64_t gstride);
extern "C" int e2_bias_act_bwd_out(e2_ctx* ctx, const e2_tensor5* dout, const e2_tensor5* out,
                                   int act, const e2_tensor5* dy, float* dbias) {
  return act_bwd_out_impl(ctx, dout, out, act, dy, dbias, 1, 0);
}
extern "C" int e2_bias_act_bwd_out_parts(e2_ctx* ctx, const e2_tensor5* dout,
                                         int64_t dout_part_stride, int dout_parts,
                                         const e2_tensor5* out, int act, const e2_tensor5* dy,
                                         float* dbias) {
  E2_REQUIRE(dout_parts >= 1 && dout_parts <= kMaxParts && (dout_parts == 1 || dout_part_stride > 0),
             "bias_act_bwd_out_parts: bad parts (1 .. %d)", kMaxParts);
  return act_bwd_out_impl(ctx, dout, out, act, dy, dbias, dout_parts, dout_part_stride);
}
static int act_bwd_out_impl(e2_ctx* ctx, const e2_tensor5* dout, const e2_tensor5* out, int act,
                            const e2_tensor5* dy, float* dbias, int gparts, int64_t gstride) {
  E2_REQUIRE(ctx, "bias_act_bwd_out: null ctx");
  E2_REQUIRE(act == E2_ACT_LIN || act == E2_ACT_RELU, "bias_act_bwd_out: bad act %d", act);
  if (int rc = check_view(dout, "bias_act_bwd_out dout")) return rc;
  if (int rc = check_view(out, "bias_act_bwd_out out")) return rc;
  if (int rc = check_view(dy, "bias_act_bwd_out dy")) return rc;
  E2_REQUIRE(dout->n == out->n && dout->c == out->c && dout->d == out->d && dout->h == out->h &&
                 dout->w == out->w && dy->n == out->n && dy->c == out->c && dy->d == out->d &&
                 dy->h == out->h && dy->w == out->w, "bias_act_bwd_out: shape mismatch");
  View5 vd = mk(dout), vo = mk(out), vdy = mk(dy);
  E2_REQUIRE((long)vd.d * vd.h * vd.w < (1L << 31), "bias_act_bwd_out: channel too large");
  View5 vq = vd; vq.w = (vd.w + 3) / 4;                 // the grid counts 4-element pieces
  const FastDiv dvw = mk_div(vq.w), dh = mk_div(vd.h);
  const unsigned chunk = pw_chunk(vq);
  hipLaunchKernelGGL(act_bwd_out_kernel, grid_chunked(vq, chunk), dim3(256), 0, ctx->stream, vd,
                     vo, act, vdy, dbias, dvw, dh, chunk, gparts, (long)gstride);
  E2_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int e2_maxpool3d_fwd(e2_ctx* ctx, const e2_tensor5* x, int pz, int py, int px,
                                const e2_tensor5* out) {
  return e2_pool_bias_act_fwd(ctx, x, nullptr, pz, py, px, E2_ACT_LIN, out);
}

extern "C" int e2_maxpool3d_bwd(e2_ctx* ctx, const e2_tensor5* dout, const e2_tensor5* x,
                                int pz, int py, int px, const e2_tensor5* dx,
                                int accumulate) {
  E2_REQUIRE(ctx, "maxpool3d_bwd: null ctx");
  return pool_bwd_common(ctx, dout, x, nullptr, pz, py, px, E2_ACT_LIN, dx, nullptr,
                         accumulate);
}

int e2i_upconv_dpre_s2d(e2_ctx* ctx, const e2_tensor5* dout, const e2_tensor5* yout, int pz,
                        int py, int px, int act, float* s2d, float* dbias) {
  View5 vd = mk(dout), vy = mk(yout);
  E2_REQUIRE((long)vd.d * vd.h * vd.w < (1L << 31), "upconv_dpre_s2d: channel too large");
  const unsigned chunk = pw_chunk(vd);
  hipLaunchKernelGGL(upconv_dpre_s2d_kernel, grid_chunked(vd, chunk), dim3(256), 0,
                     ctx->stream, vd, vy, pz, py, px, act, s2d, dbias, mk_div(vd.w),
                     mk_div(vd.h), mk_div(pz), mk_div(py), mk_div(px), chunk);
  E2_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int e2_transpose_ncdhw_to_ndhwc(e2_ctx* ctx, const e2_tensor5* src, float* dst) {
  E2_REQUIRE(ctx && dst, "transpose: null argument");
  if (int rc = check_view(src, "transpose src")) return rc;
  View5 v = mk(src);
  const long S = (long)v.d * v.h * v.w;
  dim3 grid((unsigned)((S + 31) / 32), (unsigned)((v.c + 31) / 32), (unsigned)v.n);
  hipLaunchKernelGGL(ncdhw_to_ndhwc_kernel, grid, dim3(256), 0, ctx->stream, v, dst);
  E2_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int e2_transpose_ndhwc_to_ncdhw(e2_ctx* ctx, const float* src,
                                           const e2_tensor5* dst) {
  E2_REQUIRE(ctx && src, "transpose: null argument");
  if (int rc = check_view(dst, "transpose dst")) return rc;
  View5 v = mk(dst);
  const long S = (long)v.d * v.h * v.w;
  dim3 grid((unsigned)((S + 31) / 32), (unsigned)((v.c + 31) / 32), (unsigned)v.n);
  hipLaunchKernelGGL(ndhwc_to_ncdhw_kernel, grid, dim3(256), 0, ctx->stream, src, v);
  E2_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int e2_softmax_nll_fwd(e2_ctx* ctx, const e2_tensor5* logits,
                                  const e2_tensor5* target, const e2_tensor5* probs,
                                  float* stats) {
  E2_REQUIRE(ctx && stats, "softmax_nll_fwd: null argument");
  if (int rc = check_view(logits, "softmax_nll_fwd logits")) return rc;
  if (int rc = check_view(target, "softmax_nll_fwd target")) return rc;
  if (int rc = check_view(probs, "softmax_nll_fwd probs")) return rc;
  E2_REQUIRE(target->c == 1 && target->n == logits->n && target->d == logits->d &&
                 target->h == logits->h && target->w == logits->w,
             "softmax_nll_fwd: target must be (n,1,d,h,w) matching logits");
  E2_REQUIRE(probs->c == logits->c && probs->d == logits->d && probs->h == logits->h &&
                 probs->w == logits->w && probs->n == logits->n,
             "softmax_nll_fwd: probs/logits shape mismatch");
  View5 l = mk(logits), t = mk(target), p = mk(probs);
  const long S = (long)l.d * l.h * l.w;
  dim3 grid((unsigned)((S + 255) / 256), 1, (unsigned)l.n);
  hipLaunchKernelGGL(softmax_nll_fwd_kernel, grid, dim3(256), 0, ctx->stream, l, t, p, stats);
  E2_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int e2_softmax_nll_bwd(e2_ctx* ctx, const e2_tensor5* probs,
                                  const e2_tensor5* target, const float* stats,
                                  const e2_tensor5* dlogits, float* loss_out) {
  E2_REQUIRE(ctx && stats, "softmax_nll_bwd: null argument");
  if (int rc = check_view(probs, "softmax_nll_bwd probs")) return rc;
  if (int rc = check_view(target, "softmax_nll_bwd target")) return rc;
  if (int rc = check_view(dlogits, "softmax_nll_bwd dlogits")) return rc;
  E2_REQUIRE(dlogits->c == probs->c && dlogits->d == probs->d && dlogits->h == probs->h &&
                 dlogits->w == probs->w && dlogits->n == probs->n && target->c == 1 &&
                 target->d == probs->d && target->h == probs->h && target->w == probs->w,
             "softmax_nll_bwd: shape mismatch");
  View5 p = mk(probs), t = mk(target), d = mk(dlogits);
  const long S = (long)p.d * p.h * p.w;
  dim3 grid((unsigned)((S + 255) / 256), 1, (unsigned)p.n);
  hipLaunchKernelGGL(softmax_nll_bwd_kernel, grid, dim3(256), 0, ctx->stream, p, t, stats, d,
                     loss_out, ctx->loss_sum_mode, ctx->loss_count_out);
  E2_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int e2_malis_nll(e2_ctx* ctx, const e2_tensor5* probs, const float* pos,
                            const float* neg, const float* norm, const e2_tensor5* dlogits,
                            float* loss_sum) {
  E2_REQUIRE(ctx && pos && neg && norm && loss_sum, "malis_nll: null argument");
  if (int rc = check_view(probs, "malis_nll probs")) return rc;
  E2_REQUIRE(probs->n == 1 && probs->c >= 2 && probs->c % 2 == 0 && probs->c <= 2 * 65535,
             "malis_nll: probs must be (1, 2E, d, h, w)");
  View5 p = mk(probs), d = p;
  if (dlogits) {
    if (int rc = check_view(dlogits, "malis_nll dlogits")) return rc;
    E2_REQUIRE(dlogits->n == 1 && dlogits->c == probs->c && dlogits->d == probs->d &&
                   dlogits->h == probs->h && dlogits->w == probs->w,
               "malis_nll: dlogits/probs shape mismatch");
    d = mk(dlogits);
  }
  const long S = (long)p.d * p.h * p.w;
  dim3 grid((unsigned)((S + 255) / 256), (unsigned)(p.c / 2), 1);
  hipLaunchKernelGGL(malis_nll_kernel, grid, dim3(256), 0, ctx->stream, p, pos, neg, norm, d,
                     dlogits ? 1 : 0, loss_sum);
  E2_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int e2_adam_step(e2_ctx* ctx, float* p, const float* g, float* m, float* s,
                            size_t n, const int64_t* seg_off, const float* seg_reg,
                            int n_seg, const float* hyper) {
  return e2_adam_step_ex(ctx, p, const_cast<float*>(g), m, s, n, seg_off, seg_reg, n_seg, hyper,
                         nullptr, 1.f, 0);
}

extern "C" int e2_adam_step_ex(e2_ctx* ctx, float* p, float* g, float* m, float* s,
                               size_t n, const int64_t* seg_off, const float* seg_reg,
                               int n_seg, const float* hyper, const float* gdiv, float gmul,
                               int zero_g) {
  E2_REQUIRE(ctx && p && g && m && s && seg_off && seg_reg && hyper && n_seg > 0,
             "adam_step: null argument");
  E2_REQUIRE(n_seg <= kOptMaxSeg, "adam_step: %d parameter tensors (at most %d)", n_seg, kOptMaxSeg);
  E2_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)s) & 15) == 0,
             "adam_step: arenas must be 16-byte aligned");
  const int grid = (int)std::min<size_t>(std::max<size_t>((n / 4 + 255) / 256, 1), ctx->num_cu);
  hipLaunchKernelGGL(adam_kernel, dim3(grid), dim3(256), 0, ctx->stream, p, g, m, s, n,
                     seg_off, seg_reg, n_seg, const_cast<float*>(hyper), gdiv, gmul, zero_g ? 1 : 0);
  E2_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int e2_sgd_step(e2_ctx* ctx, float* p, const float* g, float* d, size_t n,
                           const int64_t* seg_off, const float* seg_reg, int n_seg,
                           const float* hyper) {
  return e2_sgd_step_ex(ctx, p, const_cast<float*>(g), d, n, seg_off, seg_reg, n_seg, hyper,
                        nullptr, 1.f, 0);
}

extern "C" int e2_sgd_step_ex(e2_ctx* ctx, float* p, float* g, float* d, size_t n,
                              const int64_t* seg_off, const float* seg_reg, int n_seg,
                              const float* hyper, const float* gdiv, float gmul, int zero_g) {
  E2_REQUIRE(ctx && p && g && d && seg_off && seg_reg && hyper && n_seg > 0,
             "sgd_step: null argument");
  E2_REQUIRE(n_seg <= kOptMaxSeg, "sgd_step: %d parameter tensors (at most %d)", n_seg, kOptMaxSeg);
  E2_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)d) & 15) == 0, "sgd_step: arenas must be 16-byte aligned");
  const int grid = (int)std::min<size_t>(std::max<size_t>((n / 4 + 255) / 256, 1), 2 * (size_t)ctx->num_cu);
  hipLaunchKernelGGL(sgd_kernel, dim3(grid), dim3(256), 0, ctx->stream, p, g, d, n, seg_off,
                     seg_reg, n_seg, hyper, gdiv, gmul, zero_g ? 1 : 0);
  E2_CHECK_HIP(hipGetLastError());
  return 0;
}
