// bf16_prod.hip -- the bf16 operands of the NEXT kernels written by the kernel that produces the
// tensor ("the producers' epilogues"): bf16 mode's GEMMs read their operands as
// bf16 images in memory (conv_bf16.hip: channels-last pixels [n][z][kg][pixel][8];
// wgrad_bf16.hip: channel-major planes at the input's row pitch), and until round 4 every GEMM
// launch converted its f32 input first (prep_bf16_kernel / wgrad_bf16_cvt_kernel: ~20 % of the
// bf16 step).  Here the pointwise kernels between two GEMMs -- bias + activation + max-pool of a
// conv output (neural.py:705-712), and their backward (T.grad, model.py:182) -- write those
// images themselves, next to (or instead of) the f32 tensor.
//
// Thread layout: the f32 kernels of pointwise.hip give a thread four x of ONE channel; a
// channels-last pixel piece holds 8 channels of one position, so here a thread owns 8 channels x
// TWO input x (one or two pooled outputs) of one row: 8-byte row accesses per channel, one
// 16-byte store per pixel piece.
#include "common.hpp"
#include <algorithm>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float pf2 __attribute__((ext_vector_type(2), aligned(4)));

namespace {

struct BfDst {
  __bf16* cl;  int clKG, clD, clH, clW, oz, oy, ox;      // channels-last image + where (0,0,0) lands
  __bf16* pl;  long plPlane; int plPitch;                  // channel-major planes
};

struct BwP {
  const float* dout; long gsN, gsC, gsZ, gsY; int parts; long gpart;
  const float* src;  long ssN, ssC, ssZ, ssY;             // conv output (pre-bias) or activated output
  const float* bias;
  float* dy;         long dsN, dsC, dsZ, dsY;             // f32 gradient (optional)
  float* dbias;
  int N, C, Do, Ho, Wo;                                    // pooled dims (= dims of dout)
  int D;                                                   // planes of src / dy
  int act, out_mode;
  int VW;                                                  // thread columns per pooled row
  BfDst d;
};

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// backward of out = act(maxpool(y) + b): dy = dL/dy (every element equal to its window maximum
// receives the gradient; relu'(0) = 0.5), dbias += sum -- pointwise.hip's pool_bwd_fixed_kernel /
// act_bwd_out_kernel with the bf16 images as outputs.  NX = 2 / PX pooled outputs per thread.
template <int PZ, int PY, int PX>
__global__ __launch_bounds__(256) void bwd_bf16_kernel(BwP p) {
  constexpr int NO = 2 / PX;                               // pooled outputs per thread (along x)
  __shared__ float red[8][4];
  const int tid = threadIdx.x;
  const int col = blockIdx.x * 256 + tid;                  // (pooled plane, pooled row, thread column)
  const int kg = blockIdx.y;
  const int n = blockIdx.z;
  const bool live = col < p.Do * p.Ho * p.VW;
  const int zo = live ? col / (p.Ho * p.VW) : 0;
  const int rc = col - zo * (p.Ho * p.VW);
  const int yo = live ? rc / p.VW : 0;
  const int xo = live ? (rc - yo * p.VW) * NO : 0;
  const int nvo = live ? min(NO, p.Wo - xo) : 0;           // valid pooled outputs
  float gs[8];
  bf16x8 piece[PZ][PY][2];
  // phase 1: every load of the thread's 8 channels (the stores below may alias them for all the
  // compiler knows: interleaved, each channel waited for the one before it)
  float g[8][NO], w[8][PZ][PY][2], bv[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = kg * 8 + j;
    const bool cv = c < p.C && nvo > 0;
    const float* gp = p.dout + (long)n * p.gsN + (long)min(c, p.C - 1) * p.gsC + (long)zo * p.gsZ + (long)yo * p.gsY + xo;
    float u[8][NO];
#pragma unroll
    for (int q = 0; q < 8; ++q)
#pragma unroll
      for (int o = 0; o < NO; ++o) u[q][o] = (cv && q < p.parts && o < nvo) ? gp[q * p.gpart + o] : 0.f;
#pragma unroll
    for (int o = 0; o < NO; ++o)
      g[j][o] = ((u[0][o] + u[1][o]) + (u[2][o] + u[3][o])) + ((u[4][o] + u[5][o]) + (u[6][o] + u[7][o]));
    const float* sp = p.src + (long)n * p.ssN + (long)min(c, p.C - 1) * p.ssC + (long)(zo * PZ) * p.ssZ +
                      (long)(yo * PY) * p.ssY + xo * PX;
#pragma unroll
    for (int a = 0; a < PZ; ++a)
#pragma unroll
      for (int b = 0; b < PY; ++b) {
        const float* r = sp + a * p.ssZ + b * p.ssY;
        if (cv && nvo * PX == 2) { const pf2 v = *reinterpret_cast<const pf2*>(r); w[j][a][b][0] = v[0]; w[j][a][b][1] = v[1]; }
        else if (cv) { w[j][a][b][0] = r[0]; w[j][a][b][1] = -INFINITY; }
        else { w[j][a][b][0] = 0.f; w[j][a][b][1] = 0.f; }
      }
    bv[j] = (p.bias && c < p.C) ? p.bias[c] : 0.f;
  }
  // phase 2: slopes, row sums, stores
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = kg * 8 + j;
    const bool cv = c < p.C && nvo > 0;
    float m[NO];
#pragma unroll
    for (int o = 0; o < NO; ++o) m[o] = -INFINITY;
    if (cv) {
#pragma unroll
      for (int a = 0; a < PZ; ++a)
#pragma unroll
        for (int b = 0; b < PY; ++b)
#pragma unroll
          for (int e = 0; e < 2; ++e) m[e / PX] = fmaxf(m[e / PX], w[j][a][b][e]);
    }
    float s = 0.f;
#pragma unroll
    for (int o = 0; o < NO; ++o) {
      if (p.act == E2_ACT_RELU) {
        if (p.out_mode) g[j][o] *= (m[o] > 0.f) ? 1.f : (__builtin_signbit(m[o]) ? 0.f : 0.5f);
        else { const float pre = m[o] + bv[j]; g[j][o] *= (pre > 0.f) ? 1.f : ((pre == 0.f) ? 0.5f : 0.f); }
      }
      if (o < nvo && cv) s += g[j][o];
    }
    gs[j] = s;
    float* dp = p.dy ? p.dy + (long)n * p.dsN + (long)c * p.dsC + (long)(zo * PZ) * p.dsZ + (long)(yo * PY) * p.dsY + xo * PX : nullptr;
    __bf16* pp = p.d.pl ? p.d.pl + (((long)n * p.C + c) * p.D + zo * PZ) * p.d.plPlane +
                              (long)(yo * PY) * p.d.plPitch + xo * PX : nullptr;
#pragma unroll
    for (int a = 0; a < PZ; ++a)
#pragma unroll
      for (int b = 0; b < PY; ++b) {
        float v[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const bool ev = cv && (e / PX) < nvo;
          v[e] = (ev && w[j][a][b][e] == m[e / PX]) ? g[j][e / PX] : 0.f;
          piece[a][b][e][j] = (__bf16)v[e];
        }
        if (!cv) continue;
        const bool both = nvo * PX == 2;
        if (dp) {
          float* d = dp + a * p.dsZ + b * p.dsY;
          if (both) { pf2 t = {v[0], v[1]}; *reinterpret_cast<pf2*>(d) = t; }
          else d[0] = v[0];
        }
        if (pp) {
          __bf16* d = pp + a * p.d.plPlane + (long)b * p.d.plPitch;
          if (both && (((uintptr_t)d & 3) == 0)) {
            bf16x2 t = {(__bf16)v[0], (__bf16)v[1]};
            *reinterpret_cast<bf16x2*>(d) = t;
          } else {
            d[0] = (__bf16)v[0];
            if (both) d[1] = (__bf16)v[1];
          }
        }
      }
  }
  if (p.d.cl && nvo > 0) {
#pragma unroll
    for (int a = 0; a < PZ; ++a)
#pragma unroll
      for (int b = 0; b < PY; ++b)
#pragma unroll
        for (int e = 0; e < 2; ++e)
          if ((e / PX) < nvo) {
            const long pix = (long)(yo * PY + b + p.d.oy) * p.d.clW + (xo * PX + e + p.d.ox);
            __bf16* dst = p.d.cl + (((((long)n * p.d.clD + zo * PZ + a + p.d.oz) * p.d.clKG + kg) * p.d.clH) * p.d.clW + pix) * 8;
            *reinterpret_cast<bf16x8*>(dst) = piece[a][b][e];
          }
  }
  if (p.dbias != nullptr) {
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float v = wave_sum_f(gs[j]);
      if (lane == 0) red[j][wave] = v;
    }
    __syncthreads();
    if (tid < 8) {
      const int c = kg * 8 + tid;
      const float tot = (red[tid][0] + red[tid][1]) + (red[tid][2] + red[tid][3]);
      if (c < p.C && tot != 0.f) unsafeAtomicAdd(p.dbias + c, tot);
    }
  }
}

// forward: out = act(maxpool(y) + b) as f32 (optional) and as the next layer's channels-last
// bf16 input copy
struct FwP {
  const float* y; long ysN, ysC, ysZ, ysY; int parts; long ypart;
  const float* bias;
  float* out; long osN, osC, osZ, osY;
  int N, C, Do, Ho, Wo;                                    // pooled dims
  int act, VW;
  BfDst d;
};
template <int PZ, int PY, int PX>
__global__ __launch_bounds__(256) void fwd_bf16_kernel(FwP p) {
  constexpr int NO = 2;                                    // pooled outputs per thread
  typedef float pfN __attribute__((ext_vector_type(NO * PX), aligned(4)));
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col >= p.Do * p.Ho * p.VW) return;
  const int kg = blockIdx.y;
  const int n = blockIdx.z;
  const int zo = col / (p.Ho * p.VW);
  const int rc = col - zo * (p.Ho * p.VW);
  const int yo = rc / p.VW, xo = (rc - yo * p.VW) * NO;
  const int nvo = min(NO, p.Wo - xo);
  bf16x8 piece[NO];
  // phase 1: every load of the thread's 8 channels (all in flight together), phase 2: stores
  float m[8][NO];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = kg * 8 + j;
#pragma unroll
    for (int o = 0; o < NO; ++o) m[j][o] = -INFINITY;
    if (c < p.C) {
      const float* sp = p.y + (long)n * p.ysN + (long)c * p.ysC + (long)(zo * PZ) * p.ysZ + (long)(yo * PY) * p.ysY + xo * PX;
#pragma unroll
      for (int a = 0; a < PZ; ++a)
#pragma unroll
        for (int b = 0; b < PY; ++b) {
          // the row piece of the thread's two windows: NO * PX consecutive floats, one access
          const float* r = sp + a * p.ysZ + b * p.ysY;
          float v[NO * PX];
          if (nvo == NO) {
            const pfN t = *reinterpret_cast<const pfN*>(r);
#pragma unroll
            for (int e = 0; e < NO * PX; ++e) v[e] = t[e];
            for (int q = 1; q < p.parts; ++q) {      // (split-K partial sums; 16-byte pieces)
              const pfN u = *reinterpret_cast<const pfN*>(r + q * p.ypart);
#pragma unroll
              for (int e = 0; e < NO * PX; ++e) v[e] += u[e];
            }
          } else {
#pragma unroll
            for (int e = 0; e < NO * PX; ++e) {
              v[e] = -INFINITY;
              if (e < nvo * PX) {
                v[e] = r[e];
                for (int q = 1; q < p.parts; ++q) v[e] += r[q * p.ypart + e];
              }
            }
          }
#pragma unroll
          for (int e = 0; e < NO * PX; ++e) m[j][e / PX] = fmaxf(m[j][e / PX], v[e]);
        }
      const float bv = p.bias ? p.bias[c] : 0.f;
#pragma unroll
      for (int o = 0; o < NO; ++o) {
        float v = m[j][o] + bv;
        if (p.act == E2_ACT_RELU) v = fmaxf(v, 0.f);        // (as pool_fwd_fixed_kernel: the backward reads y, not out)
        m[j][o] = (o < nvo) ? v : 0.f;
      }
    } else {
#pragma unroll
      for (int o = 0; o < NO; ++o) m[j][o] = 0.f;
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = kg * 8 + j;
    if (p.out && c < p.C) {
      float* op = p.out + (long)n * p.osN + (long)c * p.osC + (long)zo * p.osZ + (long)yo * p.osY + xo;
      if (nvo == 2) { pf2 t = {m[j][0], m[j][1]}; *reinterpret_cast<pf2*>(op) = t; }
      else op[0] = m[j][0];
    }
#pragma unroll
    for (int o = 0; o < NO; ++o) piece[o][j] = (__bf16)m[j][o];
  }
  if (p.d.cl) {
#pragma unroll
    for (int o = 0; o < NO; ++o)
      if (o < nvo) {
        const long pix = (long)(yo + p.d.oy) * p.d.clW + (xo + o + p.d.ox);
        __bf16* dst = p.d.cl + (((((long)n * p.d.clD + zo + p.d.oz) * p.d.clKG + kg) * p.d.clH) * p.d.clW + pix) * 8;
        *reinterpret_cast<bf16x8*>(dst) = piece[o];
      }
  }
}

int check_dst(const e2_bf16_dst* dst, BfDst* d, int n, int c, int dd, int h, int w, const char* who) {
  d->cl = nullptr; d->pl = nullptr;
  d->clKG = d->clD = d->clH = d->clW = d->oz = d->oy = d->ox = 0; d->plPlane = 0; d->plPitch = 0;
  if (!dst) return 0;
  if (dst->cl) {
    E2_REQUIRE(((uintptr_t)dst->cl & 15) == 0, "%s: channels-last image must be 16-byte aligned", who);
    E2_REQUIRE(dst->cl_kg * 8 >= c && dst->cl_oz >= 0 && dst->cl_oy >= 0 && dst->cl_ox >= 0 &&
                   dst->cl_d >= dd + dst->cl_oz && dst->cl_h >= h + dst->cl_oy && dst->cl_w >= w + dst->cl_ox,
               "%s: a (%d,%d,%d,%d) tensor does not fit the channels-last image (%d groups, %d x %d x %d, offset %d,%d,%d)",
               who, c, dd, h, w, dst->cl_kg, dst->cl_d, dst->cl_h, dst->cl_w, dst->cl_oz, dst->cl_oy, dst->cl_ox);
    d->cl = reinterpret_cast<__bf16*>(dst->cl);
    d->clKG = dst->cl_kg; d->clD = dst->cl_d; d->clH = dst->cl_h; d->clW = dst->cl_w;
    d->oz = dst->cl_oz; d->oy = dst->cl_oy; d->ox = dst->cl_ox;
  }
  if (dst->pl) {
    E2_REQUIRE(dst->pl_pitch >= w && dst->pl_plane >= (int64_t)(h - 1) * dst->pl_pitch + w,
               "%s: rows of %d at pitch %d do not fit planes of %ld", who, w, dst->pl_pitch, (long)dst->pl_plane);
    d->pl = reinterpret_cast<__bf16*>(dst->pl);
    d->plPlane = dst->pl_plane; d->plPitch = dst->pl_pitch;
  }
  return 0;
}

}  // namespace

extern "C" int e2_pool_bias_act_bwd_bf16(e2_ctx* ctx, const e2_tensor5* dout, int64_t part_stride,
                                         int parts, const e2_tensor5* src, const float* bias,
                                         int pz, int py, int px, int act, const e2_tensor5* dy,
                                         float* dbias, const e2_bf16_dst* dst) {
  E2_REQUIRE(ctx && dout && dout->ptr && src && src->ptr, "pool_bias_act_bwd_bf16: null argument");
  E2_REQUIRE(parts >= 1 && parts <= 8, "pool_bias_act_bwd_bf16: %d parts", parts);
  E2_REQUIRE(act == E2_ACT_RELU || act == E2_ACT_LIN, "pool_bias_act_bwd_bf16: relu / lin only (act %d)", act);
  const bool out_mode = bias == nullptr;      // src = the activated output (signed zeros), no pooling
  E2_REQUIRE(!out_mode || (pz == 1 && py == 1 && px == 1), "pool_bias_act_bwd_bf16: a pooled layer needs its conv output and bias");
  E2_REQUIRE(src->n == dout->n && src->c == dout->c && dout->d == src->d / pz && dout->h == src->h / py &&
                 dout->w == src->w / px, "pool_bias_act_bwd_bf16: dout is not pool(src)");
  E2_REQUIRE(!dy || !dy->ptr || (dy->n == src->n && dy->c == src->c && dy->d == src->d && dy->h == src->h && dy->w == src->w),
             "pool_bias_act_bwd_bf16: dy shape");
  BwP p;
  p.dout = dout->ptr; p.gsN = dout->sn; p.gsC = dout->sc; p.gsZ = dout->sd; p.gsY = dout->sh;
  p.parts = parts; p.gpart = part_stride;
  p.src = src->ptr; p.ssN = src->sn; p.ssC = src->sc; p.ssZ = src->sd; p.ssY = src->sh;
  p.bias = bias;
  p.dy = (dy && dy->ptr) ? dy->ptr : nullptr;
  if (p.dy) { p.dsN = dy->sn; p.dsC = dy->sc; p.dsZ = dy->sd; p.dsY = dy->sh; }
  else p.dsN = p.dsC = p.dsZ = p.dsY = 0;
  p.dbias = dbias;
  p.N = dout->n; p.C = dout->c; p.Do = dout->d; p.Ho = dout->h; p.Wo = dout->w;
  p.act = act; p.out_mode = out_mode ? 1 : 0;
  if (int rc = check_dst(dst, &p.d, src->n, src->c, src->d, src->h, src->w, "pool_bias_act_bwd_bf16")) return rc;
  p.D = src->d;
  const int NO = 2 / px;
  E2_REQUIRE(px == 1 || px == 2, "pool_bias_act_bwd_bf16: x windows of 1 or 2");
  p.VW = (p.Wo + NO - 1) / NO;
  E2_REQUIRE((long)p.Do * p.Ho * p.VW < (1L << 30), "pool_bias_act_bwd_bf16: planes too large");
  const dim3 grid((unsigned)((p.Do * p.Ho * p.VW + 255) / 256), (unsigned)((p.C + 7) / 8), (unsigned)p.N);
  E2_REQUIRE(p.N < 65536 && (p.C + 7) / 8 < 65536, "pool_bias_act_bwd_bf16: grid too large");
#define E2_L(Z, Y, X) if (pz == Z && py == Y && px == X) { \
    hipLaunchKernelGGL((bwd_bf16_kernel<Z, Y, X>), grid, dim3(256), 0, ctx->stream, p); \
    E2_CHECK_HIP(hipGetLastError()); return 0; }
  E2_L(1, 1, 1) E2_L(1, 2, 2) E2_L(2, 1, 1) E2_L(2, 2, 2)
#undef E2_L
  e2_set_error("pool_bias_act_bwd_bf16: no instance for the window (%d,%d,%d)", pz, py, px);
  return 2;
}

extern "C" int e2_pool_bias_act_fwd_bf16(e2_ctx* ctx, const e2_tensor5* y, int64_t part_stride,
                                         int parts, const float* bias, int pz, int py, int px,
                                         int act, const e2_tensor5* out, const e2_bf16_dst* dst) {
  E2_REQUIRE(ctx && y && y->ptr && out, "pool_bias_act_fwd_bf16: null argument");
  E2_REQUIRE(parts >= 1 && parts <= 8, "pool_bias_act_fwd_bf16: %d parts", parts);
  E2_REQUIRE(act == E2_ACT_RELU || act == E2_ACT_LIN, "pool_bias_act_fwd_bf16: relu / lin only (act %d)", act);
  E2_REQUIRE(out->n == y->n && out->c == y->c && out->d == y->d / pz && out->h == y->h / py && out->w == y->w / px,
             "pool_bias_act_fwd_bf16: out is not pool(y)");
  FwP p;
  p.y = y->ptr; p.ysN = y->sn; p.ysC = y->sc; p.ysZ = y->sd; p.ysY = y->sh; p.parts = parts; p.ypart = part_stride;
  p.bias = bias;
  p.out = out->ptr; p.osN = out->sn; p.osC = out->sc; p.osZ = out->sd; p.osY = out->sh;
  p.N = out->n; p.C = out->c; p.Do = out->d; p.Ho = out->h; p.Wo = out->w;
  p.act = act;
  if (int rc = check_dst(dst, &p.d, out->n, out->c, out->d, out->h, out->w, "pool_bias_act_fwd_bf16")) return rc;
  E2_REQUIRE(!p.d.pl, "pool_bias_act_fwd_bf16: channels-last output only");
  p.VW = (p.Wo + 1) / 2;
  E2_REQUIRE((long)p.Do * p.Ho * p.VW < (1L << 30), "pool_bias_act_fwd_bf16: planes too large");
  const dim3 grid((unsigned)((p.Do * p.Ho * p.VW + 255) / 256), (unsigned)(p.d.cl ? p.d.clKG : (p.C + 7) / 8), (unsigned)p.N);
  E2_REQUIRE(p.N < 65536 && grid.y < 65536, "pool_bias_act_fwd_bf16: grid too large");
#define E2_L(Z, Y, X) if (pz == Z && py == Y && px == X) { \
    hipLaunchKernelGGL((fwd_bf16_kernel<Z, Y, X>), grid, dim3(256), 0, ctx->stream, p); \
    E2_CHECK_HIP(hipGetLastError()); return 0; }
  E2_L(1, 1, 1) E2_L(1, 2, 2) E2_L(2, 1, 1) E2_L(2, 2, 2)
#undef E2_L
  e2_set_error("pool_bias_act_fwd_bf16: no instance for the window (%d,%d,%d)", pz, py, px);
  return 2;
}
