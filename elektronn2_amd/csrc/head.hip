// head.hip -- the classifier head of the segmentation nets, fused:
//   1x1x1 conv to n_class <= 4 features, 'lin' (neural.py:662-712 with a (1,1,1)
//   kernel; examples/neuro3d*.py, unet3d_lite.py: nm.Conv(out, 2, (1,1,1),
//   activation_func='lin')) -> channel softmax (computations.py:175-176) ->
//   MultinoulliNLL with sparse targets (loss.py:261-347), and their gradients.
// As separate launches (split-K GEMM + memset, bias pass, softmax/NLL, lin-act
// backward, wgrad, dgrad) the head costs ~12 launch-latency-bound kernels for
// 0.01 GF; fused it is two streaming passes over the C-channel input:
//   forward : work-group = 64 positions x 4 channel quarters; partial logits (fp32 FMA
//             chains, coalesced reads, weight rows as scalar operands) meet in LDS,
//             softmax, probs out, loss sum and labelled count by atomics.
//   backward: work-group = tiles of 32 positions; dlogits from probs/target, the
//             input tile through LDS; thread ci accumulates dW[c][ci] over all its
//             tiles in registers (one atomic per (c,ci) per work-group at the
//             end), every thread writes its share of dx = W^T dlogits.
// Bound: HBM (read C*S*4 B; backward also writes C*S*4 B).
#include "common.hpp"
#include <algorithm>

#define E2_EPS_NLL 1e-5f
#define E2_HEAD_MAXC 4

namespace {

struct HView {
  float* p;
  int n, c, d, h, w;
  long sn, sc, sd, sh;
};
HView hv(const e2_tensor5* t) {
  return HView{t->ptr, t->n, t->c, t->d, t->h, t->w, (long)t->sn, (long)t->sc, (long)t->sd,
               (long)t->sh};
}
__device__ __forceinline__ long hidx(const HView& v, int n, int z, int y, int x) {
  return (long)n * v.sn + (long)z * v.sd + (long)y * v.sh + x;
}
__device__ __forceinline__ float h_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// ---- forward ---------------------------------------------------------------------
// work-group = 64 positions x 4 channel quarters (one wave each): the positions of a
// net's last layer are few (13,690 for C-lite@183), so the channel loop is split to
// put four times as many waves on the chip; partial logits meet in LDS.
template <int NC>
__global__ __launch_bounds__(256) void head_fwd_kernel(HView x, const float* __restrict__ w,
                                                       const float* __restrict__ bias,
                                                       HView tg, int has_target, HView pr,
                                                       float* __restrict__ stats) {
  __shared__ float part[4][NC][64];
  const int S = x.d * x.h * x.w;
  const int p = threadIdx.x & 63, cq = threadIdx.x >> 6;
  const int s = blockIdx.x * 64 + p;
  const int n = blockIdx.z;
  const bool valid = s < S;
  int xx = 0, y = 0, z = 0;
  if (valid) {
    xx = s % x.w;
    const int t = s / x.w;
    y = t % x.h; z = t / x.h;
  }
  const int per = (x.c + 3) >> 2;
  const int c0 = cq * per, c1 = min(c0 + per, x.c);
  float acc[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) acc[c] = 0.f;
  if (valid) {
    const float* xp = x.p + hidx(x, n, z, y, xx);
#pragma unroll 10
    for (int ci = c0; ci < c1; ++ci) {
      const float v = xp[(long)ci * x.sc];
#pragma unroll
      for (int c = 0; c < NC; ++c) acc[c] = fmaf(w[c * x.c + ci], v, acc[c]);
    }
  }
#pragma unroll
  for (int c = 0; c < NC; ++c) part[cq][c][p] = acc[c];
  __syncthreads();
  if (cq != 0) return;
  float lsum = 0.f, nlab = 0.f;
  if (valid) {
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      acc[c] = ((part[0][c][p] + part[1][c][p]) + (part[2][c][p] + part[3][c][p])) + bias[c];
      m = fmaxf(m, acc[c]);
    }
    float den = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) den += expf(acc[c] - m);
    const float tv = has_target ? tg.p[hidx(tg, n, z, y, xx)] : -1.f;
    float* pp = pr.p + hidx(pr, n, z, y, xx);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const float pc = expf(acc[c] - m) / den;
      pp[(long)c * pr.sc] = pc;
      if (tv == (float)c) { lsum -= logf(pc + E2_EPS_NLL); nlab += 1.f; }
    }
  }
  if (has_target) {                       // wave 0 only
    const float a = h_wave_sum(lsum), b = h_wave_sum(nlab);
    if (p == 0) {
      if (a != 0.f) unsafeAtomicAdd(stats + 0, a);
      if (b != 0.f) unsafeAtomicAdd(stats + 1, b);
    }
  }
}

// ---- backward --------------------------------------------------------------------
constexpr int HT = 32;     // positions per tile

template <int NC>
__global__ __launch_bounds__(256) void head_bwd_kernel(HView x, const float* __restrict__ w,
                                                       HView pr, HView tg,
                                                       const float* __restrict__ stats,
                                                       HView dx, int want_dx, int accumulate,
                                                       float* __restrict__ part,
                                                       float* __restrict__ loss_out,
                                                       int tilesPerN, int nTiles, int sum_mode,
                                                       float* __restrict__ count_out) {
  extern __shared__ float hs[];
  const int C = x.c;
  float* dl = hs;                         // [NC][HT]
  float* xs = hs + NC * HT;               // [C][HT + 1]
  const int tid = threadIdx.x;
  const int S = x.d * x.h * x.w;
  float inv = 1.f / (stats[1] + E2_EPS_NLL);
  if (blockIdx.x == 0 && tid == 0 && loss_out) loss_out[0] = stats[0] * inv;
  if (blockIdx.x == 0 && tid == 0 && count_out) count_out[0] = stats[1];
  if (sum_mode) inv = 1.f;                  // (e2_set_loss_grad_mode: unnormalised gradients)
  // thread ci < C (two rounds when C > 256) owns dW[c][ci]
  float aw[2][NC];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int c = 0; c < NC; ++c) aw[r][c] = 0.f;
  float ab[NC];                           // dbias partials of threads 0..HT-1
#pragma unroll
  for (int c = 0; c < NC; ++c) ab[c] = 0.f;

  for (int tile = blockIdx.x; tile < nTiles; tile += gridDim.x) {
    const int n = tile / tilesPerN;
    const int s0 = (tile - n * tilesPerN) * HT;
    const int np = min(HT, S - s0);
    // dlogits of the tile's positions
    if (tid < HT) {
      float d[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) d[c] = 0.f;
      if (tid < np) {
        const int s = s0 + tid;
        const int xx = s % x.w;
        const int t = s / x.w;
        const int y = t % x.h, z = t / x.h;
        const float tv = tg.p[hidx(tg, n, z, y, xx)];
        const float* pp = pr.p + hidx(pr, n, z, y, xx);
        float pc[NC], pt = 0.f;
        int tc = -1;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          pc[c] = pp[(long)c * pr.sc];
          if (tv == (float)c) { tc = c; pt = pc[c]; }
        }
        // dL/dp_t = -inv/(p_t+eps);  dlogit_c = p_c*(dp_c - sum_k dp_k p_k)
        const float gpt = (tc >= 0) ? (-inv / (pt + E2_EPS_NLL)) * pt : 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) d[c] = gpt * ((c == tc ? 1.f : 0.f) - pc[c]);
      }
#pragma unroll
      for (int c = 0; c < NC; ++c) { dl[c * HT + tid] = d[c]; ab[c] += d[c]; }
    }
    // the input tile, coalesced: 256/HT channel rows of HT positions per pass
    const int p = tid & (HT - 1);
    int pz = 0, py = 0, px = 0;
    if (p < np) {
      const int s = s0 + p;
      px = s % x.w;
      const int t = s / x.w;
      py = t % x.h; pz = t / x.h;
    }
    {
      const long off = hidx(x, n, pz, py, px);
      const bool pv = p < np;
#pragma unroll 8
      for (int ci = tid / HT; ci < C; ci += 256 / HT)
        xs[ci * (HT + 1) + p] = pv ? x.p[off + (long)ci * x.sc] : 0.f;
    }
    __syncthreads();
    // dW partial sums: thread = input channel
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int ci = tid + 256 * r;
      if (ci < C) {
        const float* row = xs + ci * (HT + 1);
        for (int p = 0; p < HT; ++p) {
          const float v = row[p];
#pragma unroll
          for (int c = 0; c < NC; ++c) aw[r][c] = fmaf(dl[c * HT + p], v, aw[r][c]);
        }
      }
    }
    // dx = W^T dlogits, written (or accumulated) coalesced
    if (want_dx) {
      if (p < np) {
        float* dp = dx.p + hidx(dx, n, pz, py, px);
        float d[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) d[c] = dl[c * HT + p];
#pragma unroll 8
        for (int ci = tid / HT; ci < C; ci += 256 / HT) {
          float g = 0.f;
#pragma unroll
          for (int c = 0; c < NC; ++c) g = fmaf(w[c * C + ci], d[c], g);
          float* q = dp + (long)ci * dx.sc;
          *q = accumulate ? (*q + g) : g;
        }
      }
    }
    __syncthreads();
  }
  // flush: this work-group's partial sums, part[block][NC*C + NC] (plain stores; 400+
  // same-address atomics per address serialise for tens of microseconds)
  float* mine = part + (long)blockIdx.x * (NC * C + NC);
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int ci = tid + 256 * r;
    if (ci < C) {
#pragma unroll
      for (int c = 0; c < NC; ++c) mine[c * C + ci] = aw[r][c];
    }
  }
  if (tid < 64) {           // wave 0 (threads >= HT hold zeros)
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const float sb = h_wave_sum(ab[c]);
      if (tid == 0) mine[NC * C + c] = sb;
    }
  }
}

// dw[i] += sum_b part[b][i] (i < NC*C), dbias[c] += sum_b part[b][NC*C + c]
__global__ __launch_bounds__(256) void head_reduce_kernel(const float* __restrict__ part,
                                                          int nBlocks, int total, int nW,
                                                          float* dw, float* dbias) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int per = (nBlocks + gridDim.y - 1) / gridDim.y;
  const int b0 = blockIdx.y * per, b1 = min(b0 + per, nBlocks);
  float s = 0.f;
#pragma unroll 8
  for (int b = b0; b < b1; ++b) s += part[(long)b * total + idx];
  if (s != 0.f) unsafeAtomicAdd(idx < nW ? dw + idx : dbias + (idx - nW), s);
}

int head_check(const e2_tensor5* t, const char* name) {
  E2_REQUIRE(t && t->ptr, "%s: null tensor", name);
  E2_REQUIRE(t->n > 0 && t->c > 0 && t->d > 0 && t->h > 0 && t->w > 0 && t->n < 65536,
             "%s: bad shape (%d,%d,%d,%d,%d)", name, t->n, t->c, t->d, t->h, t->w);
  return 0;
}
bool same_sp(const e2_tensor5* a, const e2_tensor5* b) {
  return a->n == b->n && a->d == b->d && a->h == b->h && a->w == b->w;
}

}  // namespace

extern "C" int e2_head_supported(int cin, int ncls) {
  return ncls >= 2 && ncls <= E2_HEAD_MAXC && cin >= 1 && cin <= 512;
}

/* probs = softmax(W x + b) over the ncls channels; with a target also
 * stats[0] += sum(-log(p_target + 1e-5)), stats[1] += #labelled (zero stats first). */
extern "C" int e2_head_fwd(e2_ctx* ctx, const e2_tensor5* x, const float* w, const float* bias,
                           int ncls, const e2_tensor5* target, const e2_tensor5* probs,
                           float* stats) {
  E2_REQUIRE(ctx && w && bias, "head_fwd: null argument");
  if (int rc = head_check(x, "head_fwd x")) return rc;
  if (int rc = head_check(probs, "head_fwd probs")) return rc;
  E2_REQUIRE(e2_head_supported(x->c, ncls), "head_fwd: unsupported cin=%d ncls=%d", x->c, ncls);
  E2_REQUIRE(probs->c == ncls && same_sp(x, probs), "head_fwd: probs shape mismatch");
  HView vt{};
  if (target) {
    if (int rc = head_check(target, "head_fwd target")) return rc;
    E2_REQUIRE(stats && target->c == 1 && same_sp(x, target), "head_fwd: target shape mismatch");
    vt = hv(target);
  }
  const long S = (long)x->d * x->h * x->w;
  E2_REQUIRE(S < (1L << 31), "head_fwd: channel too large");
  dim3 grid((unsigned)((S + 63) / 64), 1, (unsigned)x->n);
#define E2_HF(NC)                                                                      \
  hipLaunchKernelGGL((head_fwd_kernel<NC>), grid, dim3(256), 0, ctx->stream, hv(x), w, \
                     bias, vt, target ? 1 : 0, hv(probs), stats)
  if (ncls == 2) E2_HF(2); else if (ncls == 3) E2_HF(3); else E2_HF(4);
#undef E2_HF
  E2_CHECK_HIP(hipGetLastError());
  return 0;
}

static long head_grid(const e2_ctx* ctx, int n, long S) {
  const long nTiles = ((S + HT - 1) / HT) * n;
  return std::min<long>(nTiles, 2L * (ctx ? ctx->num_cu : 256));
}
extern "C" size_t e2_head_bwd_workspace_bytes(int n, int cin, int ncls, int d, int h, int w) {
  const long S = (long)d * h * w;
  const long blocks = std::min<long>(((S + HT - 1) / HT) * n, 2L * 1024);   // any CU count
  return sizeof(float) * (size_t)blocks * ((size_t)ncls * cin + ncls);
}

/* gradients of loss = stats[0]/(stats[1]+1e-5): dx (optional; accumulate_dx: +=),
 * dw[ncls*cin] and dbias[ncls] are ACCUMULATED (zero them first); loss_out optional.
 * ws: e2_head_bwd_workspace_bytes(n, cin, ncls, d, h, w) bytes. */
extern "C" int e2_head_bwd(e2_ctx* ctx, const e2_tensor5* x, const float* w,
                           const e2_tensor5* probs, const e2_tensor5* target,
                           const float* stats, const e2_tensor5* dx, int accumulate_dx,
                           float* dw, float* dbias, float* loss_out, void* ws,
                           size_t ws_bytes) {
  E2_REQUIRE(ctx && w && stats && dw && dbias && ws, "head_bwd: null argument");
  if (int rc = head_check(x, "head_bwd x")) return rc;
  if (int rc = head_check(probs, "head_bwd probs")) return rc;
  if (int rc = head_check(target, "head_bwd target")) return rc;
  const int ncls = probs->c;
  E2_REQUIRE(e2_head_supported(x->c, ncls), "head_bwd: unsupported cin=%d ncls=%d", x->c, ncls);
  E2_REQUIRE(same_sp(x, probs) && same_sp(x, target) && target->c == 1,
             "head_bwd: shape mismatch");
  HView vdx{};
  if (dx) {
    if (int rc = head_check(dx, "head_bwd dx")) return rc;
    E2_REQUIRE(dx->c == x->c && same_sp(x, dx), "head_bwd: dx shape mismatch");
    vdx = hv(dx);
  }
  const long S = (long)x->d * x->h * x->w;
  E2_REQUIRE(S < (1L << 31) - 64, "head_bwd: channel too large");
  const long tilesPerN = (S + HT - 1) / HT;
  const long nTiles = tilesPerN * x->n;
  E2_REQUIRE(nTiles < (1L << 31), "head_bwd: too many tiles");
  const int grid = (int)head_grid(ctx, x->n, S);
  const int total = ncls * x->c + ncls;
  E2_REQUIRE(ws_bytes >= sizeof(float) * (size_t)grid * total, "head_bwd: workspace too small");
  float* part = (float*)ws;
  const size_t lds = sizeof(float) * ((size_t)ncls * HT + (size_t)x->c * (HT + 1));
#define E2_HB(NC)                                                                        \
  do {                                                                                   \
    static bool attr_done = false;                                                       \
    if (!attr_done) {                                                                    \
      E2_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&head_bwd_kernel<NC>), \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
      attr_done = true;                                                                  \
    }                                                                                    \
    hipLaunchKernelGGL((head_bwd_kernel<NC>), dim3(grid), dim3(256), lds, ctx->stream,   \
                       hv(x), w, hv(probs), hv(target), stats, vdx, dx ? 1 : 0,          \
                       accumulate_dx, part, loss_out, (int)tilesPerN, (int)nTiles,       \
                       ctx->loss_sum_mode, ctx->loss_count_out);                         \
  } while (0)
  if (ncls == 2) E2_HB(2); else if (ncls == 3) E2_HB(3); else E2_HB(4);
#undef E2_HB
  E2_CHECK_HIP(hipGetLastError());
  hipLaunchKernelGGL(head_reduce_kernel, dim3(e2_cdiv(total, 256), std::min(grid, 16)), dim3(256),
                     0, ctx->stream, part, grid, total, ncls * x->c, dw, dbias);
  E2_CHECK_HIP(hipGetLastError());
  return 0;
}
