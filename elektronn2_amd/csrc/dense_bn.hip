// dense_bn.hip -- the two operators BASELINE config 1 (examples/mnist.py) needs beyond the
// conv / pool path: the Perceptron's dot product and train-mode batch normalisation.
//
//   Perceptron   neural.py:258-410   lin = dot(x.flatten(2), W), W of shape (n_in, n_f)
//                computations.py:179-213 `dot`
//   batch norm   neural.py:352-378 (Perceptron), 681-711 (Conv):
//                mean / std over every axis but 'f' (population std, + 1e-6),
//                out = act((gamma / std) * lin + b - gamma * mean / std),
//                running statistics  m <- 0.9995 m + 0.0005 mean  (same for std) as extra
//                updates of the training function
//
// Config 1 is the reference's CPU-runnable plumbing case (20 x 1 x 26 x 26 inputs, 2.3 MFLOP
// in the largest dot): these kernels are written for correctness and determinism -- one
// thread per output element, one work-group per channel, fixed summation order -- not for
// a roofline; SURVEY.md 8d owes no tuned kernel here.
#include "common.hpp"

namespace {

struct V5 {
  float* p;
  int n, c, d, h, w;
  long sn, sc, sd, sh;
};
static V5 mkv(const e2_tensor5* t) {
  return V5{t->ptr, t->n, t->c, t->d, t->h, t->w, t->sn, t->sc, t->sd, t->sh};
}

// y[i][j] = sum_k x[i][k] * w[k][j]          (x: n x k, w: k x m, y: n x m, row-major)
__global__ __launch_bounds__(256) void dense_fwd_kernel(const float* __restrict__ x,
                                                        const float* __restrict__ w,
                                                        float* __restrict__ y, int n, int k, int m) {
  const long idx = blockIdx.x * 256L + threadIdx.x;
  if (idx >= (long)n * m) return;
  const int i = (int)(idx / m), j = (int)(idx - (long)i * m);
  const float* xr = x + (long)i * k;
  float acc = 0.f;
  for (int q = 0; q < k; ++q) acc = fmaf(xr[q], w[(long)q * m + j], acc);
  y[idx] = acc;
}
// dx[i][q] (+)= sum_j dy[i][j] * w[q][j]
__global__ __launch_bounds__(256) void dense_dgrad_kernel(const float* __restrict__ dy,
                                                          const float* __restrict__ w,
                                                          float* __restrict__ dx, int n, int k, int m,
                                                          int accumulate) {
  const long idx = blockIdx.x * 256L + threadIdx.x;
  if (idx >= (long)n * k) return;
  const int i = (int)(idx / k), q = (int)(idx - (long)i * k);
  const float* g = dy + (long)i * m;
  const float* wr = w + (long)q * m;
  float acc = 0.f;
  for (int j = 0; j < m; ++j) acc = fmaf(g[j], wr[j], acc);
  dx[idx] = accumulate ? dx[idx] + acc : acc;
}
// dw[q][j] (+)= sum_i x[i][q] * dy[i][j]
__global__ __launch_bounds__(256) void dense_wgrad_kernel(const float* __restrict__ x,
                                                          const float* __restrict__ dy,
                                                          float* __restrict__ dw, int n, int k, int m,
                                                          int accumulate) {
  const long idx = blockIdx.x * 256L + threadIdx.x;
  if (idx >= (long)k * m) return;
  const int q = (int)(idx / m), j = (int)(idx - (long)q * m);
  float acc = 0.f;
  for (int i = 0; i < n; ++i) acc = fmaf(x[(long)i * k + q], dy[(long)i * m + j], acc);
  dw[idx] = accumulate ? dw[idx] + acc : acc;
}

// sum over a work-group of 256 threads, the same tree every time (deterministic)
__device__ __forceinline__ float wg_sum(float v, float* red) {
  const int t = threadIdx.x;
  red[t] = v;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (t < s) red[t] += red[t + s];
    __syncthreads();
  }
  const float r = red[0];
  __syncthreads();
  return r;
}

__device__ __forceinline__ long elem_off(const V5& v, int n, int c, unsigned s) {
  const unsigned hw = (unsigned)v.h * v.w;
  const unsigned z = s / hw, r = s - z * hw;
  const unsigned yy = r / v.w, xx = r - yy * v.w;
  return (long)n * v.sn + (long)c * v.sc + (long)z * v.sd + (long)yy * v.sh + xx;
}

// one work-group per channel.  train: statistics of x over (n, d, h, w); else the stored ones.
// save[0][c] = mean, save[1][c] = std (+ 1e-6) used; run_mean / run_std are updated in place
// when update != 0 (the training function's extra updates).
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(V5 x, const float* __restrict__ gamma,
                                                         const float* __restrict__ bias,
                                                         float* __restrict__ run_mean,
                                                         float* __restrict__ run_std, int train,
                                                         int update, int act, V5 out,
                                                         float* __restrict__ save) {
  __shared__ float red[256];
  const int c = blockIdx.x;
  const unsigned S = (unsigned)x.d * x.h * x.w;
  const long cnt = (long)x.n * S;
  float mean, sd;
  if (train) {
    float s1 = 0.f;
    for (int n = 0; n < x.n; ++n)
      for (unsigned s = threadIdx.x; s < S; s += 256) s1 += x.p[elem_off(x, n, c, s)];
    mean = wg_sum(s1, red) / (float)cnt;
    float s2 = 0.f;
    for (int n = 0; n < x.n; ++n)
      for (unsigned s = threadIdx.x; s < S; s += 256) {
        const float d = x.p[elem_off(x, n, c, s)] - mean;
        s2 += d * d;
      }
    sd = sqrtf(wg_sum(s2, red) / (float)cnt) + 1e-6f;
    if (threadIdx.x == 0 && update) {
      run_mean[c] = 0.9995f * run_mean[c] + 0.0005f * mean;
      run_std[c] = 0.9995f * run_std[c] + 0.0005f * sd;
    }
  } else {
    mean = run_mean[c];
    sd = run_std[c];
  }
  if (threadIdx.x == 0 && save) { save[c] = mean; save[x.c + c] = sd; }
  const float a = gamma[c] / sd;
  const float b = bias[c] - gamma[c] * mean / sd;
  for (int n = 0; n < x.n; ++n)
    for (unsigned s = threadIdx.x; s < S; s += 256) {
      float v = a * x.p[elem_off(x, n, c, s)] + b;
      if (act == E2_ACT_RELU) v = fmaxf(v, 0.f);
      out.p[elem_off(out, n, c, s)] = v;
    }
}

// backward of the above.  dpre = dout * act'(pre), pre recomputed from x and the saved
// statistics (relu'(0) = 0.5 as everywhere, computations.py:81-82);
//   dbias = sum dpre,  dgamma = sum dpre * xhat              (xhat = (x - mean) / std)
//   train:   dx = (gamma / std) * (dpre - mean(dpre) - xhat * mean(dpre * xhat) * std / (std - 1e-6))
//   predict: dx = (gamma / std) * dpre            (the statistics are constants)
__global__ __launch_bounds__(256) void bn_act_bwd_kernel(V5 dout, V5 x, const float* __restrict__ gamma,
                                                         const float* __restrict__ bias,
                                                         const float* __restrict__ save, int train,
                                                         int act, V5 dx, float* __restrict__ dgamma,
                                                         float* __restrict__ dbias) {
  __shared__ float red[256];
  const int c = blockIdx.x;
  const unsigned S = (unsigned)x.d * x.h * x.w;
  const long cnt = (long)x.n * S;
  const float mean = save[c], sd = save[x.c + c];
  const float g = gamma[c];
  const float a = g / sd, b = bias[c] - g * mean / sd;
  float s1 = 0.f, s2 = 0.f;
  for (int n = 0; n < x.n; ++n)
    for (unsigned s = threadIdx.x; s < S; s += 256) {
      const float xv = x.p[elem_off(x, n, c, s)];
      float d = dout.p[elem_off(dout, n, c, s)];
      if (act == E2_ACT_RELU) {
        const float pre = a * xv + b;
        d *= (pre > 0.f) ? 1.f : ((pre == 0.f) ? 0.5f : 0.f);
      }
      s1 += d;
      s2 += d * ((xv - mean) / sd);
    }
  const float sum_d = wg_sum(s1, red);
  const float sum_dx = wg_sum(s2, red);
  if (threadIdx.x == 0) {
    if (dbias) dbias[c] += sum_d;
    if (dgamma) dgamma[c] += sum_dx;
  }
  if (!dx.p) return;
  const float m_d = train ? sum_d / (float)cnt : 0.f;
  const float s0 = sd - 1e-6f;                               // the bare standard deviation
  const float m_dx = (train && s0 > 0.f) ? (sum_dx / (float)cnt) * (sd / s0) : 0.f;
  for (int n = 0; n < x.n; ++n)
    for (unsigned s = threadIdx.x; s < S; s += 256) {
      const float xv = x.p[elem_off(x, n, c, s)];
      float d = dout.p[elem_off(dout, n, c, s)];
      if (act == E2_ACT_RELU) {
        const float pre = a * xv + b;
        d *= (pre > 0.f) ? 1.f : ((pre == 0.f) ? 0.5f : 0.f);
      }
      dx.p[elem_off(dx, n, c, s)] = a * (d - m_d - ((xv - mean) / sd) * m_dx);
    }
}

static int view_ok(const e2_tensor5* t, const char* name) {
  E2_REQUIRE(t && t->ptr, "%s: null tensor", name);
  E2_REQUIRE(t->n > 0 && t->c > 0 && t->d > 0 && t->h > 0 && t->w > 0,
             "%s: empty tensor (%d,%d,%d,%d,%d)", name, t->n, t->c, t->d, t->h, t->w);
  return 0;
}
static bool same_shape(const e2_tensor5* a, const e2_tensor5* b) {
  return a->n == b->n && a->c == b->c && a->d == b->d && a->h == b->h && a->w == b->w;
}

}  // namespace

extern "C" int e2_dense_fwd(e2_ctx* ctx, const float* x, const float* w, float* y, int n, int k,
                            int m) {
  E2_REQUIRE(ctx && x && w && y && n > 0 && k > 0 && m > 0, "dense_fwd: bad argument");
  const long tot = (long)n * m;
  hipLaunchKernelGGL(dense_fwd_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0,
                     ctx->stream, x, w, y, n, k, m);
  E2_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int e2_dense_dgrad(e2_ctx* ctx, const float* dy, const float* w, float* dx, int n,
                              int k, int m, int accumulate) {
  E2_REQUIRE(ctx && dy && w && dx && n > 0 && k > 0 && m > 0, "dense_dgrad: bad argument");
  const long tot = (long)n * k;
  hipLaunchKernelGGL(dense_dgrad_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0,
                     ctx->stream, dy, w, dx, n, k, m, accumulate);
  E2_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int e2_dense_wgrad(e2_ctx* ctx, const float* x, const float* dy, float* dw, int n,
                              int k, int m, int accumulate) {
  E2_REQUIRE(ctx && x && dy && dw && n > 0 && k > 0 && m > 0, "dense_wgrad: bad argument");
  const long tot = (long)k * m;
  hipLaunchKernelGGL(dense_wgrad_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0,
                     ctx->stream, x, dy, dw, n, k, m, accumulate);
  E2_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int e2_batchnorm_act_fwd(e2_ctx* ctx, const e2_tensor5* x, const float* gamma,
                                    const float* bias, float* run_mean, float* run_std, int train,
                                    int update_running, int act, const e2_tensor5* out,
                                    float* save) {
  E2_REQUIRE(ctx && gamma && bias, "batchnorm_act_fwd: null argument");
  if (int rc = view_ok(x, "batchnorm_act_fwd x")) return rc;
  if (int rc = view_ok(out, "batchnorm_act_fwd out")) return rc;
  E2_REQUIRE(same_shape(x, out), "batchnorm_act_fwd: x / out shape mismatch");
  E2_REQUIRE(act == E2_ACT_LIN || act == E2_ACT_RELU, "batchnorm_act_fwd: bad act %d", act);
  E2_REQUIRE(train || (run_mean && run_std), "batchnorm_act_fwd: predict mode needs the stored statistics");
  E2_REQUIRE(!update_running || (train && run_mean && run_std),
             "batchnorm_act_fwd: running statistics are updated in train mode only");
  E2_REQUIRE((long)x->d * x->h * x->w < (1L << 31), "batchnorm_act_fwd: channel too large");
  hipLaunchKernelGGL(bn_act_fwd_kernel, dim3((unsigned)x->c), dim3(256), 0, ctx->stream, mkv(x),
                     gamma, bias, run_mean, run_std, train ? 1 : 0, update_running ? 1 : 0, act,
                     mkv(out), save);
  E2_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int e2_batchnorm_act_bwd(e2_ctx* ctx, const e2_tensor5* dout, const e2_tensor5* x,
                                    const float* gamma, const float* bias, const float* save,
                                    int train, int act, const e2_tensor5* dx, float* dgamma,
                                    float* dbias) {
  E2_REQUIRE(ctx && gamma && bias && save, "batchnorm_act_bwd: null argument");
  if (int rc = view_ok(dout, "batchnorm_act_bwd dout")) return rc;
  if (int rc = view_ok(x, "batchnorm_act_bwd x")) return rc;
  E2_REQUIRE(same_shape(x, dout), "batchnorm_act_bwd: x / dout shape mismatch");
  E2_REQUIRE(act == E2_ACT_LIN || act == E2_ACT_RELU, "batchnorm_act_bwd: bad act %d", act);
  V5 vdx{nullptr, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (dx) {
    if (int rc = view_ok(dx, "batchnorm_act_bwd dx")) return rc;
    E2_REQUIRE(same_shape(x, dx), "batchnorm_act_bwd: x / dx shape mismatch");
    vdx = mkv(dx);
  }
  hipLaunchKernelGGL(bn_act_bwd_kernel, dim3((unsigned)x->c), dim3(256), 0, ctx->stream, mkv(dout),
                     mkv(x), gamma, bias, save, train ? 1 : 0, act, vdx, dgamma, dbias);
  E2_CHECK_HIP(hipGetLastError());
  return 0;
}
