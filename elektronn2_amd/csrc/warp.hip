// warp.hip -- patch extraction with geometric (warp) and grey-value augmentation on
// the device (reference: data/transformations.py:337-492 warp_slice,
// :42-76 map_coordinates_nearest / _linear, data/cnndata.py:42-60 greyAugment).
//
// The reference cuts a sub-volume on the host, builds a (pz,px,py,3) float32 array of
// source coordinates with a tensordot and gathers with numba kernels, per patch, on the
// CPU.  Here the whole training volume is resident in HBM; one launch per patch computes
// the source coordinate of every destination voxel from the inverse matrix on the fly
// (fp32 FMA chain, homogeneous divide for perspective matrices) and gathers:
//   image channels : trilinear, indices by truncation (coordinates are >= 0 after the
//                    host-side corner check), weights as in map_coordinates_linear;
//   target channels: nearest neighbour with round-half-to-even (np.round) for the
//                    channels in `nearest_mask`, trilinear otherwise.
// Thread = destination voxel (y fastest: coalesced stores), loop over channels.
// Bound: HBM / L2 gather (8 reads + 1 write of 4 B per voxel and channel).
#include "common.hpp"
#include <algorithm>

namespace {

struct WarpP {
  const float* src;      // (F, Z, X, Y) view
  float* dst;            // (F, pz, px, py) dense
  int F, Z, X, Y;
  long ssC, ssZ, ssX;    // source strides (elements), Y stride 1
  int pz, px, py;
  float m[16];           // inverse matrix, row major
  int perspective;
  unsigned nearest_mask;
  int dz0, dx0, dy0;     // destination index offset (targets: the centred sub-block)
  float oz, ox, oy;      // subtracted from the source coordinate (targets: centring offset)
};

__global__ __launch_bounds__(256) void warp_gather_kernel(WarpP p) {
  const long n = (long)p.pz * p.px * p.py;
  const long i = blockIdx.x * 256L + threadIdx.x;
  if (i >= n) return;
  const int y = (int)(i % p.py);
  const long t = i / p.py;
  const int x = (int)(t % p.px), z = (int)(t / p.px);
  const float fz = (float)(z + p.dz0), fx = (float)(x + p.dx0), fy = (float)(y + p.dy0);
  float c0 = fmaf(p.m[0], fz, fmaf(p.m[1], fx, fmaf(p.m[2], fy, p.m[3])));
  float c1 = fmaf(p.m[4], fz, fmaf(p.m[5], fx, fmaf(p.m[6], fy, p.m[7])));
  float c2 = fmaf(p.m[8], fz, fmaf(p.m[9], fx, fmaf(p.m[10], fy, p.m[11])));
  if (p.perspective) {
    const float h = fmaf(p.m[12], fz, fmaf(p.m[13], fx, fmaf(p.m[14], fy, p.m[15])));
    c0 /= h; c1 /= h; c2 /= h;
  }
  c0 -= p.oz; c1 -= p.ox; c2 -= p.oy;
  // clamp: the host checked the corners, this only guards the gather against rounding
  const float u = fminf(fmaxf(c0, 0.f), (float)(p.Z - 1));
  const float v = fminf(fmaxf(c1, 0.f), (float)(p.X - 1));
  const float w = fminf(fmaxf(c2, 0.f), (float)(p.Y - 1));
  const int u0 = (int)u, v0 = (int)v, w0 = (int)w;
  const int u1 = min(u0 + 1, p.Z - 1), v1 = min(v0 + 1, p.X - 1), w1 = min(w0 + 1, p.Y - 1);
  const float du = u - (float)u0, dv = v - (float)v0, dw = w - (float)w0;
  const int un = min((int)rintf(u), p.Z - 1), vn = min((int)rintf(v), p.X - 1),
            wn = min((int)rintf(w), p.Y - 1);
  for (int f = 0; f < p.F; ++f) {
    const float* s = p.src + (long)f * p.ssC;
    float val;
    if ((p.nearest_mask >> f) & 1u) {
      val = s[(long)un * p.ssZ + (long)vn * p.ssX + wn];
    } else {
      const float* a0 = s + (long)u0 * p.ssZ;
      const float* a1 = s + (long)u1 * p.ssZ;
      const float s000 = a0[(long)v0 * p.ssX + w0], s001 = a0[(long)v0 * p.ssX + w1];
      const float s010 = a0[(long)v1 * p.ssX + w0], s011 = a0[(long)v1 * p.ssX + w1];
      const float s100 = a1[(long)v0 * p.ssX + w0], s101 = a1[(long)v0 * p.ssX + w1];
      const float s110 = a1[(long)v1 * p.ssX + w0], s111 = a1[(long)v1 * p.ssX + w1];
      const float eu = 1.f - du, ev = 1.f - dv, ew = 1.f - dw;
      val = s000 * eu * ev * ew + s100 * du * ev * ew + s010 * eu * dv * ew +
            s001 * eu * ev * dw + s101 * du * ev * dw + s011 * eu * dv * dw +
            s110 * du * dv * ew + s111 * du * dv * dw;
    }
    p.dst[(long)f * n + i] = val;
  }
}

// d[ch] = clip(d[ch] * alpha + c, 0, 1) ** gamma, in place on a dense (F, S) block
__global__ __launch_bounds__(256) void grey_augment_kernel(float* d, long S, float alpha,
                                                          float c, float gamma) {
  const long i = blockIdx.x * 256L + threadIdx.x;
  if (i >= S) return;
  float v = fmaf(d[i], alpha, c);
  v = fminf(fmaxf(v, 0.f), 1.f);
  d[i] = powf(v, gamma);
}

}  // namespace

/* dst[f][z][x][y] = interp(src[f], Minv . (z + dest_off[0], x + dest_off[1],
 * y + dest_off[2], 1) - src_off); src is a (1, F, Z, X, Y) view, dst a dense
 * (1, F, pz, px, py) tensor, Minv the 4x4 inverse warp matrix (row major, host memory);
 * channel f is gathered nearest-neighbour when bit f of nearest_mask is set (F <= 32). */
extern "C" int e2_warp_slice(e2_ctx* ctx, const e2_tensor5* src, const float* minv,
                             int perspective, unsigned nearest_mask, const int* dest_off,
                             const float* src_off, const e2_tensor5* dst) {
  E2_REQUIRE(ctx && src && src->ptr && dst && dst->ptr && minv, "warp_slice: null argument");
  E2_REQUIRE(src->n == 1 && dst->n == 1 && src->c == dst->c && src->c >= 1 && src->c <= 32,
             "warp_slice: needs n = 1 and 1..32 matching channels");
  E2_REQUIRE(dst->sh == dst->w && dst->sd == (int64_t)dst->h * dst->w &&
                 dst->sc == (int64_t)dst->d * dst->h * dst->w,
             "warp_slice: destination must be dense");
  E2_REQUIRE(src->d >= 1 && src->h >= 1 && src->w >= 1, "warp_slice: empty source");
  WarpP p;
  p.src = src->ptr; p.dst = dst->ptr;
  p.F = src->c; p.Z = src->d; p.X = src->h; p.Y = src->w;
  p.ssC = src->sc; p.ssZ = src->sd; p.ssX = src->sh;
  p.pz = dst->d; p.px = dst->h; p.py = dst->w;
  for (int i = 0; i < 16; ++i) p.m[i] = minv[i];
  p.perspective = perspective;
  p.nearest_mask = nearest_mask;
  p.dz0 = dest_off ? dest_off[0] : 0; p.dx0 = dest_off ? dest_off[1] : 0;
  p.dy0 = dest_off ? dest_off[2] : 0;
  p.oz = src_off ? src_off[0] : 0.f; p.ox = src_off ? src_off[1] : 0.f;
  p.oy = src_off ? src_off[2] : 0.f;
  const long n = (long)p.pz * p.px * p.py;
  E2_REQUIRE(n > 0 && n < (1L << 38), "warp_slice: bad patch size");
  hipLaunchKernelGGL(warp_gather_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     ctx->stream, p);
  E2_CHECK_HIP(hipGetLastError());
  return 0;
}

/* grey-value augmentation of ONE channel (dense block of n floats), in place */
extern "C" int e2_grey_augment(e2_ctx* ctx, float* d, size_t n, float alpha, float c,
                               float gamma) {
  E2_REQUIRE(ctx && d, "grey_augment: null argument");
  if (n == 0) return 0;
  hipLaunchKernelGGL(grey_augment_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     ctx->stream, d, (long)n, alpha, c, gamma);
  E2_CHECK_HIP(hipGetLastError());
  return 0;
}
