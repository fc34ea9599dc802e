// conv_first_mfma.hip -- the FIRST layer (Cin = 1, kd = 1, pooling (1,py,px) or none, bias,
// activation) on the matrix cores.  Same entry points and semantics as conv_first.hip (which
// stays as the fallback for channel counts this file has no instance for); reference ops:
// Conv._make_output conv -> pool -> +b -> act (neural.py:662-712) on the raw input, and
// T.grad of it wrt w and b (model.py:182).
//
// With one input channel the layer is 1-2 GF of arithmetic on 3 MB of input: as VALU code
// (conv_first.hip) it ran at 15-30 % of the vector peak -- 37 + 100 us of the neuro3d step.
// v_mfma_f32_4x4x1_16b_f32 with the A operand broadcast from one block (cbsz:4 abid:b) is
// an outer product D[4][64 lanes] += A[4] (x) B[64 lanes], K = 1, so a K of 16 or 36 taps
// and a channel count of 20 cost no padding (see igemm4_core.hpp):
//   forward / recompute:  lanes = 64 consecutive conv columns, A = 4 output channels of one
//       (flipped) tap, B = the input value under that tap, one ds_read_b32 per (row, tap
//       column) from the work-group's input tile in LDS.  A register holds 16 / MG taps x MG
//       channel groups, picked by abid.  py conv rows per wave: the y-pool is a max of two
//       accumulators, the x-pool a max with the neighbouring lane.
//   weight gradient: lanes = TAPS, one instruction per conv position p:
//       dW[4 channels][tap] += dy[4 channels][p] (x) x[p + tap]; B is one ds_read_b32 with a
//       per-lane tap offset, A = the masked output gradient, transposed through LDS into
//       "16 positions x 4 channels" registers (abid = position).  With <= 16 taps the 16
//       blocks of the instruction are used WITHOUT the broadcast: block = (position of a
//       group of four, tap quad), four positions per instruction, the four position classes
//       are added up at the end.  Every element equal to its window's maximum receives the
//       gradient (Theano's MaxPoolGrad), relu'(0) = 0.5.
//   Work-groups are persistent over the tiles; a wave keeps its dW / dbias partial sums in
//   registers and writes them once (workspace + the reduce kernel of conv_first.hip).
#include "common.hpp"
#include <algorithm>
#include <utility>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// compile-time loops (the abid of an MFMA is an immediate)
template <class F, int... I>
__device__ __forceinline__ void fm_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void fm_for(F&& f) {
  fm_for_impl(f, std::make_integer_sequence<int, N>{});
}
template <int ABID>
__device__ __forceinline__ f32x4 fm_mfma(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, ABID, 0);
}

namespace {

typedef __bf16 fm_bf16x8 __attribute__((ext_vector_type(8)));

struct FirstM {
  const float* x;      // (n,1,d,h,w) view
  const float* w;      // [cout][1][1][kh][kw] dense
  const float* bias;
  float* out;          // forward: pooled output
  const float* dout;   // backward: gradient of the pooled output
  int N, Cout, D, H, W;          // input dims
  int Ho, Wo;                    // pooled output dims
  long xsN, xsD, xsH;
  long osN, osC, osD, osH;       // strides of out / dout
  int act;
  int tilesX, tilesY, nTiles;
  int dbg;                       // timing ablations (debug build, E2_FM_DBG): 1 = no tile loads,
                                 // 2 = no conv MFMAs, 4 = no output stores / no dW MFMAs
  // forward, bf16 mode: the channels-last bf16 copy of `out` that the next conv's kernels read,
  // [N][D][nxKG][Ho * Wo][8] -- or nullptr (written by the producer)
  __bf16* nxb;
  int nxKG;
};

// The ablation switches exist in a `make DEBUG_ENV=1` build only.  As run-time branches on a
// kernel argument they cost the RELEASE backward 25 registers (103 -> 128: occupancy 3 -> 2
// with 3 persistent work-groups launched per CU) and +16 us per step.
#ifdef E2_DEBUG_ENV
#define FM_DBG(p, bit) (((p).dbg & (bit)) != 0)
#else
#define FM_DBG(p, bit) false
#endif

constexpr int kTW = 72;                  // LDS row stride of the input tile (64 + kw - 1 <= 69)

template <int KH, int KW, int PY, int PX, int MG>
struct Geo {
  static constexpr int T = KH * KW;
  static constexpr int TPV = 16 / MG;                 // taps per A register
  static constexpr int NAV = (T + TPV - 1) / TPV;     // A registers
  static constexpr int RT = 4 * PY + KH - 1;          // input rows of a tile (4 waves x PY rows)
  static constexpr int CH = 4 * MG;
};

// A registers of the forward product: lane 4b + i, b = tapslot * MG + g, holds
// w[4g + i][T - 1 - (v * TPV + tapslot)]  (F1: the convolution flips the kernel)
template <int KH, int KW, int PY, int PX, int MG>
__device__ __forceinline__ void load_weights(const FirstM& p, int lane,
                                             float (&aw)[Geo<KH, KW, PY, PX, MG>::NAV]) {
  using G = Geo<KH, KW, PY, PX, MG>;
  const int b = lane >> 2, i = lane & 3;
  const int slot = b / MG, g = b - slot * MG;
  const int ch = 4 * g + i;
#pragma unroll
  for (int v = 0; v < G::NAV; ++v) {
    const int tap = v * G::TPV + slot;
    aw[v] = (slot < G::TPV && tap < G::T && ch < p.Cout) ? p.w[ch * G::T + (G::T - 1 - tap)] : 0.f;
  }
}

// input tile of the work-group -> LDS, zero outside the image
template <int KH, int KW, int PY, int PX, int MG>
__device__ __forceinline__ void load_tile(const FirstM& p, int tile, float* xt, int& n, int& z,
                                          int& row0, int& col0) {
  using G = Geo<KH, KW, PY, PX, MG>;
  const int tx_ = tile % p.tilesX;
  int r = tile / p.tilesX;
  const int ty_ = r % p.tilesY; r /= p.tilesY;
  z = r % p.D;
  n = r / p.D;
  row0 = ty_ * 4 * PY;
  col0 = tx_ * 64;
  const float* src = p.x + (long)n * p.xsN + (long)z * p.xsD;
  for (int e = threadIdx.x; e < G::RT * kTW; e += 256) {
    const int i = e / kTW, j = e - i * kTW;
    const int y = row0 + i, xx = col0 + j;
    xt[e] = (y < p.H && xx < p.W && !FM_DBG(p, 1)) ? src[(long)y * p.xsH + xx] : 0.f;
  }
}

// conv values of the wave's PY rows x 64 columns: acc[g][nb][r] = channel 4g + r
template <int KH, int KW, int PY, int PX, int MG>
__device__ __forceinline__ void conv_rows(const float* xw, int lane,
                                          const float (&aw)[Geo<KH, KW, PY, PX, MG>::NAV],
                                          f32x4 (&acc)[MG][PY]) {
  using G = Geo<KH, KW, PY, PX, MG>;
#pragma unroll
  for (int g = 0; g < MG; ++g)
#pragma unroll
    for (int nb = 0; nb < PY; ++nb) acc[g][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
  fm_for<PY + KH - 1>([&](auto rr_) {
    constexpr int rr = decltype(rr_)::value;
    float bx[KW];
#pragma unroll
    for (int tx = 0; tx < KW; ++tx) bx[tx] = xw[rr * kTW + lane + tx];
    fm_for<PY>([&](auto nb_) {
      constexpr int nb = decltype(nb_)::value;
      constexpr int ty = rr - nb;
      if constexpr (ty >= 0 && ty < KH) {
        fm_for<KW>([&](auto tx_) {
          constexpr int tx = decltype(tx_)::value;
          constexpr int tap = ty * KW + tx;
          fm_for<MG>([&](auto g_) {
            constexpr int g = decltype(g_)::value;
            acc[g][nb] = fm_mfma<(tap % G::TPV) * MG + g>(aw[tap / G::TPV], bx[tx], acc[g][nb]);
          });
        });
      }
    });
  });
}

template <int KH, int KW, int PY, int PX, int MG>
__global__ __launch_bounds__(256) void firstm_fwd_kernel(FirstM p) {
  using G = Geo<KH, KW, PY, PX, MG>;
  __shared__ float xt[G::RT * kTW];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float aw[G::NAV];
  load_weights<KH, KW, PY, PX, MG>(p, lane, aw);
  float bs[MG][4];                         // biases, once (a load per use stalled every channel)
#pragma unroll
  for (int g = 0; g < MG; ++g)
#pragma unroll
    for (int r = 0; r < 4; ++r) bs[g][r] = (4 * g + r < p.Cout) ? p.bias[4 * g + r] : 0.f;
  for (int tile = blockIdx.x; tile < p.nTiles; tile += gridDim.x) {
    int n, z, row0, col0;
    __syncthreads();                       // the previous tile's reads are done
    load_tile<KH, KW, PY, PX, MG>(p, tile, xt, n, z, row0, col0);
    __syncthreads();
    f32x4 acc[MG][PY];
    if (!FM_DBG(p, 2)) conv_rows<KH, KW, PY, PX, MG>(xt + wave * PY * kTW, lane, aw, acc);
    else {
#pragma unroll
      for (int g = 0; g < MG; ++g)
#pragma unroll
        for (int nb = 0; nb < PY; ++nb) acc[g][nb] = (f32x4){xt[lane], 0.f, 0.f, 0.f};
    }
    const int prow = row0 / PY + wave;
    const int pcol = (col0 + lane) / PX;
    const bool ok = prow < p.Ho && pcol < p.Wo && (lane % PX) == 0;
    float* ob = p.out + (long)n * p.osN + (long)z * p.osD + (long)prow * p.osH + pcol;
#pragma unroll
    for (int g = 0; g < MG; ++g)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float m = acc[g][0][r];
        if constexpr (PY == 2) m = fmaxf(m, acc[g][1][r]);
        if constexpr (PX == 2) m = fmaxf(m, __shfl_xor(m, 1, 64));
        const int ch = 4 * g + r;
        float v = 0.f;
        if (ch < p.Cout) {
          v = m + bs[g][r];
          if (p.act == E2_ACT_RELU) v = fmaxf(v, 0.f);
          if (ok && !(FM_DBG(p, 4) && ch > 0)) ob[(long)ch * p.osC] = v;
        }
        acc[g][0][r] = v;
      }
    if (p.nxb && ok) {
      // the lane holds every channel of its pooled position: whole 16-byte pixel pieces
      __bf16* nb = p.nxb + ((((long)n * p.D + z) * p.nxKG) * ((long)p.Ho * p.Wo) + (long)prow * p.Wo + pcol) * 8;
#pragma unroll
      for (int kg = 0; kg < (MG + 1) / 2; ++kg) {
        if (kg >= p.nxKG) continue;
        fm_bf16x8 h;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int g = 2 * kg + (e >> 2);
          h[e] = (__bf16)(g < MG ? acc[g < MG ? g : 0][0][e & 3] : 0.f);
        }
        *reinterpret_cast<fm_bf16x8*>(nb + (long)kg * p.Ho * p.Wo * 8) = h;
      }
    }
  }
}

// part[co][T + 1][slot]: taps 0..T-1 (unflipped tap index) and the bias gradient, one slot per
// work-group
template <int KH, int KW, int PY, int PX, int MG>
__global__ __launch_bounds__(256) void firstm_bwd_kernel(FirstM p, float* __restrict__ part) {
  using G = Geo<KH, KW, PY, PX, MG>;
  constexpr int T = G::T, CH = G::CH, NP = 64 * PY;       // conv positions per wave and tile
  constexpr bool Q4 = T <= 16;                            // four positions per MFMA
  extern __shared__ __attribute__((aligned(16))) float fm_lds[];
  float* dyT = fm_lds;                               // [4 waves][NP * CH]
  float* xt = fm_lds + 4 * NP * CH;                  // [RT * kTW]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float aw[G::NAV];
  load_weights<KH, KW, PY, PX, MG>(p, lane, aw);
  // lane -> tap (+ position class pq of the four-position form): offset inside the tile
  // (lanes whose tap does not exist read tap 0, never used)
  const int pq = Q4 ? (lane >> 4) : 0;
  const int tapi = Q4 ? (lane & 15) : lane;
  const int tapl = tapi < T ? tapi : 0;
  const int tapoff = (tapl / KW) * kTW + (tapl % KW) + pq;
  f32x4 dwacc[MG];
  float db[MG][4];
#pragma unroll
  for (int g = 0; g < MG; ++g) {
    dwacc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) db[g][r] = 0.f;
  }
  float* dyw = dyT + wave * NP * CH;
  float bs[MG][4];
#pragma unroll
  for (int g = 0; g < MG; ++g)
#pragma unroll
    for (int r = 0; r < 4; ++r) bs[g][r] = (4 * g + r < p.Cout) ? p.bias[4 * g + r] : 0.f;
  for (int tile = blockIdx.x; tile < p.nTiles; tile += gridDim.x) {
    int n, z, row0, col0;
    __syncthreads();
    load_tile<KH, KW, PY, PX, MG>(p, tile, xt, n, z, row0, col0);
    const int prow = row0 / PY + wave;
    const int pcol = (col0 + lane) / PX;
    const bool ok = prow < p.Ho && pcol < p.Wo;
    const float* gp = p.dout + (long)n * p.osN + (long)z * p.osD + (long)prow * p.osH + pcol;
    // the pooled gradients of all channels, requested before the recompute hides them
    float gvs[MG][4];
#pragma unroll
    for (int g = 0; g < MG; ++g)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        gvs[g][r] = (ok && 4 * g + r < p.Cout) ? gp[(long)(4 * g + r) * p.osC] : 0.f;
    __syncthreads();
    const float* xw = xt + wave * PY * kTW;
    f32x4 acc[MG][PY];
    if (!FM_DBG(p, 2)) conv_rows<KH, KW, PY, PX, MG>(xw, lane, aw, acc);
    else {
#pragma unroll
      for (int g = 0; g < MG; ++g)
#pragma unroll
        for (int nb = 0; nb < PY; ++nb) acc[g][nb] = (f32x4){xw[lane], 0.f, 0.f, 0.f};
    }
    // dy of the conv output: the pooled gradient goes to every element equal to the window
    // maximum, through the activation's slope at the pooled pre-activation
#pragma unroll
    for (int g = 0; g < MG; ++g) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float m = acc[g][0][r];
        if constexpr (PY == 2) m = fmaxf(m, acc[g][1][r]);
        if constexpr (PX == 2) m = fmaxf(m, __shfl_xor(m, 1, 64));
        float gv = gvs[g][r];
        if (p.act == E2_ACT_RELU) {
          const float pre = m + bs[g][r];
          gv *= (pre > 0.f) ? 1.f : ((pre == 0.f) ? 0.5f : 0.f);
        }
        if ((lane % PX) == 0) db[g][r] += gv;            // once per pooled position
#pragma unroll
        for (int nb = 0; nb < PY; ++nb) acc[g][nb][r] = (acc[g][nb][r] == m) ? gv : 0.f;
      }
      // transposed into LDS: [position][channel]
#pragma unroll
      for (int nb = 0; nb < PY; ++nb)
        *reinterpret_cast<f32x4*>(dyw + (nb * 64 + lane) * CH + 4 * g) = acc[g][nb];
    }
    // (same wave writes and reads dyw: LDS operations of a wave complete in order)
    // dW[4g..4g+3][tap = lane] += dy[..][position] * x[position + tap]
    const float* xb = xw + tapoff;
    if (FM_DBG(p, 4)) continue;
    if constexpr (Q4) {
      // block (pq, tap quad): A = dy[4g + i][p0 + pq] on lanes (pq, *, i), B = x[p0 + pq + tap]
      fm_for<NP / 4>([&](auto k_) {
        constexpr int p0 = 4 * decltype(k_)::value;
        const float bv = xb[(p0 / 64) * kTW + (p0 % 64)];
#pragma unroll
        for (int g = 0; g < MG; ++g) {
          const float av = dyw[(p0 + pq) * CH + 4 * g + (lane & 3)];
          dwacc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(av, bv, dwacc[g], 0, 0, 0);
        }
      });
    } else {
      fm_for<NP / 16>([&](auto pb_) {
        constexpr int pb = decltype(pb_)::value;
        float ad[MG];
#pragma unroll
        for (int g = 0; g < MG; ++g) ad[g] = dyw[(pb * 16 + (lane >> 2)) * CH + 4 * g + (lane & 3)];
        fm_for<16>([&](auto pp_) {
          constexpr int pp = decltype(pp_)::value;
          constexpr int pos = pb * 16 + pp;
          const float bv = xb[(pos / 64) * kTW + (pos % 64)];
#pragma unroll
          for (int g = 0; g < MG; ++g) dwacc[g] = fm_mfma<pp>(ad[g], bv, dwacc[g]);
        });
      });
    }
  }
  // partial sums of the four waves -> LDS -> ONE workspace slot per work-group
  __syncthreads();
  float* red = fm_lds;                                 // [4 waves][CH][T + 1]
#pragma unroll
  for (int g = 0; g < MG; ++g)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v = dwacc[g][r];
      if constexpr (Q4) {                              // add the four position classes
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
      }
      float sb = db[g][r];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) sb += __shfl_xor(sb, o, 64);
      float* rw = red + (wave * CH + 4 * g + r) * (T + 1);
      if (lane < T) rw[lane] = v;
      if (lane == 0) rw[T] = sb;
    }
  __syncthreads();
  // slot-major: part[element][slot] -- the reduction reads an element's slots as ONE contiguous
  // run per wave (first_bwd_reduce_sm_kernel) instead of 64 strided partial sums + atomics
  float* ps = part + blockIdx.x;
  const long nS = gridDim.x;
  for (int e = threadIdx.x; e < p.Cout * (T + 1); e += 256)
    ps[e * nS] = (red[e] + red[CH * (T + 1) + e]) + (red[2 * CH * (T + 1) + e] + red[3 * CH * (T + 1) + e]);
}

template <int KH, int KW, int PY, int PX, int MG>
int launch_fwd(e2_ctx* ctx, const FirstM& p, int grid) {
  hipLaunchKernelGGL((firstm_fwd_kernel<KH, KW, PY, PX, MG>), dim3(grid), dim3(256), 0, ctx->stream, p);
  E2_CHECK_HIP(hipGetLastError());
  return 0;
}
template <int KH, int KW, int PY, int PX, int MG>
int launch_bwd(e2_ctx* ctx, const FirstM& p, int grid, float* part) {
  using G = Geo<KH, KW, PY, PX, MG>;
  const size_t lds = sizeof(float) * (4 * 64 * PY * G::CH + G::RT * kTW);
  static bool attr_done = false;
  if (!attr_done) {
    E2_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&firstm_bwd_kernel<KH, KW, PY, PX, MG>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done = true;
  }
  hipLaunchKernelGGL((firstm_bwd_kernel<KH, KW, PY, PX, MG>), dim3(grid), dim3(256), lds, ctx->stream, p, part);
  E2_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace

// geometry variant v (conv_first.hip first_supported): 1 = 4x4 pool 2x2, 2 = 6x6 pool 2x2,
// 3 = 3x3 no pooling.  Instances: MG = 5 (Cout <= 20) and MG = 8 (Cout <= 32).
int e2i_firstm_mg(int cout) { return cout <= 20 ? 5 : (cout <= 32 ? 8 : 0); }

int e2i_firstm_grid(const e2_ctx* ctx, long nTiles) {
  return (int)std::min<long>(nTiles, std::min<long>((long)ctx->num_cu * (e2_dbg_env("E2_FM_GRID") ? e2_dbg_env_int("E2_FM_GRID") : 3), 1024));
}

static void fill(FirstM& p, const e2_tensor5* x, const e2_tensor5* o, int cout, int py, int px) {
  p.x = x->ptr;
  p.N = x->n; p.Cout = cout; p.D = x->d; p.H = x->h; p.W = x->w;
  p.Ho = o->h; p.Wo = o->w;
  p.xsN = x->sn; p.xsD = x->sd; p.xsH = x->sh;
  p.osN = o->sn; p.osC = o->sc; p.osD = o->sd; p.osH = o->sh;
  p.dbg = e2_dbg_env_int("E2_FM_DBG");
  p.tilesX = e2_cdiv(o->w * px, 64);
  p.tilesY = e2_cdiv(o->h, 4);
  p.nTiles = p.N * p.D * p.tilesY * p.tilesX;
}

#define E2_FM_DISPATCH(FN, ...)                                                        \
  if (v == 1 && mg == 5) return FN<4, 4, 2, 2, 5>(__VA_ARGS__);                       \
  if (v == 1 && mg == 8) return FN<4, 4, 2, 2, 8>(__VA_ARGS__);                       \
  if (v == 2 && mg == 5) return FN<6, 6, 2, 2, 5>(__VA_ARGS__);                       \
  if (v == 2 && mg == 8) return FN<6, 6, 2, 2, 8>(__VA_ARGS__);                       \
  if (v == 3 && mg == 5) return FN<3, 3, 1, 1, 5>(__VA_ARGS__);                       \
  if (v == 3 && mg == 8) return FN<3, 3, 1, 1, 8>(__VA_ARGS__);

int e2i_firstm_fwd(e2_ctx* ctx, int v, const e2_tensor5* x, const float* w, const float* bias,
                   int cout, int py, int px, int act, const e2_tensor5* out, void* next_xb, int next_kg) {
  const int mg = e2i_firstm_mg(cout);
  FirstM p{};
  fill(p, x, out, cout, py, px);
  p.w = w; p.bias = bias; p.out = out->ptr; p.act = act;
  E2_REQUIRE(!next_xb || (next_kg * 8 >= cout && ((uintptr_t)next_xb & 15) == 0),
             "conv1(mfma): the next layer's copy holds %d channel groups for %d channels", next_kg, cout);
  p.nxb = reinterpret_cast<__bf16*>(next_xb); p.nxKG = next_kg;
  const int grid = e2i_firstm_grid(ctx, p.nTiles);
  E2_FM_DISPATCH(launch_fwd, ctx, p, grid)
  e2_set_error("conv1(mfma): no instance for variant %d, %d channels", v, cout);
  return 2;
}

// workspace: one slot of cout * (T + 1) floats per work-group of the persistent grid (<= 1024)
size_t e2i_firstm_ws_floats(long nTiles, int cout, int T) {
  return (size_t)std::min<long>(nTiles, 1024) * cout * (T + 1);
}

int e2i_firstm_bwd(e2_ctx* ctx, int v, const e2_tensor5* x, const float* w, const float* bias,
                   const e2_tensor5* dout, int py, int px, int act, float* part, int* nslots) {
  const int cout = dout->c;
  const int mg = e2i_firstm_mg(cout);
  FirstM p{};
  fill(p, x, dout, cout, py, px);
  p.w = w; p.bias = bias; p.dout = dout->ptr; p.act = act;
  const int grid = e2i_firstm_grid(ctx, p.nTiles);
  *nslots = grid;
  E2_FM_DISPATCH(launch_bwd, ctx, p, grid, part)
  e2_set_error("conv1(mfma): no instance for variant %d, %d channels", v, cout);
  return 2;
}
