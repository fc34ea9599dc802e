// tail.hip -- the TAIL of the neuro3d nets in one launch, forward and backward:
//
//   x (C1 ch.) -> 1x1x1 conv to C2 channels + bias + relu      (neural.py:662-712 with a
//                 (1,1,1) kernel = the "tensordot" branch, computations.py:330-335,377-384)
//              -> 1x1x1 conv to ncls <= 4 'lin' features -> channel softmax
//                 (computations.py:175-176) -> MultinoulliNLL, sparse targets (loss.py:261-347)
//   and T.grad of that chain (model.py:182): dlogits, the head's dW / dbias, the gradient of
//   the 1x1x1 layer's pre-activation `dpre` (its weight gradient's operand; its bias
//   gradient = the row sums), and the data gradient dx = W^T dpre.
//
// Everything here is per position -- both convs have one tap -- so a work-group that owns a
// tile of positions can run the whole chain out of LDS.  As separate launches this was
// (examples/neuro3d_lite.py at 183^2, us): 1x1x1 forward 23 + head forward 12 + head backward
// 18 + activation backward 10 + 1x1x1 data gradient 23 = 86 us for 14 us of matrix work and
// 11 MB tensors written and read back four times; neuro3d's 2,205-position tail: 61 us for
// 2.3 us of matrix work.
//
//   tile    = NP = 16 * WN consecutive positions of one sample (the tensors are dense in
//             (z, y, x), so a position is one flat index), all channels; 4 waves = WM x WN:
//             wave (wm, wn) owns 16 positions and MTW = ceil(13 / WM) of the 13 blocks of
//             16 channel rows (C1, C2 <= 208).  (WM, chunk rows) by the host (tail_cfg): WM = 2
//             with 16-row weight chunks where there are >= 256 tiles of 32 positions (76 KB of
//             LDS: two work-groups per CU); few positions (neuro3d: 2,205) -> WM = 4, so that
//             138 work-groups exist instead of 35.
//   phase A : pre[co][p] = sum_ci Wf[ci][co] x[ci][p]   fp32 MFMA 16x16x4; B from the x tile
//             in LDS, A from the forward packed image ([k][m], m contiguous: the image
//             conv_igemm.hip keeps), staged in chunks of 16 or 40 k-rows by LDS-DMA, double
//             buffered; operand reads one k-step ahead (inline asm, counted lgkmcnt).
//   epilogue: h = relu(pre + b1) written over the x tile ([channel][position]); a negative
//             pre-activation keeps its sign in the zero (-0.0): relu'(0) = 0.5 (Theano) is told
//             from 0 by it, as in the GEMM epilogues of igemm_core.hpp.
//   head    : thread (position, channel share) -> partial logits -> LDS -> softmax, probs out,
//             -log(p_target + 1e-5), dlogits (needs 1 / #labelled: every work-group counts the
//             labelled voxels of the WHOLE target itself -- 55 KB from L2 -- or reads the count
//             of a pre-pass for large targets).
//   dpre    : thread = channel row: dpre = (Wh^T dlogits) * relu'(h), in place in LDS, row sums
//             (bias gradient) and dWh[c][row] = sum_p dlogits[c][p] h[row][p] in registers ->
//             ONE partial-sum slot per work-group (no same-address atomics, DESIGN.md lesson 6).
//   phase C : dx[ci][p] = sum_co Wd[co][ci] dpre[co][p]  (B = the dpre tile, A = the data
//             gradient's packed image), tile -> LDS -> rows of NP positions to memory --
//             optionally THROUGH the activation backward of the layer that produced x (gm: dx *=
//             act'(.), its bias gradient = the row sums, written into the interior of that
//             layer's zero-padded gradient buffer).
//   A small second kernel adds the slots up (into the zeroed gradient arena) and writes the
//   loss: it is launched from the BACKWARD half of the plan, behind the arena's zero fill.
#include "common.hpp"
#include <algorithm>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_vp;
typedef const __attribute__((address_space(1))) void* gbl_vp;

#define E2_EPS_NLL 1e-5f

namespace {

constexpr int kMT = 13;              // 16-row blocks: up to 208 channels on either side
constexpr int kRows = 16 * kMT;

struct TailP {
  const float* x; long xsN, xsC;
  const float* wpf; int coPf, ciPf;  // forward image [ci][coPf]
  const float* wpd; int coPd, ciPd;  // data-gradient image [co][coPd] (m = ci)
  const float* b1;                   // bias of the 1x1x1 layer [C2]
  const float* wh; const float* bh;  // head [ncls][C2], [ncls]
  const float* tg; long tsN;         // target [N][1][S]
  float* pr; long psN, psC;          // probabilities [N][ncls][S]
  float* dpre; long dsN, dsC;        // gradient of the 1x1x1 layer's pre-activation [N][C2][S]
  float* dx; long gsN, gsC, gsD, gsH; // gradient of x [N][C1][D][H][W] (any row / plane pitch), or nullptr
  int H, W;                          // plane extents (position s -> z, y, x for dx's strides)
  // dx is written THROUGH the activation backward of the layer that produced x (gm != 0): out =
  // dx * act'(.), and the row sums (that layer's bias gradient) go to the slot.  gm = 1: slope
  // from gm_src = that layer's activated output (a zero's sign tells 0.5 from 0), 2: from gm_src
  // = its pre-activation + gm_bias[row], 3: linear activation (slope 1)
  int gm;
  const float* gm_src; long msN, msC;
  const float* gm_bias;
  float* part;                       // [PSZ][work-groups] partial sums (an element's slots contiguous)
  float* stats;                      // [0] loss sum (written by the reduce kernel), [1] #labelled
  int zero_wb;                       // 1: whole chunks of K reach past a packed image's rows
  int sum_mode;                      // 1: dlogits are not divided by the labelled count (e2_set_loss_grad_mode)
  int N, C1, C2, S;
  int tilesPerN;
  long nTarget;                      // N * S
  int count_here;                    // 1: count the labelled voxels in this kernel
  unsigned long long* stamps;        // debug build (E2_TAIL_STAMPS): 12 s_memtime stamps per work-group
  int dbg;                           // debug build (E2_TAIL_DBG), timing only: 1 = no weight DMA after
                                     // the first chunk, 2 = no MFMAs, 4 = no operand reads
};
#ifdef E2_DEBUG_ENV
#define TAIL_DBG(bit) ((p.dbg & (bit)) != 0)
#else
#define TAIL_DBG(bit) false
#endif

#ifdef E2_DEBUG_ENV
#define TAIL_STAMP(i) do { if (p.stamps && threadIdx.x == 0) p.stamps[12L * blockIdx.x + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define TAIL_STAMP(i) do {} while (0)
#endif

// KC = k-rows per staged weight chunk: 40 (five chunks per 200 channels, one work-group per CU)
// or 16 (a work-group then fits 80 KB of LDS: TWO per CU, one computes while the other loads,
// reduces or stores)
template <int WM, int KC>
struct Geo {
  static constexpr int WN = 4 / WM, NP = 16 * WN;
  // row stride of the tile: odd (thread-per-row passes) and 17 mod 32 (the four k-rows of an
  // MFMA operand read sit 17 banks apart: one 2-way conflict per read)
  static constexpr int NPP = NP == 16 ? 17 : NP + 17;
  static constexpr int MTW = (kMT + WM - 1) / WM;
  static constexpr int BM = 16 * MTW * WM;
  static constexpr int BMS = (BM % 32 == 16) ? BM : BM + 16;   // k-rows 16 banks apart
  static constexpr int NQ = 256 / NP;                          // channel shares of a position
  // tile rows: the K loops run over whole chunks, the rows past the channels hold zeros
  static constexpr int TR = ((kRows + KC - 1) / KC) * KC;
  static constexpr int TILE_F = ((TR * NPP + 3) / 4) * 4;
  static constexpr int WB_F = KC * BMS;
};
template <int WM, int NC, int KC>
constexpr size_t tail_lds_bytes() {
  using G = Geo<WM, KC>;
  return sizeof(float) * (size_t)(G::TILE_F + 2 * G::WB_F + G::NQ * NC * G::NP + NC * G::NP + 16 +
                                  (NC + 1) * kRows);
}

// slot elements: dWh [nc][c2], dbh [nc], db1 [c2], loss sum, then (gm) the parent's bias gradient [c1]
__host__ __device__ inline int tail_psz(int nc, int c2, int c1gm = 0) { return nc * c2 + nc + c2 + 1 + c1gm; }

// BF: bf16 mode (e2_set_mfma_dtype): the operands of the layer's two GEMMs -- x and the forward
// image, dpre and the data-gradient image -- are rounded to bf16 (nearest even) in registers on
// their way into the matrix core; products and sums stay f32 (bf16 x bf16 is exact in f32, so
// this IS the bf16 MFMA's arithmetic up to summation order); tensors, the head and every
// pointwise step stay f32.  Two VALU operations per operand register, next to 32-cycle MFMAs.
__device__ __forceinline__ float tail_rnd_bf16(float v) {
  union { __bf16 h[2]; unsigned u; } r;
  r.h[0] = (__bf16)v; r.h[1] = (__bf16)0.f;           // (a plain conversion: the compiler sees the VALU write)
  return __uint_as_float(r.u << 16);
}

template <int WM, int NC, int KC, bool BF>
__global__ __launch_bounds__(256) void tail_kernel(TailP p) {
  using G = Geo<WM, KC>;
  constexpr int kKC = KC;
  constexpr int WN = G::WN, NP = G::NP, NPP = G::NPP, MTW = G::MTW, BM = G::BM, BMS = G::BMS,
                NQ = G::NQ;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* T = lds;                                   // [TR][NPP]: x, then h, then dpre, then dx
  float* WB = lds + G::TILE_F;                      // 2 x [kKC][BMS] weight chunks
  float* PL = WB + 2 * G::WB_F;                     // [NQ][NC][NP] partial logits
  float* DL = PL + NQ * NC * NP;                    // [NC][NP] dlogits
  float* RED = DL + NC * NP;                        // scalars (16)
  float* WH = RED + 16;                             // [NC][kRows] head weights, then b1 [kRows]
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % WM, wn = wave / WM;
  const int n = blockIdx.x / p.tilesPerN;
  const int s0 = (blockIdx.x - n * p.tilesPerN) * NP;
  const int np = min(NP, p.S - s0);
  const int pp = tid % NP, pq = tid / NP;           // (position, channel share) of the passes

  // ---- weight chunks: image rows [c * kKC, + kKC) x columns [0, BM) -> LDS by LDS-DMA --------
  constexpr int PIECES = kKC * BMS / 4;             // 16-byte pieces of a chunk, pad columns too
  constexpr int NI = (PIECES + 255) / 256;
  // piece pi = it * 256 + tid of a chunk [k][BMS]: image row k, columns 4 (pi % (BMS / 4)) ...
  // (pad columns re-read the row's last piece); offsets for both images, once
  int wrow[NI], woffF[NI], woffD[NI], wcol[NI];
#pragma unroll
  for (int it = 0; it < NI; ++it) {
    const int pi = min(it * 256 + tid, PIECES - 1);
    wrow[it] = (pi * 4) / BMS;
    wcol[it] = min(pi * 4 - wrow[it] * BMS, BM - 4);
    woffF[it] = wrow[it] * p.coPf + wcol[it];
    woffD[it] = wrow[it] * p.coPd + wcol[it];
  }
  // the K loop has ONE trip count -- whole chunks, no early exit around the hand-scheduled
  // steps (an exit per step made hipcc keep two sets of accumulators and move all 52
  // registers at every chunk boundary).  Chunk rows past the image (only where K padded to
  // whole chunks exceeds ciP) are never staged: both buffers are zeroed once instead.
  auto stage = [&](const float* img, int coP, int ciP, const int (&woff)[NI], int c, int buf) {
    const float* wc = img + (long)c * kKC * coP;
    unsigned char* lb = reinterpret_cast<unsigned char*>(WB + buf * G::WB_F) + (wave * 64) * 16;
#pragma unroll
    for (int it = 0; it < NI; ++it)
      // (selecting between the image and a zero block per piece made the DMA -- and every load
      // queued behind it -- 30 % slower: rows past the image are SKIPPED, their LDS rows are
      // zeroed once, below)
      if (it * 256 + tid < PIECES && c * kKC + wrow[it] < ciP)
        __builtin_amdgcn_global_load_lds((gbl_vp)(wc + woff[it]), (lds_vp)(lb + it * 256 * 16), 16, 0, 0);
  };
  TAIL_STAMP(0);
  if (p.zero_wb) {
    for (int i = tid; i < 2 * G::WB_F; i += 256) WB[i] = 0.f;
    __syncthreads();
  }
  stage(p.wpf, p.coPf, p.ciPf, woffF, 0, 0);

  // ---- the x tile -> LDS: rows of NP positions, zero past the sample and past C1; every load
  // of the tile is in flight at once ------------------------------------------------------------
  {
    const float* xb = p.x + (long)n * p.xsN + s0 + pp;
    const bool pv = pp < np;
    constexpr int NR = G::TR / NQ;
    static_assert(G::TR % NQ == 0, "tile rows per thread");
    float xv[NR];
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      const int r = pq + j * NQ;
      xv[j] = (pv && r < p.C1) ? xb[(long)r * p.xsC] : 0.f;
    }
    // (the count's loads are requested BEHIND the tile's: one round trip to memory for both)
    // ---- the labelled voxels of the whole target (every work-group for itself) -----------------
    float cnt = 0.f;
    if (p.count_here) {
      // (the first cut read one float per iteration: 54 dependent round trips to L2 = ~35 us of
      // the kernel on neuro3d_lite's 13,690 targets)
      for (int n2 = 0; n2 < p.N; ++n2) {
        const float* tp = p.tg + (long)n2 * p.tsN;
        const int head = min(p.S, (int)((4 - (((uintptr_t)tp >> 2) & 3)) & 3));   // floats up to 16-B alignment
        const int nv = (p.S - head) >> 2;
        const f32x4* tv4 = reinterpret_cast<const f32x4*>(tp + head);
        for (int i0 = 0; i0 < nv; i0 += 256 * 8) {
          f32x4 v[8];
  #pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * 256 + tid;
            v[u] = i < nv ? tv4[i] : f32x4{-1.f, -1.f, -1.f, -1.f};
          }
  #pragma unroll
          for (int u = 0; u < 8; ++u)
  #pragma unroll
            for (int e = 0; e < 4; ++e)
  #pragma unroll
              for (int c = 0; c < NC; ++c) cnt += (v[u][e] == (float)c) ? 1.f : 0.f;
        }
        // the unaligned head and the tail of < 4 floats
        const int rest = p.S - head - 4 * nv;
        if (tid < head + rest) {
          const float tv = tid < head ? tp[tid] : tp[head + 4 * nv + (tid - head)];
  #pragma unroll
          for (int c = 0; c < NC; ++c) cnt += (tv == (float)c) ? 1.f : 0.f;
        }
      }
  #pragma unroll
      for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
      if (lane == 0) RED[wave] = cnt;
    }
    TAIL_STAMP(1);
    if (tid < kRows) {                          // head weights and the layer's bias, once
#pragma unroll
      for (int c = 0; c < NC; ++c) WH[c * kRows + tid] = tid < p.C2 ? p.wh[c * p.C2 + tid] : 0.f;
      WH[NC * kRows + tid] = tid < p.C2 ? p.b1[tid] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < NR; ++j) T[(pq + j * NQ) * NPP + pp] = xv[j];
  }

  // ---- one GEMM phase: acc[i] (+)= sum_k img[k][m] * T[k][position] --------------------------
  f32x4 acc[MTW];
  auto lds_a = [](const float* q) { return (unsigned)(uintptr_t)(lds_vp)q; };
  // step s of a chunk: A[i] = chunk[4 s + kq][16 (wm MTW + i) + l15], B = tile[k0 + 4 s + kq][position]
  auto rd = [&](unsigned wb, unsigned tb, int s, float (&A)[MTW], float& B) {
    if (TAIL_DBG(4)) return;
#pragma unroll
    for (int i = 0; i < MTW; ++i)
      asm volatile("ds_read_b32 %0, %1" : "=v"(A[i]) : "v"(wb + (unsigned)(s * 4 * BMS * 4 + i * 64)));
    asm volatile("ds_read_b32 %0, %1" : "=v"(B) : "v"(tb + (unsigned)(s * 4 * NPP * 4)));
  };
  auto fma = [&](float (&A)[MTW], float& B) {
#pragma unroll
    for (int i = 0; i < MTW; ++i) asm volatile("" : "+v"(A[i]));
    asm volatile("" : "+v"(B));
    if (TAIL_DBG(2)) return;
    if constexpr (BF) {
      const float Bb = tail_rnd_bf16(B);
#pragma unroll
      for (int i = 0; i < MTW; ++i)
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(tail_rnd_bf16(A[i]), Bb, acc[i], 0, 0, 0);
    } else {
#pragma unroll
    for (int i = 0; i < MTW; ++i)
      acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[i], B, acc[i], 0, 0, 0);
    }
  };
  // chunk c of a phase sits in buffer (buf0 + c) & 1; its first chunk was staged by the caller
  auto gemm = [&](const float* img, int coP, int ciP, const int (&woff)[NI], int K, int buf0,
                  bool next_d) {
#pragma unroll
    for (int i = 0; i < MTW; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nch = (K + kKC - 1) / kKC;
    for (int c = 0; c < nch; ++c) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();                         // chunk c has landed; the other buffer is free
      if (!TAIL_DBG(1)) {
      if (c + 1 < nch) stage(img, coP, ciP, woff, c + 1, (buf0 + c + 1) & 1);
      else if (next_d) stage(p.wpd, p.coPd, p.ciPd, woffD, 0, (buf0 + c + 1) & 1);   // phase C's first chunk
      }
      const unsigned wb = lds_a(WB + ((buf0 + c) & 1) * G::WB_F + (wm * MTW) * 16 + kq * BMS + l15);
      const unsigned tb = lds_a(T + (c * kKC + kq) * NPP + wn * 16 + l15);
      // operands of step s + 1 are requested before the MFMAs of step s are issued (inline-asm
      // reads, counted lgkmcnt: hipcc's own schedule waited for every pair of reads -- two
      // MFMAs per LDS round trip)
      float A0[MTW], A1[MTW], B0, B1;
      rd(wb, tb, 0, A0, B0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < kKC / 4; s += 2) {
        rd(wb, tb, s + 1, A1, B1);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(MTW + 1 < 15 ? MTW + 1 : 15) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        fma(A0, B0);
        __builtin_amdgcn_sched_barrier(0);
        if (s + 2 < kKC / 4) rd(wb, tb, s + 2, A0, B0);
        __builtin_amdgcn_sched_barrier(0);
        if (s + 2 < kKC / 4) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(MTW + 1 < 15 ? MTW + 1 : 15) : "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        fma(A1, B1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    return nch;
  };

  TAIL_STAMP(2);
  // ======== phase A: pre = Wf^T x ================================================================
  const int nchA = gemm(p.wpf, p.coPf, p.ciPf, woffF, p.C1, 0, p.dx != nullptr);
  TAIL_STAMP(3);
  __syncthreads();                             // every wave is done with the x tile
  // h = relu(pre + b1) over the tile; rows past C2 are the zero k-rows of phase C
#pragma unroll
  for (int i = 0; i < MTW; ++i) {
    const int r0 = (wm * MTW + i) * 16 + 4 * kq;
    if (r0 < kRows) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(WH + NC * kRows + r0);   // (0 past C2)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = r0 + r;
        const float t = acc[i][r] + bv[r];       // rows past C2: zero weights, zero bias -> +0.0
        T[row * NPP + wn * 16 + l15] = (t > 0.f) ? t : ((t == 0.f) ? 0.f : -0.f);
      }
    }
  }
  __syncthreads();

  TAIL_STAMP(4);
  // ======== head: logits, softmax, loss, dlogits ================================================
  {
    const int per = (p.C2 + NQ - 1) / NQ;
    const int c0 = pq * per, c1 = min(c0 + per, p.C2);
    float lg[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) lg[c] = 0.f;
#pragma unroll 10
    for (int co = c0; co < c1; ++co) {
      const float hv = fmaxf(T[co * NPP + pp], 0.f);
#pragma unroll
      for (int c = 0; c < NC; ++c) lg[c] = fmaf(WH[c * kRows + co], hv, lg[c]);
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) PL[(pq * NC + c) * NP + pp] = lg[c];
  }
  __syncthreads();
  TAIL_STAMP(5);
  float inv;
  {
    float tot = p.count_here ? ((RED[0] + RED[1]) + (RED[2] + RED[3])) : p.stats[1];
    inv = 1.f / (tot + E2_EPS_NLL);
    if (blockIdx.x == 0 && tid == 0 && p.count_here) p.stats[1] = tot;
  }
  if (wave == 0) {                             // (NP <= 64 positions: lanes 0 .. NP-1)
    float lsum = 0.f;
    float d[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) d[c] = 0.f;
    if (tid < np) {
      float lg[NC], m = -INFINITY;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        float s = 0.f;
        for (int q = 0; q < NQ; ++q) s += PL[(q * NC + c) * NP + tid];
        lg[c] = s + p.bh[c];
        m = fmaxf(m, lg[c]);
      }
      float den = 0.f;
#pragma unroll
      for (int c = 0; c < NC; ++c) den += expf(lg[c] - m);
      const float tv = p.tg[(long)n * p.tsN + s0 + tid];
      float* prp = p.pr + (long)n * p.psN + s0 + tid;
      float pc[NC], pt = 0.f;
      int tc = -1;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        pc[c] = expf(lg[c] - m) / den;
        prp[(long)c * p.psC] = pc[c];
        if (tv == (float)c) { tc = c; pt = pc[c]; lsum -= logf(pc[c] + E2_EPS_NLL); }
      }
      // dL/dp_t = -inv / (p_t + eps);  dlogit_c = p_c (dp_c - sum_k dp_k p_k)   (head.hip)
      const float gpt = (tc >= 0) ? (-(p.sum_mode ? 1.f : inv) / (pt + E2_EPS_NLL)) * pt : 0.f;
#pragma unroll
      for (int c = 0; c < NC; ++c) d[c] = gpt * ((c == tc ? 1.f : 0.f) - pc[c]);
    }
    if (tid < NP) {
#pragma unroll
      for (int c = 0; c < NC; ++c) DL[c * NP + tid] = d[c];
    }
    // this work-group's loss sum and head-bias gradient (lanes >= np hold zeros)
    float* mine = p.part + blockIdx.x;             // element e of this slot: mine[e * gridDim.x]
    const long nS = gridDim.x;
    float v = lsum;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if (lane == 0) mine[(NC * p.C2 + NC + p.C2) * nS] = v;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      float sb = d[c];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) sb += __shfl_xor(sb, o, 64);
      if (lane == 0) mine[(NC * p.C2 + c) * nS] = sb;
    }
  }
  __syncthreads();

  TAIL_STAMP(6);
  // ======== dpre = (Wh^T dlogits) * relu'(h), thread = channel row ============================
  if (tid < p.C2) {
    const int row = tid;
    float w[NC], aw[NC], db = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) { w[c] = WH[c * kRows + row]; aw[c] = 0.f; }
    float* tr = T + row * NPP;
#pragma unroll 8
    for (int q = 0; q < NP; ++q) {
      const float hv = tr[q];
      float g = 0.f;
#pragma unroll
      for (int c = 0; c < NC; ++c) g = fmaf(w[c], DL[c * NP + q], g);     // (0 past the sample)
      const float slope = (hv > 0.f) ? 1.f : (__builtin_signbit(hv) ? 0.f : 0.5f);
      const float d = g * slope;
      tr[q] = d;
      db += d;
      const float hp = fmaxf(hv, 0.f);
#pragma unroll
      for (int c = 0; c < NC; ++c) aw[c] = fmaf(DL[c * NP + q], hp, aw[c]);
    }
    float* mine = p.part + blockIdx.x;
    const long nS = gridDim.x;
#pragma unroll
    for (int c = 0; c < NC; ++c) mine[(c * p.C2 + row) * nS] = aw[c];
    mine[(NC * p.C2 + NC + row) * nS] = db;
  }
  __syncthreads();
  TAIL_STAMP(7);
  // the dpre tile to memory (the 1x1x1 layer's weight gradient reads it), rows of NP positions
  if (pp < np) {
    float* db_ = p.dpre + (long)n * p.dsN + s0 + pp;
#pragma unroll 10
    for (int r = pq; r < p.C2; r += NQ) db_[(long)r * p.dsC] = T[r * NPP + pp];
  }
  if (!p.dx) return;                           // (uniform: nothing upstream needs a gradient)

  TAIL_STAMP(8);
  // ======== phase C: dx = Wd^T dpre =============================================================
  gemm(p.wpd, p.coPd, p.ciPd, woffD, p.C2, nchA & 1, false);
  TAIL_STAMP(9);
  __syncthreads();                             // every wave is done with the dpre tile
#pragma unroll
  for (int i = 0; i < MTW; ++i) {
    const int r0 = (wm * MTW + i) * 16 + 4 * kq;
    if (r0 < kRows) {
#pragma unroll
      for (int r = 0; r < 4; ++r) T[(r0 + r) * NPP + wn * 16 + l15] = acc[i][r];
    }
  }
  __syncthreads();
  {
    const int sp = s0 + min(pp, np - 1);
    const int z = sp / (p.H * p.W), rem = sp - z * (p.H * p.W);
    const int y = rem / p.W, xx = rem - y * p.W;
    float* gb = p.dx + (long)n * p.gsN + (long)z * p.gsD + (long)y * p.gsH + xx;
    if (p.gm == 0) {
      if (pp < np) {
#pragma unroll 10
        for (int r = pq; r < p.C1; r += NQ) gb[(long)r * p.gsC] = T[r * NPP + pp];
      }
    } else {
      // through the producing layer's activation backward; its bias gradient = the row sums,
      // reduced over the NP lanes of a row (each lane holds ONE position of the row)
      const float* sb = p.gm_src ? p.gm_src + (long)n * p.msN + s0 + min(pp, np - 1) : nullptr;
      float* red = WB;                           // [C1] row sums (the weight buffers are free)
      constexpr int NR = kRows / NQ;
      // every source value of the thread requested before the first use (one load per loop
      // trip made this pass 13-26 dependent round trips: +12 us on neuro3d_lite)
      float sl[NR];
#pragma unroll
      for (int j = 0; j < NR; ++j) {
        const int r = pq + j * NQ;
        sl[j] = 1.f;
        if (sb && r < p.C1 && pp < np) sl[j] = sb[(long)r * p.msC];
      }
      if (p.gm == 2) {
#pragma unroll
        for (int j = 0; j < NR; ++j) {
          const int r = pq + j * NQ;
          const float o = sl[j] + p.gm_bias[min(r, p.C1 - 1)];
          sl[j] = (o > 0.f) ? 1.f : ((o == 0.f) ? 0.5f : 0.f);
        }
      } else if (p.gm == 1) {
#pragma unroll
        for (int j = 0; j < NR; ++j) sl[j] = (sl[j] > 0.f) ? 1.f : (__builtin_signbit(sl[j]) ? 0.f : 0.5f);
      }
#pragma unroll
      for (int j = 0; j < NR; ++j) {
        const int r = pq + j * NQ;               // (uniform per NP lanes)
        float d = 0.f;
        if (r < p.C1 && pp < np) {
          d = T[r * NPP + pp] * sl[j];
          gb[(long)r * p.gsC] = d;
        }
#pragma unroll
        for (int o = NP / 2; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
        if (pp == 0 && r < p.C1) red[r] = d;
      }
      __syncthreads();
      if (tid < p.C1)
        p.part[((long)(NC * p.C2 + NC + p.C2 + 1) + tid) * gridDim.x + blockIdx.x] = red[tid];
    }
  }
  TAIL_STAMP(10);
}

// the labelled voxels of a LARGE target, once (the tail kernel counts small ones itself)
template <int NC>
__global__ __launch_bounds__(256) void tail_count_kernel(const float* tg, long tsN, int S, long nTarget,
                                                         float* stats) {
  float cnt = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nTarget; i += (long)gridDim.x * 256) {
    const float tv = tg[(i / S) * tsN + (i % S)];
#pragma unroll
    for (int c = 0; c < NC; ++c) cnt += (tv == (float)c) ? 1.f : 0.f;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
  if ((threadIdx.x & 63) == 0 && cnt != 0.f) unsafeAtomicAdd(stats + 1, cnt);
}

// slot sums -> dwh / dbh / db1 (ADDED: the gradient arena was zeroed), stats[0] and the loss.
// One wave per element: its slots are contiguous, the sum has ONE writer (no atomics, a fixed
// summation order).  (The first cut walked the slots with one thread per element: 214
// dependent-latency loads for the loss alone, 46 us.)
__global__ __launch_bounds__(256) void tail_reduce_kernel(const float* __restrict__ part, int nWG,
                                                          int nc, int c2, float* dwh, float* dbh,
                                                          float* db1, float* stats, float* loss_out,
                                                          float* count_out, int c1gm, float* dbp) {
  const int psz = tail_psz(nc, c2, c1gm);
  const int idx = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (idx >= psz) return;
  const float* row = part + (long)idx * nWG;
  float s = 0.f;
  for (int b = lane; b < nWG; b += 64) s += row[b];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane != 0) return;
  if (idx == nc * c2 + nc + c2) {
    stats[0] = s;
    if (loss_out) loss_out[0] = s / (stats[1] + E2_EPS_NLL);
    if (count_out) count_out[0] = stats[1];
    return;
  }
  float* dst = idx < nc * c2 ? dwh + idx
             : (idx < nc * c2 + nc ? dbh + (idx - nc * c2)
             : (idx < nc * c2 + nc + c2 ? db1 + (idx - nc * c2 - nc) : dbp + (idx - nc * c2 - nc - c2 - 1)));
  *dst += s;
}

// dense in (z, y, x): a position is one flat index
bool flat_sp(const e2_tensor5* t) {
  return t->sh == t->w && t->sd == (int64_t)t->h * t->w;
}
bool same_sp(const e2_tensor5* a, const e2_tensor5* b) {
  return a->n == b->n && a->d == b->d && a->h == b->h && a->w == b->w;
}

// (WM, KC): enough 32-position tiles for two work-groups on every CU -> rows split 2 ways,
// 16-row chunks (< 80 KB of LDS); few positions (neuro3d: 2,205) -> rows split 4 ways, so that
// 138 work-groups exist instead of 35
void tail_cfg(const e2_ctx* ctx, long N, long S, int* wm, int* kc) {
  const long cus = ctx ? ctx->num_cu : 256;
  if (N * ((S + 31) / 32) >= cus) { *wm = 2; *kc = 16; }
  else if (N * ((S + 31) / 32) >= cus / 2) { *wm = 2; *kc = 40; }
  else { *wm = 4; *kc = 40; }
#ifdef E2_DEBUG_ENV
  if (const char* f = e2_dbg_env("E2_TAIL_CFG")) sscanf(f, "%d,%d", wm, kc);
#endif
}
long tail_grid(long N, long S, int wm) {
  const int np = 16 * (4 / wm);
  return N * ((S + np - 1) / np);
}

template <int WM, int NC, int KC, bool BF>
int launch_tail(e2_ctx* ctx, TailP p, long grid) {
  constexpr size_t ldsb = tail_lds_bytes<WM, NC, KC>();
  static_assert(ldsb <= 160 * 1024, "tail kernel: LDS");
  static_assert((KC / 4) % 2 == 0, "the K loop is unrolled by two steps");
  static bool attr_done = false;
  if (!attr_done) {
    E2_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&tail_kernel<WM, NC, KC, BF>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done = true;
  }
#ifdef E2_DEBUG_ENV
  // in-kernel timeline (debug build): mean cycles between the stamps over the work-groups
  const bool stamps = e2_dbg_env("E2_TAIL_STAMPS") != nullptr && !ctx->capturing;
  if (stamps) {
    E2_CHECK_HIP(hipMalloc(&p.stamps, sizeof(unsigned long long) * 12 * grid));
    E2_CHECK_HIP(hipMemsetAsync(p.stamps, 0, sizeof(unsigned long long) * 12 * grid, ctx->stream));
  }
#endif
  hipLaunchKernelGGL((tail_kernel<WM, NC, KC, BF>), dim3((unsigned)grid), dim3(256), ldsb, ctx->stream, p);
  E2_CHECK_HIP(hipGetLastError());
#ifdef E2_DEBUG_ENV
  if (stamps) {
    E2_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    std::vector<unsigned long long> h(12 * grid);
    E2_CHECK_HIP(hipMemcpy(h.data(), p.stamps, sizeof(unsigned long long) * 12 * grid, hipMemcpyDeviceToHost));
    static const char* names[10] = {"x loads + count", "x tile to LDS", "phase A", "(barrier)", "h epilogue", "logits", "softmax/loss",
                                    "dpre", "dpre store", "phase C"};
    double sum[11] = {0};
    unsigned long long t0 = ~0ull, t1 = 0;
    for (long b = 0; b < grid; ++b) {
      for (int i = 0; i < 10; ++i) sum[i] += (double)(h[12 * b + i + 1] - h[12 * b + i]);
      t0 = std::min(t0, h[12 * b]); t1 = std::max(t1, h[12 * b + 10]);
    }
    fprintf(stderr, "[e2] tail<%d,%d,%d> grid %ld, %zu B of LDS: first start -> last end %llu shader cycles\n", WM, NC, KC, grid, ldsb, t1 - t0);
    for (int i = 0; i < 10; ++i) fprintf(stderr, "   %-14s %9.0f cycles\n", i == 9 ? "phase C + dx" : names[i], sum[i] / grid);
    (void)hipFree(p.stamps);
  }
#endif
  return 0;
}

}  // namespace

extern "C" int e2_tail_supported(int c1, int c2, int ncls) {
  return ncls >= 2 && ncls <= 4 && c1 >= 1 && c1 <= kRows && c2 >= 1 && c2 <= kRows;
}

extern "C" size_t e2_tail_workspace_bytes(int n, int c1, int c2, int ncls, int d, int h, int w) {
  const long S = (long)d * h * w;
  return sizeof(float) * (size_t)tail_grid(n, S, 4) * (size_t)tail_psz(ncls, c2, c1);   // (WM = 4: most slots)
}

/* forward AND backward of [1x1x1 conv c1 -> c2, bias, relu] -> [classifier head] in one launch
 * (replaces e2_conv3d_fwd_packed_act + e2_head_fwd + e2_head_bwd + e2_bias_act_bwd_out +
 * e2_conv3d_dgrad_packed of those two layers).  wp_fwd / wp_dgrad: the 1x1x1 layer's packed
 * images (e2_conv3d_pack modes 0 / 1).  Writes probs, dpre (gradient of the layer's
 * pre-activation, dense [n][c2][positions]), dx (optional, any row / plane pitch), stats[1] =
 * #labelled, and one slot of partial sums per work-group to ws; e2_tail_reduce adds those up.
 * gm_mode != 0: dx goes THROUGH the activation backward of the layer that produced x -- dx *=
 * act'(.), that layer's bias gradient (row sums) joins the slots: 1 = relu slope from gm_src = its
 * activated output (signed zeros, e2_conv3d_fwd_packed_act), 2 = from gm_src = its pre-activation
 * + gm_bias, 3 = linear.  In bf16 mode (e2_set_mfma_dtype) the operands of the layer's two GEMMs
 * are rounded to bf16 on their way into the matrix core like those of every other conv GEMM; the
 * head, the loss and all tensors stay f32. */
extern "C" int e2_tail_fwd_bwd(e2_ctx* ctx, const e2_tensor5* x, const float* wp_fwd,
                               const float* wp_dgrad, const float* bias1, int c2,
                               const float* w_head, const float* b_head, int ncls,
                               const e2_tensor5* target, const e2_tensor5* probs,
                               const e2_tensor5* dpre, const e2_tensor5* dx, int gm_mode,
                               const e2_tensor5* gm_src, const float* gm_bias, float* stats,
                               void* ws, size_t ws_bytes, int* n_slots) {
  E2_REQUIRE(ctx && x && wp_fwd && bias1 && w_head && b_head && target && probs && dpre &&
                 stats && ws && n_slots, "tail: null argument");
  E2_REQUIRE(!dx || wp_dgrad, "tail: the data gradient needs its packed image");
  E2_REQUIRE(e2_tail_supported(x->c, c2, ncls), "tail: unsupported c1=%d c2=%d ncls=%d", x->c, c2, ncls);
  E2_REQUIRE(flat_sp(x) && flat_sp(target) && flat_sp(probs) && flat_sp(dpre),
             "tail: tensors need dense (z, y, x) planes");
  E2_REQUIRE(gm_mode >= 0 && gm_mode <= 3 && (gm_mode == 0 || dx), "tail: bad gm_mode %d", gm_mode);
  E2_REQUIRE((gm_mode != 1 && gm_mode != 2) ||
                 (gm_src && flat_sp(gm_src) && same_sp(x, gm_src) && gm_src->c == x->c),
             "tail: gm_src must be a dense tensor of x's shape");
  E2_REQUIRE(gm_mode != 2 || gm_bias, "tail: gm_mode 2 needs the producing layer's bias");
  E2_REQUIRE(same_sp(x, target) && same_sp(x, probs) && same_sp(x, dpre) && (!dx || same_sp(x, dx)) &&
                 target->c == 1 && probs->c == ncls && dpre->c == c2 && (!dx || dx->c == x->c),
             "tail: shape mismatch");
  const long S = (long)x->d * x->h * x->w;
  E2_REQUIRE(S < (1L << 30) && x->n < 65536, "tail: sample too large");
  TailP p{};
  p.x = x->ptr; p.xsN = x->sn; p.xsC = x->sc;
  p.wpf = wp_fwd; p.wpd = wp_dgrad;
  e2i_pack_dims(c2, x->c, &p.ciPf, &p.coPf);
  e2i_pack_dims(x->c, c2, &p.ciPd, &p.coPd);
  p.b1 = bias1; p.wh = w_head; p.bh = b_head;
  p.tg = target->ptr; p.tsN = target->sn;
  p.pr = probs->ptr; p.psN = probs->sn; p.psC = probs->sc;
  p.dpre = dpre->ptr; p.dsN = dpre->sn; p.dsC = dpre->sc;
  p.dx = dx ? dx->ptr : nullptr;
  if (dx) { p.gsN = dx->sn; p.gsC = dx->sc; p.gsD = dx->sd; p.gsH = dx->sh; }
  p.H = x->h; p.W = x->w;
  p.gm = gm_mode;
  p.gm_src = (gm_mode == 1 || gm_mode == 2) ? gm_src->ptr : nullptr;
  if (p.gm_src) { p.msN = gm_src->sn; p.msC = gm_src->sc; }
  p.gm_bias = gm_bias;
  p.part = (float*)ws; p.stats = stats; p.sum_mode = ctx->loss_sum_mode;
  p.N = x->n; p.C1 = x->c; p.C2 = c2; p.S = (int)S;
  p.nTarget = (long)x->n * S;
  int wm = 1, kc = 40;
  tail_cfg(ctx, x->n, S, &wm, &kc);
  const long grid = tail_grid(x->n, S, wm);
  p.tilesPerN = (int)(grid / x->n);
  E2_REQUIRE(grid < (1L << 31), "tail: grid too large");
  E2_REQUIRE(ws_bytes >= sizeof(float) * (size_t)grid * tail_psz(ncls, c2, gm_mode ? x->c : 0),
             "tail: workspace too small");
  *n_slots = (int)grid;
  p.zero_wb = (e2_cdiv(p.C1, kc) * kc > p.ciPf || e2_cdiv(p.C2, kc) * kc > p.ciPd) ? 1 : 0;
  p.count_here = p.nTarget <= (1L << 16) ? 1 : 0;
  p.dbg = e2_dbg_env_int("E2_TAIL_DBG");
  if (!p.count_here) {
    if (int rc = e2i_fill_flat(ctx, stats + 1, 1, 0.f)) return rc;
    const int cg = (int)std::min<long>((p.nTarget + 255) / 256, 1024);
#define E2_TC(NC_) hipLaunchKernelGGL((tail_count_kernel<NC_>), dim3(cg), dim3(256), 0, ctx->stream, \
                                      p.tg, p.tsN, p.S, p.nTarget, stats)
    if (ncls == 2) E2_TC(2); else if (ncls == 3) E2_TC(3); else E2_TC(4);
#undef E2_TC
    E2_CHECK_HIP(hipGetLastError());
  }
#define E2_TL(WM_, KC_)                                              \
  if (wm == WM_ && kc == KC_) {                                      \
    if (ctx->mfma_bf16) {                                            \
      if (ncls == 2) return launch_tail<WM_, 2, KC_, true>(ctx, p, grid);  \
      if (ncls == 3) return launch_tail<WM_, 3, KC_, true>(ctx, p, grid);  \
      return launch_tail<WM_, 4, KC_, true>(ctx, p, grid);           \
    }                                                                \
    if (ncls == 2) return launch_tail<WM_, 2, KC_, false>(ctx, p, grid);   \
    if (ncls == 3) return launch_tail<WM_, 3, KC_, false>(ctx, p, grid);   \
    return launch_tail<WM_, 4, KC_, false>(ctx, p, grid);            \
  }
  E2_TL(1, 40) E2_TL(2, 40) E2_TL(4, 40) E2_TL(1, 16) E2_TL(2, 16) E2_TL(4, 16)
#undef E2_TL
  e2_set_error("tail: no instance WM=%d KC=%d", wm, kc);
  return 2;
}

/* slot sums of e2_tail_fwd_bwd: dw_head[ncls * c2], db_head[ncls], db1[c2] are ADDED to (zero
 * them first), stats[0] = loss sum, loss_out (optional) = stats[0] / (stats[1] + 1e-5). */
extern "C" int e2_tail_reduce(e2_ctx* ctx, const void* ws, int n_slots, int c2, int ncls,
                              float* dw_head, float* db_head, float* db1, float* stats,
                              float* loss_out, int c1_gm, float* db_parent) {
  E2_REQUIRE(ctx && ws && dw_head && db_head && db1 && stats && n_slots > 0, "tail_reduce: bad argument");
  E2_REQUIRE(c1_gm >= 0 && (c1_gm == 0 || db_parent), "tail_reduce: the parent's bias gradient is missing");
  const int psz = tail_psz(ncls, c2, c1_gm);
  hipLaunchKernelGGL(tail_reduce_kernel, dim3(e2_cdiv(psz, 4)), dim3(256), 0,
                     ctx->stream, (const float*)ws, n_slots, ncls, c2, dw_head, db_head, db1, stats,
                     loss_out, ctx->loss_count_out, c1_gm, db_parent);
  E2_CHECK_HIP(hipGetLastError());
  return 0;
}
