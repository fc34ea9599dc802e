// conv_igemm.hip -- 3-D "valid" correlation as an implicit GEMM on the gfx950
// fp32 matrix cores (v_mfma_f32_16x16x4_f32: exact f32 fma chain).  Host side +
// weight packing; the kernels live in igemm_core.hpp / conv_igemm_k*.hip.
//
//   out[n][oc][z][y][x] (+)= sum_{dz,ty,tx,ic} Wp[dz][t][ic][oc] *
//                             in[n][ic][z+dz][y+ty][x+tx]
//
// One kernel serves
//   * Conv forward  (reference: computations.py:386-428; flip folded into Wp)
//   * Conv dgrad    (Theano's ConvGradI born at model.py:182): the same
//     correlation run over the zero-padded dy with (oc,ic) swapped, no flip
//   * UpConv forward as a 1x1x1 GEMM with a depth-to-space scatter epilogue.
//
// GEMM view: M = out channels (A operand = packed weights), N = output
// positions of one z-plane (B operand = input), K = (dz, ic, ty, tx).
// Work-group = 4 compute waves (+ 4 producer waves); each compute wave owns
// MT x NT 16x16 accumulator blocks; the four sit side by side along N, so a
// work-group covers BM = 16*MT channels x BN = 64*NT consecutive plane positions
// (q = y*Wo + x, tiles never cross a z-plane).  Because the positions are
// consecutive in the plane, the input window the tile touches is, per (ic, dz),
// ONE contiguous span of the input plane:
//   [in_off(q_first), in_off(q_last) + (kh-1)*sY + kw-1].
// The producer waves bring those spans into LDS with coalesced LDS-DMA (double
// buffered per channel chunk) and every tap (ty,tx) of every channel is served
// from LDS -- each input element is fetched once per tile instead of kh*kw times.
// The weights do NOT go through LDS: lane (l15, qd) loads its A operands
// directly from the packed image (L2/L1 resident) one pipeline step ahead.
//
// MFMA operand maps (cdna_hip_programming.md 3): 16x16x4 f32, lane l:
//   A[i = l&15][k = l>>4], B[k = l>>4][j = l&15]; D: col = l&15,
//   row = 4*(l>>4) + reg.  Lane quarter qd = l>>4 picks channel ic = 4*cg+qd of
//   the staged chunk, so per k-step the tap offset is wave-uniform and the
//   per-lane part of every operand address is loop invariant.
// LDS rows are padded so that rows qd and qd+1 sit 16 banks apart
// (stride == 16 mod 32): ds_read_b32 of a 32-lane half is conflict free.
#include "igemm4_core.hpp"
#include <stdlib.h>
#include <algorithm>
#include <vector>

// ---- weight packing ---------------------------------------------------------
// Wp[dz][cg = ic/4][t = ty*kw+tx][qd = ic%4][oc (coP)], zero padded: the K order
// the kernel consumes, so staging a chunk is a linear copy of rows.
__global__ void pack_weights_kernel(const float* __restrict__ w, float* __restrict__ wp,
                                    int Cout, int Cin, int kd, int THW, long wsO,
                                    long wsI, int flip, int ciP, int coP, int Rout,
                                    int Rin) {
  const long total = (long)kd * THW * ciP * coP;
  const int T = kd * THW;
  const int nCG = ciP >> 2;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int oc = (int)(i % coP);
    long r = i / coP;
    const int qd = (int)(r & 3); r >>= 2;
    const int t = (int)(r % THW); r /= THW;
    const int cg = (int)(r % nCG);
    const int dz = (int)(r / nCG);
    const int ic = cg * 4 + qd;
    float v = 0.f;
    if (oc < Cout && ic < Cin) {
      const int tl = dz * THW + t;
      const int tap = flip ? (T - 1 - tl) : tl;
      // Rout/Rin > 1: UpConv sub-position folded into the channel index
      v = w[(long)(oc / Rout) * wsO + (long)(ic / Rin) * wsI + tap + (oc % Rout) +
            (ic % Rin)];
    }
    wp[i] = v;
  }
}

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
  int rowsW;                         // > 0: output rows to (re)write (e2_pack_job_set_rows), else the default
};
// rows of an image the repack rewrites: the real ones + the padding a GEMM tile may fetch.  Default:
// up to the widest M tile in use (7 blocks); a caller that knows the tiling of the launch that reads
// the image says how far its tiles reach (the rest stays zero from the one-time fill: correct for
// any tiling, only not refreshed in the memory-side cache)
__device__ __forceinline__ int pack_rows(const PackJobDev& j) {
  return j.rowsW > 0 ? min(j.coP, max(j.rowsW, ((j.Cout + 15) / 16) * 16))
                     : min(j.coP, ((j.Cout + 15) / 16) * 16 + 96);
}
__global__ void pack_multi_gather_kernel(const PackJobDev* __restrict__ jobs) {
  // Only the part of an image that the kernels actually fetch is rewritten: the channel
  // groups that hold data plus the four a pipeline may prefetch past the end, and the
  // output columns up to the widest M tile in use (7 blocks) -- the rest of the padding,
  // about half of an image, is zero from its one-time fill and never read.  The padding
  // that IS read is rewritten on purpose: it is fetched with every real operand, and
  // lines that only ever sit in HBM (not refreshed in L2 by this kernel) made the
  // 200-channel GEMMs 11-14 % slower.
  const PackJobDev j = jobs[blockIdx.y];
  const int T = j.kd * j.THW;
  const int nCG = j.ciP >> 2;
  const int nCGw = min(nCG, ((j.Cin + 3) >> 2) + 4);
  const int coW = pack_rows(j);
  const long total = (long)T * nCGw * 4 * coW;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int oc = (int)(i % coW);
    long r = i / coW;
    const int qd = (int)(r & 3); r >>= 2;
    const int t = (int)(r % j.THW); r /= j.THW;
    const int cg = (int)(r % nCGw);
    const int dz = (int)(r / nCGw);
    const int ic = cg * 4 + qd;
    float v = 0.f;
    if (oc < j.Cout && ic < j.Cin) {
      const int tl = dz * j.THW + t;
      v = j.w[(long)oc * j.wsO + (long)ic * j.wsI + (j.flip ? (T - 1 - tl) : tl)];
    }
    j.wp[((((long)dz * nCG + cg) * j.THW + t) * 4 + qd) * j.coP + oc] = v;
  }
}

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
__global__ __launch_bounds__(256) void pack_multi_kernel(const PackJobDev* __restrict__ jobs) {
  // (dynamic: 33 * (32 + largest tap volume) floats, e2_conv3d_pack_multi_ex -- with the full
  // 32 KB only five work-groups fit a CU)
  extern __shared__ __attribute__((aligned(16))) float tile[];
  const PackJobDev j = jobs[blockIdx.y];
  if (j.up) {
    // UpConv (one tap, Wp[cg][qd][oc']): the whole image, as pack_weights_kernel writes it
    for (long i = blockIdx.x * 256L + threadIdx.x; i < j.total; i += (long)gridDim.x * 256) {
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
  const int coW = pack_rows(j);
  const int icW = 4 * nCGw;
  const int ICT = e2_pack_ict(T);                     // input channels per tile
  const int KT = ICT * T;                             // (ic, tap) pairs per tile
  const int nOT = (coW + 31) >> 5, nIT = (icW + ICT - 1) / ICT;
  const bool oc_major = (j.wsI == T);                 // forward image: w[oc][ic][tap] contiguous in (ic, tap)
  const int tid = threadIdx.x;
  for (int tl_ = blockIdx.x; tl_ < nOT * nIT; tl_ += gridDim.x) {
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

extern "C" size_t e2_pack_job_bytes(void) { return sizeof(PackJobDev); }

/* fill one host-side job record (to be copied into a device array) */
extern "C" int e2_pack_job_fill(void* rec, const float* w, void* wp, int cout, int cin, int kd,
                                int kh, int kw, int mode) {
  E2_REQUIRE(rec && w && wp, "pack_job_fill: null argument");
  E2_REQUIRE(kd * kh * kw <= 248, "pack_job_fill: tap volume %d too large for the tiled repack "
             "(use e2_conv3d_pack)", kd * kh * kw);
  PackJobDev* j = (PackJobDev*)rec;
  const int T = kd * kh * kw;
  j->up = 0; j->Rout = j->Rin = 1; j->rowsW = 0;
  if (mode == 2 || mode == 3) {
    // UpConv weights w[cout][cin][R = kd*kh*kw sub-positions]; mode 2: the forward GEMM's image
    // (rows oc' = co*R + r, k = ci), mode 3: the data gradient's (rows ci, k = co*R + r) --
    // what e2_upconv3d_fwd / _bwd pack per call (api.hip)
    const int R = T;
    j->w = w; j->wp = (float*)wp; j->kd = 1; j->THW = 1; j->flip = 0; j->up = 1;
    if (mode == 2) {
      j->Cout = cout * R; j->Cin = cin; j->wsO = (long)cin * R; j->wsI = R; j->Rout = R;
      e2i_pack_dims(cout * R, cin, &j->ciP, &j->coP);
    } else {
      j->Cout = cin; j->Cin = cout * R; j->wsO = R; j->wsI = (long)cin * R; j->Rin = R;
      e2i_pack_dims(cin, cout * R, &j->ciP, &j->coP);
    }
    j->total = (long)j->ciP * j->coP;
    j->dT = j->dKT = j->d32T = j->dTHW = mk_pack_div(1);
    return 0;
  }
  j->w = w; j->wp = (float*)wp; j->kd = kd; j->THW = kh * kw;
  if (mode == 0) {
    j->Cout = cout; j->Cin = cin; j->wsO = (long)cin * T; j->wsI = T; j->flip = 1;
    e2i_pack_dims(cout, cin, &j->ciP, &j->coP);
  } else {
    j->Cout = cin; j->Cin = cout; j->wsO = T; j->wsI = (long)cin * T; j->flip = 0;
    e2i_pack_dims(cin, cout, &j->ciP, &j->coP);
  }
  j->total = (long)kd * kh * kw * j->ciP * j->coP;
  const int ICT = e2_pack_ict(T);
  j->dT = mk_pack_div(T); j->dKT = mk_pack_div(ICT * T); j->d32T = mk_pack_div(32 * T);
  j->dTHW = mk_pack_div(kh * kw);
  return 0;
}

/* how far the tiles of the launch that reads this image reach along its rows (16 * MT * number of
 * M tiles of the tiling in use): the repack then rewrites only those rows instead of the default
 * (real rows rounded to 16, + 96).  Purely a bandwidth hint: rows beyond stay zero. */
extern "C" int e2_pack_job_set_rows(void* rec, int rows) {
  E2_REQUIRE(rec && rows >= 0, "pack_job_set_rows: bad argument");
  ((PackJobDev*)rec)->rowsW = rows;
  return 0;
}

/* pack this image with rows of `rows` floats instead of the formula's (e2_set_image_rows: the
 * launches that read it must be told the same length); conv images (modes 0 / 1) only */
extern "C" int e2_pack_job_set_stride(void* rec, int rows) {
  E2_REQUIRE(rec, "pack_job_set_stride: null record");
  PackJobDev* j = (PackJobDev*)rec;
  E2_REQUIRE(!j->up, "pack_job_set_stride: UpConv images keep the formula");
  E2_REQUIRE(rows % 4 == 0 && rows >= ((j->Cout + 15) / 16) * 16, "pack_job_set_stride: rows of %d floats for %d channels", rows, j->Cout);
  j->coP = rows;
  j->total = (long)j->kd * j->THW * j->ciP * j->coP;
  return 0;
}

extern "C" int e2_conv3d_pack_multi(e2_ctx* ctx, const void* jobs_dev, int njobs) {
  return e2_conv3d_pack_multi_ex(ctx, jobs_dev, njobs, 248);
}

/* max_taps: the largest kd * kh * kw among the jobs (<= 248) -- sizes the launch's LDS tile
 * (tile[k * 33 + oc], k < ICT * T < 32 + T), so that more work-groups share a CU */
extern "C" int e2_conv3d_pack_multi_ex(e2_ctx* ctx, const void* jobs_dev, int njobs, int max_taps) {
  E2_REQUIRE(ctx && jobs_dev && njobs > 0 && njobs < 65536 && max_taps > 0 && max_taps <= 248,
             "pack_multi: bad argument");
  const size_t lds = sizeof(float) * std::min<size_t>(kPackTileFloats, 33 * (size_t)(32 + max_taps));
  if (e2_dbg_env("E2_PACK_GATHER"))
    hipLaunchKernelGGL(pack_multi_gather_kernel, dim3(256, njobs), dim3(256), 0, ctx->stream,
                       (const PackJobDev*)jobs_dev);
  else
    hipLaunchKernelGGL(pack_multi_kernel, dim3(e2_dbg_env("E2_PACK_GRID") ? e2_dbg_env_int("E2_PACK_GRID") : 512, njobs),
                       dim3(256), lds, ctx->stream, (const PackJobDev*)jobs_dev);
  E2_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---- host side ----------------------------------------------------------------
// kind 0: 16x16x4 kernel, tile 16*MT channels x 64*NT positions ("MT,NT,CC,SK");
// kind 4: 4x4x1 kernel (igemm4_core.hpp), MT = channel groups of 4 per wave, WM x WN compute
// waves along the channels / positions, G work-groups per CU ("4,MG,NT,CC,SK,WM,WN,G")
struct IgemmCfg { int MT, NT, CC, SK, kind = 0, WM = 1, WN = 4, G = 1, KC = 32; };

static const int kMTs[] = {1, 2, 3, 4, 5, 6, 7, 8, 10, 13};

static bool has_fast_kw(int kw) { return kw == 1 || kw == 3 || kw == 4 || kw == 5; }

static int pad16mod32(int v) {          // smallest s >= v with s % 32 == 16
  int s = ((v + 15) / 16) * 16;
  if ((s & 31) == 0) s += 16;
  return s;
}
constexpr int generic_bmpad(int MT) { return ((16 * MT) & 31) == 16 ? 16 * MT : 16 * MT + 16; }

static int span_rows(int BN, int Wo) { return (BN + Wo - 2) / Wo; }

// exact upper bound of the span length of a BN-position tile:
// L = (q_last - q_first) + rows_crossed*(isY - Wo) + (kh-1)*isY + kw
static int span_lmax(const IgemmArgs& a, int BN) {
  return (BN - 1) + span_rows(BN, a.Wo) * ((int)a.isY - a.Wo) + (a.kh - 1) * (int)a.isY + a.kw;
}
// LDS row stride of a span.  Fast path: the DMA lanes past the span are masked off,
// the straddling lane writes <= 3 floats more.  Generic path: whole 256-float
// pieces (16 B per lane) are written.
static int span_lpad(const IgemmArgs& a, int BN) {
  if (has_fast_kw(a.kw)) return pad16mod32(span_lmax(a, BN) + 3);
  return pad16mod32(((span_lmax(a, BN) + 255) / 256) * 256);
}
// floats per LDS buffer.  Fast path: input spans only.  Generic path: spans +
// weight rows + DMA piece slack.
static size_t buf_floats(const IgemmArgs& a, int MT, int BN, int CC) {
  if (has_fast_kw(a.kw)) {    // + the group the pipeline prefetches past the end of a chunk
    const int gu = (a.kh * a.kw == 1 && CC % 16 == 0) ? 4 : 1;
    return (size_t)(CC + 4 * gu) * span_lpad(a, BN) + 64;
  }
  const int THW = a.kh * a.kw;
  return (size_t)CC * span_lpad(a, BN) + (size_t)THW * CC * generic_bmpad(MT) + 256 + 64;
}

// Pick the tiling when the caller (autotuner) forces none.  Cost model (cycles):
// a work-group's time is its MFMA count per wave at ~35 cycles, a per-chunk and a
// per-group overhead, plus prologue/epilogue; work-groups run 1-2 per CU;
// split-K pays the chip-wide fp32 atomic rate (1.3 TB/s) and a memset.
static IgemmCfg choose_cfg(const e2_ctx* ctx, const IgemmArgs& a, int* ok) {
  const int mblocks = e2_cdiv(a.Cout, 16);
  const int THW = a.kh * a.kw;
  const long Q = (long)a.Ho * a.Wo;
  const bool fast = has_fast_kw(a.kw);
  IgemmCfg best{0, 0, 0, 0};
  double bestCost = 1e300;
  const char* force = ctx->tiling[E2_TILING_IGEMM];
  if (force[0]) {
    IgemmCfg f{0, 0, 0, 0};
    int v[8];
    const int nf = sscanf(force, "%d,%d,%d,%d,%d,%d,%d,%d", &v[0], &v[1], &v[2], &v[3], &v[4], &v[5], &v[6], &v[7]);
    if (nf == 4) { f.MT = v[0]; f.NT = v[1]; f.CC = v[2]; f.SK = v[3]; *ok = 1; return f; }
    if (nf == 3 && v[0] == 1) { f.kind = 1; f.MT = v[1]; f.NT = v[2]; *ok = 1; return f; }
    if (nf == 5 && v[0] == 1) { f.kind = 1; f.MT = v[1]; f.NT = v[2]; f.KC = v[3]; *ok = 1; return f; }
    if (nf == 8 && v[0] == 4) {
      f.kind = 4; f.MT = v[1]; f.NT = v[2]; f.CC = v[3]; f.SK = v[4]; f.WM = v[5]; f.WN = v[6];
      f.G = v[7]; *ok = 1; return f;
    }
    // a forced tiling this entry point cannot run ("32,MB,NB" belongs to e2_conv3d_*_bf16):
    // fail, never fall back silently (e2hip.h, e2_set_tiling)
    *ok = -1;
    return f;
  }
  const double out_bytes = 4.0 * a.N * a.Cout * a.Do * (double)Q;
  const int cinP = ((a.Cin + 3) / 4) * 4;
  const int ccMax = fast ? 64 : 32;
  for (int MT : kMTs) {
    if (MT > mblocks && MT != 1) continue;
    const int nMT = e2_cdiv(mblocks, MT);
    if (nMT * 16 * MT > a.coP) continue;          // (the tiles would reach past the image's rows)
    for (int NT = 1; NT <= 4; NT *= 2) {
      if (NT == 4 && MT > 5) continue;
      const int BN = 64 * NT;
      const int nPT = (int)((Q + BN - 1) / BN);
      for (int CC = 4; CC <= ccMax && CC <= std::max(cinP, 4); CC += 4) {
        if (CC > 4 && e2_cdiv(a.Cin, CC) == e2_cdiv(a.Cin, CC - 4)) continue;  // no fewer chunks
        const size_t lds = 2 * buf_floats(a, MT, BN, CC) * 4;
        if (lds > 160 * 1024) break;
        const int perCU = (lds <= 80 * 1024 && MT * NT <= 8) ? 2 : 1;
        const int slots = ctx->num_cu * perCU;
        const int nChunkC = e2_cdiv(a.Cin, CC);
        const int nChunks = a.kd * nChunkC;
        const long wgs0 = (long)a.N * a.Do * nPT * nMT;
        const int gu = (fast && THW == 1 && CC % 16 == 0) ? 4 : 1;
        const double groups = (double)(CC / 4) * a.kh / gu;          // per chunk
        for (int SK = 1; SK <= 8; SK *= 2) {
          if (SK > nChunks) break;
          const long wgs = wgs0 * SK;
          const int per = e2_cdiv(nChunks, SK);
          const double mfma = (double)MT * NT * THW * (CC / 4) * 35.0;     // per chunk
          double chunk = mfma * perCU + groups * 60.0 + (fast ? 900.0 : 2200.0);
          if (fast && MT * NT * a.kw * gu < 24) chunk += groups * 400.0;   // loads not hidden
          const double wg_time = per * chunk + 6000.0 + MT * NT * 16 * 6.0;
          const double rounds = (double)((wgs + slots - 1) / slots);
          double cost = rounds * wg_time;
          if (SK > 1) cost += out_bytes * SK / 1.3e12 * 2.4e9 + out_bytes / 4e12 * 2.4e9 + 4000.0;
          if (cost < bestCost) { bestCost = cost; best = IgemmCfg{MT, NT, CC, SK}; }
        }
      }
    }
  }
  *ok = best.MT != 0;
  return best;
}

void e2i_pack_dims(int cout, int cin, int* ciP, int* coP) {
  *ciP = ((cin + 3) / 4) * 4 + 32;        // any channel-chunk size up to 32+
#ifdef E2_COP_SLACK
  *coP = ((cout + 15) / 16) * 16 + E2_COP_SLACK;   // (experiment: denser k-rows, fewer tilings fit)
#else
  *coP = ((cout + 15) / 16) * 16 + 16 * 13;   // room for any MT tiling
#endif
}

int e2i_pack_weights(e2_ctx* ctx, const float* w, float* wp, int Cout, int Cin,
                     int kd, int kh, int kw, int64_t wsO, int64_t wsI, int flip,
                     int ciP, int coP, int Rout, int Rin) {
  const long total = (long)kd * kh * kw * ciP * coP;
  int grid = (int)std::min<long>((total + 255) / 256, 4096);
  hipLaunchKernelGGL(pack_weights_kernel, dim3(grid), dim3(256), 0, ctx->stream, w, wp,
                     Cout, Cin, kd, kh * kw, (long)wsO, (long)wsI, flip, ciP, coP, Rout, Rin);
  E2_CHECK_HIP(hipGetLastError());
  return 0;
}

int e2i_igemm4_pairs(int kw, int MG, int NT) {
  switch (kw) {
    case 1: return e2i_igemm4_pairs_k1(MG, NT);
    case 3: return e2i_igemm4_pairs_k3(MG, NT);
    case 4: return e2i_igemm4_pairs_k4(MG, NT);
    case 5: return e2i_igemm4_pairs_k5(MG, NT);
  }
  return 0;
}

// zero-fill of a split-K output (atomics accumulate onto it); notes the region for callers
// that batch the fills of a whole step (e2_conv_last_zero_fill / e2_set_skip_zero_fill)
static int igemm_zero_output(e2_ctx* ctx, const IgemmArgs& a) {
  if (a.fill_base) {
    ctx->last_fill_ptr = a.fill_base; ctx->last_fill_n = a.fill_n;
    if (!ctx->skip_zero_fill) return e2i_fill_flat(ctx, a.fill_base, a.fill_n, 0.f);
    return 0;
  }
  const int R = a.upz * a.upy * a.upx;
  const int oc = a.Cout / (R > 1 ? R : 1);
  const int od = a.Do * a.upz, oh = a.Ho * a.upy, ow = a.Wo * a.upx;
  if (a.osY == ow && a.osZ == (long)oh * ow && a.osC == (long)od * oh * ow &&
      (a.N == 1 || a.osN == (long)oc * od * oh * ow)) {
    ctx->last_fill_ptr = a.out; ctx->last_fill_n = (size_t)a.N * oc * od * oh * ow;
    if (!ctx->skip_zero_fill)
      if (int rc = e2i_fill_flat(ctx, a.out, ctx->last_fill_n, 0.f)) return rc;
    return 0;
  }
  e2_tensor5 v{a.out, a.N, oc, od, oh, ow, a.osN, a.osC, a.osZ, a.osY};
  return e2i_fill_view(ctx, &v, 0.f);
}

// the 4x4x1-MFMA kernel (igemm4_core.hpp)
static int igemm4_conv(e2_ctx* ctx, const IgemmArgs& a, const IgemmCfg& c) {
  E2_REQUIRE(has_fast_kw(a.kw), "igemm4: tap rows of %d have no instance", a.kw);
  const int U = e2i_igemm4_pairs(a.kw, c.MT, c.NT);
  E2_REQUIRE(U > 0, "igemm4: no instance MG=%d NT=%d", c.MT, c.NT);
  E2_REQUIRE(c.WM >= 1 && c.WN >= 1 && c.WM * c.WN <= 12, "igemm4: WM x WN compute waves, at most 12");
  E2_REQUIRE(c.CC >= 4 && c.CC % 4 == 0 && c.CC % U == 0 && ((c.CC / U) * a.kh) % 2 == 0,
             "igemm4: CC must be a multiple of 4 and of %d with an even number of steps per chunk", U);
  E2_REQUIRE(c.SK >= 1 && c.G >= 1 && c.G <= 8, "igemm4: SK must be >= 1, G in 1..8");
  IgemmP p;
  p.in = a.in; p.wp = a.wp; p.out = a.out;
  p.Cin = a.Cin; p.Cout = a.Cout; p.kd = a.kd; p.kh = a.kh; p.kw = a.kw;
  p.THW = a.kh * a.kw;
  p.Do = a.Do; p.Ho = a.Ho; p.Wo = a.Wo; p.Q = a.Ho * a.Wo;
  p.isN = a.isN; p.isC = a.isC; p.isZ = a.isZ; p.isY = a.isY;
  p.osN = a.osN; p.osC = a.osC; p.osZ = a.osZ; p.osY = a.osY;
  p.ciP = a.ciP; p.coP = a.coP;
  const int BN = 64 * c.NT * c.WN, BM = 4 * c.MT * c.WM;
  p.Lpad = pad16mod32(span_lmax(a, BN) + 3);
  p.CC = c.CC;
  p.Din = a.Do + a.kd - 1;
  p.dbg = 0;
  p.N = a.N;
  p.nPT = e2_cdiv(p.Q, BN);
  p.nMT = e2_cdiv(a.Cout, BM);
  p.nChunkC = e2_cdiv(a.Cin, c.CC);
  {   // K splits: every split gets at least one chunk
    const int nChunks = a.kd * p.nChunkC;
    int sk = a.bias ? 1 : std::min(c.SK, nChunks);
    if (a.parts_max > 1) sk = std::min(sk, a.parts_max);
    const int per = e2_cdiv(nChunks, sk);
    p.splitK = e2_cdiv(nChunks, per);
  }
  const bool parts4 = a.parts_max > 1 && p.splitK > 1 && a.upz * a.upy * a.upx == 1;
  p.parts = parts4 ? 1 : 0; p.partStride = parts4 ? (long)a.part_stride : 0;
  if (a.nparts) *a.nparts = parts4 ? p.splitK : 1;
  p.atomic = (p.splitK > 1 && !parts4) ? 1 : 0;
  p.upz = a.upz; p.upy = a.upy; p.upx = a.upx; p.zpad = a.zpad;
  p.bufFloats = c.CC * p.Lpad + 64;
  p.stamps = nullptr; p.bias = a.bias; p.act = a.act; p.bf16 = 0; p.wide = 0;
  // channels past Cin are staged as copies of the last one: their weight rows must exist and be zero
  E2_REQUIRE(p.nChunkC * c.CC <= a.ciP, "igemm4: CC=%d pads Cin=%d beyond the packed image", c.CC, a.Cin);
  // a wave reads 64-channel rows starting at its first channel
  const int NA = (4 * c.MT + 63) / 64;
  E2_REQUIRE((p.nMT - 1) * BM + (c.WM - 1) * 4 * c.MT + 64 * NA <= a.coP, "igemm4: tile exceeds the packed coP");
  const size_t lds = 3 * (size_t)p.bufFloats * 4;
  E2_REQUIRE(lds <= 160 * 1024, "igemm4: tiling needs %zu B of LDS", lds);
  const long tiles = (long)a.N * p.splitK * p.nMT * p.Do * p.nPT;
  E2_REQUIRE(tiles < (1L << 31), "igemm4: too many tiles");
  const long grid = std::min<long>(tiles, (long)c.G * ctx->num_cu);
  ctx->last_fill_ptr = nullptr; ctx->last_fill_n = 0;
  if (p.atomic)
    if (int rc = igemm_zero_output(ctx, a)) return rc;
  if (e2_dbg_env("E2_VERBOSE"))
    fprintf(stderr, "[e2] igemm4 Cin=%d Cout=%d k=%d,%d,%d out=%d,%d,%d MG=%d NT=%d CC=%d SK=%d WM=%d WN=%d U=%d tiles=%ld grid=%ld lds=%zu\n",
            a.Cin, a.Cout, a.kd, a.kh, a.kw, a.Do, a.Ho, a.Wo, c.MT, c.NT, c.CC, p.splitK, c.WM, c.WN, U, tiles, grid, lds);
  Igemm4Extra x{c.WM, c.WN, (int)tiles};
  switch (a.kw) {
    case 1: return e2i_igemm4_launch_k1(ctx, p, x, c.MT, c.NT, (int)grid, lds);
    case 3: return e2i_igemm4_launch_k3(ctx, p, x, c.MT, c.NT, (int)grid, lds);
    case 4: return e2i_igemm4_launch_k4(ctx, p, x, c.MT, c.NT, (int)grid, lds);
    default: return e2i_igemm4_launch_k5(ctx, p, x, c.MT, c.NT, (int)grid, lds);
  }
}

int e2i_igemm_conv(e2_ctx* ctx, const IgemmArgs& a) {
  E2_REQUIRE(a.Do > 0 && a.Ho > 0 && a.Wo > 0 && a.Cin > 0 && a.Cout > 0,
             "igemm: empty problem");
  E2_REQUIRE(a.isY < (1 << 20), "igemm: input row stride too large");
  int ok = 0;
  IgemmCfg c = choose_cfg(ctx, a, &ok);
  E2_REQUIRE(ok >= 0, "igemm: the forced tiling '%s' is not one this launch can run (packed conv "
             "launches take \"MT,NT,CC,SK\", \"4,MG,NT,CC,SK,WM,WN,G\", \"1,MT,NT\" or \"1,MT,NT,KC,0\")",
             ctx->tiling[E2_TILING_IGEMM]);
  E2_REQUIRE(ok, "igemm: no tiling fits LDS (Cin=%d Cout=%d k=%dx%dx%d W=%d)", a.Cin,
             a.Cout, a.kd, a.kh, a.kw, a.Wo);
  if (a.gm_done) *a.gm_done = 0;
  if (a.up_bias_done) *a.up_bias_done = 0;
  if (a.nparts) *a.nparts = 1;
  const int src = ctx->tiling[E2_TILING_IGEMM][0] ? E2_SRC_FORCED : E2_SRC_MODEL;   // (no fallback here)
  if (c.kind == 4) {
    e2_note_launch(ctx, "igemm4", src, "4,%d,%d,%d,%d,%d,%d,%d", c.MT, c.NT, c.CC, c.SK, c.WM, c.WN, c.G);
    return igemm4_conv(ctx, a, c);
  }
  if (c.kind == 1) {                                             // "1,MT,NT": conv_pw.hip
    E2_REQUIRE(!ctx->mfma_bf16, "pointwise conv: an f32 kernel, not offered in bf16 mode");
    e2_note_launch(ctx, "pw_gemm", src, "1,%d,%d,%d,0", c.MT, c.NT, c.KC);
    return e2i_pw_conv(ctx, a, c.MT, c.NT, c.KC);
  }
  const bool fast = has_fast_kw(a.kw);
  e2_note_launch(ctx, !fast ? "igemm_generic" : (ctx->mfma_bf16 ? "igemm_bf16r" : "igemm"), src, "%d,%d,%d,%d",
                 c.MT, c.NT, c.CC, c.SK);
  // the gradient-mask epilogue lives in the specialised-width 16x16x4 kernel
  const bool gm = a.gm && fast && a.upz * a.upy * a.upx == 1 && a.Wo >= 4;
  E2_REQUIRE(c.CC >= 4 && c.CC % 4 == 0 && c.CC <= (fast ? 64 : 32),
             "igemm: CC must be a multiple of 4, at most %d", fast ? 64 : 32);
  E2_REQUIRE(c.NT == 1 || c.NT == 2 || c.NT == 4, "igemm: NT must be 1, 2 or 4");
  E2_REQUIRE(c.SK >= 1, "igemm: SK must be >= 1");
  IgemmP p;
  p.in = a.in; p.wp = a.wp; p.out = a.out;
  p.Cin = a.Cin; p.Cout = a.Cout; p.kd = a.kd; p.kh = a.kh; p.kw = a.kw;
  p.THW = a.kh * a.kw;
  p.Do = a.Do; p.Ho = a.Ho; p.Wo = a.Wo; p.Q = a.Ho * a.Wo;
  p.isN = a.isN; p.isC = a.isC; p.isZ = a.isZ; p.isY = a.isY;
  p.osN = a.osN; p.osC = a.osC; p.osZ = a.osZ; p.osY = a.osY;
  p.ciP = a.ciP; p.coP = a.coP;
  const int BN = 64 * c.NT;
  p.Lpad = span_lpad(a, BN);
  p.CC = c.CC;
  p.Din = a.Do + a.kd - 1;
  p.dbg = e2_dbg_env_int("E2_IGEMM_DBG");
  p.N = a.N;
  p.nPT = e2_cdiv(p.Q, BN);
  p.nMT = e2_cdiv(e2_cdiv(a.Cout, 16), c.MT);
  p.nChunkC = e2_cdiv(a.Cin, c.CC);
  p.splitK = a.bias ? 1 : std::min(c.SK, a.kd * p.nChunkC);     // the fused epilogue cannot split K
  const bool parts = a.parts_max > 1 && a.upz * a.upy * a.upx == 1;
  if (parts) p.splitK = std::min(p.splitK, a.parts_max);
  p.parts = (parts && p.splitK > 1) ? 1 : 0;
  p.partStride = p.parts ? (long)a.part_stride : 0;
  if (a.nparts) *a.nparts = p.parts ? p.splitK : 1;
  p.atomic = (p.splitK > 1 && !p.parts) ? 1 : 0;
  p.upz = a.upz; p.upy = a.upy; p.upx = a.upx; p.zpad = a.zpad;
  p.bufFloats = (int)buf_floats(a, c.MT, BN, c.CC);
  E2_REQUIRE(p.nMT * 16 * c.MT <= a.coP, "igemm: packed coP too small");
  if (fast) E2_REQUIRE(((a.Cin + 15) / 16) * 16 <= a.ciP, "igemm: packed ciP too small");
  else E2_REQUIRE(p.nChunkC * c.CC <= a.ciP, "igemm: packed ciP too small");
  size_t lds = 2 * (size_t)p.bufFloats * 4;
  E2_REQUIRE(lds <= 160 * 1024, "igemm: forced tiling needs %zu B of LDS", lds);
  // wide epilogue (fast path): dense output rows, plain stores
  const size_t tile_lds = (size_t)4 * 16 * c.MT * (16 * c.NT + 4) * 4 + (gm ? 4 * 16 * c.MT * 4 : 0);
  p.wide = (fast && (p.splitK == 1 || p.parts) && a.upz * a.upy * a.upx == 1 && (a.osY == a.Wo || a.Wo >= 4) &&
            std::max(lds, tile_lds) <= 160 * 1024 && !e2_dbg_env("E2_IGEMM_NARROW")) ? 1 : 0;
  if (p.wide) lds = std::max(lds, tile_lds);
  p.bias = a.bias; p.act = a.act;
  if (gm) {
    p.gm = 1; p.gm_src = a.gm_src; p.gsN = a.gsN; p.gsC = a.gsC; p.gsZ = a.gsZ;
    p.gm_dbias = a.gm_dbias; p.gm_bias = a.gm_bias;
    if (a.gm_done) *a.gm_done = 1;
  }
  E2_REQUIRE(!a.bias || p.wide, "igemm: the fused bias/act epilogue needs dense output rows, a "
             "specialised kernel width and %zu B of LDS", tile_lds);
  const long grid = (long)a.N * p.splitK * p.nMT * p.Do * p.nPT;
  E2_REQUIRE(grid < (1L << 31), "igemm: grid too large");
  ctx->last_fill_ptr = nullptr; ctx->last_fill_n = 0;
  if (p.atomic)
    if (int rc = igemm_zero_output(ctx, a)) return rc;
  // debug: per-work-group timeline stamps (fast path only), printed after a sync
  p.stamps = nullptr;
  static unsigned long long* stamp_buf = nullptr;
  const bool want_stamps = e2_dbg_env("E2_IGEMM_STAMPS") != nullptr && fast && !ctx->capturing;
  if (want_stamps) {
    if (!stamp_buf) E2_CHECK_HIP(hipMalloc(&stamp_buf, 8 * sizeof(unsigned long long) * 65536));
    E2_REQUIRE(grid <= 65536, "igemm stamps: grid too large");
    E2_CHECK_HIP(hipMemsetAsync(stamp_buf, 0, 8 * sizeof(unsigned long long) * grid, ctx->stream));
    p.stamps = stamp_buf;
  }
  // 1x1 taps: four channel groups per pipeline step when the chunk allows it
  const int GU = (fast && p.THW == 1 && c.CC % 16 == 0) ? 4 : 1;
  if (e2_dbg_env("E2_VERBOSE"))
    fprintf(stderr, "[e2] igemm%s Cin=%d Cout=%d k=%d,%d,%d out=%d,%d,%d MT=%d NT=%d CC=%d SK=%d GU=%d grid=%ld lds=%zu\n",
            (ctx->mfma_bf16 && fast) ? "(bf16)" : "", a.Cin, a.Cout, a.kd, a.kh, a.kw, a.Do, a.Ho, a.Wo, c.MT, c.NT, c.CC, p.splitK, GU, grid, lds);
  int rc;
  p.bf16 = (ctx->mfma_bf16 && fast) ? 1 : 0;     // the generic-width kernel stays f32
  if (p.bf16) {
    switch (a.kw) {
      case 1: rc = e2i_igemm_launch_k1_bf(ctx, p, c.MT, c.NT, GU, (int)grid, lds); break;
      case 3: rc = e2i_igemm_launch_k3_bf(ctx, p, c.MT, c.NT, GU, (int)grid, lds); break;
      case 4: rc = e2i_igemm_launch_k4_bf(ctx, p, c.MT, c.NT, GU, (int)grid, lds); break;
      default: rc = e2i_igemm_launch_k5_bf(ctx, p, c.MT, c.NT, GU, (int)grid, lds); break;
    }
  } else
  switch (a.kw) {
    case 1: rc = e2i_igemm_launch_k1(ctx, p, c.MT, c.NT, GU, (int)grid, lds); break;
    case 3: rc = e2i_igemm_launch_k3(ctx, p, c.MT, c.NT, GU, (int)grid, lds); break;
    case 4: rc = e2i_igemm_launch_k4(ctx, p, c.MT, c.NT, GU, (int)grid, lds); break;
    case 5: rc = e2i_igemm_launch_k5(ctx, p, c.MT, c.NT, GU, (int)grid, lds); break;
    default: rc = e2i_igemm_launch_generic(ctx, p, c.MT, c.NT, (int)grid, lds);
  }
  if (rc == 0 && want_stamps) {
    std::vector<unsigned long long> h(8 * grid);
    E2_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    E2_CHECK_HIP(hipMemcpy(h.data(), stamp_buf, 8 * sizeof(unsigned long long) * grid, hipMemcpyDeviceToHost));
    unsigned long long r0 = ~0ull, r1 = 0;
    double s_first = 0, s_loop = 0, s_epi = 0, s_tot = 0, max_tot = 0;
    for (long b = 0; b < grid; ++b) {
      const unsigned long long* s = &h[8 * b];
      r0 = std::min(r0, s[0]); r1 = std::max(r1, s[5]);
      s_first += (double)(s[2] - s[1]); s_loop += (double)(s[3] - s[2]);
      s_epi += (double)(s[4] - s[3]); s_tot += (double)(s[4] - s[1]);
      max_tot = std::max(max_tot, (double)(s[4] - s[1]));
    }
    double last_start = 0;
    for (long b = 0; b < grid; ++b) last_start = std::max(last_start, (double)(h[8 * b] - r0));
    fprintf(stderr, "[e2 stamps] grid=%ld  span(first start..last end)=%.2f us  last block start +%.2f us | per block (cycles, mean): "
            "to first barrier %.0f, main loop %.0f, epilogue %.0f, total %.0f (max %.0f)\n",
            grid, (double)(r1 - r0) * 0.01, last_start * 0.01, s_first / grid, s_loop / grid, s_epi / grid,
            s_tot / grid, max_tot);
    // in-kernel clock (MI355X_MICROARCH.md "DVFS give-back" item 6): shader cycles
    // (s_memtime) per 100 MHz tick (s_memrealtime) between a work-group's first and last
    // stamp, median over the work-groups
    std::vector<double> clk;
    for (long b = 0; b < grid; ++b) {
      const unsigned long long* s = &h[8 * b];
      if (s[5] > s[0]) clk.push_back((double)(s[4] - s[1]) / (double)(s[5] - s[0]) * 0.1);
    }
    if (!clk.empty()) {
      std::sort(clk.begin(), clk.end());
      fprintf(stderr, "[e2 stamps] in-kernel clock: median %.3f GHz (min %.3f, max %.3f) over %zu work-groups\n",
              clk[clk.size() / 2], clk.front(), clk.back(), clk.size());
    }
  }
  return rc;
}
