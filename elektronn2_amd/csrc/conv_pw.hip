// conv_pw.hip -- the 1x1x1 conv (the reference's "tensordot" branch, computations.py:330-335,
// 377-384) forward and data gradient as a plain GEMM of its own:
//
//   out[m][pos] = sum_k Wp[k][m] * in[k][pos]        (+ bias[m], activation)
//
// Wp is the packed image conv_igemm.hip already keeps for either direction ([k][m], m
// contiguous, zero rows / columns in the padding).  The implicit-GEMM kernel serves this layer
// at 29 us for 7 us of MFMA (DESIGN.md "known losses" item 5): with one tap there are only
// Cin / 4 k-steps, and every one of a wave's MT MFMAs per step needs its own 256-byte weight
// fragment straight from L2 -- the texture path, not the matrix pipe, sets the pace.  Here
// the weights go through LDS instead: the work-group stages chunks of 32 k-rows x 16*MT
// channels by LDS-DMA (double buffered, the next chunk lands under the current one's MFMAs),
// the four waves read their fragments with ds_read_b32 one step ahead (counted lgkmcnt), and
// only the activations -- each position belongs to ONE wave -- come from global memory, a
// whole chunk ahead.
//   work-group = 4 waves x NT blocks of 16 consecutive positions of one z-plane, all 16*MT
//   channels of an M tile; grid = N * Do * ceil(Q / (64 NT)) * ceil(Cout / (16 MT)).
// MFMA maps (16x16x4 f32): A[i = l & 15][k = l >> 4], B[k = l >> 4][j = l & 15],
// D: col = l & 15 (position), row = 4 * (l >> 4) + reg (channel).
// Tiling "1,MT,NT" (chunks of 32 k-rows) or "1,MT,NT,KC,0" (KC = 32 / 64 / 128 k-rows per
// chunk) through e2_set_tiling(E2_TILING_IGEMM, ...); never chosen untuned.  A chunk is the
// unit of the software pipeline -- the next chunk's weights (DMA) and activations (registers)
// are in flight while this one's MFMAs run -- so its MFMA time has to cover the memory
// latency: with 4 x 1 blocks a 32-row chunk is 0.4 us of MFMAs, and the small GEMMs that
// cannot fill the chip with work-groups (UpConv, K = 1-2 k) stalled at every chunk.
// UpConv (round 3): the same GEMM with M = Cout * R rows and the depth-to-space scatter in the
// epilogue -- row m = co * R + r goes to out[co][pz z + rz][py y + ry][px x + rx] -- and the
// bias of channel m / R (+ activation) fused, which the implicit-GEMM forms leave to one more
// launch.
#include "common.hpp"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_vp;
typedef const __attribute__((address_space(1))) void* gbl_vp;

namespace {


struct PwP {
  const float* in; const float* wp; float* out;
  const float* bias; int act;
  int N, Cin, Cout, Do, Wo, Q;
  long isN, isC, isZ, isY, osN, osC, osZ, osY;
  int coP, ciP, nPT, nMT, nChunks;
  int upz, upy, upx, R;                  // UpConv scatter (R = upz * upy * upx; 1 = plain conv)
};

constexpr int bms_of(int MT) { return ((16 * MT) & 31) == 16 ? 16 * MT : 16 * MT + 16; }

template <int MT, int NT, int KC>
__global__ __launch_bounds__(256) void pw_gemm_kernel(PwP p) {
  constexpr int kKC = KC, kSteps = KC / 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int BMS = bms_of(MT);               // LDS row stride (floats), 16 mod 32: the four
                                                // k-rows of a fragment read sit 16 banks apart
  constexpr int PIECES = kKC * BMS / 4;         // 16-byte pieces of a chunk
  constexpr int NI = (PIECES + 255) / 256;      // DMA instructions per thread and chunk
  constexpr unsigned BUFB = kKC * BMS * 4;
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  int b = blockIdx.x;
  const int mt = b % p.nMT; b /= p.nMT;         // M tiles of one position tile are neighbours
  const int pt = b % p.nPT; b /= p.nPT;
  const int z = b % p.Do;
  const int n = b / p.Do;
  const int m0 = mt * 16 * MT;

  // ---- per-lane addresses -------------------------------------------------------------
  long xoff[NT], ooff[NT];
  bool qok[NT];
#pragma unroll
  for (int nb = 0; nb < NT; ++nb) {
    const int q = pt * 64 * NT + (wave * NT + nb) * 16 + l15;
    qok[nb] = q < p.Q;
    const int qc = min(q, p.Q - 1);
    const int y = qc / p.Wo, x = qc - y * p.Wo;
    xoff[nb] = (long)n * p.isN + (long)z * p.isZ + (long)y * p.isY + x;
    ooff[nb] = (long)n * p.osN + (long)(z * p.upz) * p.osZ + (long)(y * p.upy) * p.osY + x * p.upx;
  }
  // weight pieces this thread stages: piece pi = it * 256 + tid covers floats 4 pi .. 4 pi + 3
  // of the chunk [k][BMS]; the pad columns of a row (BMS > 16 MT) re-read its last piece
  long woff[NI];
  int wrow[NI];                                  // the piece's k-row inside its chunk
#pragma unroll
  for (int it = 0; it < NI; ++it) {
    const int pi = min(it * 256 + tid, PIECES - 1);
    const int k = (pi * 4) / BMS, m = min(pi * 4 - k * BMS, 16 * MT - 4);
    woff[it] = (long)k * p.coP + m0 + m;
    wrow[it] = k;
  }
  const unsigned abase = (unsigned)(uintptr_t)(lds_vp)(lds + (kq * BMS + l15) * 4);

  f32x4 acc[MT][NT];
#pragma unroll
  for (int mb = 0; mb < MT; ++mb)
#pragma unroll
    for (int nb = 0; nb < NT; ++nb) acc[mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto stage = [&](int c, int buf) {
    const float* wc = p.wp + (long)c * kKC * p.coP;
    unsigned char* lb = lds + buf * BUFB + (wave * 64) * 16;
#pragma unroll
    for (int it = 0; it < NI; ++it)       // (rows past the packed image: never read -- steps_of)
      if (it * 256 + tid < PIECES && c * kKC + wrow[it] < p.ciP)
        __builtin_amdgcn_global_load_lds((gbl_vp)(wc + woff[it]), (lds_vp)(lb + it * 256 * 16), 16, 0, 0);
  };
  // activations of chunk c: B[k = 4 s + kq][position]; channels past the last one re-read it
  // (their weight rows are zero, the value only has to be finite)
  auto loadB = [&](int c, float (&B)[NT][kSteps]) {
#pragma unroll
    for (int s = 0; s < kSteps; ++s) {
      const int k = min(c * kKC + 4 * s + kq, p.Cin - 1);
#pragma unroll
      for (int nb = 0; nb < NT; ++nb)       // (asm: hipcc would wait for EVERYTHING in flight, the
                                            // next chunk's DMA included, before the first MFMA)
        asm volatile("global_load_dword %0, %1, off" : "=v"(B[nb][s]) : "v"(p.in + (long)k * p.isC + xoff[nb]));
    }
  };
  auto touchB = [&](float (&B)[NT][kSteps]) {   // after the counted wait: pins the uses behind it
#pragma unroll
    for (int nb = 0; nb < NT; ++nb)
#pragma unroll
      for (int s = 0; s < kSteps; ++s) asm volatile("" : "+v"(B[nb][s]));
  };
  auto readA = [&](unsigned base, int s, float (&A)[MT]) {
#pragma unroll
    for (int mb = 0; mb < MT; ++mb)
      asm volatile("ds_read_b32 %0, %1" : "=v"(A[mb]) : "v"(base + (unsigned)(s * 4 * BMS * 4 + mb * 64)));
  };
  auto fma = [&](float (&A)[MT], float (&B)[NT][kSteps], int s) {
#pragma unroll
    for (int mb = 0; mb < MT; ++mb) asm volatile("" : "+v"(A[mb]));
#pragma unroll
    for (int mb = 0; mb < MT; ++mb)
#pragma unroll
      for (int nb = 0; nb < NT; ++nb)
        acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[mb], B[nb][s], acc[mb][nb], 0, 0, 0);
  };
  // nst: k-steps of this chunk that hold real channels (the last chunk's zero rows are skipped
  // pairwise: the loop is unrolled by two)
  auto compute = [&](int buf, float (&B)[NT][kSteps], int nst) {
    const unsigned base = abase + buf * BUFB;
    float A0[MT], A1[MT];
    readA(base, 0, A0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < kSteps; s += 2) {
      if (s >= nst) break;
      readA(base, s + 1, A1);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(MT < 15 ? MT : 15) : "memory");
      __builtin_amdgcn_sched_barrier(0);
      fma(A0, B, s);
      __builtin_amdgcn_sched_barrier(0);
      if (s + 2 < kSteps) readA(base, s + 2, A0);
      __builtin_amdgcn_sched_barrier(0);
      if (s + 2 < kSteps) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(MT < 15 ? MT : 15) : "memory");
      else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      fma(A1, B, s + 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // (the read-ahead of a skipped step)
  };
  auto steps_of = [&](int c) { return min(kSteps, (p.Cin - c * kKC + 3) >> 2); };

  float B0[NT][kSteps], B1[NT][kSteps];
  stage(0, 0);
  loadB(0, B0);
  for (int c = 0; c < p.nChunks; c += 2) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();               // chunk c is in LDS; buffer 1 is free
    asm volatile("" ::: "memory");
    touchB(B0);
    if (c + 1 < p.nChunks) { stage(c + 1, 1); loadB(c + 1, B1); }
    __builtin_amdgcn_sched_barrier(0);
    compute(0, B0, steps_of(c));
    if (c + 1 >= p.nChunks) break;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    touchB(B1);
    if (c + 2 < p.nChunks) { stage(c + 2, 0); loadB(c + 2, B0); }
    __builtin_amdgcn_sched_barrier(0);
    compute(1, B1, steps_of(c + 1));
  }

  // ---- epilogue: 16 consecutive positions per 64-byte store segment ------------------------
  // the tile's biases go through LDS (one round trip instead of one per accumulator row)
  float* bl = reinterpret_cast<float*>(lds);
  if (p.bias) {
    __builtin_amdgcn_s_barrier();               // every wave is done with the weight buffers
    if (tid < 16 * MT) bl[tid] = p.bias[min(m0 + tid, p.Cout - 1) / p.R];
    __syncthreads();
  }
#pragma unroll
  for (int mb = 0; mb < MT; ++mb) {
    f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
    if (p.bias) bv = *reinterpret_cast<const f32x4*>(bl + 16 * mb + 4 * kq);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = m0 + 16 * mb + 4 * kq + r;
      if (co >= p.Cout) continue;
      long roff = (long)co * p.osC;
      if (p.R > 1) {                              // UpConv: row = (channel, sub-position)
        const int cc = co / p.R, rr = co - cc * p.R;
        const int rz = rr / (p.upy * p.upx), r2 = rr - rz * (p.upy * p.upx);
        const int ry = r2 / p.upx, rx = r2 - ry * p.upx;
        roff = (long)cc * p.osC + (long)rz * p.osZ + (long)ry * p.osY + rx;
      }
#pragma unroll
      for (int nb = 0; nb < NT; ++nb) {
        if (!qok[nb]) continue;
        float t = acc[mb][nb][r];
        if (p.bias) {
          t += bv[r];
          // relu keeps the sign of a negative pre-activation in the zero it stores (-0.0):
          // the backward pass tells relu'(0) = 0.5 from 0 by it (igemm_core.hpp, wide epilogue)
          if (p.act == E2_ACT_RELU) t = (t > 0.f) ? t : ((t == 0.f) ? 0.f : -0.f);
        }
        p.out[roff + ooff[nb]] = t;
      }
    }
  }
}

template <int MT, int NT, int KC>
int launch(e2_ctx* ctx, const PwP& p, long grid) {
  constexpr size_t ldsb = 2 * (size_t)KC * bms_of(MT) * 4;
  if constexpr (ldsb > 160 * 1024 || NT * (KC / 4) * 2 > 64) {      // (two register sets of a chunk's activations: beyond 64 they spill)
    e2_set_error("pointwise conv: %d x %d blocks with %d-row chunks do not fit (LDS %zu B)", MT, NT, KC, ldsb);
    return 2;
  } else {
    static bool attr_done = false;
    if (!attr_done && ldsb > 64 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pw_gemm_kernel<MT, NT, KC>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
      if (e != hipSuccess) { e2_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return 1; }
      attr_done = true;
    }
    hipLaunchKernelGGL((pw_gemm_kernel<MT, NT, KC>), dim3((unsigned)grid), dim3(256), ldsb, ctx->stream, p);
    E2_CHECK_HIP(hipGetLastError());
    return 0;
  }
}

template <int MT, int NT>
int launch_kc(e2_ctx* ctx, const PwP& p, long grid, int KC) {
  if (KC == 32) return launch<MT, NT, 32>(ctx, p, grid);
  if (KC == 64) return launch<MT, NT, 64>(ctx, p, grid);
  if (KC == 128) return launch<MT, NT, 128>(ctx, p, grid);
  e2_set_error("pointwise conv: chunks of %d k-rows (32, 64 or 128)", KC);
  return 2;
}

}  // namespace

int e2i_pw_conv(e2_ctx* ctx, const IgemmArgs& a, int MT, int NT, int KC) {
  E2_REQUIRE(a.kd == 1 && a.kh == 1 && a.kw == 1, "pointwise conv: kernel %dx%dx%d is not 1x1x1", a.kd, a.kh, a.kw);
  E2_REQUIRE(NT == 1 || NT == 2, "pointwise conv: NT must be 1 or 2");
  // (called with a partial-sum budget it writes the complete result: *nparts stays 1)
  PwP p;
  p.in = a.in; p.wp = a.wp; p.out = a.out; p.bias = a.bias; p.act = a.act;
  p.upz = a.upz; p.upy = a.upy; p.upx = a.upx; p.R = a.upz * a.upy * a.upx;
  E2_REQUIRE(p.R >= 1 && a.Cout % p.R == 0, "pointwise conv: %d rows for %d sub-positions", a.Cout, p.R);
  if (p.R > 1 && a.up_bias) {              // UpConv: bias[m / R] (+ act) in the epilogue
    p.bias = a.up_bias; p.act = a.up_act;
    if (a.up_bias_done) *a.up_bias_done = 1;
  }
  p.ciP = a.ciP;
  p.N = a.N; p.Cin = a.Cin; p.Cout = a.Cout; p.Do = a.Do; p.Wo = a.Wo; p.Q = a.Ho * a.Wo;
  p.isN = a.isN; p.isC = a.isC; p.isZ = a.isZ; p.isY = a.isY;
  p.osN = a.osN; p.osC = a.osC; p.osZ = a.osZ; p.osY = a.osY;
  p.coP = a.coP;
  p.nPT = e2_cdiv(p.Q, 64 * NT);
  p.nMT = e2_cdiv(e2_cdiv(a.Cout, 16), MT);
  p.nChunks = e2_cdiv(a.Cin, KC);
  E2_REQUIRE(p.nMT * 16 * MT <= a.coP, "pointwise conv: packed coP too small");
  E2_REQUIRE(a.Cin <= a.ciP, "pointwise conv: packed ciP too small");
  const long grid = (long)a.N * p.Do * p.nPT * p.nMT;
  E2_REQUIRE(grid < (1L << 31), "pointwise conv: grid too large");
  ctx->last_fill_ptr = nullptr; ctx->last_fill_n = 0;
#define E2_L(M, N_) if (MT == M && NT == N_) return launch_kc<M, N_>(ctx, p, grid, KC);
  E2_L(4, 1) E2_L(4, 2) E2_L(5, 1) E2_L(5, 2) E2_L(6, 1) E2_L(6, 2) E2_L(7, 1) E2_L(7, 2)
  E2_L(8, 1) E2_L(8, 2) E2_L(10, 1) E2_L(10, 2) E2_L(13, 1) E2_L(13, 2) E2_L(16, 1)
#undef E2_L
  e2_set_error("pointwise conv: no instance MT=%d NT=%d", MT, NT);
  return 2;
}
