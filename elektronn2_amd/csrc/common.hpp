// Internal helpers shared by the libe2hip translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string>
#include "../../include/e2hip.h"

struct e2_ctx {
  int device;
  hipStream_t stream;
  int num_cu;
  bool capturing;
  float* zeros;        // 1 KiB of zeros in device memory (masked-lane DMA source)
  hipEvent_t fork_ev[32];   // dependency events of e2_stream_fork / e2_stream_join
  int fork_next;
  int mfma_bf16;       // e2_set_mfma_dtype: 1 = conv GEMMs round their operands to bf16
  int skip_zero_fill;  // e2_set_skip_zero_fill: split-K outputs were zeroed by the caller
  float* last_fill_ptr;     // flat region the last conv launch zero-filled (n = 0: none)
  size_t last_fill_n;
  char tiling[2][64];       // e2_set_tiling: forced tiling of the igemm / wgrad launches ("" = cost model)
  int loss_sum_mode;        // e2_set_loss_grad_mode: 1 = NLL gradients are NOT divided by the labelled count
  float* loss_count_out;    // ... and the count is also written here (the slot behind the gradient arena)
  int input_slack;          // e2_set_input_slack: finite readable bytes behind the x of the launches that follow
  int image_rows;           // e2_set_image_rows: floats per k-row of the packed images the next launches read (0 = formula)
  char last_launch[160];    // e2_last_launch: "<kernel family> <tiling that ran> <forced|model|fallback>"
  unsigned tiling_fallbacks;   // launches since e2_ctx_create whose forced tiling was NOT the one that ran
};

// what decided the tiling of a launch (third word of e2_last_launch)
enum { E2_SRC_MODEL = 0, E2_SRC_FORCED = 1, E2_SRC_FALLBACK = 2 };
// record the conv GEMM launch that is about to be issued (api.hip)
void e2_note_launch(e2_ctx* ctx, const char* family, int src, const char* fmt, ...);

// Debug switches (timing ablations, in-kernel stamps, verbose launch log) are compiled in
// only with -DE2_DEBUG_ENV (make DEBUG_ENV=1): the release library never reads the process
// environment -- tilings arrive through e2_set_tiling().
#ifdef E2_DEBUG_ENV
#include <stdlib.h>
static inline const char* e2_dbg_env(const char* name) { return getenv(name); }
#else
static inline const char* e2_dbg_env(const char*) { return nullptr; }
#endif
static inline int e2_dbg_env_int(const char* name) {
  const char* v = e2_dbg_env(name);
  return v ? atoi(v) : 0;
}

void e2_set_error(const char* fmt, ...);

#define E2_CHECK_HIP(expr)                                                   \
  do {                                                                       \
    hipError_t _e = (expr);                                                  \
    if (_e != hipSuccess) {                                                  \
      e2_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr,             \
                   hipGetErrorString(_e));                                   \
      return 1;                                                              \
    }                                                                        \
  } while (0)

#define E2_REQUIRE(cond, ...)                                                \
  do {                                                                       \
    if (!(cond)) {                                                           \
      e2_set_error(__VA_ARGS__);                                             \
      return 2;                                                              \
    }                                                                        \
  } while (0)

static inline int e2_cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t e2_cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- internal launchers (defined in the .hip files) ---------------------

// Generic strided "valid" correlation as implicit GEMM on fp32 MFMA.
//   out[n][oc][z][y][x] (+)= sum_ic sum_t Wp[dz][t][ic][oc] * in[n][ic][z+dz][y+ty][x+tx]
// Wp is the PACKED weight image produced by e2i_pack_weights (flip folded in).
struct IgemmArgs {
  const float* in;
  const float* wp;
  float* out;
  int N, Cin, Cout;
  int kd, kh, kw;
  int Do, Ho, Wo;          // output spatial
  int64_t isN, isC, isZ, isY;
  int64_t osN, osC, osZ, osY;
  int ciP, coP;            // padded channel counts of the packed image
  // upconv scatter mode: out channel oc' = co*R + r, stored at
  // out[n][co][pz*z+rz][py*y+ry][px*x+rx]; R = pz*py*px (1 = off)
  int upz, upy, upx;
  // fused epilogue (conv forward of a layer that does not pool): out = act(conv + bias[oc]);
  // relu writes -0.0 where the pre-activation was NEGATIVE and +0.0 where it was exactly
  // zero, so that the backward can tell relu'(0) = 0.5 from 0 without the pre-activation
  const float* bias = nullptr;
  int act = 0;
  // UpConv (upz * upy * upx > 1): bias[oc' / R] + activation in the epilogue IF the chosen
  // kernel can (the pointwise GEMM, conv_pw.hip); *up_bias_done reports it, else the caller
  // runs its bias / activation pass on the output
  const float* up_bias = nullptr;
  int up_act = 0;
  int* up_bias_done = nullptr;
  // gradient-mask epilogue (data gradient fused with the activation backward of the layer
  // that produced this conv's input): out = conv * act'(gm_src), gm_dbias += column sums.
  // gm_src: that layer's activated output with dense rows (nullptr = linear activation).
  // e2i_igemm_conv reports in *gm_done whether the chosen kernel did it (else the caller
  // runs the activation backward in place on the output).
  int gm = 0;
  const float* gm_src = nullptr;
  int64_t gsN = 0, gsC = 0, gsZ = 0;
  float* gm_dbias = nullptr;
  const float* gm_bias = nullptr;   // non-null: gm_src is the PRE-activation, slope from gm_src + gm_bias[c]
  int* gm_done = nullptr;
  // split-K zero-fill of an output that is the interior of a zero-padded scratch buffer:
  // the whole buffer may be filled flat (batched by the caller), nullptr = fill the view
  float* fill_base = nullptr;
  size_t fill_n = 0;
  // data gradient: `in` is dy zero-padded by kd - 1 planes at both ends of z (the contract
  // of e2_conv3d_dgrad*): the kernels skip the tap planes that read only that border
  int zpad = 0;
  // split-K with partial-sum stores instead of atomics (e2_conv3d_*_packed_parts): up to
  // parts_max splits, split s stores to out + s * part_stride; *nparts receives the number
  // of parts written (1: `out` holds the complete result).  parts_max <= 1: off.
  int parts_max = 0;
  int64_t part_stride = 0;
  int* nparts = nullptr;
};
int e2i_igemm_conv(e2_ctx*, const IgemmArgs& a);
int e2i_pw_conv(e2_ctx*, const IgemmArgs& a, int MT, int NT, int KC);   // 1x1x1 GEMM with LDS-staged weights (conv_pw.hip)

// Wp[dz][t][ic(ciP)][oc(coP)] = w[(oc/Rout)*wsO + (ic/Rin)*wsI + tap + oc%Rout + ic%Rin],
// tap = (dz*kh*kw+t), reversed when flip.  Zero in the padding.  Cout/Cin are
// the GEMM's channel counts (already multiplied by Rout/Rin).  Rout/Rin > 1 fold
// UpConv's sub-position r into the out / in channel index (kd=kh=kw=1 then).
int e2i_pack_weights(e2_ctx*, const float* w, float* wp, int Cout, int Cin,
                     int kd, int kh, int kw, int64_t wsO, int64_t wsI, int flip,
                     int ciP, int coP, int Rout, int Rin);
void e2i_pack_dims(int cout, int cin, int* ciP, int* coP);

struct WgradArgs {
  const float* x;     // input activations view
  const float* dy;    // output-gradient view (unpadded dims)
  float* dw;          // [M][Cin*T] dense, accumulated atomically (pre-zeroed)
  int N, Cin, Cout;
  int kd, kh, kw;
  int Do, Ho, Wo;
  int64_t xsN, xsC, xsZ, xsY;
  int64_t dsN, dsC, dsZ, dsY;
  int flip;           // store tap index reversed
  int upR;            // >1: rows are (co*R + r); store dw[co][col][r] (UpConv)
  int accumulate;     // 1: dw += grad (caller zeroed it); 0: dw = grad
  int dy_padded;      // 1: dy is the interior of a zero-padded buffer (row gaps hold
                      //    zeros, >= 128 readable bytes follow its last element)
};
int e2i_upconv_dpre_s2d(e2_ctx*, const e2_tensor5* dout, const e2_tensor5* yout, int pz,
                        int py, int px, int act, float* s2d, float* dbias);
int e2i_wgrad_conv(e2_ctx*, const WgradArgs& a);
// conv_wgrad_direct.hip: dy operand straight from global memory (needs dy_padded)
int e2i_wgrad_direct(e2_ctx*, const WgradArgs& a, int MT, int NT, int BP, int PS, int WK, int xcd = 0);
int e2i_wgrad_direct_lpad(const WgradArgs& a, int BP);
// conv_pw_wgrad.hip: 1x1x1 / UpConv weight gradient as a GEMM with K-contiguous operands
int e2i_pw_wgrad(e2_ctx*, const WgradArgs& a, int MT, int NT, int S);
int e2i_pw_wgrad_ks(e2_ctx*, const WgradArgs& a, int MT, int NT, int S);
int e2i_wgrad_ks(e2_ctx*, const WgradArgs& a, int MT, int NT, int S);     // the same for kernels with taps
size_t e2i_wgrad_direct_buf_floats(const WgradArgs& a, int NT, int BP, int WK);

// view helpers (pointwise.hip)
int e2i_fill_view(e2_ctx*, const e2_tensor5* v, float value);
int e2i_fill_flat(e2_ctx*, float* ptr, size_t n, float value);   // kernel fill (graph safe)
