/* e2hip.h -- C ABI of libe2hip.so: MI355X (gfx950) kernels for the 3-D
 * conv / pool / upconv training step behind ELEKTRONN2's neuromancer
 * Conv / Pool / UpConv nodes.
 *
 * The reference has NO FFI for this path: its seam is the Theano op calls in
 * elektronn2/neuromancer/computations.py and the one theano.function() call in
 * graphutils.py:376-387.  Each entry point below names the reference call it
 * replaces (paths relative to /root/reference/elektronn2/).
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on error; the message is
 *     available from e2_last_error() (thread-local).
 *   - all data pointers are DEVICE pointers owned by the caller; workspace sizes
 *     come from e2_*_workspace_bytes().  The library's only device allocation is
 *     one 1 KiB page of zeros per context, made in e2_ctx_create and released in
 *     e2_ctx_destroy (source of masked DMA lanes); a -DE2_DEBUG_ENV build may also
 *     allocate in-kernel timeline buffers when asked to through the environment.
 *   - the release library never reads the process environment: tilings arrive
 *     through e2_set_tiling().
 *   - layouts: activations (b,f,z,x,y) = NCDHW, fp32; weights
 *     (n_f, n_in, kz, kx, ky) = KCDHW, fp32  (neural.py:615-623 'dnn' order).
 *     Tensors are described by e2_tensor5: sizes + element strides, so crops,
 *     channel-concat slices and zero-padded gradient buffers are views.
 *     The innermost (y / "W") stride must be 1.
 *   - one hipStream_t per context; launches are asynchronous on it; no
 *     internal threads; nothing here synchronises the device, so every call is
 *     legal inside hipStreamBeginCapture/EndCapture (e2_graph_*).
 */
#ifndef E2HIP_H
#define E2HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct e2_ctx e2_ctx;

/* 5-D fp32 tensor view. n = batch, c = features, d/h/w = z/x/y spatial.
 * Strides in ELEMENTS; the w stride is implicitly 1. */
typedef struct e2_tensor5 {
  float*  ptr;
  int32_t n, c, d, h, w;
  int64_t sn, sc, sd, sh;
} e2_tensor5;

enum { E2_ACT_LIN = 0, E2_ACT_RELU = 1 };
enum { E2_MFMA_F32 = 0, E2_MFMA_BF16 = 1 };
enum { E2_TILING_IGEMM = 0, E2_TILING_WGRAD = 1 };

/* ---- context / stream / errors --------------------------------------- */
int  e2_ctx_create(int device, e2_ctx** out);
int  e2_ctx_destroy(e2_ctx* ctx);
/* stream: a hipStream_t (as void*); NULL = the default stream. */
int  e2_ctx_set_stream(e2_ctx* ctx, void* stream);
const char* e2_last_error(void);
int  e2_version(void);
/* Arithmetic of the convolution GEMMs (no reference counterpart: Theano computes in
 * float32 throughout).  E2_MFMA_F32 (default): f32 operands, exact f32
 * products and sums.  E2_MFMA_BF16: the operands of the packed-weight forward / data-
 * gradient kernels (tap rows of 1, 3, 4 or 5) and of e2_conv3d_wgrad_pad are rounded
 * to bf16 (nearest even) on their way into the matrix core; products and sums stay f32;
 * tensors in memory stay f32.  The first-layer, head, generic-width kernels and the plain
 * e2_conv3d_wgrad entry point always compute in f32 (UpConv follows the setting, and so do the
 * two GEMMs of e2_tail_fwd_bwd's 1x1x1 layer; its head stays f32).  Not to be changed while a graph is being captured. */
int  e2_set_mfma_dtype(e2_ctx* ctx, int dtype);
int  e2_get_mfma_dtype(const e2_ctx* ctx);
/* Tiling of the conv GEMM launches (no reference counterpart; stands where Theano's
 * per-op algorithm choice `dnn_conv3d(..., algo=...)` stood, computations.py:30,391).
 * kind E2_TILING_IGEMM: cfg = "MT,NT,CC,SK" for the packed forward / dgrad / UpConv
 * launches (16x16x4 MFMA kernel: 16*MT output channels x 64*NT positions per work-group,
 * CC input channels per LDS chunk, SK-way split of K), or cfg = "4,MG,NT,CC,SK,WM,WN,G"
 * (4x4x1 MFMA kernel, persistent: 4*MG channels x 64*NT positions per WAVE, WM x WN <= 12
 * compute waves per work-group along the channels / positions, G work-groups per CU
 * walking the tiles), or, for 1x1x1 kernels and UpConv, cfg = "1,MT,NT" / "1,MT,NT,KC,0" (the
 * GEMM with LDS-staged weights of conv_pw.hip: 16*MT channels x 64*NT positions per
 * work-group, KC = 32 / 64 / 128 reduction channels per pipeline chunk, 32 in the short form;
 * for UpConv its epilogue scatters depth-to-space and applies bias + activation);
 * kind E2_TILING_WGRAD: cfg = "MT,NT,WK,BP,PS" for
 * e2_conv3d_wgrad / e2_conv3d_wgrad_pad (WK 1 / 14 direct kernel, 0 / 4 LDS-staged, 7 =
 * "MT,NT,7,0,S": 1x1x1 kernels and UpConv, the GEMM with K-contiguous operands of
 * conv_pw_wgrad.hip, 2 x 2 waves of MT x NT blocks, S position splits; 8 = "MT,NT,8,0,S": the
 * same GEMM with ONE 16 MT x 16 NT tile of dW per work-group whose four waves split the positions
 * and sum their partial tiles through LDS; 9 = "MT,NT,9,0,S": that kernel for a conv WITH taps --
 * every (input channel, tap) column of dW is a K-contiguous row of x at the tap's constant shift;
 * e2_conv3d_wgrad_pad only: needs the gradient at the input's row pitch, (kh - 1) input rows
 * + kw - 1 >= 31 zeros behind a gradient plane, f32 mode and e2_set_input_slack(ctx, >= 128);
 * BP positions per tile, PS position splits).  The forms 7, 8 and 9 are exempt from the rule
 * below: where their layout requirements are not met the call takes the cost model's choice --
 * a FALLBACK, which e2_last_launch() reports as such and e2_tiling_fallbacks() counts (the tuning
 * keys hold the row pitch, not the plane pitch, so a shipped entry can meet a view it cannot
 * run; a caller that wants the error checks e2_last_launch after the call).  The setting holds for every following
 * launch of that kind on this context until changed; cfg NULL or "" returns the choice
 * to the library's cost model.  A tiling the problem cannot use (LDS, instance list)
 * makes the launch fail with an error, never silently fall back.  A launch captured
 * into a graph keeps the tiling that was set at capture time. */
int  e2_set_tiling(e2_ctx* ctx, int kind, const char* cfg);
/* Which kernel the last conv GEMM launch of this context ran (packed forward / data gradient /
 * UpConv, weight gradient, f32 and bf16 forms; same seam: Theano reports the chosen `algo` of a
 * dnn conv op in its profile, computations.py:30,391).  buf receives
 * "<kernel family> <tiling that ran> <source>", e.g. "wgrad_ks 13,2,9,0,4 forced":
 *   family: igemm | igemm_bf16r | igemm_generic | igemm4 | pw_gemm | conv_bf16 |
 *           wgrad_ks | pw_wgrad_ks | pw_wgrad | wgrad_direct | wgrad_direct_bf16r | wgrad_lds |
 *           wgrad_bf16
 *   source: "forced"   = the e2_set_tiling string was honoured,
 *           "model"    = no string was set, the library's cost model chose,
 *           "fallback" = a string was set but this launch could not run it and took the cost
 *                        model's choice (weight-gradient forms 7 / 8 / 9 only; every other
 *                        unrunnable string is an ERROR of the launch).
 * Host-side bookkeeping, legal during capture; "" before the first launch.
 * e2_tiling_fallbacks: number of "fallback" launches since e2_ctx_create. */
int  e2_last_launch(e2_ctx* ctx, char* buf, int n);
unsigned e2_tiling_fallbacks(const e2_ctx* ctx);

/* ---- conv  (computations.py:364-428 conv(), 3-D branch; F1: true
 *      convolution, kernel flipped in every spatial dim, 'valid') --------- */
/* bytes of workspace conv fwd / dgrad need for the packed weight image. */
size_t e2_conv3d_workspace_bytes(int cout, int cin, int kd, int kh, int kw);

/* y = conv_valid_flip(x, w).  y must be (n, cout, d-kd+1, h-kh+1, w-kw+1).
 * w: dense KCDHW [cout][cin][kd][kh][kw].  ws: workspace (device). */
int e2_conv3d_fwd(e2_ctx*, const e2_tensor5* x, const float* w,
                  int cout, int kd, int kh, int kw,
                  const e2_tensor5* y, void* ws, size_t ws_bytes);

/* dx = d(loss)/dx given dy (replaces Theano's GpuDnnConv3dGradI /
 * conv3d2d grad born at model.py:182).  dy_pad is dy stored in the interior
 * of a buffer zero-padded by (kd-1, kh-1, kw-1) on every side:
 * dy_pad dims = (n, cout, do+2(kd-1), ho+2(kh-1), wo+2(kw-1)); the border
 * MUST be zero.  dx dims = (n, cin, do+kd-1, ho+kh-1, wo+kw-1). */
int e2_conv3d_dgrad(e2_ctx*, const e2_tensor5* dy_pad, const float* w,
                    int cin, int kd, int kh, int kw,
                    const e2_tensor5* dx, void* ws, size_t ws_bytes);

/* Split form of the two calls above: the packed weight image depends only on
 * w, so a caller that runs many convolutions with the same weights (or wants
 * the repack off the critical path) packs once.  mode 0 = forward image,
 * mode 1 = dgrad image.  ws must hold e2_conv3d_workspace_bytes(). */
int e2_conv3d_pack(e2_ctx*, const float* w, int cout, int cin, int kd, int kh,
                   int kw, int mode, void* ws, size_t ws_bytes);
int e2_conv3d_fwd_packed(e2_ctx*, const e2_tensor5* x, const void* wp, int cout,
                         int kd, int kh, int kw, const e2_tensor5* y);
int e2_conv3d_dgrad_packed(e2_ctx*, const e2_tensor5* dy_pad, const void* wp,
                           int cin, int kd, int kh, int kw, const e2_tensor5* dx);
/* Conv layer WITHOUT pooling, fused: out = act(conv(x, w) + bias[oc]) in the GEMM's
 * epilogue (neural.py:662-712 with pool_shape all ones).  Needs dense output rows
 * and a kernel width of 1, 3, 4 or 5 (e2_conv3d_fwd_packed + e2_pool_bias_act_fwd
 * otherwise).  relu stores -0.0 where the pre-activation was negative and +0.0
 * where it was exactly 0 (equal values; the sign is what e2_bias_act_bwd_out reads
 * to reproduce Theano's relu'(0) = 0.5). */
int e2_conv3d_fwd_packed_act(e2_ctx*, const e2_tensor5* x, const void* wp, int cout,
                             int kd, int kh, int kw, const float* bias, int act,
                             const e2_tensor5* out);
/* its backward: dy = dout * act'(out) (dy may be the interior of the padded gradient
 * buffer), dbias[oc] += sum(dy) when dbias != NULL. */
int e2_bias_act_bwd_out(e2_ctx*, const e2_tensor5* dout, const e2_tensor5* out, int act,
                        const e2_tensor5* dy, float* dbias);
/* e2_conv3d_dgrad_packed FUSED with e2_bias_act_bwd_out of the layer that produced this
 * conv's input (a conv layer without pooling whose only consumer is this conv; T.grad's
 * chain ConvGradI -> relu', model.py:182):
 *   dy_prev = dgrad(dy_pad, w) * act_prev'(out_prev),  dbias_prev[c] += sum(dy_prev[c])
 * written straight into the INTERIOR of that layer's zero-padded gradient buffer
 * dy_pad_prev (n, cin, d + 2 pd, h + 2 ph, w + 2 pw), interior at (pd, ph, pw); its border
 * must be (and stays) zero.  out_prev (n, cin, d, h, w) = that layer's activated output
 * as e2_conv3d_fwd_packed_act stored it (bias_prev == NULL), or -- for a layer whose forward
 * ran as e2_conv3d_fwd_packed + e2_pool_bias_act_fwd with a (1,1,1) window -- its
 * PRE-activation conv output, the slope then being act'(out_prev + bias_prev[c]).  The mask is applied in the GEMM's epilogue (also
 * under split-K: it is linear in the partial sums; a contiguous dy_pad_prev is then
 * zero-filled whole); tilings without that epilogue run the two steps one after the other,
 * in place.  dbias_prev may be NULL. */
int e2_conv3d_dgrad_packed_actbwd(e2_ctx*, const e2_tensor5* dy_pad, const void* wp, int cin,
                                  int kd, int kh, int kw, const e2_tensor5* out_prev,
                                  int act_prev, const float* bias_prev,
                                  const e2_tensor5* dy_pad_prev, int pd, int ph, int pw,
                                  float* dbias_prev);

/* dw[cout][cin][kd][kh][kw] = d(loss)/dw (replaces GpuDnnConv3dGradW).
 * dy is the UNPADDED view (n, cout, do, ho, wo) (it may be the interior view
 * of a padded buffer).  dw is overwritten. */
int e2_conv3d_wgrad(e2_ctx*, const e2_tensor5* x, const e2_tensor5* dy,
                    float* dw, int kd, int kh, int kw);
/* same, but dw += gradient (no internal memset): for callers that zero one
 * flat gradient arena per step. */
int e2_conv3d_wgrad_acc(e2_ctx*, const e2_tensor5* x, const e2_tensor5* dy,
                        float* dw, int kd, int kh, int kw);
/* Fast path of the same gradient for the training plan: dy_pad is the ZERO-PADDED
 * gradient buffer that e2_conv3d_dgrad_packed reads, shape (n, cout, do+2(kd-1),
 * ho+2(kh-1), wo+2(kw-1)); the interior holds dy, the borders MUST be zero, and
 * at least 128 readable bytes must follow the buffer's last element (the kernel
 * fetches dy straight from memory, 16 bytes per lane, and masks what lies past a
 * plane).  accumulate: 0 = dw is overwritten, 1 = dw += gradient. */
int e2_conv3d_wgrad_pad(e2_ctx*, const e2_tensor5* x, const e2_tensor5* dy_pad,
                        float* dw, int kd, int kh, int kw, int accumulate);

/* Repack MANY weight tensors in one launch (after an optimiser step): fill one
 * record per (tensor, mode) with e2_pack_job_fill on the host, copy the records
 * (e2_pack_job_bytes() each) to the device, then call e2_conv3d_pack_multi.  It
 * rewrites the weight-carrying part of every image only: the images must have been
 * zero-filled (or packed by e2_conv3d_pack) once before the first call.  mode 0 / 1: a
 * conv's forward / data-gradient image; mode 2 / 3: an UpConv's (w [cout][cin][kd][kh][kw]
 * with the factors in place of the kernel extents; see e2_upconv3d_fwd_packed). */
size_t e2_pack_job_bytes(void);
int e2_pack_job_fill(void* rec, const float* w, void* wp, int cout, int cin,
                     int kd, int kh, int kw, int mode);
/* Row length of the packed conv weight images (default: cout rounded to 16, + 208 floats -- any M
 * tiling fits).  e2_pack_job_set_stride packs ONE image of e2_conv3d_pack_multi with rows of `rows`
 * floats (a multiple of 4, >= cout rounded to 16; modes 0 / 1 only); e2_set_image_rows announces
 * that length to the e2_conv3d_pack / e2_conv3d_{fwd,dgrad}_packed* calls that follow (0 = the
 * formula again): the launch that reads an image must be told the length it was packed with.  A
 * tiling whose tiles reach past the rows is an error of the launch, never an over-read.  Purpose:
 * once the tiling of a launch is known its image needs 224 instead of 416 floats per row for 200
 * channels -- the repack writes a third less, the weight rows lie closer (DESIGN finding 52).
 * UpConv images and e2_tail_fwd_bwd's images always use the formula. */
int e2_set_image_rows(e2_ctx* ctx, int rows);
int e2_pack_job_set_stride(void* rec, int rows);
/* optional, after e2_pack_job_fill: `rows` = how far the M tiles of the launch that reads this image
 * reach (number of M tiles x 16 MT of its tiling).  The repack rewrites the real rows + the padding
 * rows up to there instead of its default (rows rounded to 16, + 96: any tiling); rows beyond stay
 * zero from the one-time fill -- correct for every tiling, a bandwidth hint only. */
int e2_pack_job_set_rows(void* rec, int rows);
int e2_conv3d_pack_multi(e2_ctx*, const void* jobs_dev, int njobs);
/* the same with the launch's LDS tile sized for the largest kd * kh * kw among the jobs */
int e2_conv3d_pack_multi_ex(e2_ctx*, const void* jobs_dev, int njobs, int max_taps);

/* ---- first layer, fused (Conv node on a 1-channel input: conv -> pool -> +b ->
 *      act in one pass; neural.py:662-712).  Supported: kd = 1, pool z = 1 and
 *      (kh,kw,py,px) in {(4,4,2,2), (6,6,2,2)} -- the neuro3d nets' first layers;
 *      e2_conv1_supported() tells.  The backward RECOMPUTES the conv values
 *      instead of reading a stored conv output; dw / dbias are accumulated
 *      (zero them first). ---------------------------------------------------- */
int e2_conv1_supported(int cin, int kd, int kh, int kw, int pz, int py, int px);
int e2_conv1_pool_act_fwd(e2_ctx*, const e2_tensor5* x, const float* w,
                          const float* bias, int cout, int kh, int kw, int py,
                          int px, int act, const e2_tensor5* out);
/* workspace of the backward: per-tile partial sums, (n, cout, d, ho, wo) = dims of
 * the POOLED output gradient */
size_t e2_conv1_bwd_workspace_bytes(int n, int cout, int d, int ho, int wo, int kh, int kw);
int e2_conv1_pool_act_bwd(e2_ctx*, const e2_tensor5* x, const float* w,
                          const float* bias, const e2_tensor5* dout, int kh,
                          int kw, int py, int px, int act, float* dw, float* dbias,
                          void* ws, size_t ws_bytes);

/* ---- split-K without atomics: partial sums added up by the consumer ------------------
 * (No reference counterpart: inside Theano's compiled function the conv, the pooling and the
 * bias / activation that follows are separate ops, computations.py:364-428, 538-631,
 * neural.py:705-712; this is how the conv hands its result to the next op here.)
 * A small layer fills the chip only when its K range is split over several work-groups.
 * The plain entry points then ACCUMULATE with float atomics into a zero-filled output
 * (~1.3 TB/s chip-wide, and a fill launch).  The _parts forms store split s with plain
 * stores to y.ptr + s * part_stride instead (the caller provides max_parts such slabs,
 * nothing needs to be zero) and report in *nparts how many were written (1: y holds the
 * complete result -- the tiling did not split K); the pointwise kernel that consumes the
 * tensor anyway adds the parts up on its way:
 *   e2_pool_bias_act_fwd_parts   y = sum of the parts (left in part 0), out = act(pool(y) + b)
 *   e2_pool_bias_act_bwd_parts / e2_bias_act_bwd_out_parts   dout = sum of the parts
 * (at most 8 parts: a consumer requests every part of an element before it adds them up)
 * Partial sums are added up for the pooling windows (1,1,1), (1,2,2), (2,1,1), (2,2,2). */
int e2_conv3d_fwd_packed_parts(e2_ctx*, const e2_tensor5* x, const void* wp, int cout, int kd,
                               int kh, int kw, const e2_tensor5* y, int64_t part_stride,
                               int max_parts, int* nparts);
int e2_conv3d_dgrad_packed_parts(e2_ctx*, const e2_tensor5* dy_pad, const void* wp, int cin,
                                 int kd, int kh, int kw, const e2_tensor5* dx,
                                 int64_t part_stride, int max_parts, int* nparts);
int e2_pool_bias_act_fwd_parts(e2_ctx*, const e2_tensor5* y, int64_t part_stride, int nparts,
                               const float* bias, int pz, int py, int px, int act,
                               const e2_tensor5* out);
int e2_pool_bias_act_bwd_parts(e2_ctx*, const e2_tensor5* dout, int64_t dout_part_stride,
                               int dout_parts, const e2_tensor5* y, const float* bias, int pz,
                               int py, int px, int act, const e2_tensor5* dy, float* dbias);
int e2_bias_act_bwd_out_parts(e2_ctx*, const e2_tensor5* dout, int64_t dout_part_stride,
                              int dout_parts, const e2_tensor5* out, int act,
                              const e2_tensor5* dy, float* dbias);

/* ---- pool + bias + activation  (computations.py:538-631 pooling();
 *      neural.py:705-712; computations.py:57-134 apply_activation) -------- */
/* out = act(maxpool(y, pool) + bias[c]) ; pool == stride, floor semantics. */
int e2_pool_bias_act_fwd(e2_ctx*, const e2_tensor5* y, const float* bias,
                         int pz, int py, int px, int act,
                         const e2_tensor5* out);
/* Given dout = dL/dout: dy = dL/dy (every element equal to its window max
 * receives the gradient -- Theano CPU tie rule; relu'(0) = 0.5) written to the
 * view dy (typically the interior of the zero-padded dgrad buffer), and
 * dbias[c] += sum(dL/dpre).  dbias must be zeroed by the caller. */
int e2_pool_bias_act_bwd(e2_ctx*, const e2_tensor5* dout, const e2_tensor5* y,
                         const float* bias, int pz, int py, int px, int act,
                         const e2_tensor5* dy, float* dbias);

/* stand-alone max-pool (neural.py:1520-1523 Pool node) */
int e2_maxpool3d_fwd(e2_ctx*, const e2_tensor5* x, int pz, int py, int px,
                     const e2_tensor5* out);
/* dx (+)= pool-backward(dout); accumulate != 0 adds into dx. */
int e2_maxpool3d_bwd(e2_ctx*, const e2_tensor5* dout, const e2_tensor5* x,
                     int pz, int py, int px, const e2_tensor5* dx,
                     int accumulate);

/* ---- UpConv  (neural.py:989-1072; computations.py:216-255 upconv(),
 *      749-782 unpooling_nd; F2: y[n,co,p*i+r] = sum_ci w[co,ci,r] x[n,ci,i]) */
/* (n,d,h,w) = dims of the UpConv INPUT x. */
size_t e2_upconv3d_workspace_bytes(int cout, int cin, int pz, int py, int px,
                                   int n, int d, int h, int w);
/* y = act(upconv(x, w) + bias); w [cout][cin][pz][py][px]; bias may be NULL
 * (then act must be LIN). */
int e2_upconv3d_fwd(e2_ctx*, const e2_tensor5* x, const float* w,
                    const float* bias, int cout, int pz, int py, int px,
                    int act, const e2_tensor5* y, void* ws, size_t ws_bytes);
/* dpre = dout * act'(y) is formed internally from the forward OUTPUT y
 * (relu: y > 0).  dx overwritten; dw overwritten; dbias overwritten.
 * Any of dx / dw / dbias may be NULL to skip it. */
int e2_upconv3d_bwd(e2_ctx*, const e2_tensor5* x, const float* w,
                    const e2_tensor5* y, const e2_tensor5* dout,
                    int pz, int py, int px, int act,
                    const e2_tensor5* dx, float* dw, float* dbias,
                    void* ws, size_t ws_bytes);
/* The same with the two packed weight images kept by the caller and refreshed once per step
 * by e2_conv3d_pack_multi (e2_pack_job_fill modes 2 = forward image, 3 = data-gradient image;
 * e2_upconv3d_image_bytes each) instead of a repack launch inside every call; ws of the
 * backward (e2_upconv3d_workspace_bytes) still holds the space-to-depth image.  accumulate
 * != 0: dw and dbias are ADDED to -- the caller cleared them (the training plan zeroes the
 * whole gradient arena with one launch), two fill launches less; dx is overwritten. */
size_t e2_upconv3d_image_bytes(int cout, int cin, int pz, int py, int px);
int e2_upconv3d_fwd_packed(e2_ctx*, const e2_tensor5* x, const float* wp_fwd,
                           const float* bias, int cout, int pz, int py, int px, int act,
                           const e2_tensor5* y);
int e2_upconv3d_bwd_packed(e2_ctx*, const e2_tensor5* x, const float* wp_dgrad,
                           const e2_tensor5* y, const e2_tensor5* dout, int pz, int py, int px,
                           int act, const e2_tensor5* dx, float* dw, float* dbias, void* ws,
                           size_t ws_bytes, int accumulate);

/* ---- layout / copies (computations.py:398-401,414-428 dimshuffles;
 *      neural.py:1152-1168 Crop; node_basic.py:1433-1440 Concat) ---------- */
/* dst[n][d][h][w][c] (dense NDHWC) <- src (NCDHW view) and back; staged
 * through LDS tiles so both sides are coalesced. */
int e2_transpose_ncdhw_to_ndhwc(e2_ctx*, const e2_tensor5* src, float* dst);
int e2_transpose_ndhwc_to_ncdhw(e2_ctx*, const float* src, const e2_tensor5* dst);
/* dst (+)= src, both arbitrary views of identical sizes. */
int e2_copy5(e2_ctx*, const e2_tensor5* src, const e2_tensor5* dst,
             int accumulate);
int e2_fill(e2_ctx*, float* ptr, size_t n, float value);

/* Batching the zero-fills of a captured step (no reference counterpart).  A split-K conv
 * launch (forward / data gradient with a small output) zero-fills its output and then
 * accumulates with atomics; each such fill is a ~5 us kernel.  A caller that replays the
 * same launches every step can (1) ask after a launch which flat region it zeroed
 * (e2_conv_last_zero_fill: *n = 0 if none), (2) zero all of them with ONE
 * e2_fill_multi(ptrs, counts: device arrays of nregions 64-bit entries) at the start of
 * the step, and (3) wrap the launches in e2_set_skip_zero_fill(ctx, 1) ... (ctx, 0):
 * "the output is already zero".  Nothing else may write the region in between. */
int e2_fill_multi(e2_ctx*, const void* ptrs_dev, const void* counts_dev, int nregions,
                  float value);
int e2_set_skip_zero_fill(e2_ctx*, int on);

/* Several steps in ONE captured graph (no reference counterpart: training/trainer.py:186-194 hands
 * one batch to model.trainingstep per iteration and reads the loss back before the next).  Between
 * two graph launches the device idles for ~19 us however short the host's part is (DESIGN finding
 * 54); a graph of k steps has that gap once per k steps.  What differs between the k copies of the
 * step -- the batch and the loss -- goes through rings in device memory indexed by a launch count.
 * e2_step_prologue is the first launch of such a step; with L = its launches on `state` so far:
 *   dst[0 .. slot_floats) = ring[L % n_slots]            (ring != NULL; the producer -- data/batch.py's
 *                            sampler, or copies from the host -- fills slots ahead of their step)
 *   hist[(L - 1) % hist_slots][0 .. n_vals) = src[..]     (hist != NULL, L > 0: what the step BEFORE
 *                            this one left in src -- its loss; the newest loss is still in src)
 * state: two zeroed 64-bit words in device memory owned by the caller ([0] = L; advanced by the
 * work-group that finishes last).  ring / dst 16-byte aligned, slot_floats a multiple of 4;
 * launches on one state must be ordered (one stream). */
int e2_step_prologue(e2_ctx*, const float* ring, int n_slots, size_t slot_floats, float* dst,
                     const float* src, int n_vals, float* hist, int hist_slots, void* state);
/* the conv launches that follow may read up to `bytes` (finite, readable) bytes behind the last
 * element of their input x (0 withdraws the promise): required (>= 128) by the weight-gradient
 * tiling "MT,NT,9,0,S", csrc/conv_pw_wgrad.hip */
int e2_set_input_slack(e2_ctx*, int bytes);
int e2_conv_last_zero_fill(const e2_ctx*, void** ptr, size_t* n);

/* ---- loss (computations.py:175-176 softmax; loss.py:261-347
 *      MultinoulliNLL(target_is_sparse); loss.py:1357-1363 AggregateLoss) - */
/* ---- classifier head, fused: 1x1x1 conv to ncls <= 4 'lin' features + channel softmax
 *      + MultinoulliNLL (sparse target) and their gradients (neural.py:662-712 with a
 *      (1,1,1) kernel, computations.py:175-176, loss.py:261-347).  w is the dense
 *      (ncls, cin, 1,1,1) weight tensor.  stats = {sum of -log(p_t + 1e-5), #labelled},
 *      zero it before e2_head_fwd; loss = stats[0] / (stats[1] + 1e-5).  target NULL:
 *      probabilities only.  e2_head_bwd ACCUMULATES into dw / dbias; its workspace holds
 *      per-work-group partial sums (e2_head_bwd_workspace_bytes). --------------------- */
int e2_head_supported(int cin, int ncls);
int e2_head_fwd(e2_ctx*, const e2_tensor5* x, const float* w, const float* bias, int ncls,
                const e2_tensor5* target, const e2_tensor5* probs, float* stats);
size_t e2_head_bwd_workspace_bytes(int n, int cin, int ncls, int d, int h, int w);
int e2_head_bwd(e2_ctx*, const e2_tensor5* x, const float* w, const e2_tensor5* probs,
                const e2_tensor5* target, const float* stats, const e2_tensor5* dx,
                int accumulate_dx, float* dw, float* dbias, float* loss_out, void* ws,
                size_t ws_bytes);

/* ---- the TAIL of the neuro3d nets, forward AND backward in one launch (csrc/tail.hip):
 *      x -> [1x1x1 conv to c2 channels + bias + relu] -> [classifier head as above], i.e. the
 *      last two Conv nodes of examples/neuro3d.py:61-63 / neuro3d_lite.py:57-59 under
 *      Softmax + MultinoulliNLL, and T.grad of that chain (model.py:182).  Replaces
 *      e2_conv3d_fwd_packed_act + e2_head_fwd + e2_head_bwd + e2_bias_act_bwd_out +
 *      e2_conv3d_dgrad_packed of the two layers: both convs have ONE tap, so a work-group runs
 *      the whole chain for its tile of positions out of LDS.
 *      x: (n, c1, d, h, w) with dense (z, y, x) planes, c1, c2 <= 208; wp_fwd / wp_dgrad: the
 *      1x1x1 layer's packed images (e2_conv3d_pack modes 0 / 1; wp_dgrad only with dx);
 *      w_head: dense (ncls, c2), ncls <= 4; target (n, 1, d, h, w) float class ids (< 0:
 *      unlabelled).  Writes probs (n, ncls, ...), dpre = d loss / d (pre-activation of the
 *      1x1x1 layer) (n, c2, ...) -- the operand of that layer's weight gradient --, dx
 *      (optional) = d loss / d x, stats[1] = #labelled, and one slot of partial sums per
 *      work-group to ws (e2_tail_workspace_bytes; *n_slots slots).
 *      e2_tail_reduce ADDS the slots into dw_head (ncls * c2), db_head (ncls), db1 (c2) --
 *      zero those first -- and writes stats[0] = sum of -log(p_t + 1e-5) and loss_out =
 *      stats[0] / (stats[1] + 1e-5).  f32 mode only. ------------------------------------ */
int e2_tail_supported(int c1, int c2, int ncls);
size_t e2_tail_workspace_bytes(int n, int c1, int c2, int ncls, int d, int h, int w);
/* gm_mode != 0: dx is written THROUGH the activation backward of the layer that produced x
 * (e2_bias_act_bwd_out / e2_pool_bias_act_bwd with a (1,1,1) window): dx *= act'(.), and that
 * layer's bias gradient (the row sums) joins the slots (e2_tail_reduce: c1_gm = c1, db_parent);
 * dx may then be the interior of that layer's zero-padded gradient buffer (any row / plane
 * pitch).  1: relu slope read off gm_src = its activated output (signed zeros, as
 * e2_conv3d_fwd_packed_act stores them), 2: off gm_src = its pre-activation + gm_bias[c],
 * 3: linear activation. */
int e2_tail_fwd_bwd(e2_ctx*, const e2_tensor5* x, const float* wp_fwd, const float* wp_dgrad,
                    const float* bias1, int c2, const float* w_head, const float* b_head,
                    int ncls, const e2_tensor5* target, const e2_tensor5* probs,
                    const e2_tensor5* dpre, const e2_tensor5* dx, int gm_mode,
                    const e2_tensor5* gm_src, const float* gm_bias, float* stats, void* ws,
                    size_t ws_bytes, int* n_slots);
int e2_tail_reduce(e2_ctx*, const void* ws, int n_slots, int c2, int ncls, float* dw_head,
                   float* db_head, float* db1, float* stats, float* loss_out, int c1_gm,
                   float* db_parent);

/* probs = softmax_c(logits); loss_sum += sum_pos -log(p[target]+1e-5);
 * n_lab += #labelled (target in [0,C)).  stats = {loss_sum, n_lab} must be
 * zeroed by the caller.  target: (n,1,d,h,w) float class ids. */
int e2_softmax_nll_fwd(e2_ctx*, const e2_tensor5* logits,
                       const e2_tensor5* target, const e2_tensor5* probs,
                       float* stats);
/* dlogits = d(loss)/d(logits) with loss = loss_sum/(n_lab+1e-5);
 * also writes loss_out[0] = loss (device scalar). */
int e2_softmax_nll_bwd(e2_ctx*, const e2_tensor5* probs,
                       const e2_tensor5* target, const float* stats,
                       const e2_tensor5* dlogits, float* loss_out);

/* MALIS NLL (neuromancer/loss.py:560-690 MalisNLL; malis/malisop.py:19-123): probs is
 * (1, 2E, d, h, w) -- E independent 2-class softmaxes, channel 2e = "disconnected",
 * 2e+1 = affinity of edge e; pos / neg: dense (E, d, h, w) float MALIS counts (from
 * e2_malis_loss_weights, constants for the gradient); norm[0] = 1/(n_tot + 1e-5) on
 * the device.  loss_sum[0] += -sum(pos*log(p_aff+1e-5) + neg*log(p_dis+1e-5))*norm[0]
 * (terms with a zero count contribute 0: xlogy0, loss.py:26-28); dlogits (nullable):
 * d(loss)/d(logits) through the pair softmax, same shape as probs. */
int e2_malis_nll(e2_ctx*, const e2_tensor5* probs, const float* pos, const float* neg,
                 const float* norm, const e2_tensor5* dlogits, float* loss_sum);

/* ---- optimiser (optimiser.py:273-334 Adam; 135-165 SGD) on one flat
 *      parameter arena.  wd_mult[i] = weight-decay multiplier per element
 *      segment is given by seg tables: for segment s, elements
 *      [seg_off[s], seg_off[s+1]) use decay multiplier seg_reg[s]. --------- */
/* hyper: 8 device floats {lr, mom, beta2, wd, t, factor, -, arrival counter}.  The
 * kernel reads t, steps with t + 1 (bias factor sqrt(1-beta2^t)/(1-mom^t)) and stores
 * the new t for the next call; hyper[7] must be zero before the first call; all four
 * arenas 16-byte aligned, n_seg <= 1024. */
int e2_adam_step(e2_ctx*, float* p, const float* g, float* m, float* s,
                 size_t n, const int64_t* seg_off, const float* seg_reg,
                 int n_seg, const float* hyper /* device: lr,mom,beta2,wd,t,.. */);
int e2_sgd_step(e2_ctx*, float* p, const float* g, float* d, size_t n,
                const int64_t* seg_off, const float* seg_reg, int n_seg,
                const float* hyper /* device: lr,mom,_,wd,_ */);

/* Data-parallel normalisation: the reference divides the NLL by the labelled voxels of the
 * WHOLE batch (loss.py:342-344).  With sum_mode != 0 the NLL backward launches that follow
 * (e2_tail_fwd_bwd, e2_head_bwd, e2_softmax_nll_bwd) leave the gradient UNNORMALISED -- loss
 * values are unaffected -- and e2_tail_reduce / e2_head_bwd / e2_softmax_nll_bwd also write this
 * rank's labelled count to count_out (a slot behind the gradient arena, so that it rides in
 * the same all-reduce); sum-all-reduce + e2_adam_step_ex(gdiv = the summed count) then IS the
 * whole-batch gradient, with no elementwise launch around the collective.  (0, NULL) restores
 * the per-rank normalisation. */
int e2_set_loss_grad_mode(e2_ctx*, int sum_mode, float* count_out);

/* the same with the gradient scaled on the way in and the arena cleared on the way out:
 * the update sees g * gmul / (gdiv ? gdiv[0] + 1e-5 : 1); zero_g != 0 leaves g ZERO for the
 * next backward pass (no fill launch).  gdiv: device scalar -- the data-parallel step sums
 * unnormalised gradients (and the labelled-voxel counts, in a slot behind the arena) over the
 * ranks and divides here, which is the reference's whole-batch normalisation
 * (loss.py:342-344) without elementwise launches around the collective. */
int e2_adam_step_ex(e2_ctx*, float* p, float* g, float* m, float* s, size_t n,
                    const int64_t* seg_off, const float* seg_reg, int n_seg,
                    const float* hyper, const float* gdiv, float gmul, int zero_g);
int e2_sgd_step_ex(e2_ctx*, float* p, float* g, float* d, size_t n, const int64_t* seg_off,
                   const float* seg_reg, int n_seg, const float* hyper, const float* gdiv,
                   float gmul, int zero_g);

/* The Adam update of optimiser.py:301-329 that also WRITES the packed weight images of every
 * conv (what e2_conv3d_pack_multi does at the head of a step): a work-group owns a tile of one
 * weight tensor -- 32 output channels x IC input channels x the taps of one kernel plane --,
 * updates it (arithmetic and order of e2_adam_step_ex: bit-identical p / m / s), and writes the
 * new values from LDS as the tile of the forward image and of the data-gradient image, each along
 * its contiguous axis, zeros in the fetched padding exactly where e2_conv3d_pack_multi writes them.
 * One launch instead of two, the weights read once (csrc/update_pack.hip).
 *   e2_upd_job_fill: one record per conv weight tensor w[cout][cin][kd][kh][kw] at element offset
 *     `off` of the arenas; wp_f / wp_d = its e2_conv3d_pack images of mode 0 / 1 (either may be
 *     NULL); reg = weight-decay multiplier (0 = none); tile0 = tiles of the records before it;
 *     *ntiles = this record's tiles, *lds_bytes = LDS a launch holding it needs.
 *   e2_upd_rest_fill: a run of `n` elements at `off` without images (biases, first layer, head,
 *     UpConv): the plain update.
 *   e2_adam_pack_step: jobs_dev / rest_dev = device arrays of those records; every trainable
 *     element must be covered exactly once by the two lists (caller's contract); lds_bytes = the
 *     largest *lds_bytes; hyper / gdiv / gmul / zero_g as e2_adam_step_ex, except that hyper
 *     holds 24 floats here ([8..23]: arrival counters, zero before the first launch). */
size_t e2_upd_job_bytes(void);
size_t e2_upd_rest_bytes(void);
int e2_upd_job_fill(void* rec, long off, void* wp_f, void* wp_d, int cout, int cin, int kd, int kh,
                    int kw, float reg, int tile0, int* ntiles, size_t* lds_bytes);
int e2_upd_rest_fill(void* rec, long off, long n, float reg);
int e2_adam_pack_step(e2_ctx*, float* p, float* g, float* m, float* s, const void* jobs_dev,
                      int njobs, int ntiles, const void* rest_dev, int nrest, float* hyper,
                      const float* gdiv, float gmul, int zero_g, size_t lds_bytes);

/* ---- step capture (replaces theano.function, graphutils.py:376-387) ---- */
typedef struct e2_graph e2_graph;
int e2_graph_begin(e2_ctx*);                 /* hipStreamBeginCapture      */
int e2_graph_end(e2_ctx*, e2_graph** out);   /* EndCapture + Instantiate   */
int e2_graph_launch(e2_ctx*, e2_graph*);     /* hipGraphLaunch on the stream*/
int e2_graph_destroy(e2_graph*);
/* the captured graph as a GraphViz file (hipGraphDebugDotPrint; verbose: with the kernels' names) --
 * how the fork / join structure of a plan with a side stream was looked at (DESIGN finding 54) */
int e2_graph_debug_dot(e2_graph*, const char* path, int verbose);

/* ---- timing helpers (HIP events on the context's stream) -------------- */
/* ---- patch extraction with warp + grey augmentation on the device (SURVEY 8f-1;
 *      data/transformations.py:337-492, :42-76; data/cnndata.py:42-60).
 *      dst[f][z][x][y] = interp(src[f], Minv . (z + dest_off[0], x + dest_off[1],
 *      y + dest_off[2], 1) - src_off): trilinear, or nearest (np.round) for the
 *      channels whose bit is set in nearest_mask.  src: (1, F <= 32, Z, X, Y) view of
 *      the resident volume; dst: dense (1, F, pz, px, py); Minv, dest_off (3 ints),
 *      src_off (3 floats) in HOST memory (NULL offsets = 0).  The caller checks the
 *      patch corners against the volume first (WarpingOOBError on the host side). */
int e2_warp_slice(e2_ctx*, const e2_tensor5* src, const float* minv, int perspective,
                  unsigned nearest_mask, const int* dest_off, const float* src_off,
                  const e2_tensor5* dst);
/* d = clip(d * alpha + c, 0, 1) ** gamma on one dense channel of n floats, in place */
int e2_grey_augment(e2_ctx*, float* d, size_t n, float alpha, float c, float gamma);

/* ---- MALIS (host code, no GPU involved; malis/_malis_lib.cpp:38-167 via
 *      malis/_malis.pyx:42-123).  e2_malis_loss_weights: counts[e] = number of voxel
 *      pairs whose maximin edge in the affinity graph is e and whose ground-truth ids are
 *      equal (pos != 0) or different (pos == 0); seg ids 0 = unlabelled; node indices out
 *      of [0, n_vert) mark absent edges.  e2_malis_connected_components: seg[v] = 1 +
 *      representative under edges with |weight| > 1e-5, components of <= size_thresh
 *      voxels -> 0.  Return 0 on success, 2 on bad arguments. ------------------------- */
int e2_malis_loss_weights(int n_vert, const int32_t* seg, int n_edge, const int32_t* node1,
                          const int32_t* node2, const float* edge_weight, int pos,
                          uint64_t* counts);
int e2_malis_connected_components(int n_vert, int n_edge, const int32_t* node1,
                                  const int32_t* node2, const float* edge_weight,
                                  int size_thresh, int32_t* seg);

typedef struct e2_event e2_event;
int e2_event_create(e2_event** out);
int e2_event_record(e2_ctx*, e2_event*);
int e2_event_elapsed_ms(e2_event* start, e2_event* stop, float* ms); /* syncs stop */
int e2_event_destroy(e2_event*);
int e2_stream_synchronize(e2_ctx*);
/* Second stream for independent work (wgrad of a layer next to the dgrad chain):
 * fork = `side` waits for everything issued on the context's stream so far;
 * join = the context's stream waits for everything issued on `side`.  Both work
 * while the context's stream is being captured (the side stream becomes a parallel
 * branch of the graph; join it before e2_graph_end).  Launch on the side stream by
 * pointing the context at it with e2_ctx_set_stream and back. */
int e2_stream_fork(e2_ctx*, void* side_stream);
int e2_stream_join(e2_ctx*, void* side_stream);

/* ---- conv forward / data gradient with bf16 operands in memory ----------
 * Same arithmetic as E2_MFMA_BF16 (operands rounded to bf16, nearest even; f32 products and
 * sums; f32 tensors in and out) on a kernel built for v_mfma_f32_32x32x16_bf16: the call
 * converts its input to bf16 planes of 16-byte pixels (8 channels each) and the canonical
 * weights w (n_f, n_in, kd, kh, kw) to bf16 filter rows in the caller's workspace
 * (e2_conv3d_bf16_workspace_bytes for the layer: covers both directions); the GEMM stages a
 * work-group's input window in LDS once per kernel plane (LDS-DMA) and reads every tap from
 * it at a shifted pixel, filter rows come from L2.  fwd: optional fused bias + activation
 * (bias NULL: plain conv), out any strided view.  dgrad: dy_pad zero-padded as for
 * e2_conv3d_dgrad.  All reduction channels of one kernel plane must fit LDS (an error says
 * so; the operand-rounding form of E2_MFMA_BF16 has no such limit).  Tiling:
 * e2_set_tiling(E2_TILING_IGEMM, "32,MB,NB") = MB x NB blocks of 32 channels x 32 positions
 * per wave (default 2 x 2).  No reference counterpart (the reference is f32). */
size_t e2_conv3d_bf16_workspace_bytes(int n, int cin, int d, int h, int w, int cout, int kd,
                                      int kh, int kw);
int e2_conv3d_fwd_bf16(e2_ctx*, const e2_tensor5* x, const float* w, int cout, int kd, int kh,
                       int kw, const float* bias, int act, const e2_tensor5* out, void* ws,
                       size_t ws_bytes);
int e2_conv3d_dgrad_bf16(e2_ctx*, const e2_tensor5* dy_pad, const float* w, int cin, int kd,
                         int kh, int kw, const e2_tensor5* dx, void* ws, size_t ws_bytes);
/* The weight gradient in the same arithmetic (operands rounded to bf16, f32 sums), K = the
 * positions: dy (the UNPADDED gradient view, any strides) is converted to bf16 channel-major
 * planes at the input's row pitch, x to bf16 channels-last pixels, both staged in LDS per
 * 64 positions; dy rows are read with ds_read_b128, the shifted input windows with the
 * transposed LDS read ds_read_b64_tr_b16; the tile is flushed into dw with f32 atomics
 * (accumulate 0: dw is zeroed first).  Tiling: e2_set_tiling(E2_TILING_WGRAD,
 * "32,MB,NB,R,S"): MB (1..2) blocks of 32 out channels x NB (1..4) taps of a kernel row per
 * wave; R = 0: the work-group spans 128 input channels of one kernel row, R = 1: 32 input
 * channels of four kernel rows; S position splits (0 = one work-group per CU). */
size_t e2_conv3d_wgrad_bf16_workspace_bytes(int n, int cin, int d, int h, int w, int cout,
                                            int kd, int kh, int kw);
int e2_conv3d_wgrad_bf16(e2_ctx*, const e2_tensor5* x, const e2_tensor5* dy, float* dw, int kd,
                         int kh, int kw, int accumulate, void* ws, size_t ws_bytes);
/* One conversion less per layer and step: e2_conv3d_fwd_bf16_keep writes the bf16
 * channels-last copy of its input into a buffer of the caller's (e2_conv3d_bf16_xkeep_bytes,
 * zero-filled ONCE by the caller: the pixels behind the last plane stay zero) instead of the
 * shared workspace, and e2_conv3d_wgrad_bf16_xcl reads that copy (kg_per_plane = channel
 * groups of 8 per plane in it = ceil(cin / 16) * 2; x_shape: the input's extents, its
 * pointer and strides are not used) instead of converting x again. */
size_t e2_conv3d_bf16_xkeep_bytes(int n, int cin, int d, int h, int w, int kh, int kw);
int e2_conv3d_fwd_bf16_keep(e2_ctx*, const e2_tensor5* x, const float* w, int cout, int kd,
                            int kh, int kw, const float* bias, int act, const e2_tensor5* out,
                            void* ws, size_t ws_bytes, void* xkeep, size_t xkeep_bytes);
int e2_conv3d_wgrad_bf16_xcl(e2_ctx*, const e2_tensor5* x_shape, const void* xcl,
                             int kg_per_plane, const e2_tensor5* dy, float* dw, int kd, int kh,
                             int kw, int accumulate, void* ws, size_t ws_bytes);

/* ---- the producers' epilogues ---------------------------------------------
 * No conversion pass at all: the bf16 operand images of a GEMM are written by the kernel that
 * produces the tensor, the filter rows of every layer by ONE launch at the start of the step,
 * and the GEMM entry points take them ready-made.  Reference seam unchanged: the tensors are
 * those of neural.py:662-712 (conv -> pool -> bias -> activation) and of T.grad of that chain
 * (model.py:182); only WHERE their bf16 copies are made moves.
 *
 * e2_bf16_dst: where a producer puts the bf16 copies of the tensor it writes.
 *   cl  channels-last image [n][cl_d][cl_kg][cl_h][cl_w][8] (16-byte pixel pieces of 8 channels,
 *       cl_kg * 8 >= channels; the layout conv_bf16 / wgrad_bf16 read); element (z, y, x) of the
 *       tensor lands at (z + cl_oz, y + cl_oy, x + cl_ox) -- the interior of a zero-padded
 *       gradient image.  Everything the producer does not write (border, padding channel
 *       groups, slack) must be zero from a one-time fill.
 *   pl  channel-major planes [n][c][d][pl_plane] with row pitch pl_pitch (the dy operand of
 *       e2_conv3d_wgrad_bf16_ex: pl_pitch = the layer INPUT's row length, pl_plane from
 *       e2_conv3d_wgrad_bf16_geometry); gaps stay zero from a one-time fill.
 * Either may be NULL. */
typedef struct e2_bf16_dst {
  void* cl;
  int cl_kg, cl_d, cl_h, cl_w;
  int cl_oz, cl_oy, cl_ox;
  void* pl;
  int64_t pl_plane;
  int pl_pitch;
} e2_bf16_dst;
/* e2_pool_bias_act_bwd (bias != NULL: src = the conv output y, slope from max + bias) and
 * e2_bias_act_bwd_out (bias == NULL, window (1,1,1): src = the activated output, slope from its
 * signed zeros) with dout as `parts` partial sums part_stride apart; writes dy (f32; dy == NULL
 * or dy->ptr == NULL: not written), dbias += row sums (NULL: skipped) and the bf16 copies named
 * by dst.  relu / lin; windows (1,1,1), (1,2,2), (2,1,1), (2,2,2). */
int e2_pool_bias_act_bwd_bf16(e2_ctx*, const e2_tensor5* dout, int64_t part_stride, int parts,
                              const e2_tensor5* src, const float* bias, int pz, int py, int px,
                              int act, const e2_tensor5* dy, float* dbias, const e2_bf16_dst* dst);
/* e2_pool_bias_act_fwd_parts writing out (f32; out->ptr == NULL: not written) and the
 * channels-last bf16 copy of out (dst->cl: the NEXT conv layer's kept input copy). */
int e2_pool_bias_act_fwd_bf16(e2_ctx*, const e2_tensor5* y, int64_t part_stride, int parts,
                              const float* bias, int pz, int py, int px, int act,
                              const e2_tensor5* out, const e2_bf16_dst* dst);
/* e2_conv1_pool_act_fwd writing also the channels-last bf16 copy of out (cout <= 32) */
int e2_conv1_pool_act_fwd_bf16(e2_ctx*, const e2_tensor5* x, const float* w, const float* bias,
                               int cout, int kh, int kw, int py, int px, int act,
                               const e2_tensor5* out, void* next_xb, int next_kg);
/* The filter rows of a launch, packed ahead of it for the tile "32,mb,nb" (mode 0: forward
 * image of w (nf, nin, kd, kh, kw), rows = nf; mode 1: data-gradient image, rows = nin); in_w /
 * out_w: the row lengths of the GEMM's input and output (forward: x and y; data gradient: the
 * padded dy and dx).  Jobs are plain records (e2_bf16_wjob_bytes each) filled on the host,
 * copied to the device by the caller and run by ONE launch. */
size_t e2_conv3d_bf16_wb_bytes(int rows, int kk, int kd, int kh, int kw, int in_w, int out_w,
                               int mb, int nb);
size_t e2_bf16_wjob_bytes(void);
int e2_bf16_wjob_fill(void* rec, const float* w, int nf, int nin, int kd, int kh, int kw, int mode,
                      int in_w, int out_w, int mb, int nb, void* wb, size_t wb_bytes);
int e2_conv3d_bf16_pack_w_multi(e2_ctx*, const void* jobs_dev, int njobs);
/* e2_conv3d_fwd_bf16_keep / e2_conv3d_dgrad_bf16 with ready-made operands: x_ready != 0: xkeep
 * already holds this step's copy of x (x->ptr may be NULL); wb != NULL: the packed filter rows
 * (w may be NULL; the forced tile must be the one they were packed for); next_xb != NULL
 * (forward): the channels-last bf16 copy of out, [n][d][next_kg][h * w][8] with next_kg =
 * ceil(cout / 16) * 2, is written by the epilogue (padding channels as zeros).  dy_cl: the
 * channels-last image of the padded gradient (dy_pad gives the shape).  With every operand
 * ready ws may be NULL. */
int e2_conv3d_fwd_bf16_ex(e2_ctx*, const e2_tensor5* x, const float* w, int cout, int kd, int kh,
                          int kw, const float* bias, int act, const e2_tensor5* out, void* ws,
                          size_t ws_bytes, void* xkeep, size_t xkeep_bytes, int x_ready,
                          const void* wb, void* next_xb, int next_kg);
int e2_conv3d_dgrad_bf16_ex(e2_ctx*, const e2_tensor5* dy_pad, const float* w, int cin, int kd,
                            int kh, int kw, const e2_tensor5* dx, void* ws, size_t ws_bytes,
                            const void* dy_cl, const void* wb);
/* e2_conv3d_wgrad_bf16_xcl with dy as ready-made planes (dyc: e2_bf16_dst.pl of the kernel that
 * produced dy; dy gives the shape) and the f32 sums in a buffer of the caller's (sums:
 * sums_bytes of e2_conv3d_wgrad_bf16_geometry, zero-filled ONCE -- the call leaves it zero);
 * xcl, dyc, sums may each be NULL (then made / held in ws as before). */
int e2_conv3d_wgrad_bf16_geometry(int n, int cin, int d, int h, int w, int cout, int kd, int kh,
                                  int kw, int64_t* plane_d, size_t* dyc_bytes, size_t* sums_bytes);
int e2_conv3d_wgrad_bf16_ex(e2_ctx*, const e2_tensor5* x_shape, const void* xcl, int kg_per_plane,
                            const e2_tensor5* dy, const void* dyc, float* sums, float* dw, int kd,
                            int kh, int kw, int accumulate, void* ws, size_t ws_bytes);

/* ---- BASELINE config 1 (examples/mnist.py:29-56): Perceptron and batch normalisation ----
 * Correctness-first kernels for the reference's CPU-runnable plumbing case.
 *
 * Perceptron (neural.py:258-410, computations.py:179-213 `dot`): plain row-major matrices,
 *   y (n x m) = x (n x k) . w (k x m)              w has the reference's (n_in, n_f) layout
 *   dx (n x k) (+)= dy (n x m) . w^T               accumulate: 0 overwrite, 1 add
 *   dw (k x m) (+)= x^T . dy
 * bias / activation of a Perceptron without batch norm: e2_pool_bias_act_fwd / _bwd with a
 * (1,1,1) window on the (n, m, 1, 1, 1) view. */
int e2_dense_fwd(e2_ctx*, const float* x, const float* w, float* y, int n, int k, int m);
int e2_dense_dgrad(e2_ctx*, const float* dy, const float* w, float* dx, int n, int k, int m,
                   int accumulate);
int e2_dense_wgrad(e2_ctx*, const float* x, const float* dy, float* dw, int n, int k, int m,
                   int accumulate);
/* Batch normalisation + bias + activation (neural.py:352-378 Perceptron, 681-711 Conv; applied
 * to the pooled conv output / the dot product):
 *   train != 0: mean, std = statistics of x over every axis but the channel (population std,
 *               + 1e-6); update_running != 0 additionally performs the training function's
 *               extra updates  run_mean <- 0.9995 run_mean + 0.0005 mean  (same for run_std)
 *   train == 0: mean, std = run_mean, run_std ('predict' mode)
 *   out = act((gamma / std) * x + bias - gamma * mean / std)
 * save (2 * c floats, may be NULL in forward-only use) receives the mean and std used.
 * Backward: dx (may be NULL), dgamma += , dbias += (either may be NULL); in train mode the
 * gradient flows through the batch statistics (as T.grad does), in predict mode they are
 * constants.  relu'(0) = 0.5. */
int e2_batchnorm_act_fwd(e2_ctx*, const e2_tensor5* x, const float* gamma, const float* bias,
                         float* run_mean, float* run_std, int train, int update_running,
                         int act, const e2_tensor5* out, float* save);
int e2_batchnorm_act_bwd(e2_ctx*, const e2_tensor5* dout, const e2_tensor5* x,
                         const float* gamma, const float* bias, const float* save, int train,
                         int act, const e2_tensor5* dx, float* dgamma, float* dbias);

#ifdef __cplusplus
}
#endif
#endif /* E2HIP_H */
