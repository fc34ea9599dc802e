// api.hip -- extern "C" entry points of libe2hip.so (see include/e2hip.h).
#include "common.hpp"
#include <stdarg.h>
#include <string.h>

static thread_local char g_err[1024] = "";

void e2_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* e2_last_error(void) { return g_err; }
extern "C" int e2_version(void) { return 1; }

/* 0 = f32 operands (exact f32 MFMA, the default), 1 = operands of the conv GEMMs rounded
 * to bf16 on their way into the matrix core, f32 accumulation */
extern "C" int e2_set_mfma_dtype(e2_ctx* ctx, int dtype) {
  E2_REQUIRE(ctx, "e2_set_mfma_dtype: null context");
  E2_REQUIRE(dtype == E2_MFMA_F32 || dtype == E2_MFMA_BF16, "e2_set_mfma_dtype: unknown dtype %d", dtype);
  E2_REQUIRE(!ctx->capturing, "e2_set_mfma_dtype: not while capturing a graph");
  ctx->mfma_bf16 = (dtype == E2_MFMA_BF16) ? 1 : 0;
  return 0;
}
extern "C" int e2_get_mfma_dtype(const e2_ctx* ctx) {
  return (ctx && ctx->mfma_bf16) ? E2_MFMA_BF16 : E2_MFMA_F32;
}

extern "C" int e2_ctx_create(int device, e2_ctx** out) {
  E2_REQUIRE(out, "e2_ctx_create: null out");
  int ndev = 0;
  E2_CHECK_HIP(hipGetDeviceCount(&ndev));
  E2_REQUIRE(device >= 0 && device < ndev, "e2_ctx_create: device %d of %d", device, ndev);
  E2_CHECK_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  E2_CHECK_HIP(hipGetDeviceProperties(&prop, device));
  E2_REQUIRE(strncmp(prop.gcnArchName, "gfx950", 6) == 0,
             "e2_ctx_create: device %d is %s; this library is built for gfx950 only",
             device, prop.gcnArchName);
  e2_ctx* c = new e2_ctx;
  c->device = device;
  c->stream = nullptr;
  c->num_cu = prop.multiProcessorCount;
  c->capturing = false;
  c->zeros = nullptr;
  c->fork_next = 0;
  c->mfma_bf16 = 0;
  c->skip_zero_fill = 0;
  c->loss_sum_mode = 0;
  c->loss_count_out = nullptr;
  c->input_slack = 0;
  c->image_rows = 0;
  c->last_fill_ptr = nullptr;
  c->last_fill_n = 0;
  c->tiling[0][0] = c->tiling[1][0] = 0;
  c->last_launch[0] = 0;
  c->tiling_fallbacks = 0;
  for (int i = 0; i < 32; ++i) c->fork_ev[i] = nullptr;
  if (hipMalloc(&c->zeros, 1024) != hipSuccess || hipMemset(c->zeros, 0, 1024) != hipSuccess) {
    delete c;
    e2_set_error("e2_ctx_create: cannot allocate the zero page");
    return 1;
  }
  *out = c;
  return 0;
}

/* Data-parallel normalisation (loss.py:342-344 divides by the labelled voxels of the WHOLE
 * batch): with sum_mode != 0 the NLL backward launches that follow (e2_tail_fwd_bwd, e2_head_bwd,
 * e2_softmax_nll_bwd) leave the gradient UNNORMALISED (loss values are unaffected) and write
 * this rank's labelled count to count_out (optional: e2_tail_reduce / e2_head_bwd /
 * e2_softmax_nll_bwd), so that sum-all-reduce + e2_adam_step_ex(gdiv = summed count) is the
 * whole-batch gradient.  (0, NULL) restores the per-rank normalisation. */
extern "C" int e2_set_loss_grad_mode(e2_ctx* ctx, int sum_mode, float* count_out) {
  E2_REQUIRE(ctx, "set_loss_grad_mode: null context");
  ctx->loss_sum_mode = sum_mode ? 1 : 0;
  ctx->loss_count_out = count_out;
  return 0;
}

/* The caller vouches that the input tensor x of the conv launches that follow is followed by at
 * least `bytes` readable bytes holding FINITE values (its own allocation's zeroed slack, or more
 * of the buffer a view was cut from): the weight gradient "MT,NT,9,0,S" lets the last 32-position
 * unit of a plane run past the plane's end -- the gradient's zero border times whatever x holds
 * there -- and is offered only with >= 128.  0 (the default) withdraws the promise. */
extern "C" int e2_set_input_slack(e2_ctx* ctx, int bytes) {
  E2_REQUIRE(ctx && bytes >= 0, "e2_set_input_slack: bad argument");
  ctx->input_slack = bytes;
  return 0;
}

/* Row length of the packed weight images (no reference counterpart).  An image holds, per
 * (plane, channel group, tap, channel-in-group), one row of output channels; by default a row is
 * `cout` rounded to 16 plus 208 floats, so that ANY M tiling of the GEMM kernels fits -- 416 floats
 * for 200 channels, of which the 7 x 2 tiles fetch 224.  A caller that knows the tiling of the
 * launch that reads an image may pack it with SHORTER rows (e2_pack_job_set_stride, or this
 * setting around e2_conv3d_pack) and must then announce the same length here around every
 * e2_conv3d_{fwd,dgrad}_packed* launch that reads it: the repack writes a third less and the
 * GEMM's weight rows lie closer together (neuro3d@185 -1 %, the U-Nets -0.4 ... -0.7 %, DESIGN
 * finding 52).  rows = 0 returns to the formula.  A tiling whose tiles reach past the announced
 * row length is an ERROR of the launch ("packed coP too small"), never a silent over-read.
 * UpConv images and the images of e2_tail_fwd_bwd always use the formula. */
extern "C" int e2_set_image_rows(e2_ctx* ctx, int rows) {
  E2_REQUIRE(ctx && rows >= 0 && rows % 4 == 0, "e2_set_image_rows: rows must be a non-negative multiple of 4");
  ctx->image_rows = rows;
  return 0;
}

// dims of the packed image a conv launch reads: the formula, or the announced row length
static int image_dims(const e2_ctx* ctx, int cout, int cin, int* ciP, int* coP) {
  e2i_pack_dims(cout, cin, ciP, coP);
  if (ctx->image_rows > 0) {
    E2_REQUIRE(ctx->image_rows >= ((cout + 15) / 16) * 16, "packed image: announced rows of %d floats for %d channels",
               ctx->image_rows, cout);
    *coP = ctx->image_rows;
  }
  return 0;
}

extern "C" int e2_set_tiling(e2_ctx* ctx, int kind, const char* cfg) {
  E2_REQUIRE(ctx, "e2_set_tiling: null context");
  E2_REQUIRE(kind == E2_TILING_IGEMM || kind == E2_TILING_WGRAD, "e2_set_tiling: unknown kind %d", kind);
  if (!cfg) cfg = "";
  E2_REQUIRE(strlen(cfg) < sizeof(ctx->tiling[0]), "e2_set_tiling: configuration string too long");
  if (cfg[0]) {
    int v[5], n = sscanf(cfg, "%d,%d,%d,%d,%d", &v[0], &v[1], &v[2], &v[3], &v[4]);
    int v5 = 0, v6 = 0, v7 = 0;
    if (n == 5 && kind == E2_TILING_IGEMM) n += sscanf(cfg, "%*d,%*d,%*d,%*d,%*d,%d,%d,%d", &v5, &v6, &v7);
    E2_REQUIRE(kind == E2_TILING_IGEMM ? (n == 4 || (n == 3 && (v[0] == 32 || v[0] == 1)) || (n == 8 && v[0] == 4)) : n == 5,
               "e2_set_tiling: '%s' is not %s", cfg,
               kind == E2_TILING_IGEMM ? "\"MT,NT,CC,SK\", \"4,MG,NT,CC,SK,WM,WN,G\", \"1,MT,NT\" or \"32,MB,NB\"" : "\"MT,NT,WK,BP,PS\"");
  }
  strcpy(ctx->tiling[kind], cfg);
  return 0;
}

void e2_note_launch(e2_ctx* ctx, const char* family, int src, const char* fmt, ...) {
  char til[96];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(til, sizeof(til), fmt, ap);
  va_end(ap);
  snprintf(ctx->last_launch, sizeof(ctx->last_launch), "%s %s %s", family, til,
           src == E2_SRC_FORCED ? "forced" : (src == E2_SRC_FALLBACK ? "fallback" : "model"));
  if (src == E2_SRC_FALLBACK) ++ctx->tiling_fallbacks;
}

/* Which kernel the last conv GEMM launch of this context ran (forward / data gradient / UpConv /
 * weight gradient, f32 and bf16 forms): "<kernel family> <tiling> <source>", source = "forced"
 * (the e2_set_tiling string was honoured), "model" (none was set: the cost model chose) or
 * "fallback" (a string was set but the launch could not run it and took the cost model's
 * choice: only the weight-gradient forms 7 / 8 / 9 may, see e2_set_tiling).  Host-side
 * bookkeeping only: legal during capture; empty before the first launch. */
extern "C" int e2_last_launch(e2_ctx* ctx, char* buf, int n) {
  E2_REQUIRE(ctx && buf && n > 0, "e2_last_launch: bad argument");
  snprintf(buf, (size_t)n, "%s", ctx->last_launch);
  return 0;
}

/* launches since e2_ctx_create whose forced tiling was not the one that ran */
extern "C" unsigned e2_tiling_fallbacks(const e2_ctx* ctx) { return ctx ? ctx->tiling_fallbacks : 0; }

extern "C" int e2_ctx_destroy(e2_ctx* ctx) {
  if (ctx && ctx->zeros) (void)hipFree(ctx->zeros);
  if (ctx)
    for (int i = 0; i < 32; ++i)
      if (ctx->fork_ev[i]) (void)hipEventDestroy(ctx->fork_ev[i]);
  delete ctx;
  return 0;
}

// Fork / join a second stream (also while the context's stream is being captured:
// the side stream joins the capture, the launches on it become a parallel branch of
// the graph).  Independent kernels -- the weight gradient of a layer and the data
// gradient that feeds the layer below -- then share the chip: the work-groups of one
// fill the CUs the other leaves idle (tails, 220-work-group grids on 256 CUs).
static int fork_event(e2_ctx* ctx, hipEvent_t* ev) {
  const int i = ctx->fork_next;
  ctx->fork_next = (i + 1) & 31;
  if (!ctx->fork_ev[i]) E2_CHECK_HIP(hipEventCreateWithFlags(&ctx->fork_ev[i], hipEventDisableTiming));
  *ev = ctx->fork_ev[i];
  return 0;
}
extern "C" int e2_stream_fork(e2_ctx* ctx, void* side) {
  E2_REQUIRE(ctx && side && ctx->stream, "stream_fork: null argument / default stream");
  hipEvent_t ev;
  if (int rc = fork_event(ctx, &ev)) return rc;
  E2_CHECK_HIP(hipEventRecord(ev, ctx->stream));
  E2_CHECK_HIP(hipStreamWaitEvent(reinterpret_cast<hipStream_t>(side), ev, 0));
  return 0;
}
extern "C" int e2_stream_join(e2_ctx* ctx, void* side) {
  E2_REQUIRE(ctx && side && ctx->stream, "stream_join: null argument / default stream");
  hipEvent_t ev;
  if (int rc = fork_event(ctx, &ev)) return rc;
  E2_CHECK_HIP(hipEventRecord(ev, reinterpret_cast<hipStream_t>(side)));
  E2_CHECK_HIP(hipStreamWaitEvent(ctx->stream, ev, 0));
  return 0;
}

extern "C" int e2_ctx_set_stream(e2_ctx* ctx, void* stream) {
  E2_REQUIRE(ctx, "e2_ctx_set_stream: null ctx");
  ctx->stream = reinterpret_cast<hipStream_t>(stream);
  return 0;
}

// ---------------------------------------------------------------------------
// conv
// ---------------------------------------------------------------------------
static int view_ok(const e2_tensor5* t, const char* name) {
  E2_REQUIRE(t && t->ptr, "%s: null tensor", name);
  E2_REQUIRE(t->n > 0 && t->c > 0 && t->d > 0 && t->h > 0 && t->w > 0,
             "%s: empty tensor (%d,%d,%d,%d,%d)", name, t->n, t->c, t->d, t->h, t->w);
  return 0;
}

extern "C" size_t e2_conv3d_workspace_bytes(int cout, int cin, int kd, int kh, int kw) {
  // one packed image, big enough for either orientation (fwd or dgrad)
  int ciP, coP, ciP2, coP2;
  e2i_pack_dims(cout, cin, &ciP, &coP);
  e2i_pack_dims(cin, cout, &ciP2, &coP2);
  const size_t a = (size_t)ciP * coP, b = (size_t)ciP2 * coP2;
  return sizeof(float) * (size_t)kd * kh * kw * (a > b ? a : b);
}

// mode 0: forward image (flip, oc = cout, ic = cin); mode 1: dgrad image
// (no flip, oc = cin, ic = cout).
extern "C" int e2_conv3d_pack(e2_ctx* ctx, const float* w, int cout, int cin, int kd,
                              int kh, int kw, int mode, void* ws, size_t ws_bytes) {
  E2_REQUIRE(ctx && w && ws, "conv3d_pack: null argument");
  E2_REQUIRE(ws_bytes >= e2_conv3d_workspace_bytes(cout, cin, kd, kh, kw),
             "conv3d_pack: workspace too small");
  const int T = kd * kh * kw;
  int ciP, coP;
  if (mode == 0) {
    if (int rc = image_dims(ctx, cout, cin, &ciP, &coP)) return rc;
    return e2i_pack_weights(ctx, w, (float*)ws, cout, cin, kd, kh, kw, (int64_t)cin * T, T,
                            1, ciP, coP, 1, 1);
  }
  if (int rc = image_dims(ctx, cin, cout, &ciP, &coP)) return rc;
  return e2i_pack_weights(ctx, w, (float*)ws, cin, cout, kd, kh, kw, T, (int64_t)cin * T, 0,
                          ciP, coP, 1, 1);
}

static int conv_fwd_packed(e2_ctx* ctx, const e2_tensor5* x, const void* wp, int cout, int kd,
                           int kh, int kw, const e2_tensor5* y, int64_t part_stride,
                           int max_parts, int* nparts) {
  E2_REQUIRE(ctx && wp, "conv3d_fwd: null argument");
  if (int rc = view_ok(x, "conv3d_fwd x")) return rc;
  if (int rc = view_ok(y, "conv3d_fwd y")) return rc;
  E2_REQUIRE(kd >= 1 && kh >= 1 && kw >= 1, "conv3d_fwd: bad kernel %d,%d,%d", kd, kh, kw);
  E2_REQUIRE(y->n == x->n && y->c == cout && y->d == x->d - kd + 1 &&
                 y->h == x->h - kh + 1 && y->w == x->w - kw + 1,
             "conv3d_fwd: y is (%d,%d,%d,%d,%d), expected (%d,%d,%d,%d,%d)", y->n, y->c,
             y->d, y->h, y->w, x->n, cout, x->d - kd + 1, x->h - kh + 1, x->w - kw + 1);
  IgemmArgs a;
  a.in = x->ptr; a.wp = (const float*)wp; a.out = y->ptr;
  a.N = x->n; a.Cin = x->c; a.Cout = cout;
  a.kd = kd; a.kh = kh; a.kw = kw;
  a.Do = y->d; a.Ho = y->h; a.Wo = y->w;
  a.isN = x->sn; a.isC = x->sc; a.isZ = x->sd; a.isY = x->sh;
  a.osN = y->sn; a.osC = y->sc; a.osZ = y->sd; a.osY = y->sh;
  if (int rc = image_dims(ctx, cout, x->c, &a.ciP, &a.coP)) return rc;
  a.upz = a.upy = a.upx = 1;
  a.parts_max = max_parts; a.part_stride = part_stride; a.nparts = nparts;
  return e2i_igemm_conv(ctx, a);
}
extern "C" int e2_conv3d_fwd_packed(e2_ctx* ctx, const e2_tensor5* x, const void* wp,
                                    int cout, int kd, int kh, int kw, const e2_tensor5* y) {
  return conv_fwd_packed(ctx, x, wp, cout, kd, kh, kw, y, 0, 0, nullptr);
}
extern "C" int e2_conv3d_fwd_packed_parts(e2_ctx* ctx, const e2_tensor5* x, const void* wp,
                                          int cout, int kd, int kh, int kw, const e2_tensor5* y,
                                          int64_t part_stride, int max_parts, int* nparts) {
  E2_REQUIRE(nparts && max_parts >= 1 && (max_parts == 1 || part_stride > 0),
             "conv3d_fwd_parts: bad parts arguments");
  return conv_fwd_packed(ctx, x, wp, cout, kd, kh, kw, y, part_stride, max_parts, nparts);
}

extern "C" int e2_conv3d_fwd_packed_act(e2_ctx* ctx, const e2_tensor5* x, const void* wp,
                                        int cout, int kd, int kh, int kw, const float* bias,
                                        int act, const e2_tensor5* out) {
  E2_REQUIRE(ctx && wp && bias, "conv3d_fwd_act: null argument");
  E2_REQUIRE(act == E2_ACT_LIN || act == E2_ACT_RELU, "conv3d_fwd_act: bad act %d", act);
  if (int rc = view_ok(x, "conv3d_fwd_act x")) return rc;
  if (int rc = view_ok(out, "conv3d_fwd_act out")) return rc;
  E2_REQUIRE(kd >= 1 && kh >= 1 && kw >= 1, "conv3d_fwd_act: bad kernel %d,%d,%d", kd, kh, kw);
  E2_REQUIRE(out->n == x->n && out->c == cout && out->d == x->d - kd + 1 &&
                 out->h == x->h - kh + 1 && out->w == x->w - kw + 1,
             "conv3d_fwd_act: out is (%d,%d,%d,%d,%d), expected (%d,%d,%d,%d,%d)", out->n,
             out->c, out->d, out->h, out->w, x->n, cout, x->d - kd + 1, x->h - kh + 1,
             x->w - kw + 1);
  E2_REQUIRE(out->sh == out->w && (kw == 1 || kw == 3 || kw == 4 || kw == 5),
             "conv3d_fwd_act: needs dense output rows and a kernel width of 1, 3, 4 or 5");
  IgemmArgs a;
  a.in = x->ptr; a.wp = (const float*)wp; a.out = out->ptr;
  a.N = x->n; a.Cin = x->c; a.Cout = cout;
  a.kd = kd; a.kh = kh; a.kw = kw;
  a.Do = out->d; a.Ho = out->h; a.Wo = out->w;
  a.isN = x->sn; a.isC = x->sc; a.isZ = x->sd; a.isY = x->sh;
  a.osN = out->sn; a.osC = out->sc; a.osZ = out->sd; a.osY = out->sh;
  if (int rc = image_dims(ctx, cout, x->c, &a.ciP, &a.coP)) return rc;
  a.upz = a.upy = a.upx = 1;
  a.bias = bias; a.act = act;
  return e2i_igemm_conv(ctx, a);
}

static int conv_dgrad_packed(e2_ctx* ctx, const e2_tensor5* dy_pad, const void* wp, int cin,
                             int kd, int kh, int kw, const e2_tensor5* dx, int64_t part_stride,
                             int max_parts, int* nparts) {
  E2_REQUIRE(ctx && wp, "conv3d_dgrad: null argument");
  if (int rc = view_ok(dy_pad, "conv3d_dgrad dy_pad")) return rc;
  if (int rc = view_ok(dx, "conv3d_dgrad dx")) return rc;
  E2_REQUIRE(dx->n == dy_pad->n && dx->c == cin && dx->d == dy_pad->d - kd + 1 &&
                 dx->h == dy_pad->h - kh + 1 && dx->w == dy_pad->w - kw + 1,
             "conv3d_dgrad: dx is (%d,%d,%d,%d,%d) but padded dy (%d,%d,%d,%d,%d) with "
             "kernel %d,%d,%d gives (%d,%d,%d,%d,%d)",
             dx->n, dx->c, dx->d, dx->h, dx->w, dy_pad->n, dy_pad->c, dy_pad->d, dy_pad->h,
             dy_pad->w, kd, kh, kw, dy_pad->n, cin, dy_pad->d - kd + 1, dy_pad->h - kh + 1,
             dy_pad->w - kw + 1);
  IgemmArgs a;
  a.in = dy_pad->ptr; a.wp = (const float*)wp; a.out = dx->ptr;
  a.N = dx->n; a.Cin = dy_pad->c; a.Cout = cin;
  a.kd = kd; a.kh = kh; a.kw = kw;
  a.Do = dx->d; a.Ho = dx->h; a.Wo = dx->w;
  a.isN = dy_pad->sn; a.isC = dy_pad->sc; a.isZ = dy_pad->sd; a.isY = dy_pad->sh;
  a.osN = dx->sn; a.osC = dx->sc; a.osZ = dx->sd; a.osY = dx->sh;
  if (int rc = image_dims(ctx, cin, dy_pad->c, &a.ciP, &a.coP)) return rc;
  a.upz = a.upy = a.upx = 1;
  a.zpad = kd - 1;
  a.parts_max = max_parts; a.part_stride = part_stride; a.nparts = nparts;
  return e2i_igemm_conv(ctx, a);
}
extern "C" int e2_conv3d_dgrad_packed(e2_ctx* ctx, const e2_tensor5* dy_pad, const void* wp,
                                      int cin, int kd, int kh, int kw,
                                      const e2_tensor5* dx) {
  return conv_dgrad_packed(ctx, dy_pad, wp, cin, kd, kh, kw, dx, 0, 0, nullptr);
}
extern "C" int e2_conv3d_dgrad_packed_parts(e2_ctx* ctx, const e2_tensor5* dy_pad,
                                            const void* wp, int cin, int kd, int kh, int kw,
                                            const e2_tensor5* dx, int64_t part_stride,
                                            int max_parts, int* nparts) {
  E2_REQUIRE(nparts && max_parts >= 1 && (max_parts == 1 || part_stride > 0),
             "conv3d_dgrad_parts: bad parts arguments");
  return conv_dgrad_packed(ctx, dy_pad, wp, cin, kd, kh, kw, dx, part_stride, max_parts, nparts);
}

extern "C" int e2_conv3d_dgrad_packed_actbwd(e2_ctx* ctx, const e2_tensor5* dy_pad,
                                             const void* wp, int cin, int kd, int kh, int kw,
                                             const e2_tensor5* out_prev, int act_prev,
                                             const float* bias_prev,
                                             const e2_tensor5* dy_pad_prev, int pd, int ph,
                                             int pw, float* dbias_prev) {
  E2_REQUIRE(ctx && wp, "conv3d_dgrad_actbwd: null argument");
  E2_REQUIRE(act_prev == E2_ACT_LIN || act_prev == E2_ACT_RELU, "conv3d_dgrad_actbwd: bad act %d", act_prev);
  if (int rc = view_ok(dy_pad, "conv3d_dgrad_actbwd dy_pad")) return rc;
  if (int rc = view_ok(out_prev, "conv3d_dgrad_actbwd out_prev")) return rc;
  if (int rc = view_ok(dy_pad_prev, "conv3d_dgrad_actbwd dy_pad_prev")) return rc;
  E2_REQUIRE(pd >= 0 && ph >= 0 && pw >= 0, "conv3d_dgrad_actbwd: negative padding");
  const e2_tensor5* o = out_prev; const e2_tensor5* q = dy_pad_prev;
  E2_REQUIRE(o->n == dy_pad->n && o->c == cin && o->d == dy_pad->d - kd + 1 &&
                 o->h == dy_pad->h - kh + 1 && o->w == dy_pad->w - kw + 1,
             "conv3d_dgrad_actbwd: out_prev is (%d,%d,%d,%d,%d) but padded dy (%d,%d,%d,%d,%d) "
             "with kernel %d,%d,%d gives (%d,%d,%d,%d,%d)", o->n, o->c, o->d, o->h, o->w,
             dy_pad->n, dy_pad->c, dy_pad->d, dy_pad->h, dy_pad->w, kd, kh, kw, dy_pad->n, cin,
             dy_pad->d - kd + 1, dy_pad->h - kh + 1, dy_pad->w - kw + 1);
  E2_REQUIRE(q->n == o->n && q->c == o->c && q->d == o->d + 2 * pd && q->h == o->h + 2 * ph &&
                 q->w == o->w + 2 * pw,
             "conv3d_dgrad_actbwd: dy_pad_prev is (%d,%d,%d,%d,%d), expected (%d,%d,%d,%d,%d)",
             q->n, q->c, q->d, q->h, q->w, o->n, o->c, o->d + 2 * pd, o->h + 2 * ph, o->w + 2 * pw);
  e2_tensor5 dx = *q;                           // the interior view
  dx.ptr = q->ptr + (int64_t)pd * q->sd + (int64_t)ph * q->sh + pw;
  dx.d = o->d; dx.h = o->h; dx.w = o->w;
  IgemmArgs a;
  a.in = dy_pad->ptr; a.wp = (const float*)wp; a.out = dx.ptr;
  a.N = dx.n; a.Cin = dy_pad->c; a.Cout = cin;
  a.kd = kd; a.kh = kh; a.kw = kw;
  a.Do = dx.d; a.Ho = dx.h; a.Wo = dx.w;
  a.isN = dy_pad->sn; a.isC = dy_pad->sc; a.isZ = dy_pad->sd; a.isY = dy_pad->sh;
  a.osN = dx.sn; a.osC = dx.sc; a.osZ = dx.sd; a.osY = dx.sh;
  if (int rc = image_dims(ctx, cin, dy_pad->c, &a.ciP, &a.coP)) return rc;
  a.upz = a.upy = a.upx = 1;
  a.zpad = kd - 1;
  int done = 0;
  if (o->sh == o->w) {                          // the epilogue reads dense mask rows
    a.gm = 1; a.gm_src = (act_prev == E2_ACT_RELU) ? o->ptr : nullptr;
    a.gsN = o->sn; a.gsC = o->sc; a.gsZ = o->sd;
    a.gm_dbias = dbias_prev; a.gm_bias = bias_prev; a.gm_done = &done;
  }
  if (q->sh == q->w && q->sd == (int64_t)q->h * q->w && q->sc == (int64_t)q->d * q->sd &&
      (q->n == 1 || q->sn == (int64_t)q->c * q->sc)) {
    a.fill_base = q->ptr; a.fill_n = (size_t)q->n * q->c * q->d * q->h * q->w;
  }
  if (int rc = e2i_igemm_conv(ctx, a)) return rc;
  if (done) return 0;
  if (bias_prev)      // pre-activation form: a (1,1,1) "pool" backward, in place
    return e2_pool_bias_act_bwd(ctx, &dx, out_prev, bias_prev, 1, 1, 1, act_prev, &dx, dbias_prev);
  return e2_bias_act_bwd_out(ctx, &dx, out_prev, act_prev, &dx, dbias_prev);
}

extern "C" int e2_conv3d_fwd(e2_ctx* ctx, const e2_tensor5* x, const float* w, int cout,
                             int kd, int kh, int kw, const e2_tensor5* y, void* ws,
                             size_t ws_bytes) {
  E2_REQUIRE(ctx && w && ws, "conv3d_fwd: null argument");
  if (int rc = view_ok(x, "conv3d_fwd x")) return rc;
  if (int rc = e2_conv3d_pack(ctx, w, cout, x->c, kd, kh, kw, 0, ws, ws_bytes)) return rc;
  return e2_conv3d_fwd_packed(ctx, x, ws, cout, kd, kh, kw, y);
}

extern "C" int e2_conv3d_dgrad(e2_ctx* ctx, const e2_tensor5* dy_pad, const float* w,
                               int cin, int kd, int kh, int kw, const e2_tensor5* dx,
                               void* ws, size_t ws_bytes) {
  E2_REQUIRE(ctx && w && ws, "conv3d_dgrad: null argument");
  if (int rc = view_ok(dy_pad, "conv3d_dgrad dy_pad")) return rc;
  if (int rc = e2_conv3d_pack(ctx, w, dy_pad->c, cin, kd, kh, kw, 1, ws, ws_bytes)) return rc;
  return e2_conv3d_dgrad_packed(ctx, dy_pad, ws, cin, kd, kh, kw, dx);
}

static int wgrad_impl(e2_ctx* ctx, const e2_tensor5* x, const e2_tensor5* dy, float* dw,
                      int kd, int kh, int kw, int accumulate);
extern "C" int e2_conv3d_wgrad(e2_ctx* ctx, const e2_tensor5* x, const e2_tensor5* dy,
                               float* dw, int kd, int kh, int kw) {
  return wgrad_impl(ctx, x, dy, dw, kd, kh, kw, 0);
}
extern "C" int e2_conv3d_wgrad_acc(e2_ctx* ctx, const e2_tensor5* x, const e2_tensor5* dy,
                                   float* dw, int kd, int kh, int kw) {
  return wgrad_impl(ctx, x, dy, dw, kd, kh, kw, 1);
}
static int wgrad_impl(e2_ctx* ctx, const e2_tensor5* x, const e2_tensor5* dy, float* dw,
                      int kd, int kh, int kw, int accumulate) {
  E2_REQUIRE(ctx && dw, "conv3d_wgrad: null argument");
  if (int rc = view_ok(x, "conv3d_wgrad x")) return rc;
  if (int rc = view_ok(dy, "conv3d_wgrad dy")) return rc;
  E2_REQUIRE(dy->n == x->n && dy->d == x->d - kd + 1 && dy->h == x->h - kh + 1 &&
                 dy->w == x->w - kw + 1,
             "conv3d_wgrad: dy spatial (%d,%d,%d) != x (%d,%d,%d) - k + 1", dy->d, dy->h,
             dy->w, x->d, x->h, x->w);
  WgradArgs a;
  a.x = x->ptr; a.dy = dy->ptr; a.dw = dw;
  a.N = x->n; a.Cin = x->c; a.Cout = dy->c;
  a.kd = kd; a.kh = kh; a.kw = kw;
  a.Do = dy->d; a.Ho = dy->h; a.Wo = dy->w;
  a.xsN = x->sn; a.xsC = x->sc; a.xsZ = x->sd; a.xsY = x->sh;
  a.dsN = dy->sn; a.dsC = dy->sc; a.dsZ = dy->sd; a.dsY = dy->sh;
  a.flip = 1;
  a.upR = 1;
  a.accumulate = accumulate;
  a.dy_padded = 0;
  return e2i_wgrad_conv(ctx, a);
}

extern "C" int e2_conv3d_wgrad_pad(e2_ctx* ctx, const e2_tensor5* x, const e2_tensor5* dy_pad,
                                   float* dw, int kd, int kh, int kw, int accumulate) {
  E2_REQUIRE(ctx && dw, "conv3d_wgrad_pad: null argument");
  if (int rc = view_ok(x, "conv3d_wgrad_pad x")) return rc;
  if (int rc = view_ok(dy_pad, "conv3d_wgrad_pad dy_pad")) return rc;
  E2_REQUIRE(kd >= 1 && kh >= 1 && kw >= 1, "conv3d_wgrad_pad: bad kernel");
  const int Do = dy_pad->d - 2 * (kd - 1), Ho = dy_pad->h - 2 * (kh - 1),
            Wo = dy_pad->w - 2 * (kw - 1);
  E2_REQUIRE(dy_pad->n == x->n && Do == x->d - kd + 1 && Ho == x->h - kh + 1 &&
                 Wo == x->w - kw + 1 && Do > 0 && Ho > 0 && Wo > 0,
             "conv3d_wgrad_pad: padded dy (%d,%d,%d) does not match x (%d,%d,%d), kernel %d,%d,%d",
             dy_pad->d, dy_pad->h, dy_pad->w, x->d, x->h, x->w, kd, kh, kw);
  WgradArgs a;
  a.x = x->ptr;
  a.dy = dy_pad->ptr + (int64_t)(kd - 1) * dy_pad->sd + (int64_t)(kh - 1) * dy_pad->sh + (kw - 1);
  a.dw = dw;
  a.N = x->n; a.Cin = x->c; a.Cout = dy_pad->c;
  a.kd = kd; a.kh = kh; a.kw = kw;
  a.Do = Do; a.Ho = Ho; a.Wo = Wo;
  a.xsN = x->sn; a.xsC = x->sc; a.xsZ = x->sd; a.xsY = x->sh;
  a.dsN = dy_pad->sn; a.dsC = dy_pad->sc; a.dsZ = dy_pad->sd; a.dsY = dy_pad->sh;
  a.flip = 1;
  a.upR = 1;
  a.accumulate = accumulate;
  a.dy_padded = 1;
  return e2i_wgrad_conv(ctx, a);
}

// ---------------------------------------------------------------------------
// UpConv: 1x1x1 GEMMs + depth-to-space scatter / space-to-depth gather
// ---------------------------------------------------------------------------
static size_t up_pack_floats(int cout, int cin, int R) {
  int ciP, coP, ciP2, coP2;
  e2i_pack_dims(cout * R, cin, &ciP, &coP);      // fwd: oc' = co*R + r
  e2i_pack_dims(cin, cout * R, &ciP2, &coP2);    // dgrad: ic' = co*R + r
  const size_t a = (size_t)ciP * coP, b = (size_t)ciP2 * coP2;
  return a > b ? a : b;
}

extern "C" size_t e2_upconv3d_workspace_bytes(int cout, int cin, int pz, int py, int px,
                                              int n, int d, int h, int w) {
  const int R = pz * py * px;
  // packed weights + the space-to-depth image of dpre used by the backward
  // (+ alignment of the image and the 128 readable bytes the direct weight-gradient
  // kernel wants behind its last element)
  return sizeof(float) * (up_pack_floats(cout, cin, R) +
                          (size_t)n * cout * R * d * h * w) + 512;
}

// wp != nullptr: the packed image exists (a pack job of mode 2, e2_pack_job_fill); else the
// canonical weights w are packed into the workspace first
static int upconv_fwd(e2_ctx* ctx, const e2_tensor5* x, const float* w, const float* wp,
                      const float* bias, int cout, int pz, int py, int px, int act,
                      const e2_tensor5* y, void* ws, size_t ws_bytes) {
  E2_REQUIRE(ctx && (wp || (w && ws)), "upconv3d_fwd: null argument");
  if (int rc = view_ok(x, "upconv3d_fwd x")) return rc;
  if (int rc = view_ok(y, "upconv3d_fwd y")) return rc;
  E2_REQUIRE(pz >= 1 && py >= 1 && px >= 1, "upconv3d_fwd: bad factors");
  E2_REQUIRE(y->n == x->n && y->c == cout && y->d == x->d * pz && y->h == x->h * py &&
                 y->w == x->w * px, "upconv3d_fwd: y shape mismatch");
  E2_REQUIRE(bias || act == E2_ACT_LIN, "upconv3d_fwd: act needs bias");
  const int R = pz * py * px;
  IgemmArgs a;
  e2i_pack_dims(cout * R, x->c, &a.ciP, &a.coP);
  if (!wp) {
    E2_REQUIRE(ws_bytes >= sizeof(float) * up_pack_floats(cout, x->c, R),
               "upconv3d_fwd: workspace too small");
    if (int rc = e2i_pack_weights(ctx, w, (float*)ws, cout * R, x->c, 1, 1, 1,
                                  (int64_t)x->c * R, R, 0, a.ciP, a.coP, R, 1))
      return rc;
    wp = (const float*)ws;
  }
  a.in = x->ptr; a.wp = wp; a.out = y->ptr;
  a.N = x->n; a.Cin = x->c; a.Cout = cout * R;
  a.kd = a.kh = a.kw = 1;
  a.Do = x->d; a.Ho = x->h; a.Wo = x->w;
  a.isN = x->sn; a.isC = x->sc; a.isZ = x->sd; a.isY = x->sh;
  a.osN = y->sn; a.osC = y->sc; a.osZ = y->sd; a.osY = y->sh;
  a.upz = pz; a.upy = py; a.upx = px;
  int fused = 0;                 // (the pointwise GEMM applies bias + act in its epilogue)
  a.up_bias = bias; a.up_act = act; a.up_bias_done = &fused;
  if (int rc = e2i_igemm_conv(ctx, a)) return rc;
  if (bias && !fused) return e2_pool_bias_act_fwd(ctx, y, bias, 1, 1, 1, act, y);
  return 0;
}

extern "C" int e2_upconv3d_fwd(e2_ctx* ctx, const e2_tensor5* x, const float* w,
                               const float* bias, int cout, int pz, int py, int px, int act,
                               const e2_tensor5* y, void* ws, size_t ws_bytes) {
  E2_REQUIRE(w && ws, "upconv3d_fwd: null argument");
  return upconv_fwd(ctx, x, w, nullptr, bias, cout, pz, py, px, act, y, ws, ws_bytes);
}

extern "C" size_t e2_upconv3d_image_bytes(int cout, int cin, int pz, int py, int px) {
  return sizeof(float) * up_pack_floats(cout, cin, pz * py * px) + 256;
}

extern "C" int e2_upconv3d_fwd_packed(e2_ctx* ctx, const e2_tensor5* x, const float* wp_fwd,
                                      const float* bias, int cout, int pz, int py, int px,
                                      int act, const e2_tensor5* y) {
  E2_REQUIRE(wp_fwd, "upconv3d_fwd_packed: null image");
  return upconv_fwd(ctx, x, nullptr, wp_fwd, bias, cout, pz, py, px, act, y, nullptr, 0);
}

// accumulate: dw and dbias are ADDED to (the caller zeroed them: the plan clears the whole
// gradient arena with one launch at the start of the backward pass) -- two fills less
// wp_d != nullptr: the data gradient's packed image exists (pack job of mode 3)
static int upconv_bwd(e2_ctx* ctx, const e2_tensor5* x, const float* w,
                      const e2_tensor5* y, const e2_tensor5* dout, int pz, int py,
                      int px, int act, const e2_tensor5* dx, float* dw,
                      float* dbias, void* ws, size_t ws_bytes, int accumulate,
                      const float* wp_d = nullptr) {
  // the weights are read by the data gradient only: an UpConv whose parent needs no gradient
  // (dx == nullptr, e.g. directly on an Input) passes neither w nor an image
  E2_REQUIRE(ctx && ws && (w || wp_d || !dx), "upconv3d_bwd: null argument");
  if (int rc = view_ok(x, "upconv3d_bwd x")) return rc;
  if (int rc = view_ok(y, "upconv3d_bwd y")) return rc;
  if (int rc = view_ok(dout, "upconv3d_bwd dout")) return rc;
  const int cout = y->c, cin = x->c, R = pz * py * px;
  E2_REQUIRE(dout->n == y->n && dout->c == cout && dout->d == y->d && dout->h == y->h &&
                 dout->w == y->w && y->d == x->d * pz && y->h == x->h * py &&
                 y->w == x->w * px, "upconv3d_bwd: shape mismatch");
  E2_REQUIRE(ws_bytes >= e2_upconv3d_workspace_bytes(cout, cin, pz, py, px, x->n, x->d,
                                                     x->h, x->w),
             "upconv3d_bwd: workspace too small");
  float* wp = (float*)ws;
  size_t off = (up_pack_floats(cout, cin, R) + 63) & ~(size_t)63;
  float* s2d = wp + off;
  if (dbias && !accumulate)
    if (int rc = e2i_fill_flat(ctx, dbias, (size_t)cout, 0.f)) return rc;
  if (int rc = e2i_upconv_dpre_s2d(ctx, dout, y, pz, py, px, act, s2d, dbias)) return rc;
  const long S = (long)x->d * x->h * x->w;
  if (dx) {
    if (int rc = view_ok(dx, "upconv3d_bwd dx")) return rc;
    E2_REQUIRE(dx->n == x->n && dx->c == cin && dx->d == x->d && dx->h == x->h &&
                   dx->w == x->w, "upconv3d_bwd: dx shape mismatch");
    IgemmArgs a;
    e2i_pack_dims(cin, cout * R, &a.ciP, &a.coP);
    // Wp[ic' = co*R + r][oc = ci] = w[co][ci][r]
    if (!wp_d)
      if (int rc = e2i_pack_weights(ctx, w, wp, cin, cout * R, 1, 1, 1, R, (int64_t)cin * R,
                                    0, a.ciP, a.coP, 1, R))
        return rc;
    a.in = s2d; a.wp = wp_d ? wp_d : wp; a.out = dx->ptr;
    a.N = x->n; a.Cin = cout * R; a.Cout = cin;
    a.kd = a.kh = a.kw = 1;
    a.Do = x->d; a.Ho = x->h; a.Wo = x->w;
    a.isN = (long)cout * R * S; a.isC = S; a.isZ = (long)x->h * x->w; a.isY = x->w;
    a.osN = dx->sn; a.osC = dx->sc; a.osZ = dx->sd; a.osY = dx->sh;
    a.upz = a.upy = a.upx = 1;
    if (int rc = e2i_igemm_conv(ctx, a)) return rc;
  }
  if (dw) {
    WgradArgs g;
    g.x = x->ptr; g.dy = s2d; g.dw = dw;
    g.N = x->n; g.Cin = cin; g.Cout = cout * R;
    g.kd = g.kh = g.kw = 1;
    // 1x1x1 taps: the dense space-to-depth image IS its own zero-padded form (no border,
    // no gap columns), and the workspace leaves 128 B behind it: the direct kernel applies
    g.dy_padded = 1;
    g.Do = x->d; g.Ho = x->h; g.Wo = x->w;
    g.xsN = x->sn; g.xsC = x->sc; g.xsZ = x->sd; g.xsY = x->sh;
    g.dsN = (long)cout * R * S; g.dsC = S; g.dsZ = (long)x->h * x->w; g.dsY = x->w;
    g.flip = 0;
    g.upR = R;
    g.accumulate = accumulate;
    if (int rc = e2i_wgrad_conv(ctx, g)) return rc;
  }
  return 0;
}

extern "C" int e2_upconv3d_bwd(e2_ctx* ctx, const e2_tensor5* x, const float* w,
                               const e2_tensor5* y, const e2_tensor5* dout, int pz, int py,
                               int px, int act, const e2_tensor5* dx, float* dw,
                               float* dbias, void* ws, size_t ws_bytes) {
  return upconv_bwd(ctx, x, w, y, dout, pz, py, px, act, dx, dw, dbias, ws, ws_bytes, 0);
}

extern "C" int e2_upconv3d_bwd_packed(e2_ctx* ctx, const e2_tensor5* x, const float* wp_dgrad,
                                      const e2_tensor5* y, const e2_tensor5* dout, int pz,
                                      int py, int px, int act, const e2_tensor5* dx, float* dw,
                                      float* dbias, void* ws, size_t ws_bytes, int accumulate) {
  E2_REQUIRE(wp_dgrad || !dx, "upconv3d_bwd_packed: null image");
  return upconv_bwd(ctx, x, nullptr, y, dout, pz, py, px, act, dx, dw, dbias, ws, ws_bytes,
                    accumulate ? 1 : 0, wp_dgrad);
}

// ---------------------------------------------------------------------------
// graph capture / events
// ---------------------------------------------------------------------------
struct e2_graph {
  hipGraph_t graph;
  hipGraphExec_t exec;
};

extern "C" int e2_graph_begin(e2_ctx* ctx) {
  E2_REQUIRE(ctx, "graph_begin: null ctx");
  E2_REQUIRE(ctx->stream != nullptr,
             "graph_begin: capture needs a non-default stream (e2_ctx_set_stream)");
  E2_CHECK_HIP(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
  ctx->capturing = true;
  return 0;
}

extern "C" int e2_graph_end(e2_ctx* ctx, e2_graph** out) {
  E2_REQUIRE(ctx && out, "graph_end: null argument");
  hipGraph_t g = nullptr;
  ctx->capturing = false;
  E2_CHECK_HIP(hipStreamEndCapture(ctx->stream, &g));
  hipGraphExec_t ex = nullptr;
  E2_CHECK_HIP(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
  e2_graph* r = new e2_graph{g, ex};
  *out = r;
  return 0;
}

extern "C" int e2_graph_launch(e2_ctx* ctx, e2_graph* g) {
  E2_REQUIRE(ctx && g, "graph_launch: null argument");
  E2_CHECK_HIP(hipGraphLaunch(g->exec, ctx->stream));
  return 0;
}

extern "C" int e2_graph_debug_dot(e2_graph* g, const char* path, int verbose) {
  E2_REQUIRE(g && path, "graph_debug_dot: null argument");
  E2_CHECK_HIP(hipGraphDebugDotPrint(g->graph, path, verbose ? hipGraphDebugDotFlagsKernelNodeParams : 0));
  return 0;
}

extern "C" int e2_graph_destroy(e2_graph* g) {
  if (!g) return 0;
  (void)hipGraphExecDestroy(g->exec);
  (void)hipGraphDestroy(g->graph);
  delete g;
  return 0;
}

struct e2_event { hipEvent_t ev; };

extern "C" int e2_event_create(e2_event** out) {
  E2_REQUIRE(out, "event_create: null out");
  hipEvent_t ev;
  E2_CHECK_HIP(hipEventCreate(&ev));
  *out = new e2_event{ev};
  return 0;
}
extern "C" int e2_event_record(e2_ctx* ctx, e2_event* e) {
  E2_REQUIRE(ctx && e, "event_record: null argument");
  E2_CHECK_HIP(hipEventRecord(e->ev, ctx->stream));
  return 0;
}
extern "C" int e2_event_elapsed_ms(e2_event* start, e2_event* stop, float* ms) {
  E2_REQUIRE(start && stop && ms, "event_elapsed_ms: null argument");
  E2_CHECK_HIP(hipEventSynchronize(stop->ev));
  E2_CHECK_HIP(hipEventElapsedTime(ms, start->ev, stop->ev));
  return 0;
}
extern "C" int e2_event_destroy(e2_event* e) {
  if (!e) return 0;
  (void)hipEventDestroy(e->ev);
  delete e;
  return 0;
}
extern "C" int e2_stream_synchronize(e2_ctx* ctx) {
  E2_REQUIRE(ctx, "stream_synchronize: null ctx");
  E2_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  return 0;
}
