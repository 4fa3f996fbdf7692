/*
 * vcg.h — C ABI of libvcg.so, the MI355X (gfx950) native kernels behind the
 * VAE-CycleGAN training step.
 *
 * The reference (Baverne/VAE-CYCLEGAN-Implementation) has no FFI of its own:
 * every op below replaces a torch call site inside Networks.py / Losses.py.
 * The citation after each entry point is the reference line it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (torch's caching
 *     allocator); the library never allocates, frees or synchronises;
 *   - activations are fp32 NHWC with a channel pitch that is a multiple of 4
 *     (3-channel images are stored with pitch 4, pad channel == 0);
 *   - conv weights live in the reference's OIHW layout (state_dict parity) and
 *     are repacked once per optimizer step into the GEMM layout Wf[K][Cout];
 *   - all launches go to `stream` (a hipStream_t passed as void*);
 *   - return 0 on success, negative on error; vcg_last_error() has the text.
 *     Nothing throws across this boundary.
 */
#ifndef VCG_H
#define VCG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VCG_ABI_VERSION 6   /* 6 (round 4): vcg_amax_valid, the explicit-argument forms vcg_*_h.  5: vcg_conv_reads_wf, cd[VCG_CD_PACK_FLAGS].  4: vcg_amax_measure.  3 (round 3): packed weights and kept forward state carry fp16 x 2 planes + amax
                               words (sizes changed); vcg_amax_hint / vcg_amax_last.  round 2: vcg_adam_step takes 1 - beta; vcg_conv_fwd_in,
                               vcg_conv_wgrad_saved, input transforms, profiling */

/* Operand magnitudes (round 3) ------------------------------------------------
 * The MFMA kernels compute fp32 products as three fp16 MFMAs: every operand is scaled by a power of two taken from the
 * largest magnitude ("amax") of its tensor and split into two fp16 pieces (csrc/vcg_common.h).  An entry point that reads
 * a tensor whose amax it does not know measures it with one pass over the tensor.  The entry points that WRITE activations
 * and gradients (vcg_in_apply, vcg_in_bwd, vcg_act_bwd) publish the amax of what they wrote as a by-product:
 *   vcg_amax_last()        handle of the amax of the tensor the last such call on this thread wrote (0: none); reading resets it;
 *   vcg_amax_hint(x, dy)   handles for the x / dy operands of the NEXT vcg_conv_fwd / _fwd_in / _dgrad / _wgrad(_saved) call
 *                          on this thread (0 = unknown: measured); consumed by that call.
 * A handle names device-side state of the library's code object; it stays valid for ~15 000 later library calls on the device
 * (a training step makes ~1 000) and is refused — the tensor is measured — once it is older.  Passing a handle that belongs to
 * another tensor scales that operand wrongly: hand over only what vcg_amax_last returned for exactly that tensor. */
void vcg_amax_hint(uint64_t x_amax, uint64_t dy_amax);
uint64_t vcg_amax_last(void);
/* Measure a tensor nobody published an amax for (n floats at t, 16-byte aligned), once, on `stream`: the handle serves every
 * later call that reads the tensor on a stream ordered after this one (a forward and the weight gradient that re-reads x; the
 * weight and data gradient of one dy) instead of each of them measuring it again.  0: no slot on this device (callers pass 0 on:
 * the consumers measure). */
uint64_t vcg_amax_measure(const float* t, size_t n, void* stream);
/* 1 while `handle` would still be honoured on the current device, 0 once it is too old (or is not a handle): a caller that keeps
 * handles across many calls (ops.py keeps one per tensor object) asks before handing one on and measures again — refreshing what it
 * keeps — instead of carrying a handle every consumer refuses.  A handle is honoured while its slot is >= 6144 generations from
 * reuse: the check runs when a kernel is enqueued, the kernel reads the slot up to one training step (~1 500 generations) later. */
int vcg_amax_valid(uint64_t handle);
/* A handle describes the tensor's CONTENTS at the time it was published: a caller that lets anything write into the tensor
 * afterwards (in-place ops, buffer reuse) must drop the handle (ops.py keys it on torch's version counter and the data pointer).
 *
 * Explicit-argument forms (ABI v6) — the same calls with the operand handles as arguments and the handle of what the call wrote
 * returned through a pointer, for callers that do not want per-thread state between two calls; vcg_amax_hint / vcg_amax_last
 * remain as the shim they are built on:
 *   vcg_conv_fwd_in_h(..., x_amax, stream)             == vcg_amax_hint(x_amax, 0);  vcg_conv_fwd_in(...)
 *   vcg_conv_dgrad_h(..., dy_amax, stream)             == vcg_amax_hint(0, dy_amax); vcg_conv_dgrad(...)
 *   vcg_conv_wgrad_saved_h(..., x_amax, dy_amax, stream)
 *   vcg_in_apply_h / vcg_in_bwd_h / vcg_act_bwd_h(..., &amax_of_what_was_written, stream)   (pointer may be NULL)
 * declared next to their base forms below. */

/* conv descriptor: int32[16] ------------------------------------------------ */
enum {
  VCG_CD_N = 0,      /* batch */
  VCG_CD_H = 1,      /* physical input height */
  VCG_CD_W = 2,      /* physical input width */
  VCG_CD_CIN = 3,    /* physical input channel pitch (multiple of 4) */
  VCG_CD_COUT = 4,   /* physical output channel pitch (multiple of 4) */
  VCG_CD_KH = 5,
  VCG_CD_KW = 6,
  VCG_CD_STRIDE = 7, /* 1 or 2 */
  VCG_CD_PAD = 8,
  VCG_CD_REFLECT = 9, /* 1 = padding_mode='reflect', 0 = zeros */
  VCG_CD_UPS = 10,   /* 1, or 2 = nn.PixelUnshuffle(2) folded into the gather */
  VCG_CD_ACT = 11,   /* epilogue activation: VCG_ACT_* */
  VCG_CD_CIN_LOGICAL = 12, /* channels of the physical input that carry data (3 for images) */
  VCG_CD_COUT_LOGICAL = 13,
  VCG_CD_PACK_FLAGS = 14,  /* vcg_pack_weight only: bit 0 = leave the fp32 Wf block out (vcg_conv_reads_wf); 0 everywhere else */
  VCG_CD_LEN = 16
};

enum { VCG_ACT_NONE = 0, VCG_ACT_RELU = 1, VCG_ACT_LEAKY02 = 2, VCG_ACT_TANH = 3, VCG_ACT_SIGMOID = 4 };   /* CaSb's choices, Networks.py:62-73 */

int vcg_abi_version(void);
const char* vcg_last_error(void);

/* Diagnostic (bench.py's `roofline` object; the training path never enables it): while enabled, every MFMA kernel launch
   is bracketed by HIP events on its launch stream.  vcg_profile_read WAITS for those events (the one call of this library
   that synchronises) and writes one line per device kernel: name \t launches \t total ms \t executed FLOPs
   (fp32-equivalent: 2*M*N*K of the GEMM the launch ran); returns the bytes written — or, with buf NULL, the bytes a
   following call will need (the table is kept until it has been copied out). */
int vcg_profile_enable(int on);
long vcg_profile_read(char* buf, size_t cap);

/* layout ------------------------------------------------------------------- */
/* x.to(channels-last); images get channel pitch P (pad channels zeroed).     */
int vcg_nchw_to_nhwc(const float* src, float* dst, int N, int C, int H, int W, int P, void* stream);
int vcg_nhwc_to_nchw(const float* src, float* dst, int N, int C, int H, int W, int P, void* stream);
int vcg_fill(float* dst, float value, size_t n, void* stream);

/* nn.Conv2d(padding_mode='reflect') — Networks.py:60,87,101,104,122,136,145 -- */
/* OIHW -> the packed buffer every conv entry point below takes as `wf` (repacked once per optimizer step):
     Wf[K][Cout]          K ordered (kh,kw,i,j,c), so PixelUnshuffle (Networks.py:86) needs no data movement;
                          the B operand of the data-gradient and weight-gradient GEMMs
     U, Ud[16][..]        3x3 / stride-1 layers: the Winograd F(2x2,3x3) transforms G g G^T of the kernel (forward,
                          [xi][Cout][K]) and of the flipped kernel (data gradient, [xi][K][Cout]), ALREADY SCALED AND SPLIT into
                          the two fp16 pieces the GEMMs multiply with ("blocked planes": 128 bytes per row and 32-wide block)
     Wk, Wkd              7x7 layers with <= 4 channels on one side: kw folded into the GEMM's N (forward / data gradient)
     WFT, WFD planes      other layers with >= 64 output channels: the transpose of Wf (forward) and its rows per tap (data
                          gradient), pre-split the same way for the direct split-operand kernels
   vcg_pack_weight_floats: floats the caller must provide for `wf` (spatial fields of cd are ignored).
   vcg_conv_reads_wf(cd): 1 if vcg_conv_fwd / _fwd_in / _dgrad at THIS geometry (N, H, W count) read the fp32 Wf block of the
     pack — the thin layers, tiles the split-operand kernels do not take, maps Winograd cannot take — 0 if every direction runs
     from the pre-split planes.  A caller whose layer answers 0 for every geometry it is used at may set cd[VCG_CD_PACK_FLAGS] = 1
     in the descriptor it hands to vcg_pack_weight: the Wf block is then not written (round 3: 0.6 ms and 1 GB per training step
     for the D / R / U layers, whose Wf nothing reads).  Reading a Wf that was left out is undefined: ask again when the geometry
     changes, and repack in full if the answer does. */
size_t vcg_pack_weight_floats(const int32_t* cd);
int vcg_conv_reads_wf(const int32_t* cd);
int vcg_pack_weight(const float* w_oihw, float* wf, const int32_t* cd, void* stream);
/* y = act(conv(x) + bias).  fp32 in, fp32 out, fp32 accumulate; the GEMMs run on the 16-bit matrix pipe
   (v_mfma_f32_{32x32x16,16x16x32}_f16) with every fp32 operand scaled by a power of two and split into two fp16 pieces, three
   products per multiply-add (fp32-level rounding, csrc/vcg_common.h); thin layers use v_mfma_f32_32x32x2_f32.
   3x3 / stride-1 layers go through Winograd F(2x2,3x3) (16 batched GEMMs, V and M in `ws`); layers with few
   output tiles slice K across workgroups into fp32 slabs in `ws` (vcg_conv_fwd_workspace bytes, may be 0). */
size_t vcg_conv_fwd_workspace(const int32_t* cd);
int vcg_conv_fwd(const float* x, const float* wf, const float* bias, float* y,
                 const int32_t* cd, void* ws, size_t ws_bytes, void* stream);

/* The same convolution when an InstanceNorm follows it (CaSb with norm=True, /root/reference/Networks.py:93-95): y as
   above AND mean / rstd (N x Cout each) of y over its pixels, biased variance, rstd = 1 / sqrt(var + eps) — what
   vcg_in_stats(y) returns.  Where the conv's launch plan allows (Winograd output transform; direct split-operand tiles
   with Ho*Wo % 128 == 0; the LDS-slab pixel blocks) the statistics' partial sums (in double) are written by the conv's own
   epilogue, so y is not read again; otherwise the separate reduction pass runs.  mean == rstd == NULL: no statistics,
   the plain forward.  `ws`: vcg_conv_fwd_in_workspace(cd) bytes (vcg_conv_fwd_workspace when mean is NULL).
   `saved` (may be NULL): vcg_conv_saved_floats(cd) floats in which the call leaves forward state that the weight gradient
   of the same (x, cd) can reuse instead of recomputing it — vcg_conv_wgrad_saved below; 0 floats: nothing to keep. */
size_t vcg_conv_fwd_in_workspace(const int32_t* cd);
size_t vcg_conv_saved_floats(const int32_t* cd);
int vcg_conv_fwd_in(const float* x, const float* wf, const float* bias, float* y, float* mean, float* rstd, float eps,
                    float* saved, const int32_t* cd, void* ws, size_t ws_bytes, void* stream);
int vcg_conv_fwd_in_h(const float* x, const float* wf, const float* bias, float* y, float* mean, float* rstd, float eps,
                      float* saved, const int32_t* cd, void* ws, size_t ws_bytes, uint64_t x_amax, void* stream);
/* The fused Conv + InstanceNorm + activation hand-off (round 4; /root/reference/Networks.py:93-95, 110-115: a block's
   InstanceNorm output feeds the next block's conv).  `t_prev` is the RAW output of the previous block's conv — what
   vcg_conv_fwd_in left in y, with its statistics pre_mean / pre_rstd ([N][Cin]) — and this conv's input gather computes
   pre_act((t_prev - mean) * rstd) on the way in: the normalised tensor is never written or read (vcg_in_apply is not called for
   it).  Everything else as vcg_conv_fwd_in (y, this conv's own statistics, `saved`).  The operand's magnitude is bounded by
   sqrt(H * W) instead of measured.  Exists where vcg_conv_pre_ok(cd) says 1 — the Winograd input transform (the D2..D4, R and U1
   layers at the training sizes); elsewhere it returns an error and the caller runs vcg_in_apply + vcg_conv_fwd_in.  The weight
   gradient of such a layer must come from `saved` (vcg_conv_wgrad_saved): there is no normalised x to re-read. */
int vcg_conv_pre_ok(const int32_t* cd);
int vcg_conv_fwd_in_pre(const float* t_prev, const float* pre_mean, const float* pre_rstd, int pre_act, const float* wf,
                        const float* bias, float* y, float* mean, float* rstd, float eps, float* saved, const int32_t* cd,
                        void* ws, size_t ws_bytes, void* stream);
/* dx = conv^T(dy) including the adjoint of the reflect padding.              */
size_t vcg_conv_dgrad_workspace(const int32_t* cd);
int vcg_conv_dgrad(const float* dy, const float* wf, float* dx, const int32_t* cd,
                   void* ws, size_t ws_bytes, void* stream);
int vcg_conv_dgrad_h(const float* dy, const float* wf, float* dx, const int32_t* cd,
                     void* ws, size_t ws_bytes, uint64_t dy_amax, void* stream);
/* gw_oihw += x^T dy (split-K slabs in ws, deterministic reduce); gbias += sum(dy).
   gbias may be NULL.                                                         */
size_t vcg_conv_wgrad_workspace(const int32_t* cd);
int vcg_conv_wgrad(const float* x, const float* dy, float* gw_oihw, float* gbias,
                   const int32_t* cd, void* ws, size_t ws_bytes, void* stream);
/* vcg_conv_wgrad with the forward state vcg_conv_fwd_in kept in `saved` (NULL: identical to vcg_conv_wgrad). */
int vcg_conv_wgrad_saved(const float* x, const float* dy, float* gw_oihw, float* gbias, const float* saved,
                         const int32_t* cd, void* ws, size_t ws_bytes, void* stream);
int vcg_conv_wgrad_saved_h(const float* x, const float* dy, float* gw_oihw, float* gbias, const float* saved,
                           const int32_t* cd, void* ws, size_t ws_bytes, uint64_t x_amax, uint64_t dy_amax, void* stream);

/* nn.InstanceNorm2d(eps=1e-5, affine=False) — Networks.py:61,88,102,105,123 -- */
size_t vcg_in_workspace(int N, int HW, int C);
int vcg_in_stats(const float* t, float* mean, float* rstd, int N, int HW, int C, float eps,
                 void* ws, size_t ws_bytes, void* stream);
/* out = post_act((t-mean)*rstd) [+ residual]; shuffle=1 stores through
   nn.PixelShuffle(2) (Networks.py:121): out is (N,2H,2W,C/4).                */
int vcg_in_apply(const float* t, const float* mean, const float* rstd, const float* residual,
                 float* out, int N, int H, int W, int C, int post_act, int shuffle, void* stream);
int vcg_in_apply_h(const float* t, const float* mean, const float* rstd, const float* residual,
                   float* out, int N, int H, int W, int C, int post_act, int shuffle, uint64_t* out_amax, void* stream);
/* dt = epi_act'(t) * IN-backward(post_act'(.) * g); g is in `out` layout.   */
int vcg_in_bwd(const float* g, const float* t, const float* mean, const float* rstd, float* dt,
               int N, int H, int W, int C, int epi_act, int post_act, int shuffle,
               void* ws, size_t ws_bytes, void* stream);
int vcg_in_bwd_h(const float* g, const float* t, const float* mean, const float* rstd, float* dt,
                 int N, int H, int W, int C, int epi_act, int post_act, int shuffle,
                 void* ws, size_t ws_bytes, uint64_t* dt_amax, void* stream);
/* vcg_in_bwd that also ACCUMULATES sum over (n, pixel) of dt into gbias[0 .. c_log): the bias gradient of the convolution in front
   of this InstanceNorm when an activation sits between the two (conv -> ReLU -> IN, Networks.py:93-95, 110-111, 128-130; with no
   activation between them it is identically zero) — from the pass that writes dt, instead of a pass of its own inside
   vcg_conv_wgrad (hand that call gbias = NULL).  dt_amax: as vcg_in_bwd_h (may be NULL).  Same workspace. */
int vcg_in_bwd_bias(const float* g, const float* t, const float* mean, const float* rstd, float* dt,
                    int N, int H, int W, int C, int epi_act, int post_act, int shuffle, float* gbias, int c_log,
                    void* ws, size_t ws_bytes, uint64_t* dt_amax, void* stream);
/* nn.PixelShuffle(2) — Networks.py:121 — as a copy: (N,H,W,C) -> (N,2H,2W,C/4); inverse=1 is its backward */
int vcg_pixel_shuffle(const float* src, float* dst, int N, int H, int W, int C, int inverse, void* stream);
/* dt = g * act'(t) where t is the activation OUTPUT (blocks without a norm). */
int vcg_act_bwd(const float* g, const float* t, float* dt, size_t n, int act, void* stream);
int vcg_act_bwd_h(const float* g, const float* t, float* dt, size_t n, int act, uint64_t* dt_amax, void* stream);

/* Helpers of the fused mu / logvar convolution (Networks.py:219-222: both convolutions read one map — they run as ONE
   convolution with 2 x latent output channels).  NHWC rows of ca + cb floats <-> rows of ca and rows of cb (multiples of 4;
   vcg_chan_cat: a NULL source reads as zeros); vcg_add_into: dst += src, src = 0 (n a multiple of 4, 16-byte aligned) — the
   fused kernel's weight gradient handed to the two parameters' own gradient buffers. */
int vcg_chan_split(const float* src, float* a, float* b, size_t rows, int ca, int cb, void* stream);
int vcg_chan_cat(const float* a, const float* b, float* dst, size_t rows, int ca, int cb, void* stream);
int vcg_add_into(float* dst, float* src, size_t n, void* stream);

/* VariationalEncoderBlock.forward — Networks.py:219-227 -------------------- */
/* lvc = clamp(lv,-10,10); z = mu + eps*exp(0.5*lvc). eps==NULL: eps is drawn
   on device (Philox4x32-10 + Box-Muller, (seed, offset)) and written to eps_out. */
int vcg_reparam_fwd(const float* mu, const float* lv, const float* eps, float* eps_out,
                    float* z, float* lvc, size_t n, uint64_t seed, uint64_t offset, void* stream);
/* dmu = gz ; dlv = (gz*eps*0.5*exp(0.5*lvc) + glvc) * [-10<=lv<=10]; glvc may be NULL */
int vcg_reparam_bwd(const float* gz, const float* glvc, const float* eps, const float* lv,
                    float* dmu, float* dlv, size_t n, void* stream);
int vcg_randn(float* out, size_t n, uint64_t seed, uint64_t offset, void* stream);
int vcg_rand_uniform(float* out, size_t n, uint64_t seed, uint64_t offset, void* stream);

/* Losses.py ---------------------------------------------------------------- */
size_t vcg_reduce_workspace(size_t n);
/* nn.L1Loss — Losses.py:21-24,34-39,53-65: out[0] = sum|a-b| / n_logical     */
int vcg_l1_fwd(const float* a, const float* b, float* out, size_t n_phys, size_t n_logical,
               void* ws, size_t ws_bytes, void* stream);
/* ga = sign(a-b) * gout[0] / n_logical ; gb (optional) = -ga                 */
int vcg_l1_bwd(const float* a, const float* b, const float* gout, float* ga, float* gb,
               size_t n_phys, size_t n_logical, void* stream);
/* nn.MSELoss vs a constant — Losses.py:78-83,97-102: out[0]=mean((d-c)^2), out[1]=mean(d) */
int vcg_mse_const_fwd(const float* d, float target, float* out, size_t n, void* stream);
int vcg_mse_const_bwd(const float* d, float target, const float* gout, float* gd, size_t n, void* stream);
/* KLDivergenceLoss — Losses.py:115-121                                       */
int vcg_kl_fwd(const float* mu, const float* lv, float* out, size_t n,
               void* ws, size_t ws_bytes, void* stream);
int vcg_kl_bwd(const float* mu, const float* lv, const float* gout, float* gmu, float* glv,
               size_t n, void* stream);
/* out[0] = sum_i w[i]*(*s[i]) ; the composite loss lines Networks.py:941,2012-2018 */
int vcg_lincomb_fwd(const float* const* s, const float* w, int count, float* out, void* stream);

/* spectral_norm(nn.Conv2d(512,1,16)) — Networks.py:248 ---------------------- */
/* one power iteration exactly as torch.nn.utils.spectral_norm in train mode:
   v<-normalize(W^T u), u<-normalize(W v), sigma=u.(W v); wsn (NHWC-K order) = W/sigma.
   ws: >= 8 bytes of device scratch (two scalars handed from the reduction to the all-CU apply pass) */
int vcg_sn_prepare(const float* w_orig_oihw, float* u, float* v, float* sigma, float* wsn_k,
                   int C, int KH, int KW, int update_uv, void* ws, size_t ws_bytes, void* stream);
/* out[n] = <x[n,:], wsn_k> + bias[0]                                          */
int vcg_fullmap_fwd(const float* x, const float* wsn_k, const float* bias, float* out,
                    int N, size_t K, void* stream);
/* dx[n,:] = g[n]*wsn_k (dx may be NULL)                                       */
int vcg_fullmap_dgrad(const float* g, const float* wsn_k, float* dx, int N, size_t K, void* stream);
/* gw_orig_oihw += d(W/sigma)^T applied to (sum_n g[n] x[n,:]); gbias[0] += sum g */
int vcg_fullmap_wgrad(const float* g, const float* x, const float* wsn_k, const float* sigma,
                      const float* u, const float* v, float* gw_orig_oihw, float* gbias,
                      int N, int C, int KH, int KW, void* ws, size_t ws_bytes, void* stream);

/* Input transforms on the device — the torchvision pipelines of train.py:184-190, 248-262, 309-319 ------------------------- */
/* RandomHorizontal/VerticalFlip -> RandomResizedCrop(S, BICUBIC) | Resize((S,S)) -> ToTensor for N decoded uint8 HWC images
   packed in `arena`.  params[n][16] (device, int32): source offset lo, hi; source H, W; crop box y0, x0, h, w in
   FLIPPED-image coordinates; flip_h; flip_v; filter (0 bicubic, 1 bilinear); source kind (0: uint8 RGB in `arena`, offset in
   bytes; 1: float4 pixels in [0, 1] in `fsrc` — what vcg_input_prejitter left — offset in pixels; fsrc may be NULL when no
   sample uses it).  Pillow's antialiased convolution resize in floating point.  out: (N, S, S, 4) fp32, channel 3 = 0 — the
   layout every network entry point takes.                                                                                     */
int vcg_input_resample(const unsigned char* arena, const float* fsrc, const int32_t* params, float* out, int N, int S,
                       void* stream);
/* torchvision ColorJitter (tensor-path formulas) in place on (N, S, S, 4).  jitter[n][8] (device, fp32): enabled, brightness,
   contrast, saturation, hue factors, order code o0 + 4 o1 + 16 o2 + 64 o3 (0 brightness, 1 contrast, 2 saturation, 3 hue). */
int vcg_input_color_jitter(float* img, const float* jitter, int N, int S, void* stream);
/* ColorJitter on WHOLE decoded frames, before any crop — hypersim's colour modality (Data_Manager.py:164-171 applies
   color_transform first, then the spatial transform): unpacks N uint8 frames into float4 pixels in `fbuf` and jitters them
   there.  frames[n][8] (device, int32): arena byte offset lo, hi; pixel count; fbuf pixel offset lo, hi.  var[n][4]: fbuf
   pixel offset lo, hi; pixel count.  jitter[n][8] as above.                                                                   */
int vcg_input_prejitter(const unsigned char* arena, const int32_t* frames, const float* jitter, const int32_t* var,
                        float* fbuf, int N, void* stream);

/* torch.optim.Adam.step — call sites Networks.py:312,894,1928-1935 ---------- */
/* single-tensor torch formula on one flat buffer:
   m += (1-b1)(g-m); v = b2 v + (1-b2) g^2; p -= step_size * m / (sqrt(v)/bc2_sqrt + eps)
   one_minus_beta1/2: 1 - beta computed in double by the caller and rounded once, as torch passes them to lerp_ / addcmul_ */
int vcg_adam_step(float* p, const float* g, float* m, float* v, size_t n,
                  float step_size, float beta1, float beta2, float one_minus_beta1, float one_minus_beta2,
                  float eps, float bc2_sqrt, float grad_scale, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VCG_H */
