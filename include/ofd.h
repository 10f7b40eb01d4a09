/* libofd_hip -- C-ABI of the MI355X-native FlowDiffuser hot path (gfx950 only).
 *
 * This is the drop-in boundary: plain pointers and sizes, no torch types.  All pointers are
 * DEVICE pointers unless a parameter says "host".  Every call enqueues work on `stream`
 * (a hipStream_t passed as void*, e.g. torch.cuda.current_stream().cuda_stream) and returns
 * 0 on success or a negative ofd_status; ofd_last_error() gives the message of the last
 * failure on the calling thread.  No call synchronises the device.  The caller owns all
 * buffers; the library owns nothing but the handles it returns.
 *
 * Reference interfaces replaced (paths relative to the reference repo,
 * algorithms/diffusion_animation/):
 *   softsplat_new.py:255-269,352-423  cuda_launch("softsplat_out")      -> ofd_splat_fwd
 *   softsplat_new.py:489-565          cuda_launch("softsplat_ingrad")   -> ofd_splat_bwd_in
 *   softsplat_new.py:600-700          cuda_launch("softsplat_flowgrad") -> ofd_splat_bwd_flow
 *   warp.py:121-156   warp_forward_flow (NaN handling + holes)          -> ofd_warp_prep / ofd_warp_holes
 *   warp.py:95-119    warp_backward_flow (2x F.grid_sample + mask)      -> ofd_grid_warp_fwd, ofd_grid_warp_bwd
 *   denoising_diffusion.py:363-417    Unet.forward                      -> ofd_unet_forward
 *   denoising_diffusion.py:666-698    p_mean_variance + p_sample update -> ofd_ddpm_update
 *   denoising_diffusion.py:750-767    ddim_sample update                -> ofd_ddim_update
 *   denoising_diffusion.py:806-812    q_sample                          -> ofd_q_sample
 *   warp.py:260-271 + denoising_diffusion.py:908,973  nan_mse + nanmean -> ofd_nan_mse_sum
 */
#ifndef OFD_H
#define OFD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    OFD_OK = 0,
    OFD_ERR_ARG = -1,       /* bad shape / null pointer / unsupported configuration */
    OFD_ERR_HIP = -2,       /* a HIP runtime call failed */
    OFD_ERR_WORKSPACE = -3, /* workspace too small */
    OFD_ERR_STATE = -4      /* handle used out of order (e.g. forward before weights) */
} ofd_status;

int ofd_version(void);               /* (major << 16) | minor */
const char* ofd_last_error(void);    /* thread-local, never NULL */

/* ---------------------------------------------------------------- forward splat (SS) ------
 * Tensors are contiguous NCHW fp32.  in: (B,C,H,W), flow: (B,2,H,W) with channel 0 = x
 * displacement, out: (B,C,H/scale,W/scale).  out is fully overwritten (no zero-fill needed).
 * radius: half-width in source pixels of the window an output tile scans (>= the expected
 * max |flow| + 1; samples that land farther are still handled, through a slower list pass).
 * workspace: ofd_splat_workspace_bytes(B,H,W) bytes. */
size_t ofd_splat_workspace_bytes(int B, int H, int W);
int ofd_splat_fwd(const float* in, const float* flow, float* out, int B, int C, int H, int W,
                  int scale, int offset_x, int offset_y, int radius,
                  void* workspace, size_t workspace_bytes, void* stream);
/* optional debug output of softsplat_out's integer corner (x0,y0) per source pixel, int32
 * (B,H,W,2), -2^30 where the sample is skipped (non-finite).  Used by the index parity tests. */
int ofd_splat_corners(const float* flow, int32_t* corners, int B, int H, int W,
                      int scale, int offset_x, int offset_y, void* stream);
int ofd_splat_bwd_in(const float* flow, const float* outgrad, float* ingrad, int B, int C, int H, int W,
                     int scale, int offset_x, int offset_y, void* stream);
int ofd_splat_bwd_flow(const float* in, const float* flow, const float* outgrad, float* flowgrad,
                       int B, int C, int H, int W, int scale, int offset_x, int offset_y, void* stream);

/* All L*L offsets of a scale-L splat at once (the photometric pyramid of flow_learner.py:159-206 evaluates every offset of 10
 * levels on the same image and flow).  T: (B, C, L*(H/L), L*(W/L)) with
 *     T[n, c, L*cy + b, L*cx + a] = ofd_splat_fwd(..., scale = L, offset_x = a, offset_y = b)[n, c, cy, cx].
 * One scale-1 splat of the pixels whose targets avoid the reference's border branches for every offset + one separable
 * tent filter of half-width L + a scatter of the remaining pixels with the reference's own remap; L in 1..16.
 * bwd: ingrad (B,C,H,W) and / or flowgrad (B,2,H,W) (either may be NULL) = the sums over (a, b) of ofd_splat_bwd_in /
 * ofd_splat_bwd_flow with the matching slices of dT.  Workspace: ofd_splat_pyramid_workspace_bytes. */
size_t ofd_splat_pyramid_workspace_bytes(int B, int C, int H, int W);
int ofd_splat_pyramid_fwd(const float* in, const float* flow, float* T, int B, int C, int H, int W, int L, int radius,
                          void* workspace, size_t workspace_bytes, void* stream);
int ofd_splat_pyramid_bwd(const float* in, const float* flow, const float* dT, float* ingrad, float* flowgrad,
                          int B, int C, int H, int W, int L, void* workspace, size_t workspace_bytes, void* stream);

/* Photometric loss of one pyramid level on two ofd_splat_pyramid_fwd results in "soft" form (flow_learner.py:176-190):
 * Tin, Ttg: (B, C+1, Ht, Wt), last channel = splatted weight.  filled = Tin_c / (w + 1e-7) where w > 0, NaN elsewhere
 * (fill_holes_nan, WP:273-276); tgt = Ttg_c / (w_t + 1e-7); Charbonnier sqrt(d^2 + 1e-6) (WP:278-287) over the pairs without NaN,
 * per offset: sums / counts [L][L] (row = offset_y, col = offset_x), fp64.  The level loss is mean(sums / counts).
 * bwd: dTin = d(level loss)/dTin x gscale[0] (device scalar). */
int ofd_pyramid_charbonnier_fwd(const float* Tin, const float* Ttg, double* sums, double* counts, int B, int C, int Ht, int Wt,
                                int L, void* stream);
int ofd_pyramid_charbonnier_bwd(const float* Tin, const float* Ttg, const double* counts, const float* gscale, float* dTin,
                                int B, int C, int Ht, int Wt, int L, void* stream);

/* warp_forward_flow pieces (WP:121-156).
 * prep : first (B,C,H,W) -> ten_in (B,C+1,H,W) = cat(nan_to_zero(first) * w, w), w = 0 where any
 *        channel of the pixel is NaN else 1.  square != 0 squares the values (get_variance).
 * holes: splat result (B,C+1,Ho,Wo) -> img (B,C,Ho,Wo); mode 0 = raw sums ("linear_unn"),
 *        mode 1 = divide by (weight + 1e-7) ("linear"); set_nans != 0 puts NaN where weight <= 0. */
int ofd_warp_prep(const float* first, float* ten_in, int B, int C, int H, int W, int square, void* stream);
int ofd_warp_holes(const float* splat, float* img, int B, int C, int Ho, int Wo, int mode, int set_nans, void* stream);

/* ------------------------------------------------------------ grid_sample warp (WP:95-119) -
 * second: (B,C,H,W), flow: (B,2,H,W) exactly as handed to the reference's warp(mode='backward'):
 * after its flip(1) channel 1 displaces x and channel 0 displaces y (WP:105-106).
 * out, mask: (B,C,H,W); mask may be NULL.  bilinear, zeros padding, align_corners=True. */
int ofd_grid_warp_fwd(const float* second, const float* flow, float* out, float* mask,
                      int B, int C, int H, int W, void* stream);
/* Backward of the call above (what autograd derives for WP:95-119; the thresholded mask has no gradient).
 * grad_second (B,C,H,W): bilinear scatter of grad_out to the forward's four corners (the splat kernel on grid_sample's
 *   coordinates; needs a workspace of ofd_splat_workspace_bytes(B,H,W); radius as in ofd_splat_fwd, >= the largest |flow|
 *   keeps the far-corner list empty).  NULL: skipped.
 * grad_flow (B,2,H,W): ATen's grid gradient chained through the reference's normalisation (its (W-1)/2 and 2/(W-1) factors
 *   are applied in the reference's order); channel 1 receives d/dx.  NULL: skipped (second may then be NULL). */
int ofd_grid_warp_bwd(const float* second, const float* flow, const float* grad_out, float* grad_second, float* grad_flow,
                      int B, int C, int H, int W, int radius, void* workspace, size_t workspace_bytes, void* stream);
/* integer north-west corner (ix_nw, iy_nw) per pixel, int32 (B,H,W,2): index parity tests. */
int ofd_grid_warp_corners(const float* flow, int32_t* corners, int B, int H, int W, void* stream);

/* ------------------------------------------------------------------- diffusion elementwise -
 * x_t, model_out, noise, out: (B,C,H,W) fp32 with n_per_sample = C*H*W.  Coefficients are
 * per-sample fp32 arrays of length B gathered by the host from the schedule (DD:422-425). */
int ofd_q_sample(const float* x0, const float* noise, const float* sqrt_ac, const float* sqrt_1mac,
                 float* out, int B, size_t n_per_sample, void* stream);
/* x_{t-1} = c1*clamp(model_out) + c2*x_t + sigma*noise (sigma = exp(0.5*logvar); pass noise=NULL
 * or sigma=0 at t == 0).  x_start (clamped model_out) is written when non-NULL. */
int ofd_ddpm_update(const float* x_t, const float* model_out, const float* noise,
                    const float* coef1, const float* coef2, const float* sigma,
                    float* out, float* x_start, int B, size_t n_per_sample, void* stream);
/* DDIM (DD:750-767): x0 = clamp(model_out); eps = (sr*x_t - x0)/srm1;
 * out = x0*sqrt_an + c*eps + sigma*noise; last != 0 returns x0. */
int ofd_ddim_update(const float* x_t, const float* model_out, const float* noise,
                    const float* sqrt_recip_ac, const float* sqrt_recipm1_ac,
                    const float* sqrt_alpha_next, const float* c, const float* sigma, int last,
                    float* out, float* x_start, int B, size_t n_per_sample, void* stream);
/* sum and count over positions where neither pred nor target is NaN of (pred-target)^2.
 * result (device): ofd_nan_mse_result_doubles() doubles -- [0] sum, [1] count, the rest scratch (per-workgroup partial sums, added
 * in a fixed order: the same inputs give the same sum bit for bit). */
size_t ofd_nan_mse_result_doubles(void);
int ofd_nan_mse_sum(const float* pred, const float* target, size_t n, double* result, void* stream);
/* backward of result[0] / result[1] (the nanmean): dpred = 2 (pred - target) * gout[0] / result[1] on finite pairs, else 0;
 * result is what ofd_nan_mse_sum left, gout a device scalar */
int ofd_nan_mse_grad(const float* pred, const float* target, size_t n, const double* result, const float* gout,
                     float* dpred, void* stream);

/* ---------------------------------------------------------------- optimiser (FD:131-134) ---
 * torch.optim.Adam(lr, weight_decay) semantics (L2 in the gradient) preceded by
 * clip_grad_norm_(max_norm) (exp_base.py:192,205), multi-tensor, no host synchronisation.
 * table: device array of {float* param; const float* grad; float* exp_avg; float* exp_avg_sq;
 * uint64 numel}; tasks: device arrays (tensor index, chunk index) with ofd_adam_chunk() elements
 * per chunk.  sqnorm_acc (n_tasks + 1 doubles: [0] the squared norm, then one partial sum per task, added in a
 * fixed order -- the same gradients give the same clip coefficient bit for bit), clip_coef (1 float) and optional
 * total_norm are device scratch/outputs.  step counts from 1.  max_norm <= 0 disables clipping. */
int ofd_adam_chunk(void);
int ofd_adam_step(const void* table, const unsigned* task_tensor, const unsigned* task_chunk, int n_tasks,
                  double* sqnorm_acc, float* clip_coef, float* total_norm, float max_norm, float lr,
                  float beta1, float beta2, float eps, float weight_decay, int step, void* stream);

/* ------------------------------------------------------------------------ UNet (DD:272-417) -
 * Handle-based executor of the whole forward: one call runs every kernel of the network on
 * `stream`.  Parameters keep the reference's state-dict names. */
typedef struct ofd_unet ofd_unet;

typedef struct {
    int dim;            /* 64 (must be a multiple of 64) */
    int channels;       /* UNet input channels = x channels + cond channels (5 or 9) */
    int out_dim;        /* 2 */
    int eps_mode;       /* 0: eps 1e-5 everywhere (reference precision 32);
                           1: per-site eps the reference uses under bf16 autocast (DD:107,122) */
    int no_time;        /* 0: Unet(time_in=True) (the diffusion UNet).  1: Unet(time_in=False) (DD:306-324, FD:110 with
                           is_diffusion=False; flow_learner.py:93-98): no time MLP, ResnetBlocks without scale/shift
                           (DD:192-195), the `t` argument of the forward calls is ignored and may be NULL */
} ofd_unet_config;

int ofd_unet_create(const ofd_unet_config* cfg, ofd_unet** out);
void ofd_unet_destroy(ofd_unet* u);
/* number of parameter tensors, and name / element count of the i-th (reference state-dict order) */
int ofd_unet_num_params(const ofd_unet* u);
const char* ofd_unet_param_name(const ofd_unet* u, int i);
size_t ofd_unet_param_numel(const ofd_unet* u, int i);
int ofd_unet_param_shape(const ofd_unet* u, int i, int* dims4);   /* returns ndim (1..4) */
/* copy fp32 parameter i from a device buffer (reference layout, e.g. OIHW) into the engine */
int ofd_unet_set_param(ofd_unet* u, int i, const float* dev_src, size_t numel, void* stream);
/* weight standardisation + bf16 re-layout of everything set so far; call after set_param */
/* Zero-copy alternative to ofd_unet_set_param: the executor reads its parameters from the caller's flat fp32
 * device buffer (parameter i at floats [ofd_unet_param_offset(i), +numel), ofd_unet_param_floats in all; 16-byte
 * aligned, must outlive the handle or the next bind).  Call ofd_unet_prepare again whenever its contents change. */
int ofd_unet_bind_param_buffer(ofd_unet* u, float* dev_params, size_t floats);
int ofd_unet_prepare(ofd_unet* u, void* stream);
size_t ofd_unet_workspace_bytes(const ofd_unet* u, int B, int H, int W);
/* x: (B,Cx,H,W) fp32, cond: (B,Cc,H,W) fp32 or NULL (Cx+Cc == channels), t: (B,) int64,
 * out: (B,out_dim,H,W) fp32.  H and W must be multiples of 8.
 * Memory of `out` (and of every device pointer of this header): ordinary device memory of the GPU the stream belongs to, as
 * hipMalloc / torch's caching allocator return it (coarse-grained).  When H*W is a multiple of 128 the final 1x1 conv (DD:361) is
 * computed on the tile of final_res_block's res_conv and reaches `out` as hardware fp32 atomic adds onto a buffer the call zeroes first
 * (two addends per element, order-free); fine-grained / host-coherent / managed allocations, where such atomics are not guaranteed, are
 * not supported for `out` -- ofd_unet_set_debug_taps(u, 1) selects the unfused final conv (plain stores) for such a buffer. */
int ofd_unet_forward(ofd_unet* u, const float* x, int Cx, const float* cond, int Cc, const int64_t* t,
                     float* out, int B, int H, int W, void* workspace, size_t workspace_bytes, void* stream);
/* debug: copy a named intermediate of the LAST forward (e.g. "init_conv", "downs.0.0",
 * "mid_attn") as NCHW fp32 into dst (numel checked).  Returns OFD_ERR_ARG for unknown names. */
/* hipGraph replay of ofd_unet_forward (launch-bound regimes: small images, the 1000-step sampling loop).  When enabled,
 * the inputs are copied to fixed staging buffers inside the workspace, the first call of a (workspace, shape, stream)
 * configuration runs eagerly, the second is captured into a graph and every later one is a single hipGraphLaunch.
 * Bit-identical results.  Ignored while profiling is on (per-kernel events need individual launches). */
int ofd_unet_set_graph(ofd_unet* u, int enabled);
/* Two half-batch forwards on two streams (even B >= 2; ignored while profiling or graph replay is on).  enabled: 1 on, 0 off, -1 (the
 * handle's default) on for batches of at least 2^21 pixels in all -- at the BASELINE size the step is 1.3 % faster, small problems are
 * launch-bound and keep one stream.  Samples are independent
 * in every kernel of the network (GroupNorm, LinearAttention and attention are per sample: DD:172-268), so samples [0, B/2) run on
 * `stream` and samples [B/2, B) run the same launch sequence on a second, library-owned stream that starts `offset_blocks` blocks
 * (ResnetBlock / attention block granularity; < 0 keeps the current value) behind the first: the HBM-bound kernels of one half share
 * the chip with the MFMA-bound kernels of the other.  `stream` waits for both halves before the call's successors run.  Results are
 * bit-identical to the one-stream forward per sample.  ofd_unet_workspace_bytes already covers the two half contexts. */
int ofd_unet_set_split_streams(ofd_unet* u, int enabled, int offset_blocks);
int ofd_unet_read_tap(ofd_unet* u, const char* name, float* dst, size_t numel, void* stream);
/* debug: materialise EVERY named intermediate of the inference forward.  Off (default), final_res_block's 64-channel output is not:
 * the final 1x1 conv (DD:361) then rides on the tile of its producer (out_dim 2, H*W a multiple of 128) and "final_res_block" is
 * not a tap.  The two forms sum the 64 products of a pixel in different orders (equal to fp32 rounding). */
int ofd_unet_set_debug_taps(ofd_unet* u, int enabled);
/* deterministic backward (default: the environment variable OFD_DETERMINISTIC, else off).  The backward accumulates parameter
 * gradients from many workgroups with float atomics, whose order -- and so the last bits of the sums -- changes from run to run
 * (torch's cuDNN weight gradients behave the same way in the reference).  Enabled, every such accumulation goes through a 64-bit
 * fixed-point shadow (csrc/det.h: value * 2^38, integer atomics, order-independent) that is flushed into the fp32 gradient
 * before anyone reads it: two backward passes over the same tape give bit-identical gradients.  Costs 8 bytes per parameter and
 * accumulator of extra memory and a few small launches per layer.  ofd_unet_deterministic_misses: accumulations (since creation)
 * that found no shadow and fell back to float atomics (0 for the UNet of this library; -1 on error). */
int ofd_unet_set_deterministic(ofd_unet* u, int enabled);
long ofd_unet_deterministic_misses(ofd_unet* u);
/* per-kernel-class device time of forwards run with profiling enabled (HIP events on the
 * stream the kernels are launched on).  classes: see ofd_unet_prof_name(). */
int ofd_unet_set_profiling(ofd_unet* u, int enabled);
int ofd_unet_prof_count(const ofd_unet* u);
const char* ofd_unet_prof_name(const ofd_unet* u, int i);
/* resolves the recorded events (host-synchronises) -> ms, launches and flops/bytes summed since last reset */
int ofd_unet_prof_read(ofd_unet* u, int i, double* ms, long long* launches, double* flops, double* bytes);
int ofd_unet_prof_reset(ofd_unet* u);
/* optional: append one CSV row per launch (class,label,ms,flops,bytes) to `path` whenever the
 * events are resolved; NULL/"" disables */
int ofd_unet_prof_dump_path(ofd_unet* u, const char* path);

/* ---------------------------------------------------------------- single ops (parity tests) -
 * NHWC bf16 activations (uint16_t* = raw bf16 bits).  These are the kernels the executor
 * launches, exposed one by one so that each can be checked against the oracle. */
typedef struct {
    const void* src;        /* bf16 NHWC */
    int channels;           /* channels taken from this source (multiple of 64; 7x7: 16) */
    int src_channels;       /* channel count (pixel stride) of the source tensor */
    int ch_offset;          /* first channel taken */
    int upsample;           /* 1: source is (H/2,W/2), read with nearest x2 (DD:91) */
    int unshuffle;          /* 1: source is (2H,2W); this entry is sub-pixel (p1,p2) of DD:97 */
    int p1, p2;
} ofd_conv_src;

typedef struct {
    int B, H, W;            /* OUTPUT spatial size */
    int ksize;              /* 1, 3 or 7; 2 = one phase of an up-sampled 3x3 (see up2_phase) */
    int n_src;              /* 1..4 */
    ofd_conv_src src[4];
    int Cout;               /* multiple of 64 */
    const void* weight;     /* bf16, engine layout (ofd_conv_weight_layout) */
    const float* bias;      /* fp32 [Cout] or NULL */
    const float* in_scale;  /* optional prologue: y = silu(x*in_scale[b][c] + in_shift[b][c]) */
    const float* in_shift;  /* fp32 [B][Cin] */
    const void* residual;   /* optional epilogue add: bf16 NHWC (B,H,W,Cout) */
    const void* res_act;    /* optional epilogue add of silu(r*res_scale[b][c] + res_shift[b][c]) */
    const float* res_scale;
    const float* res_shift;
    void* out;              /* bf16 NHWC (B,H,W,Cout) */
    float* gn_partial;      /* optional: GroupNorm partial sums [b][8 groups][8x32 tile][4 wave slots][Cout/64][2] (sum, sum
                               of squares of the stored values; group-major so that ofd_gn_finalize reads one contiguous
                               run per (sample, group)), ofd_conv_gn_partial_count floats */
    void* out2;             /* split output (data gradients of a conv over two concatenated sources): channels */
    const void* residual2;  /*   [0, split) -> out / residual with pixel stride split, [split, Cout) -> out2 / residual2 */
    int split;              /*   with stride Cout - split; multiple of 64, 0 = off */
    int up2_phase;          /* ksize 2 only: 1 + py*2 + px.  Upsample(x2, nearest) + 3x3 (DD:89-93) as four 2x2 convs on
                               the LOW-RES source: B,H,W are the low-res size, `out` is the (2H, 2W) tensor and this
                               launch writes its pixels (2y+py, 2x+px); weights from ofd_conv_upsample_phase_weight_prep
                               (+ (up2_phase-1) * 4*Cin*Cout elements).  5 = all four phases in one launch (`weight` =
                               the base of the four kernels): the phases of a pixel tile run together on one XCD */
} ofd_conv_args;

int ofd_conv_forward(const ofd_conv_args* a, void* stream);
/* the same 3x3 convolution with every 2x2 block of output pixels SUMMED in the epilogue: `out` (and `residual`, added there) is the
 * (H/2, W/2) tensor.  This is the data gradient of Upsample(x2, nearest) + conv (DD:89-93) taken straight into the low-resolution
 * source's gradient: a = the data-gradient configuration above (source dY at (H, W), tap-flipped transposed weights).  H, W even,
 * Cout a multiple of 64 and not the 64 -> 64 case, no split output / GroupNorm statistics / activation residual. */
int ofd_conv_forward_pool2(const ofd_conv_args* a, void* stream);
size_t ofd_conv_gn_partial_count(int B, int H, int W, int Cout);   /* floats */
/* OIHW fp32 -> engine bf16 layout [tap][Cin/8][Cout][8]; eps < 0: plain conv, else weight
 * standardisation with that eps (DD:109-112).  cin_pad: Cin rounded up (7x7: 16).
 * unshuffle != 0: Cin index is c*4+p1*2+p2 (DD:97) and is regrouped as (p1,p2) major. */
size_t ofd_conv_weight_elems(int Cout, int Cin_pad, int ksize);
int ofd_conv_weight_prep(const float* w_oihw, void* w_out, int Cout, int Cin, int Cin_pad, int ksize,
                         float ws_eps, int unshuffle, void* stream);

/* ---------------------------------------------------------------- conv backward (training) --
 * data gradient  : run ofd_conv_forward on dY (source, Cout channels) with the weights produced by
 *                  ofd_conv_dgrad_weight_prep (tap-flipped, in/out transposed); output has Cin channels;
 *                  ofd_grad_scatter applies the adjoint of concat / up-sample (mode 1) / unshuffle (mode 2).
 * weight gradient: ofd_conv_wgrad accumulates dW[tap][Cin][Cout] (fp32, zeroed by the caller) from the
 *                  forward's input sources and dY; ofd_conv_wgrad_finish converts to the OIHW parameter
 *                  gradient, through weight standardisation when ws_eps >= 0 (DD:109-112).
 * bias gradient  : ofd_channel_sum (out[C] += column sums; zero it first). */
/* 3x3 OIHW fp32 -> the four collapsed 2x2 kernels of the phase decomposition, 4 x [4 taps][Cin/8][Cout][8] bf16 */
int ofd_conv_upsample_phase_weight_prep(const float* w_oihw, void* w_out, int Cout, int Cin, void* stream);
int ofd_conv_dgrad_weight_prep(const void* w_fwd, void* w_t, int Cout, int Cin, int ksize, void* stream);
int ofd_conv_wgrad(const ofd_conv_args* fwd, const void* dy, float* dw_acc, void* stream);
int ofd_conv7_wgrad(const void* x16, const void* dy, float* dw_acc, int B, int H, int W, void* stream);
int ofd_conv_wgrad_finish(const float* dw_acc, const float* w_oihw, float* dst_oihw, int Cout, int Cin, int Cin_pad,
                          int ksize, float ws_eps, int unshuffle, int accumulate, void* stream);
int ofd_grad_scatter(const void* D, int Ctot, int ch_off, void* dst, int C, int B, int H, int W, int mode,
                     int p1, int p2, int accumulate, void* stream);
int ofd_channel_sum(const void* dy, float* out, size_t npix, int C, void* stream);

/* --------------------------------------------------- Block / norm backward (training) --------
 * ofd_gn_silu_backward: backward of out = SiLU(GroupNorm(h) * (scale+1) + shift) (DD:181-187) given
 *   g = dL/dout.  (a, s) is the folded per-(sample, channel) affine of the forward, stats = [B][8][{mean,
 *   rstd}].  Writes dh; ADDS into dgamma / dbeta (fp32 [C]); writes the scale|shift gradient rows into dss
 *   (same indexing as ss; NULL when the block has no time-embedding input).  dconv_bias (may be NULL):
 *   ADDS sum_pixels dh, the bias gradient of the convolution that produced h.  workspace:
 *   ofd_gn_bwd_workspace_floats floats.
 * ofd_layernorm_c_backward: DD:116-125; dx written (or added when accumulate != 0), dg ADDED.
 * ofd_final_conv_backward: DD:361; dx written, dw / db ADDED. */
size_t ofd_gn_bwd_workspace_floats(int B, int H, int W, int C);
int ofd_gn_silu_backward(const void* g, const void* h, const float* a, const float* s, const float* stats,
                         const float* gamma, const float* beta, const float* ss, int ss_stride, int ss_offset,
                         void* dh, float* dgamma, float* dbeta, float* dss, float* dconv_bias, float* workspace,
                         int B, int H, int W, int C, void* stream);
int ofd_affine_silu(const void* h, const float* a, const float* s, void* out, int B, int H, int W, int C, void* stream);
int ofd_layernorm_c_backward(const void* x, const float* g, const void* dy, void* dx, float* dg, size_t npix, int C,
                             float eps, int accumulate, void* stream);
int ofd_final_conv_backward(const void* x, const float* w, const float* dy, void* dx, float* dw, float* db,
                            int B, int H, int W, int C, int out_dim, void* stream);

/* --------------------------------------------------- attention cores, forward + backward -------
 * qkv / dqkv: [B][n][384] bf16 (q | k | v, 4 heads x 32); out / dout: [B][n][128] bf16.
 * LinearAttention core (DD:229-242): forward keeps ctx [B*4][32][32] and ml [B*4][64] (max and 1/sum of
 *   the softmax over pixels) for the backward.  workspace: ofd_la_workspace_floats / ofd_la_bwd_workspace_floats.
 * Softmax attention (DD:256-268): forward keeps lse [B*4][n]; backward needs delta scratch [B*4][n]. */
size_t ofd_la_workspace_floats(int B, int n);
size_t ofd_la_bwd_workspace_floats(int B, int n);
int ofd_linear_attention_core(const void* qkv, void* out, float* ctx, float* ml, float* workspace, int B, int n, void* stream);
int ofd_linear_attention_core_backward(const void* qkv, const void* dout, const float* ctx, const float* ml, void* dqkv,
                                       float* workspace, int B, int n, void* stream);
int ofd_flash_attention(const void* qkv, void* out, float* lse, int B, int n, void* stream);
int ofd_flash_attention_backward(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv,
                                 float* delta, int B, int n, void* stream);

/* --------------------------------------------------- UNet training step -----------------------
 * What autograd does for the reference's training_step (flow_diffuser.py:218-235 ->
 * denoising_diffusion.py:823-891 -> Unet.forward DD:363-417), as two calls:
 *   ofd_unet_train_forward: the forward with every intermediate kept in `workspace`
 *       (ofd_unet_train_workspace_bytes; the workspace must stay untouched until the backward);
 *   ofd_unet_backward: dout = dL/d(out) [B][out_dim][H][W] fp32 -> gradients of all parameters, written
 *       (not accumulated) into the flat fp32 buffer bound with ofd_unet_bind_grad_buffer.  Parameter i
 *       lives at floats [ofd_unet_param_offset(i), +numel) of that buffer (ofd_unet_param_floats in all).
 *       on_ready (may be NULL) is called on the host, in backward order, with each [begin, end) float
 *       range of the buffer as soon as the launches that produce it are enqueued on `stream` -- the
 *       hook for overlapping the data-parallel all-reduce with the rest of the backward. */
typedef void (*ofd_grad_ready_fn)(size_t begin, size_t end, void* user);
size_t ofd_unet_train_workspace_bytes(ofd_unet* u, int B, int H, int W);
size_t ofd_unet_param_floats(const ofd_unet* u);
size_t ofd_unet_param_offset(const ofd_unet* u, int i);
int ofd_unet_bind_grad_buffer(ofd_unet* u, float* dev_grads, size_t floats);
int ofd_unet_train_forward(ofd_unet* u, const float* x, int Cx, const float* cond, int Cc, const int64_t* t,
                           float* out, int B, int H, int W, void* workspace, size_t workspace_bytes, void* stream);
int ofd_unet_backward(ofd_unet* u, const float* dout, ofd_grad_ready_fn on_ready, void* user, void* stream);

/* --------------------------------------------------- forward building blocks of the UNet executor ----
 * ofd_layernorm_c (DD:116-125): out = LN_channels(x) * g (+ residual), NHWC bf16, npix rows of C channels.
 * ofd_time_mlp (DD:139-151, 319-324): sinusoidal embedding -> Linear -> GELU -> Linear; also SiLU(temb).
 * ofd_gn_finalize (DD:181-185): per-(tile, wave, 8-channel) partial sums of a conv epilogue
 *   (ofd_conv_gn_partial_count floats) -> the folded GroupNorm affine (a, s) per (sample, channel), and
 *   optionally the statistics [B][8][{mean, rstd}] the backward needs. */
int ofd_layernorm_c(const void* x, const float* g, const void* residual, void* out, size_t npix, int C, float eps, void* stream);
int ofd_time_mlp(const int64_t* t, const float* w1, const float* b1, const float* w2, const float* b2,
                 float* temb, float* temb_silu, int B, int dim, void* stream);
int ofd_gn_finalize(const float* partial, int B, int H, int W, int C, const float* gamma, const float* beta,
                    const float* ss, int ss_stride, int ss_offset, float* a_out, float* s_out, float* stats_out, void* stream);

/* Batched training augmentation in one pass (replaces the per-sample torchvision pipeline of augmentation.py:6-76 behind
 * FlowDiffuser.preprocess(aug=True), flow_diffuser.py:137-138).  img / tgt (B,3,H,W), flow (B,2,H,W) fp32 NCHW; params (B,16) fp32:
 * 0 jitter on, 1 brightness, 2 contrast, 3 saturation, 4 grayscale on, 5 blur on, 6 sigma, 7 h-flip, 8 v-flip, 9 crop on,
 * 10 oy, 11 ox, 12 ch, 13 cw (crop window origin / size as fractions of the image; 1, 1 without a crop).  means_ws: B*2 x 8 bytes of device scratch.
 * reference_semantics == 0 (default of the plugin): geometrically consistent flow -- a flip negates the component along the flipped
 * axis, a crop divides each component by its axis' window fraction.  != 0: the reference's own arithmetic on the flow channels --
 * flips negate the other channel (augmentation.py:37-45), the crop multiplies channel 0 by ch and channel 1 by cw (augmentation.py:47-48). */
int ofd_augment(const float* img, const float* tgt, const float* flow, const float* params, void* means_ws, float* out_img,
                float* out_tgt, float* out_flow, int B, int H, int W, int reference_semantics, void* stream);
/* the (B, 16) table above from (B, 14) uniform draws in [0, 1) (the caller's RNG: torch's device generator in the plugin), one launch:
 * columns 0, 4, 5, 7, 8, 9 are the decisions u < p (p = 0.4, 0.1, 0.2, 0.3, 0.3, 0.15: augmentation.py:8-35 of the reference), 1-3 = 1 +- 0.1,
 * 6 = max(u / 2, 0.05), the crop window from RandomResizedCrop's scale (0.8, 1.0) and log-uniform ratio (0.9, 1.1). */
int ofd_augment_table(const float* uniforms, float* params, int B, void* stream);

/* the four statistics FlowDiffuser.training_step logs for cond and for flow (flow_diffuser.py:218-235: torch.min, torch.max, torch.mean,
 * torch.mean(torch.std(x, dim=0))) of x (B, n_per_sample) fp32 in one pass: out4 = {min, max, mean, mean over elements of the unbiased standard
 * deviation across the batch} (device); ws: ofd_batch_stats_ws_doubles() doubles of device scratch (partial sums, added in a fixed order). */
size_t ofd_batch_stats_ws_doubles(void);
int ofd_batch_stats(const float* x, int B, size_t n_per_sample, double* ws, float* out4, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* OFD_H */
