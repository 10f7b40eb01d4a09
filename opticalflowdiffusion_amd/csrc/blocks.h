// Internal launchers of the UNet building-block kernels (blocks.hip, conv_igemm.hip).
#pragma once
#include "common.h"

namespace ofd {

struct MlpDesc {            // one ResnetBlock's time-embedding Linear (DD:193-196)
    const float* weight;    // [n_out][tdim]
    const float* bias;      // [n_out]
    int n_out;              // 2 * Cout
    int offset;             // into the per-sample scale/shift row
};

struct FcFuse { const float* w; const float* b; float* out; };      // ConvParams::fc_* (the UNet's final 1x1 conv on the final res_conv's tile)
bool conv_fc_fuse_supported(const ofd_conv_args* a, int out_dim);
int conv_forward_impl(const ofd_conv_args* a, hipStream_t s, int cout0 = 0, int pool2 = 0, const bf16_t* residual_b = nullptr, const FcFuse* fc = nullptr);      // cout0: see ConvParams::cout0 (honoured by the streaming 1x1 kernel only
                                                                                   // for 64 -> 384 with cout0 = 128; everything else computes all channels)
bool conv_pool2_supported(const ofd_conv_args* a);
bool conv_residual_b_supported(const ofd_conv_args* a);                            // residual_b: ConvParams::residual_b (3x3 through conv_wp.hip only)                                 // pool2: ConvParams::pool2 (3x3 through conv_wp.hip only)
// conv backward (conv_bwd.hip)
int k_wt_transpose(const bf16_t* w, bf16_t* wt, int taps, int Cin, int Cout, hipStream_t s);
int k_conv_wgrad(const ofd_conv_args* a, const bf16_t* dy, float* dw, hipStream_t s, float* dbias = nullptr);
int k_conv7_wgrad(const bf16_t* x16, const bf16_t* dy, float* dw, int B, int H, int W, hipStream_t s, float* dbias = nullptr, int channels = 16);
int k_channel_sum(const bf16_t* dy, float* out, size_t npix, int C, hipStream_t s);
int k_wgrad_finish(const float* acc, const float* w_raw, float* dst, int Cout, int Cin, int Cin_pad, int ksize, float ws_eps, int unshuffle,
                   int accumulate, hipStream_t s);
int k_grad_scatter(const bf16_t* D, int Ctot, int ch_off, bf16_t* dst, int C, int B, int H, int W, int mode, int p1, int p2, int accumulate,
                   hipStream_t s);

struct ofd_weight_prep_desc {      // one conv of conv_weight_prep_batched_kernel / wt_transpose_batched_kernel
    const float* w;                // OIHW fp32 (prep) | prepared bf16 weights (transpose: reinterpret)
    void* out;
    int Cout, Cin, Cin_pad, ksize;
    float ws_eps;
    int unshuffle;
    int block0;                    // first block of this conv in the batched grid
};
int k_conv_weight_prep_batched(const ofd_weight_prep_desc* d_descs, int n, int total_blocks, hipStream_t s);
int k_wt_transpose_batched(const ofd_weight_prep_desc* d_descs, int n, int total_blocks, hipStream_t s);
int k_pack_input(const float* x, int Cx, const float* cond, int Cc, bf16_t* out, int B, int H, int W, hipStream_t s, int cpad = 16);
int k_time_mlp(const int64_t* t, const float* w1, const float* b1, const float* w2, const float* b2, float* temb,
               float* temb_silu, int B, int dim, hipStream_t s);
int k_block_mlp(const float* temb_silu, const MlpDesc* descs, int n_desc, float* ss, int B, int tdim, int ss_stride, hipStream_t s);
int k_gn_finalize(const float* partial, int B, int H, int W, int C, const float* gamma, const float* beta, const float* ss,
                  int ss_stride, int ss_offset, float* a_out, float* s_out, hipStream_t s, float* stats_out = nullptr);
int k_resblock_out(const bf16_t* h, const float* a, const float* sft, const bf16_t* x, bf16_t* out, int B, int H, int W, int C, hipStream_t s);
int k_layernorm_c(const bf16_t* x, const float* g, const bf16_t* res, bf16_t* out, size_t npix, int C, float eps, hipStream_t s);
int la_parts(int B, int n);
int k_linear_attention_core(const bf16_t* qkv, float* partial, float* ctx, bf16_t* out, int B, int n, hipStream_t s, float* ml_out = nullptr,
                            const bf16_t* wo = nullptr, const float* bo = nullptr, bf16_t* o2 = nullptr, int C = 0, const bf16_t* xn = nullptr,
                            const bf16_t* wq = nullptr);
int la_fused_blocks(int n, int B);
int la_fwd_parts(int B, int n);
int k_linear_attention_fused_train(const bf16_t* x, const bf16_t* wq, const bf16_t* wkv, const bf16_t* woutp, const float* bias, const float* g_pre,
                                   const float* g2, float* partial, bf16_t* ctxfrag, float* ctx, float* ml, bf16_t* xn, bf16_t* qkv, bf16_t* o2,
                                   bf16_t* y, int B, int n, int C, float eps_pre, float eps_post, hipStream_t s);
int k_la_weight_prep(const float* wqkv, const float* g, const float* wout, bf16_t* wq, bf16_t* wkv, bf16_t* woutp, int C, hipStream_t s);
int k_linear_attention_fused(const bf16_t* x, const bf16_t* wq, const bf16_t* wkv, const bf16_t* woutp, const float* bias, const float* g2,
                             float* partial, bf16_t* ctxfrag, bf16_t* y, int B, int n, int C, float eps_pre, float eps_post, hipStream_t s);
int k_flash_attention(const bf16_t* qkv, bf16_t* out, int B, int n, hipStream_t s, float* lse = nullptr);
int k_final_conv(const bf16_t* x, const float* w, const float* bias, float* out, int B, int H, int W, int C, int out_dim, hipStream_t s);
int k_nhwc_to_nchw(const bf16_t* x, float* out, int B, int H, int W, int C, hipStream_t s);
int k_nchw_to_nhwc(const float* x, bf16_t* out, int B, int H, int W, int C, hipStream_t s);

// training-path kernels (train_ops.hip)
int k_affine_silu(const bf16_t* h, const float* a, const float* s, bf16_t* out, int B, int H, int W, int C, hipStream_t st);
size_t gn_bwd_workspace_floats(int B, int H, int W, int C);
int k_gn_silu_backward(const bf16_t* g, const bf16_t* h, const float* a, const float* s, const float* stats, const float* gamma,
                       const float* beta, const float* ss, int ss_stride, int ss_offset, bf16_t* dh, float* dgamma, float* dbeta, float* dss,
                       float* workspace, int B, int H, int W, int C, hipStream_t st, float* dconv_bias = nullptr);
int k_layernorm_c_bwd(const bf16_t* x, const float* gw, const bf16_t* dy, bf16_t* dx, float* dg, size_t npix, int C, float eps, int accumulate,
                      hipStream_t st, const bf16_t* extra = nullptr);
int k_final_conv_bwd(const bf16_t* x, const float* w, const float* dy, bf16_t* dx, float* dw, float* db, int B, int H, int W, int C, int out_dim,
                     hipStream_t st);
int k_block_mlp_bwd(const float* dss, const float* temb_silu, const float* weight, int n_out, int offset, float* dweight, float* dbias, float* dts,
                    int B, int tdim, int ss_stride, hipStream_t st);
int k_time_mlp_bwd(const int64_t* t, const float* temb, const float* dts, const float* w1, const float* b1, const float* w2, float* dw1, float* db1,
                   float* dw2, float* db2, float* scratch, int B, int dim, hipStream_t st);
int k_grad_add(bf16_t* dst, const bf16_t* src, size_t elems, int accumulate, hipStream_t st);

// attention backward (attn_bwd.hip)
size_t la_bwd_workspace_floats(int B, int n);
int k_linear_attention_core_bwd(const bf16_t* qkv, const bf16_t* dout, const float* ctx, const float* ml, bf16_t* dqkv, float* workspace, int B, int n,
                                hipStream_t s, const bf16_t* xn = nullptr, const bf16_t* wt = nullptr, float* dw = nullptr, bf16_t* dxn = nullptr,
                                const bf16_t* wo_fwd = nullptr, const bf16_t* wo_t = nullptr, float* dwo = nullptr, float* dbo = nullptr,
                                const bf16_t* wq = nullptr);
int k_flash_attention_bwd(const bf16_t* qkv, const bf16_t* o, const bf16_t* dout, const float* lse, bf16_t* dqkv, float* delta, int B, int n,
                          hipStream_t s);

}  // namespace ofd
