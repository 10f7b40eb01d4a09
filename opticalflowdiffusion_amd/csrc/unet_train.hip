// Training executor: the forward of unet.hip with every intermediate kept (the tape), and the
// backward that replays the tape in reverse -- what autograd does for the reference's
// training_step (flow_diffuser.py:218-235 -> denoising_diffusion.py:823-891 -> Unet.forward).
// Gradients of the parameters land in ONE flat fp32 buffer laid out like the parameter buffer
// (bound by the caller, so torch's .grad tensors are views of it and the data-parallel all-reduce
// works on contiguous ranges); a host callback reports each range as soon as its launches are
// enqueued, in backward order, so the caller can overlap the collectives with the rest.
// Host code only: kernels live in conv_igemm.hip / conv_bwd.hip / train_ops.hip / attn_bwd.hip.
#include <set>
#include "unet_exec.h"

using namespace ofd;

namespace ofd {

struct Bwd {
    Ctx& c;
    std::set<const void*> written;
    bool has(const Tensor& t) const { return written.count(t.p) != 0; }
    void mark(const Tensor& t) { written.insert(t.p); }
    Tensor stmp(int C, int H, int W) {
        Tensor t;
        t.p = (bf16_t*)c.alloc(c.scratch, c.scratch_used, c.scratch_cap, (size_t)c.B * H * W * C * 2);
        t.C = C; t.H = H; t.W = W;
        return t;
    }
    float* stmpf(size_t n) { return (float*)c.alloc(c.scratch, c.scratch_used, c.scratch_cap, n * 4); }
};

// ---- deterministic backward (det.h) ----------------------------------------------------------------
// uploads `ctx` to the three translation units whose kernels accumulate gradients (stream-ordered)
static int det_upload(const DetCtx* ctx, hipStream_t s) {
    if (det_set_ctx_conv_bwd(ctx, s) != 0 || det_set_ctx_la_core(ctx, s) != 0 || det_set_ctx_train_ops(ctx, s) != 0) {
        set_error("unet_backward: deterministic-mode context upload failed");
        return OFD_ERR_HIP;
    }
    return OFD_OK;
}
// shadows allocated / zeroed, the context of this backward switched on: [d_grads | d_wacc | dts rows]
static int det_begin(Ctx& c, float* dts, size_t n_dts) {
    ofd_unet* u = c.u;
    const size_t need = u->n_param_floats + u->n_wbuf + n_dts;
    if (u->fx_cap < need) {
        if (u->d_fx) (void)hipFree(u->d_fx);
        u->d_fx = nullptr; u->fx_cap = 0;
        if (hipMalloc(&u->d_fx, need * sizeof(long long)) != hipSuccess) { set_error("unet_backward: %zu bytes of fixed-point shadow", need * sizeof(long long)); return OFD_ERR_HIP; }
        u->fx_cap = need;
    }
    if (!u->d_det_miss) {
        if (hipMalloc(&u->d_det_miss, sizeof(unsigned)) != hipSuccess || hipMemset(u->d_det_miss, 0, sizeof(unsigned)) != hipSuccess) {
            set_error("unet_backward: miss counter"); return OFD_ERR_HIP;
        }
    }
    if (hipMemsetAsync(u->d_fx, 0, need * sizeof(long long), c.s) != hipSuccess) { set_error("unet_backward: memset failed"); return OFD_ERR_HIP; }
    DetCtx& h = u->det_host;
    h.on = 1; h.misses = u->d_det_miss;
    h.r[0] = {u->d_grads, u->d_fx, u->n_param_floats};
    h.r[1] = {u->d_wacc, u->d_fx + u->n_param_floats, u->n_wbuf};
    h.r[2] = {dts, u->d_fx + u->n_param_floats + u->n_wbuf, n_dts};
    return det_upload(&h, c.s);
}
// the context off again (the standalone operators of the C-ABI share it)
static int det_end(Ctx& c) {
    static const DetCtx off{};
    return det_upload(&off, c.s);
}
// a conv's fixed-point weight-gradient accumulator -> its fp32 accumulator, before wgrad_finish reads it
static int det_flush_wacc(Ctx& c, const ConvDesc& d) {
    ofd_unet* u = c.u;
    if (!u->deterministic || c.dry) return OFD_OK;
    return k_det_flush(u->d_fx + u->n_param_floats + d.w_off, u->d_wacc + d.w_off, (size_t)d.ksize * d.ksize * d.Cin_pad * d.Cout, c.s);
}

static void fill_args(ofd_conv_args& a, int B, int H, int W, int ksize, int Cout, const std::vector<SrcSpec>& srcs) {
    a.B = B; a.H = H; a.W = W; a.ksize = ksize; a.n_src = (int)srcs.size(); a.Cout = Cout;
    for (size_t i = 0; i < srcs.size(); ++i) {
        a.src[i].src = srcs[i].t.p;
        a.src[i].channels = srcs[i].t.C;
        a.src[i].src_channels = srcs[i].t.C;
        a.src[i].ch_offset = 0;
        a.src[i].upsample = srcs[i].upsample;
        a.src[i].unshuffle = srcs[i].unshuffle;
        a.src[i].p1 = srcs[i].p1;
        a.src[i].p2 = srcs[i].p2;
    }
}

// backward of one convolution: bias and weight gradients into the flat gradient buffer; returns the
// gradient w.r.t. the concatenated (virtual) input [B][H][W][cin] in scratch (+ `add` when given)
// gradient w.r.t. a single same-resolution source goes straight into that source's gradient buffer; over two
// concatenated same-resolution sources the conv epilogue splits its output between their two buffers
static bool plain_src(const SrcSpec& s) { return !s.upsample && !s.unshuffle && s.t.g; }
static bool direct_target(const std::vector<SrcSpec>& srcs) {
    if (srcs.size() == 1) return plain_src(srcs[0]);
    return srcs.size() == 2 && plain_src(srcs[0]) && plain_src(srcs[1]) && srcs[0].t.C % 64 == 0 && srcs[1].t.C % 64 == 0 && srcs[0].t.p != srcs[1].t.p;
}

static Tensor conv_backward(Bwd& b, const std::string& prefix, const std::vector<SrcSpec>& srcs, const bf16_t* dy, int H, int W, bool need_dx,
                            const bf16_t* add, bool bias_done = false, bool to_source = false, const float* in_scale = nullptr,
                            const float* in_shift = nullptr) {
    Ctx& c = b.c;
    ofd_unet* u = c.u;
    Tensor D;
    if (c.rc != OFD_OK) return D;
    const ConvDesc& d = u->convs[u->cindex.at(prefix)];
    const int B = c.B, taps = d.ksize * d.ksize;
    int cin = 0;
    for (auto& s : srcs) cin += s.t.C;
    const double px = (double)B * H * W;
    // bias gradient = column sums of dY: taken from the dY tiles the weight-gradient kernel stages anyway
    // (block convs: the GroupNorm backward already summed dh)
    float* gb = bias_done ? nullptr : u->G(prefix + ".bias");
    const size_t nacc = (size_t)taps * d.Cin_pad * d.Cout;
    float* acc = u->d_wacc + d.w_off;          // zeroed once per backward (run_backward): no memset launch per conv
    (void)nacc;
    c.begin(d.ksize == 3 ? PC_WGRAD3 : PC_WGRAD1, 2.0 * px * d.Cout * (double)d.Cin * taps, px * 2.0 * (d.Cout + cin), prefix + " wgrad");
    if (d.ksize == 7) {
        RUN(k_conv7_wgrad(srcs[0].t.p, dy, acc, B, H, W, c.s, gb, srcs[0].t.C));
    } else {
        ofd_conv_args a{};
        fill_args(a, B, H, W, d.ksize, d.Cout, srcs);
        a.in_scale = in_scale; a.in_shift = in_shift;         // the weight gradient sees SiLU(affine(src)) (block2: act1 from h1)
        RUN(k_conv_wgrad(&a, dy, acc, c.s, gb));
    }
    RUN(det_flush_wacc(c, d));
    RUN(k_wgrad_finish(acc, u->P(d.wname), u->G(d.wname), d.Cout, d.Cin, d.Cin_pad, d.ksize, d.ws_eps, d.unshuffle, 0, c.s));
    c.end();
    if (!need_dx) return D;
    const bool direct = to_source && direct_target(srcs);
    const bf16_t* res_b = nullptr;
    if (direct && srcs.size() == 2) {
        // split epilogue: (+)= into both sources' gradients; `add` (if any) must already be inside them
        if (add) { set_error("conv_backward: split output takes no extra addend"); c.rc = OFD_ERR_STATE; return D; }
        const Tensor &t0 = srcs[0].t, &t1 = srcs[1].t;
        ofd_conv_args a{};
        a.B = B; a.H = H; a.W = W; a.ksize = d.ksize; a.n_src = 1; a.Cout = cin;
        a.src[0].src = dy; a.src[0].channels = d.Cout; a.src[0].src_channels = d.Cout;
        a.weight = u->d_wtbuf + d.w_off;
        a.out = t0.g; a.out2 = t1.g; a.split = t0.C;
        a.residual = b.has(t0) ? t0.g : nullptr;
        a.residual2 = b.has(t1) ? t1.g : nullptr;
        c.begin(d.ksize == 3 ? PC_DGRAD3 : PC_DGRAD1, 2.0 * px * d.Cout * (double)d.Cin * taps, px * 2.0 * (d.Cout + cin), prefix + " dgrad (split)");
        RUN(conv_forward_impl(&a, c.s));
        c.end();
        b.mark(t0); b.mark(t1);
        D = t0; D.p = t0.g;
        return D;
    }
    // Upsample(x2) + 3x3: the data gradient w.r.t. the LOW-resolution source = the 2x2 sum-pool of the gradient w.r.t. the virtual up-sampled
    // tensor; the conv epilogue pools (ConvParams::pool2) and adds into the source's gradient -- no full-resolution tensor, no scatter pass
    if (to_source && !add && srcs.size() == 1 && srcs[0].upsample && srcs[0].t.g && d.ksize == 3) {
        const Tensor& t = srcs[0].t;
        ofd_conv_args a{};
        a.B = B; a.H = H; a.W = W; a.ksize = 3; a.n_src = 1; a.Cout = cin;
        a.src[0].src = dy; a.src[0].channels = d.Cout; a.src[0].src_channels = d.Cout;
        a.weight = u->d_wtbuf + d.w_off;
        a.residual = b.has(t) ? t.g : nullptr;
        a.out = t.g;
        if (conv_pool2_supported(&a)) {
            c.begin(PC_DGRAD3, 2.0 * px * d.Cout * (double)d.Cin * taps, px * 2.0 * d.Cout + px * 0.5 * cin, prefix + " dgrad (pooled)");
            RUN(conv_forward_impl(&a, c.s, 0, 1));
            c.end();
            b.mark(t);
            D = t; D.p = t.g;
            return D;
        }
    }
    if (direct) {
        // (+)= into the source's gradient: the conv epilogue adds `residual`, which may alias the output
        const Tensor& t = srcs[0].t;
        if (b.has(t) && add) {
            // gradient already in the buffer AND an addend (identity residual): both ride on the conv epilogue when the kernel has the second
            // residual input; otherwise a separate add pass first
            ofd_conv_args probe{};
            probe.ksize = d.ksize; probe.n_src = 1; probe.Cout = cin; probe.residual = t.g;
            if (conv_residual_b_supported(&probe)) {
                res_b = add;
            } else {
                RUN(k_grad_add(t.g, add, (size_t)B * H * W * cin, 1, c.s));
            }
            add = t.g;
        } else if (b.has(t)) {
            add = t.g;
        }
        D = t; D.p = t.g;
        b.mark(t);
    } else {
        D = b.stmp(cin, H, W);
    }
    if (c.rc != OFD_OK) return D;
    ofd_conv_args a{};
    a.B = B; a.H = H; a.W = W; a.ksize = d.ksize; a.n_src = 1; a.Cout = cin;
    a.src[0].src = dy; a.src[0].channels = d.Cout; a.src[0].src_channels = d.Cout;
    a.weight = u->d_wtbuf + d.w_off;
    a.residual = add;
    a.out = D.p;
    c.begin(d.ksize == 3 ? PC_DGRAD3 : PC_DGRAD1, 2.0 * px * d.Cout * (double)d.Cin * taps, px * 2.0 * (d.Cout + cin), prefix + " dgrad");
    RUN(conv_forward_impl(&a, c.s, 0, 0, res_b));
    c.end();
    return D;
}

// adjoint of the loader: slices (and 2x2 sum-pool / pixel-shuffle) of D go to the sources' gradients
static void scatter_to_sources(Bwd& b, const Tensor& D, const std::vector<SrcSpec>& srcs) {
    Ctx& c = b.c;
    std::vector<int> acc;
    for (auto& s : srcs) acc.push_back(b.has(s.t) ? 1 : 0);
    int off = 0;
    c.begin(PC_MISC, 0, (double)c.B * D.H * D.W * D.C * 4, "grad_scatter");
    for (size_t i = 0; i < srcs.size(); ++i) {
        const SrcSpec& s = srcs[i];
        const int mode = s.upsample ? 1 : (s.unshuffle ? 2 : 0);
        RUN(k_grad_scatter(D.p, D.C, off, s.t.g, s.t.C, c.B, D.H, D.W, mode, s.p1, s.p2, acc[i], c.s));
        b.mark(s.t);
        off += s.t.C;
    }
    c.end();
}

static void resblock_backward(Bwd& b, const TapeRec& r, float* dss, float* dts) {
    Ctx& c = b.c;
    ofd_unet* u = c.u;
    const std::string& name = r.name;
    const int B = c.B, H = r.out.H, W = r.out.W, Cout = r.out.C;
    int cin = 0;
    for (auto& s : r.srcs) cin += s.t.C;
    const bf16_t* dout = r.out.g;
    float* ws = b.stmpf(gn_bwd_workspace_floats(B, H, W, Cout));
    // act1 = SiLU(GN1(h1) (scale + 1) + shift) feeds only block2's weight gradient.  For the 64-channel (full-resolution) blocks
    // that kernel applies the affine + SiLU while staging h1, as the forward conv does (one co-block: every input tile is staged
    // once; saves a write and a read of a 0.92 GB tensor per block).  Wider layers re-stage an input tile once per co-block and
    // would repeat the transcendental work: they read a materialised act1.  Same-box A/B of the training step at B=16,
    // 440x1024: never 162.5 ms, C <= 64: 160.9, C <= 128: 161.8, always: 162.3.
    static const int act1_max = getenv("OFD_FUSE_ACT1_MAXC") ? atoi(getenv("OFD_FUSE_ACT1_MAXC")) : 64;      // A/B switch
    const bool fuse_act1 = Cout <= act1_max;
    Tensor act1 = r.act1;                 // (the forward's own, when it materialised one)
    const bool have_act1 = act1.p != nullptr;
    if (!fuse_act1 && !have_act1) act1 = b.stmp(Cout, H, W);
    if (c.rc != OFD_OK) return;
    const double ew = (double)B * H * W * Cout * 2;
    // out = SiLU(GN2(h2)) + res
    c.begin(PC_GNBWD, 0, ew * 5, name + ".block2 gn-silu bwd");
    RUN(k_gn_silu_backward(dout, r.h2.p, r.a2, r.s2, r.st2, u->P(name + ".block2.norm.weight"), u->P(name + ".block2.norm.bias"), nullptr, 0, 0,
                           r.h2.g, u->G(name + ".block2.norm.weight"), u->G(name + ".block2.norm.bias"), nullptr, ws, B, H, W, Cout, c.s,
                           u->G(name + ".block2.proj.bias")));
    c.end();
    // h2 = conv2(act1), act1 = SiLU(GN1(h1) * (scale + 1) + shift): recomputed, the forward fused it into conv2's loader
    SrcSpec sa;
    if (have_act1) {
        sa.t = act1;
    } else if (fuse_act1) {
        sa.t = r.h1;
    } else {
        c.begin(PC_GNBWD, 0, ew * 2, name + " act1 recompute");
        RUN(k_affine_silu(r.h1.p, r.a1, r.s1, act1.p, B, H, W, Cout, c.s));
        c.end();
        sa.t = act1;
    }
    const bool wg_pro = fuse_act1 && !have_act1;      // the weight gradient applies the affine + SiLU while staging h1
    Tensor dact1 = conv_backward(b, name + ".block2.proj", {sa}, r.h2.g, H, W, true, nullptr, true, false, wg_pro ? r.a1 : nullptr,
                                 wg_pro ? r.s1 : nullptr);
    c.begin(PC_GNBWD, 0, ew * 5, name + ".block1 gn-silu bwd");
    const bool timed = !u->cfg.no_time;       // Unet(time_in=False): block1 has no scale/shift and no time projection
    RUN(k_gn_silu_backward(dact1.p, r.h1.p, r.a1, r.s1, r.st1, u->P(name + ".block1.norm.weight"), u->P(name + ".block1.norm.bias"),
                           timed ? c.ss : nullptr, timed ? u->ss_stride : 0, timed ? u->ss_offset.at(name) : 0, r.h1.g,
                           u->G(name + ".block1.norm.weight"), u->G(name + ".block1.norm.bias"), timed ? dss : nullptr,
                           ws, B, H, W, Cout, c.s, u->G(name + ".block1.proj.bias")));
    if (timed)
        RUN(k_block_mlp_bwd(dss, u->ts.temb_silu, u->P(name + ".mlp.1.weight"), 2 * Cout, u->ss_offset.at(name), u->G(name + ".mlp.1.weight"),
                            u->G(name + ".mlp.1.bias"), dts, B, u->cfg.dim * 4, u->ss_stride, c.s));
    c.end();
    Tensor D;
    if (cin == Cout) {
        const bool direct = direct_target(r.srcs);
        D = conv_backward(b, name + ".block1.proj", r.srcs, r.h1.g, H, W, true, dout, true, direct);     // + identity residual
        if (direct) return;
    } else if (direct_target(r.srcs)) {
        // both data gradients (block1.proj, res_conv) accumulate straight into the sources' gradient buffers
        conv_backward(b, name + ".block1.proj", r.srcs, r.h1.g, H, W, true, nullptr, true, true);
        conv_backward(b, name + ".res_conv", r.srcs, dout, H, W, true, nullptr, false, true);
        return;
    } else {
        Tensor D1 = conv_backward(b, name + ".block1.proj", r.srcs, r.h1.g, H, W, true, nullptr, true);
        D = conv_backward(b, name + ".res_conv", r.srcs, dout, H, W, true, D1.p);
    }
    if (c.rc != OFD_OK) return;
    scatter_to_sources(b, D, r.srcs);
}

static void attn_tail_backward(Bwd& b, const TapeRec& r, const std::string& qkv_prefix, const std::string& norm_g, const bf16_t* dy,
                               const Tensor* dxn_done = nullptr) {
    // shared by both attention blocks: to_qkv conv backward (unless the core backward has done it: dxn_done), then PreNorm's LayerNorm,
    // x.g (+)= dy + LN'(dxn)
    Ctx& c = b.c;
    ofd_unet* u = c.u;
    const int H = r.x.H, W = r.x.W, C = r.x.C;
    const size_t npix = (size_t)c.B * H * W;
    SrcSpec sx; sx.t = r.xn;
    Tensor Dx = dxn_done ? *dxn_done : conv_backward(b, qkv_prefix, {sx}, r.qkv.g, H, W, true, nullptr);
    if (c.rc != OFD_OK) return;
    c.begin(PC_LN, 0, (double)npix * C * 10, r.name + " prenorm bwd");
    RUN(k_layernorm_c_bwd(r.x.p, u->P(norm_g), Dx.p, r.x.g, u->G(norm_g), npix, C, site_eps(u, r.name + ".fn.norm"), b.has(r.x) ? 1 : 0, c.s, dy));
    b.mark(r.x);
    c.end();
}

static void linattn_backward(Bwd& b, const TapeRec& r) {
    Ctx& c = b.c;
    ofd_unet* u = c.u;
    const std::string& name = r.name;
    const int B = c.B, H = r.x.H, W = r.x.W, C = r.x.C, n = H * W;
    const size_t npix = (size_t)B * n;
    const bf16_t* dy = r.out.g;
    float* ws = b.stmpf(la_bwd_workspace_floats(B, n));
    if (c.rc != OFD_OK) return;
    c.begin(PC_LN, 0, (double)npix * C * 6, name + " to_out.1 bwd");
    RUN(k_layernorm_c_bwd(r.o2.p, u->P(name + ".fn.fn.to_out.1.g"), dy, r.o2.g, u->G(name + ".fn.fn.to_out.1.g"), npix, C,
                          site_eps(u, name + ".fn.fn.to_out.1"), 0, c.s));
    c.end();
    SrcSpec sao; sao.t = r.ao;
    const bool fuse_qkv = la_bwd_fuse_qkv(), fuse_dao = la_bwd_fuse_dao();
    const bool fd = C == 64 && fuse_qkv && fuse_dao;       // the core backward forms dout = Wo^T do2 itself, and the to_out.0 weight / bias gradients
    Tensor Dao;                                            // come out of its pixel reduction (la_core.hip lc_bwd_combine_kernel): no to_out.0 backward at all
    if (!fd) Dao = conv_backward(b, name + ".fn.fn.to_out.0", {sao}, r.o2.g, H, W, true, nullptr);
    if (c.rc != OFD_OK) return;
    if (C == 64 && fuse_qkv) {
        // the to_qkv backward rides on the core backward's dqkv tile (la_core.hip lc_bwd_apply_kernel<true>): no dqkv tensor
        const ConvDesc& d = u->convs[u->cindex.at(name + ".fn.fn.to_qkv")];
        float* acc = u->d_wacc + d.w_off;
        Tensor Dx = b.stmp(C, H, W);
        if (c.rc != OFD_OK) return;
        c.begin(PC_LABWD, npix * (4.0 * 4 * 32 * 32 * 2 + 4.0 * 384 * 64), (double)npix * (384 + 128 + 64 + 64) * 2, name + " core + to_qkv bwd");
        const ConvDesc& dto = u->convs[u->cindex.at(name + ".fn.fn.to_out.0")];
        if (fd) {
            RUN(k_linear_attention_core_bwd(r.qkv.p, r.o2.g, r.ctx, r.ml, nullptr, ws, B, n, c.s, r.xn.p, u->d_wtbuf + d.w_off, acc, Dx.p,
                                            u->d_wbuf + dto.w_off, u->d_wtbuf + dto.w_off, u->d_wacc + dto.w_off, u->G(name + ".fn.fn.to_out.0.bias"),
                                            (la_train_no_ao(C) && la_recompute_q()) ? u->d_wbuf + d.w_off : nullptr));
            RUN(det_flush_wacc(c, dto));
            RUN(k_wgrad_finish(u->d_wacc + dto.w_off, u->P(dto.wname), u->G(dto.wname), dto.Cout, dto.Cin, dto.Cin_pad, dto.ksize, dto.ws_eps, dto.unshuffle, 0, c.s));
        }
        else RUN(k_linear_attention_core_bwd(r.qkv.p, Dao.p, r.ctx, r.ml, nullptr, ws, B, n, c.s, r.xn.p, u->d_wtbuf + d.w_off, acc, Dx.p));
        RUN(det_flush_wacc(c, d));
        RUN(k_wgrad_finish(acc, u->P(d.wname), u->G(d.wname), d.Cout, d.Cin, d.Cin_pad, d.ksize, d.ws_eps, d.unshuffle, 0, c.s));
        c.end();
        attn_tail_backward(b, r, name + ".fn.fn.to_qkv", name + ".fn.norm.g", dy, &Dx);
        return;
    }
    c.begin(PC_LABWD, npix * 4.0 * 4 * 32 * 32 * 2, (double)npix * (384 * 2 + 128) * 2, name + " core bwd");
    RUN(k_linear_attention_core_bwd(r.qkv.p, Dao.p, r.ctx, r.ml, r.qkv.g, ws, B, n, c.s));
    c.end();
    attn_tail_backward(b, r, name + ".fn.fn.to_qkv", name + ".fn.norm.g", dy);
}

static void midattn_backward(Bwd& b, const TapeRec& r) {
    Ctx& c = b.c;
    const int B = c.B, H = r.x.H, W = r.x.W, n = H * W;
    const bf16_t* dy = r.out.g;
    float* delta = b.stmpf((size_t)B * 4 * n);
    SrcSpec sao; sao.t = r.ao;
    Tensor Dao = conv_backward(b, "mid_attn.fn.fn.to_out", {sao}, dy, H, W, true, nullptr);
    if (c.rc != OFD_OK) return;
    c.begin(PC_FLASHBWD, 10.0 * B * 4.0 * (double)n * n * 32 * 1.4, (double)B * n * (384 * 2 + 256) * 2, "mid_attn core bwd");
    RUN(k_flash_attention_bwd(r.qkv.p, r.ao.p, Dao.p, r.lse, r.qkv.g, delta, B, n, c.s));
    c.end();
    attn_tail_backward(b, r, "mid_attn.fn.fn.to_qkv", "mid_attn.fn.norm.g", dy);
}

struct TrainLayout {
    size_t small_floats, off_dss, off_dts, off_tm;     // float offsets inside the small area
    size_t small_b, persist_b, scratch_b;
};

static TrainLayout small_layout(const ofd_unet* u, int B) {
    TrainLayout L{};
    const size_t tdim = (size_t)u->cfg.dim * 4;
    size_t n = (size_t)B * u->ss_stride + 2 * B * tdim;          // ss | temb | temb_silu (as the inference forward)
    n = (n + 63) / 64 * 64;
    L.off_dss = n; n += (size_t)B * u->ss_stride;
    L.off_dts = n; n += B * tdim;
    L.off_tm = n;  n += (size_t)B * (u->cfg.dim + 3 * tdim);
    L.small_floats = n;
    L.small_b = (n * 4 + 4095) / 4096 * 4096;
    return L;
}

static int run_backward(Ctx& c, const float* dout, const TrainLayout& L, float* fsmall, ofd_grad_ready_fn cb, void* user) {
    ofd_unet* u = c.u;
    Bwd b{c, {}};
    const int B = c.B, H = u->ts.H, W = u->ts.W, dim = u->cfg.dim;
    float* dss = fsmall + L.off_dss;
    float* dts = fsmall + L.off_dts;
    float* tm = fsmall + L.off_tm;
    const bool det = u->deterministic && !c.dry;
    auto notify = [&](const std::string& prefix) {
        if (c.dry || c.rc != OFD_OK || (!cb && !det)) return;
        auto it = u->prange.find(prefix);
        if (it == u->prange.end()) return;
        // deterministic mode: the range's fixed-point shadow lands in the fp32 gradients before anyone is told they are ready
        if (det) RUN(k_det_flush(u->d_fx + it->second.first, u->d_grads + it->second.first, it->second.second - it->second.first, c.s));
        if (cb && c.rc == OFD_OK) cb(it->second.first, it->second.second, user);
    };
    if (!c.dry) {
        if (hipMemsetAsync(u->d_wacc, 0, u->n_wbuf * sizeof(float), c.s) != hipSuccess ||
            hipMemsetAsync(u->d_grads, 0, u->n_param_floats * 4, c.s) != hipSuccess ||
            hipMemsetAsync(dss, 0, ((size_t)B * u->ss_stride + (size_t)B * dim * 4) * 4, c.s) != hipSuccess) {
            set_error("unet_backward: memset failed");
            return OFD_ERR_HIP;
        }
        if (det) RUN(det_begin(c, dts, (size_t)B * dim * 4));
        if (c.rc != OFD_OK) return c.rc;
    }
    c.begin(PC_MISC, 0, 0, "final_conv bwd");
    RUN(k_final_conv_bwd(u->ts.xf.p, u->P("final_conv.weight"), dout, u->ts.xf.g, u->G("final_conv.weight"), u->G("final_conv.bias"), B, H, W, dim,
                         u->cfg.out_dim, c.s));
    c.end();
    b.mark(u->ts.xf);
    notify("final_conv");
    for (auto it = u->tape.rbegin(); it != u->tape.rend() && c.rc == OFD_OK; ++it) {
        const TapeRec& r = *it;
        c.reset_scratch();
        if (!b.has(r.out)) {
            set_error("unet_backward: no gradient reached %s", r.name.c_str());
            if (det) (void)det_end(c);
            return OFD_ERR_STATE;
        }
        switch (r.kind) {
            case TK_RES: resblock_backward(b, r, dss, dts); break;
            case TK_LINATTN: linattn_backward(b, r); break;
            case TK_MIDATTN: midattn_backward(b, r); break;
            default: {
                const bool first = r.name == "init_conv";
                const bool direct = direct_target(r.srcs);
                const bool up1 = r.srcs.size() == 1 && r.srcs[0].upsample && r.srcs[0].t.g;      // Upsample + conv: pooled epilogue if the kernel serves it
                Tensor D = conv_backward(b, r.name, r.srcs, r.out.g, r.out.H, r.out.W, !first, nullptr, false, direct || up1);
                const bool landed = direct || (up1 && D.p == r.srcs[0].t.g);
                if (!first && !landed && c.rc == OFD_OK) scatter_to_sources(b, D, r.srcs);
            }
        }
        notify(r.name);
    }
    if (u->cfg.no_time) {
        if (det) { RUN(k_det_flush(u->d_fx, u->d_grads, u->n_param_floats, c.s)); const int e = det_end(c); if (c.rc == OFD_OK) c.rc = e; }
        return c.rc;
    }
    if (det) RUN(k_det_flush(u->d_fx + u->n_param_floats + u->n_wbuf, dts, (size_t)B * dim * 4, c.s));      // every block's Linear has added its share
    c.begin(PC_MISC, 0, 0, "time_mlp bwd");
    RUN(k_time_mlp_bwd(u->ts.t, u->ts.temb, dts, u->P("time_mlp.1.weight"), u->P("time_mlp.1.bias"), u->P("time_mlp.3.weight"),
                       u->G("time_mlp.1.weight"), u->G("time_mlp.1.bias"), u->G("time_mlp.3.weight"), u->G("time_mlp.3.bias"), tm, B, dim, c.s));
    c.end();
    notify("time_mlp");
    if (det) {          // whatever no notified range covers, then the context off (also after a failed launch: it is shared)
        RUN(k_det_flush(u->d_fx, u->d_grads, u->n_param_floats, c.s));
        const int e = det_end(c);
        if (c.rc == OFD_OK) c.rc = e;
    }
    return c.rc;
}

// size planning: a dry training forward + backward that allocates and counts but launches nothing
static int plan(ofd_unet* u, int B, int H, int W, TrainLayout& L) {
    L = small_layout(u, B);
    auto hit = u->plans.find(std::make_tuple(B, H, W));
    if (hit != u->plans.end()) {
        L.persist_b = hit->second[1];
        L.scratch_b = hit->second[2];
        return OFD_OK;
    }
    Ctx c;
    c.u = u; c.s = nullptr; c.B = B; c.train = true; c.dry = true;
    c.persist = (char*)4096; c.persist_cap = (size_t)1 << 60;
    c.scratch = (char*)4096 + ((size_t)1 << 61); c.scratch_cap = (size_t)1 << 60;
    c.grad_offset = (size_t)1 << 59;
    c.ss = nullptr;
    std::vector<TapeRec> saved_tape = u->tape;
    TrainState saved_ts = u->ts;
    auto saved_taps = u->taps;
    const int saved_B = u->last_B;
    int rc = run_forward(c, nullptr, u->cfg.channels, nullptr, 0, nullptr, nullptr, H, W, nullptr, nullptr);
    if (rc == OFD_OK) {
        L.persist_b = (c.persist_used + 4095) / 4096 * 4096;
        c.reset_scratch();
        rc = run_backward(c, nullptr, L, nullptr, nullptr, nullptr);
        L.scratch_b = (c.scratch_high + 4095) / 4096 * 4096 + 4096;
    }
    u->tape = saved_tape; u->ts = saved_ts; u->taps = saved_taps; u->last_B = saved_B;
    if (rc == OFD_OK) u->plans[std::make_tuple(B, H, W)] = {L.small_b, L.persist_b, L.scratch_b};
    return rc;
}

}  // namespace ofd

extern "C" size_t ofd_unet_train_workspace_bytes(ofd_unet* u, int B, int H, int W) {
    if (!u || B <= 0 || H <= 0 || W <= 0 || H % 8 || W % 8) return 0;
    TrainLayout L;
    if (plan(u, B, H, W, L) != OFD_OK) return 0;
    return L.small_b + 2 * L.persist_b + L.scratch_b + 4096;
}

extern "C" int ofd_unet_set_deterministic(ofd_unet* u, int enabled) {
    OFD_CHECK_ARG(u, "unet_set_deterministic: null handle");
    u->deterministic = enabled != 0;
    return OFD_OK;
}
extern "C" long ofd_unet_deterministic_misses(ofd_unet* u) {
    if (!u) return -1;
    if (!u->d_det_miss) return 0;
    unsigned v = 0;
    if (hipMemcpy(&v, u->d_det_miss, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return (long)v;
}

extern "C" size_t ofd_unet_param_floats(const ofd_unet* u) { return u ? u->n_param_floats : 0; }
extern "C" size_t ofd_unet_param_offset(const ofd_unet* u, int i) {
    return (u && i >= 0 && i < (int)u->params.size()) ? u->params[i].offset : 0;
}

extern "C" int ofd_unet_bind_grad_buffer(ofd_unet* u, float* dev_grads, size_t floats) {
    OFD_CHECK_ARG(u && dev_grads, "unet_bind_grad_buffer: null argument");
    OFD_CHECK_ARG(floats >= u->n_param_floats, "unet_bind_grad_buffer: %zu floats, need %zu", floats, u->n_param_floats);
    OFD_CHECK_ARG(((uintptr_t)dev_grads & 15) == 0, "unet_bind_grad_buffer: buffer must be 16-byte aligned");
    u->d_grads = dev_grads;
    return OFD_OK;
}

static int prepare_train(ofd_unet* u, hipStream_t s) {
    if (u->prange.empty()) {
        // parameter range of every op prefix the backward reports: the longest registered op name that
        // prefixes the parameter name ("downs.0.0" resblock, "downs.0.2" attention, "downs.0.3.1" conv ...)
        std::vector<std::string> ops = {"init_conv", "final_conv", "mid_attn"};
        if (!u->cfg.no_time) ops.push_back("time_mlp");
        for (auto& n : u->resblocks) ops.push_back(n);
        for (auto& kv : u->cindex) {
            const std::string& p = kv.first;
            bool inside = false;
            for (auto& n : u->resblocks) if (p.rfind(n + ".", 0) == 0) inside = true;
            if (p.rfind("mid_attn.", 0) == 0) inside = true;
            const size_t pos = p.find(".fn.fn.");
            if (pos != std::string::npos) { ops.push_back(p.substr(0, pos)); inside = true; }
            if (!inside) ops.push_back(p);
        }
        for (auto& prm : u->params) {
            std::string best;
            for (auto& o : ops)
                if (prm.name.rfind(o + ".", 0) == 0 && o.size() > best.size()) best = o;
            if (best.empty()) { set_error("unet_train: parameter %s belongs to no op", prm.name.c_str()); return OFD_ERR_STATE; }
            const size_t b0 = prm.offset, b1 = prm.offset + (prm.numel + 3) / 4 * 4;
            auto it = u->prange.find(best);
            if (it == u->prange.end()) u->prange[best] = {b0, b1};
            else { it->second.first = std::min(it->second.first, b0); it->second.second = std::max(it->second.second, b1); }
        }
    }
    if (!u->d_wtbuf) OFD_HIP(hipMalloc(&u->d_wtbuf, u->n_wbuf * sizeof(bf16_t)));
    if (!u->d_wacc) OFD_HIP(hipMalloc(&u->d_wacc, u->n_wbuf * sizeof(float)));
    if (!u->wt_prepared) {
        if (!u->d_tr) {                                       // one launch for all convs: blocks in proportion to the weight count
            std::vector<ofd_weight_prep_desc> h;
            int blocks = 0;
            for (auto& cd : u->convs) {
                if (cd.ksize == 7) continue;                  // first layer: no data gradient
                ofd_weight_prep_desc d{(const float*)(u->d_wbuf + cd.w_off), u->d_wtbuf + cd.w_off, cd.Cout, cd.Cin, cd.Cin_pad, cd.ksize, -1.0f, 0, blocks};
                h.push_back(d);
                const size_t total = (size_t)cd.ksize * cd.ksize * cd.Cin_pad * cd.Cout;
                blocks += (int)std::min<size_t>((total + 2047) / 2048, 256);
            }
            OFD_HIP(hipMalloc(&u->d_tr, h.size() * sizeof(ofd_weight_prep_desc)));
            OFD_HIP(hipMemcpy(u->d_tr, h.data(), h.size() * sizeof(ofd_weight_prep_desc), hipMemcpyHostToDevice));
            u->n_tr = (int)h.size();
            u->tr_blocks = blocks;
        }
        {
            int rc = k_wt_transpose_batched(u->d_tr, u->n_tr, u->tr_blocks, s);
            if (rc != OFD_OK) return rc;
        }
        u->wt_prepared = true;
    }
    return OFD_OK;
}

extern "C" int ofd_unet_train_forward(ofd_unet* u, const float* x, int Cx, const float* cond, int Cc, const int64_t* t, float* out, int B, int H,
                                      int W, void* workspace, size_t workspace_bytes, void* stream) {
    OFD_CHECK_ARG(u && x && (t || u->cfg.no_time) && out && workspace, "unet_train_forward: null argument");
    OFD_CHECK_ARG(B > 0 && H > 0 && W > 0 && H % 8 == 0 && W % 8 == 0, "unet_train_forward: H=%d W=%d must be positive multiples of 8", H, W);
    OFD_CHECK_ARG(Cx + (cond ? Cc : 0) == u->cfg.channels, "unet_train_forward: %d + %d input channels, UNet has %d", Cx, cond ? Cc : 0, u->cfg.channels);
    OFD_CHECK_ARG(((uintptr_t)workspace & 255) == 0, "unet_train_forward: workspace must be 256-byte aligned");
    if (!u->prepared) { set_error("unet_train_forward: call ofd_unet_prepare after setting parameters"); return OFD_ERR_STATE; }
    u->ts.valid = false;
    TrainLayout L;
    int rc = plan(u, B, H, W, L);
    if (rc != OFD_OK) return rc;
    const size_t need = L.small_b + 2 * L.persist_b + L.scratch_b + 4096;
    if (workspace_bytes < need) { set_error("unet_train_forward: workspace %zu < %zu", workspace_bytes, need); return OFD_ERR_WORKSPACE; }
    rc = prepare_train(u, (hipStream_t)stream);
    if (rc != OFD_OK) return rc;
    Ctx c;
    c.u = u; c.s = (hipStream_t)stream; c.B = B; c.train = true;
    char* w = (char*)workspace;
    float* fsmall = (float*)w;
    c.ss = fsmall;
    float* temb = fsmall + (size_t)B * u->ss_stride;
    float* temb_silu = temb + (size_t)B * u->cfg.dim * 4;
    c.persist = w + L.small_b; c.persist_cap = L.persist_b;
    c.grad_offset = L.persist_b;
    c.scratch = c.persist + 2 * L.persist_b; c.scratch_cap = workspace_bytes - (size_t)(c.scratch - w);
    rc = run_forward(c, x, Cx, cond, Cc, t, out, H, W, temb, temb_silu);
    if (rc != OFD_OK) return rc;
    u->ts.workspace = w; u->ts.workspace_bytes = workspace_bytes;
    u->ts.valid = true;
    return OFD_OK;
}

extern "C" int ofd_unet_backward(ofd_unet* u, const float* dout, ofd_grad_ready_fn on_ready, void* user, void* stream) {
    OFD_CHECK_ARG(u && dout, "unet_backward: null argument");
    if (!u->ts.valid) { set_error("unet_backward: no training forward to differentiate (call ofd_unet_train_forward first)"); return OFD_ERR_STATE; }
    if (!u->d_grads) { set_error("unet_backward: bind a gradient buffer first (ofd_unet_bind_grad_buffer)"); return OFD_ERR_STATE; }
    const int B = u->ts.B;
    TrainLayout L;
    int rc = plan(u, B, u->ts.H, u->ts.W, L);
    if (rc != OFD_OK) return rc;
    Ctx c;
    c.u = u; c.s = (hipStream_t)stream; c.B = B; c.train = true;
    char* w = u->ts.workspace;
    float* fsmall = (float*)w;
    c.ss = u->ts.ss;
    c.persist = w + L.small_b; c.persist_cap = L.persist_b; c.persist_used = u->ts.persist_used;
    c.grad_offset = L.persist_b;
    c.scratch = c.persist + 2 * L.persist_b; c.scratch_cap = u->ts.workspace_bytes - (size_t)(c.scratch - w);
    rc = run_backward(c, dout, L, fsmall, on_ready, user);
    u->ts.valid = false;                      // the tape is consumed (activations' scratch was reused)
    return rc;
}
