// UNet executor: one C call runs the whole forward of denoising_diffusion.py:363-417 as a fixed
// sequence of HIP launches on the caller's stream.  Parameters keep the reference's state-dict
// names and order (denoising_diffusion.py:272-361; registration order downs, ups, mid).
// Host code only: every kernel lives in conv_igemm.hip / blocks.hip.
#include <cstdlib>
#include "unet_exec.h"

using namespace ofd;

namespace ofd {

float site_eps(const ofd_unet* u, const std::string& site) {
    // eps the reference uses at this site (DD:107, DD:122).  eps_mode 0 = fp32 activations
    // everywhere; eps_mode 1 = dtype flow of bf16 autocast: 1e-3 only where the site's input is
    // a bare conv output (see DESIGN.md "Numerics contract").
    if (u->cfg.eps_mode == 0) return 1e-5f;
    auto ends = [&](const char* s) { std::string t(s); return site.size() >= t.size() && site.compare(site.size() - t.size(), t.size(), t) == 0; };
    if (ends(".fn.fn.to_out.1")) return 1e-3f;
    if (ends(".0.block1.proj") && site.rfind("downs.", 0) == 0) return 1e-3f;
    if (site == "mid_block1.block1.proj" || site == "final_res_block.block1.proj") return 1e-3f;
    return 1e-5f;
}

static void add_param(ofd_unet* u, const std::string& name, std::vector<int> shape) {
    size_t numel = 1;
    for (int d : shape) numel *= (size_t)d;
    Param p{name, (int)shape.size(), {1, 1, 1, 1}, numel, u->n_param_floats, false};
    for (size_t i = 0; i < shape.size(); ++i) p.shape[i] = shape[i];
    u->pindex[name] = (int)u->params.size();
    u->params.push_back(p);
    u->n_param_floats += (numel + 3) / 4 * 4;
}

static void add_conv(ofd_unet* u, const std::string& prefix, int co, int ci, int k, bool bias, float ws_eps, int unshuffle = 0) {
    add_param(u, prefix + ".weight", {co, ci, k, k});
    if (bias) add_param(u, prefix + ".bias", {co});
    ConvDesc c{prefix + ".weight", co, ci, (k == 7) ? 16 : ci, k, ws_eps, unshuffle, u->n_wbuf};
    u->n_wbuf += (size_t)k * k * c.Cin_pad * co;
    if (k == 7 && ci <= 8) {
        c.pack8_off = (long)u->n_wbuf;
        u->n_wbuf += ofd_conv_weight_elems(co, 8, 7);
    }
    u->cindex[prefix] = (int)u->convs.size();
    u->convs.push_back(c);
}

static void add_resblock(ofd_unet* u, const std::string& name, int ci, int co) {
    const int tdim = u->cfg.dim * 4;
    if (!u->cfg.no_time) {
        add_param(u, name + ".mlp.1.weight", {co * 2, tdim});
        add_param(u, name + ".mlp.1.bias", {co * 2});
    }
    add_conv(u, name + ".block1.proj", co, ci, 3, true, site_eps(u, name + ".block1.proj"));
    add_param(u, name + ".block1.norm.weight", {co});
    add_param(u, name + ".block1.norm.bias", {co});
    add_conv(u, name + ".block2.proj", co, co, 3, true, site_eps(u, name + ".block2.proj"));
    add_param(u, name + ".block2.norm.weight", {co});
    add_param(u, name + ".block2.norm.bias", {co});
    if (ci != co) add_conv(u, name + ".res_conv", co, ci, 1, true, -1.0f);
    u->ss_offset[name] = u->ss_stride;
    u->ss_stride += 2 * co;
}

static void add_linattn(ofd_unet* u, const std::string& name, int c) {
    add_conv(u, name + ".fn.fn.to_qkv", 384, c, 1, false, -1.0f);
    add_conv(u, name + ".fn.fn.to_out.0", c, 128, 1, true, -1.0f);
    add_param(u, name + ".fn.fn.to_out.1.g", {1, c, 1, 1});
    add_param(u, name + ".fn.norm.g", {1, c, 1, 1});
    if (c <= 128) {
        u->la_fused[name] = {u->n_labuf, c};
        u->n_labuf += (size_t)512 * c;
        if (c == 64) u->n_labuf += (size_t)384 * c;      // + the plain (gain not folded in) wq | wkv of the fused TRAINING forward
    }
}

static void build_registry(ofd_unet* u) {
    const int dim = u->cfg.dim;
    u->dims = {dim, dim, dim * 2, dim * 4, dim * 8};
    const int tdim = dim * 4;
    add_conv(u, "init_conv", dim, u->cfg.channels, 7, true, -1.0f);
    if (!u->cfg.no_time) {
        add_param(u, "time_mlp.1.weight", {tdim, dim});
        add_param(u, "time_mlp.1.bias", {tdim});
        add_param(u, "time_mlp.3.weight", {tdim, tdim});
        add_param(u, "time_mlp.3.bias", {tdim});
    }
    for (int i = 0; i < 4; ++i) {
        const int ci = u->dims[i], co = u->dims[i + 1];
        const std::string p = "downs." + std::to_string(i);
        add_resblock(u, p + ".0", ci, ci);
        add_resblock(u, p + ".1", ci, ci);
        add_linattn(u, p + ".2", ci);
        if (i < 3) add_conv(u, p + ".3.1", co, ci * 4, 1, true, -1.0f, 1);
        else add_conv(u, p + ".3", co, ci, 3, true, -1.0f);
    }
    for (int i = 0; i < 4; ++i) {
        const int ci = u->dims[3 - i], co = u->dims[4 - i];
        const std::string p = "ups." + std::to_string(i);
        add_resblock(u, p + ".0", co + ci, co);
        add_resblock(u, p + ".1", co + ci, co);
        add_linattn(u, p + ".2", co);
        if (i < 3) {
            add_conv(u, p + ".3.1", ci, co, 3, true, -1.0f);
            ConvDesc& d = u->convs.back();              // Upsample(x2) + 3x3 also as four 2x2 phase kernels (forward)
            d.phase_off = (long)u->n_wbuf;
            u->n_wbuf += (size_t)16 * co * ci;
        }
        else add_conv(u, p + ".3", ci, co, 3, true, -1.0f);
    }
    const int mid = u->dims[4];
    add_resblock(u, "mid_block1", mid, mid);
    add_conv(u, "mid_attn.fn.fn.to_qkv", 384, mid, 1, false, -1.0f);
    add_conv(u, "mid_attn.fn.fn.to_out", mid, 128, 1, true, -1.0f);
    add_param(u, "mid_attn.fn.norm.g", {1, mid, 1, 1});
    add_resblock(u, "mid_block2", mid, mid);
    add_resblock(u, "final_res_block", dim * 2, dim);
    add_param(u, "final_conv.weight", {u->cfg.out_dim, dim, 1, 1});
    add_param(u, "final_conv.bias", {u->cfg.out_dim});
    // forward order of the ResnetBlocks = order of their scale/shift rows (any fixed order works)
    for (auto& kv : u->ss_offset) u->resblocks.push_back(kv.first);
}

void conv(Ctx& c, const std::string& prefix, const std::vector<SrcSpec>& srcs, Tensor out, const float* in_scale,
                 const float* in_shift, const bf16_t* residual, const bf16_t* res_act, const float* res_scale,
                 const float* res_shift, float* gn_partial, int cout0, const FcFuse* fc) {
    if (c.rc != OFD_OK) return;
    const ConvDesc& d = c.u->convs[c.u->cindex.at(prefix)];
    ofd_conv_args a{};
    a.B = c.B; a.H = out.H; a.W = out.W; a.ksize = d.ksize; a.n_src = (int)srcs.size(); a.Cout = d.Cout;
    int cin = 0;
    for (size_t i = 0; i < srcs.size(); ++i) {
        a.src[i].src = srcs[i].t.p;
        a.src[i].channels = srcs[i].t.C;
        a.src[i].src_channels = srcs[i].t.C;
        a.src[i].ch_offset = 0;
        a.src[i].upsample = srcs[i].upsample;
        a.src[i].unshuffle = srcs[i].unshuffle;
        a.src[i].p1 = srcs[i].p1;
        a.src[i].p2 = srcs[i].p2;
        cin += srcs[i].t.C;
    }
    const bool pack8 = d.ksize == 7 && cin == 8 && d.pack8_off >= 0;
    if (cin != d.Cin_pad && !pack8) { set_error("conv %s: got %d input channels, expected %d", prefix.c_str(), cin, d.Cin_pad); c.rc = OFD_ERR_ARG; return; }
    a.weight = c.u->d_wbuf + (pack8 ? (size_t)d.pack8_off : d.w_off);
    a.bias = c.u->P(prefix + ".bias");
    a.in_scale = in_scale; a.in_shift = in_shift; a.residual = residual; a.res_act = res_act;
    a.res_scale = res_scale; a.res_shift = res_shift; a.out = out.p; a.gn_partial = gn_partial;
    const double px = (double)c.B * out.H * out.W;
    const double flops = 2.0 * px * d.Cout * (double)d.Cin * d.ksize * d.ksize;   // counted as the reference executes
    const double bytes = px * 2.0 * (d.Cout + (double)cin / ((srcs[0].upsample) ? 4 : 1)) + (double)d.ksize * d.ksize * d.Cin_pad * d.Cout * 2.0;
    const bool pp = d.ksize == 3 && d.Cout == 64 && cin == 64 && srcs.size() == 1 && !srcs[0].upsample && !residual && !res_act;
    const int cls3 = pp ? PC_CONV3_PP : (d.Cout % 128 == 0 ? PC_CONV3 : PC_CONV3_64);
    c.begin(d.ksize == 3 ? cls3 : (d.ksize == 1 ? PC_CONV1 : PC_CONV7), flops, bytes,
            prefix + " " + std::to_string(d.Cin) + "->" + std::to_string(d.Cout) + " @" + std::to_string(out.H) + "x" + std::to_string(out.W));
    RUN(conv_forward_impl(&a, c.s, cout0, 0, nullptr, fc));
    c.end();
}

static Tensor resblock(Ctx& c, const std::string& name, const std::vector<Tensor>& in, int Cout, bool keep_out = true, const FcFuse* fc = nullptr) {
    ofd_unet* u = c.u;
    const int H = in[0].H, W = in[0].W, B = c.B;
    std::vector<SrcSpec> srcs;
    int cin = 0;
    for (auto& t : in) { SrcSpec s; s.t = t; srcs.push_back(s); cin += t.C; }
    Tensor h1 = c.tmp(Cout, H, W), h2 = c.tmp(Cout, H, W);
    const size_t np = ofd_conv_gn_partial_count(B, H, W, Cout);
    float* p1 = c.tmpf(np);
    float* p2 = c.tmpf(np);
    // training keeps the folded affines and the statistics for the GroupNorm backward
    float* a1 = c.train ? c.keepf((size_t)B * Cout) : c.tmpf((size_t)B * Cout);
    float* s1 = c.train ? c.keepf((size_t)B * Cout) : c.tmpf((size_t)B * Cout);
    float* a2 = c.train ? c.keepf((size_t)B * Cout) : c.tmpf((size_t)B * Cout);
    float* s2 = c.train ? c.keepf((size_t)B * Cout) : c.tmpf((size_t)B * Cout);
    float* st1 = c.train ? c.keepf((size_t)B * 16) : nullptr;
    float* st2 = c.train ? c.keepf((size_t)B * 16) : nullptr;
    Tensor out = keep_out ? c.keep(Cout, H, W) : c.tmp(Cout, H, W);
    if (c.rc != OFD_OK) return out;
    conv(c, name + ".block1.proj", srcs, h1, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, p1);
    c.begin(PC_GN, 0, (double)np * 4);
    RUN(k_gn_finalize(p1, B, H, W, Cout, u->P(name + ".block1.norm.weight"), u->P(name + ".block1.norm.bias"), c.ss, u->ss_stride,
                      u->ss_offset.at(name), a1, s1, c.s, st1));
    c.end();
    SrcSpec hs; hs.t = h1;
    // inference, layers of several 128-channel output blocks: every block's workgroups would re-apply SiLU(GroupNorm(h1)) to the same staged tile
    // (an exp + a reciprocal per element, 2-4 times over); the small tensors of those levels are activated ONCE by an elementwise pass instead
    static const int act1_min = getenv("OFD_ACT1_MATERIALIZE_MIN") ? atoi(getenv("OFD_ACT1_MATERIALIZE_MIN")) : 256;
    // training: the weight gradient of block2.proj reads a materialised act1 for every layer wider than 64 channels anyway (unet_train.hip): the
    // forward makes it (and multiplies by it) instead of the backward
    static const int act1_train_min = getenv("OFD_ACT1_TRAIN_MIN") ? atoi(getenv("OFD_ACT1_TRAIN_MIN")) : 128;
    const int amin = c.train ? act1_train_min : act1_min;
    Tensor act1;
    if (amin > 0 && Cout >= amin) {
        act1 = c.tmp(Cout, H, W);
        if (c.rc != OFD_OK) return out;
        c.begin(PC_MISC, 0, (double)B * H * W * Cout * 4.0);
        RUN(k_affine_silu(h1.p, a1, s1, act1.p, B, H, W, Cout, c.s));
        c.end();
        hs.t = act1;
        conv(c, name + ".block2.proj", {hs}, h2, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, p2);
    } else
    conv(c, name + ".block2.proj", {hs}, h2, a1, s1, nullptr, nullptr, nullptr, nullptr, p2);
    c.begin(PC_GN, 0, (double)np * 4);
    RUN(k_gn_finalize(p2, B, H, W, Cout, u->P(name + ".block2.norm.weight"), u->P(name + ".block2.norm.bias"), nullptr, 0, 0, a2, s2, c.s, st2));
    c.end();
    if (cin != Cout) {
        conv(c, name + ".res_conv", srcs, out, nullptr, nullptr, nullptr, h2.p, a2, s2, nullptr, 0, fc);   // DD:214 fused into the 1x1 (fc: + the final 1x1 conv, `out` not written)
    } else {
        c.begin(PC_RESOUT, 0, (double)B * H * W * Cout * 6.0);
        RUN(k_resblock_out(h2.p, a2, s2, in[0].p, out.p, B, H, W, Cout, c.s));
        c.end();
    }
    if (c.train) {
        TapeRec r;
        r.kind = TK_RES; r.name = name; r.srcs = srcs; r.out = out; r.h1 = h1; r.h2 = h2;
        r.a1 = a1; r.s1 = s1; r.a2 = a2; r.s2 = s2; r.st1 = st1; r.st2 = st2; r.act1 = act1;
        u->tape.push_back(r);
    }
    return out;
}

static Tensor linattn(Ctx& c, const std::string& name, Tensor x) {
    ofd_unet* u = c.u;
    const int H = x.H, W = x.W, B = c.B, C = x.C, n = H * W;
    const size_t npix = (size_t)B * n;
    auto fit = u->la_fused.find(name);
    if (fit != u->la_fused.end() && !c.train) {
        // fused two-pass block (la_fused.hip): no LayerNorm / qkv / head-output tensors in HBM
        const int nparts = la_fused_blocks(n, B);
        float* partial = c.tmpf((size_t)B * 4 * nparts * 1088);
        bf16_t* ctxfrag = (bf16_t*)c.tmpf((size_t)B * 2048);
        Tensor y = c.keep(C, H, W);
        if (c.rc != OFD_OK) return y;
        const bf16_t* base = u->d_labuf + fit->second.first;
        c.begin(PC_LINATTN, npix * 2.0 * (C * 384.0 + 2 * 4 * 32 * 32 + 128.0 * C), (double)npix * C * 2 * 3,
                name + " fused C=" + std::to_string(C) + " @" + std::to_string(H) + "x" + std::to_string(W));
        RUN(k_linear_attention_fused(x.p, base, base + (size_t)128 * C, base + (size_t)384 * C, u->P(name + ".fn.fn.to_out.0.bias"),
                                     u->P(name + ".fn.fn.to_out.1.g"), partial, ctxfrag, y.p, B, n, C, site_eps(u, name + ".fn.norm"),
                                     site_eps(u, name + ".fn.fn.to_out.1"), c.s));
        c.end();
        return y;
    }
    Tensor xn = c.tmp(C, H, W), qkv = c.tmp(384, H, W), ao = c.tmp(128, H, W), o2 = c.tmp(C, H, W);
    const int nparts = la_fwd_parts(B, n);
    if (c.train && fit != u->la_fused.end() && la_train_fused(C)) {
        // training, 64 channels: the two fused passes leave the tape themselves (xn, k | v, o2, ctx, ml): no LayerNorm kernels, no to_qkv conv
        float* partial = c.tmpf((size_t)B * 4 * nparts * 1088);
        bf16_t* ctxfrag = (bf16_t*)c.tmpf((size_t)B * 2048);
        float* ctx = c.keepf((size_t)B * 4 * 1024);
        float* ml = c.keepf((size_t)B * 4 * 64);
        Tensor y = c.keep(C, H, W);
        if (c.rc != OFD_OK) return y;
        const bf16_t* base = u->d_labuf + fit->second.first;
        c.begin(PC_LINATTN, npix * 2.0 * (C * 640.0 + 2 * 4 * 32 * 32 + 128.0 * C), (double)npix * C * 2 * (2 + 1 + 4 + 2),
                name + " fused (training) C=" + std::to_string(C) + " @" + std::to_string(H) + "x" + std::to_string(W));
        RUN(k_linear_attention_fused_train(x.p, base + (size_t)512 * C, base + (size_t)640 * C, base + (size_t)384 * C, u->P(name + ".fn.fn.to_out.0.bias"),
                                           u->P(name + ".fn.norm.g"), u->P(name + ".fn.fn.to_out.1.g"), partial, ctxfrag, ctx, ml, xn.p, qkv.p, o2.p, y.p,
                                           B, n, C, site_eps(u, name + ".fn.norm"), site_eps(u, name + ".fn.fn.to_out.1"), c.s));
        c.end();
        TapeRec r;
        r.kind = TK_LINATTN; r.name = name; r.x = x; r.xn = xn; r.qkv = qkv; r.ao = ao; r.o2 = o2; r.out = y; r.ctx = ctx; r.ml = ml;
        u->tape.push_back(r);
        return y;
    }
    float* partial = c.tmpf((size_t)B * 4 * nparts * 1088);
    float* ctx = c.train ? c.keepf((size_t)B * 4 * 1024) : c.tmpf((size_t)B * 4 * 1024);
    float* ml = c.train ? c.keepf((size_t)B * 4 * 64) : nullptr;
    Tensor y = c.keep(C, H, W);
    if (c.rc != OFD_OK) return y;
    c.begin(PC_LN, 0, (double)npix * C * 4);
    RUN(k_layernorm_c(x.p, u->P(name + ".fn.norm.g"), nullptr, xn.p, npix, C, site_eps(u, name + ".fn.norm"), c.s));
    c.end();
    SrcSpec s; s.t = xn;
    // training, 64 channels, every fusion on and OFD_LA_RECOMPUTE_Q: the passes that need q re-derive it from xn -- only k and v are computed
    const bool rq_conv = c.train && la_train_no_ao(C) && la_recompute_q();
    conv(c, name + ".fn.fn.to_qkv", {s}, qkv, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, rq_conv ? 128 : 0);
    c.begin(PC_LINATTN, 4.0 * npix * 4 * 32 * 32, (double)npix * (384 + 128 + 128) * 2);
    if (C <= 128 && la_fuse_to_out()) {     // to_out.0 rides on the head-output tile of the core's second pass (la_core.hip lc_out_kernel)
        const ConvDesc& d = u->convs[u->cindex.at(name + ".fn.fn.to_out.0")];
        // training, 64 channels, all fusions on: nothing reads the head outputs again (the backward derives what it needs from ctx and q)
        bf16_t* ao_out = (c.train && la_train_no_ao(C)) ? nullptr : ao.p;
        const bool rq = c.train && la_train_no_ao(C) && la_recompute_q();      // q re-derived from xn in every pass that needs it
        const ConvDesc& dq = u->convs[u->cindex.at(name + ".fn.fn.to_qkv")];
        RUN(k_linear_attention_core(qkv.p, partial, ctx, ao_out, B, n, c.s, ml, u->d_wbuf + d.w_off, u->P(name + ".fn.fn.to_out.0.bias"), o2.p, C,
                                    rq ? xn.p : nullptr, rq ? u->d_wbuf + dq.w_off : nullptr));
        c.end();
    } else {
        RUN(k_linear_attention_core(qkv.p, partial, ctx, ao.p, B, n, c.s, ml));
        c.end();
        SrcSpec s2; s2.t = ao;
        conv(c, name + ".fn.fn.to_out.0", {s2}, o2, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
    }
    c.begin(PC_LN, 0, (double)npix * C * 6);
    RUN(k_layernorm_c(o2.p, u->P(name + ".fn.fn.to_out.1.g"), x.p, y.p, npix, C, site_eps(u, name + ".fn.fn.to_out.1"), c.s));
    c.end();
    if (c.train) {
        TapeRec r;
        r.kind = TK_LINATTN; r.name = name; r.x = x; r.xn = xn; r.qkv = qkv; r.ao = ao; r.o2 = o2; r.out = y; r.ctx = ctx; r.ml = ml;
        u->tape.push_back(r);
    }
    return y;
}

static Tensor midattn(Ctx& c, Tensor x) {
    ofd_unet* u = c.u;
    const int H = x.H, W = x.W, B = c.B, C = x.C, n = H * W;
    Tensor xn = c.tmp(C, H, W), qkv = c.tmp(384, H, W), ao = c.tmp(128, H, W);
    float* lse = c.train ? c.keepf((size_t)B * 4 * n) : nullptr;
    Tensor y = c.keep(C, H, W);
    if (c.rc != OFD_OK) return y;
    const size_t npix = (size_t)B * n;
    c.begin(PC_LN, 0, (double)npix * C * 4);
    RUN(k_layernorm_c(x.p, u->P("mid_attn.fn.norm.g"), nullptr, xn.p, npix, C, site_eps(u, "mid_attn.fn.norm"), c.s));
    c.end();
    SrcSpec s; s.t = xn;
    conv(c, "mid_attn.fn.fn.to_qkv", {s}, qkv, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
    c.begin(PC_FLASH, 4.0 * B * 4.0 * (double)n * n * 32, (double)npix * (384 + 128) * 2);
    RUN(k_flash_attention(qkv.p, ao.p, B, n, c.s, lse));
    c.end();
    SrcSpec s2; s2.t = ao;
    conv(c, "mid_attn.fn.fn.to_out", {s2}, y, nullptr, nullptr, x.p, nullptr, nullptr, nullptr, nullptr);   // + x (DD:87)
    if (c.train) {
        TapeRec r;
        r.kind = TK_MIDATTN; r.name = "mid_attn"; r.x = x; r.xn = xn; r.qkv = qkv; r.ao = ao; r.out = y; r.lse = lse;
        u->tape.push_back(r);
    }
    return y;
}

// a bare convolution between blocks (init, down / up sampling): recorded for the backward
static void plain_conv(Ctx& c, const std::string& prefix, const std::vector<SrcSpec>& srcs, Tensor out) {
    conv(c, prefix, srcs, out, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
    if (c.train) {
        TapeRec r;
        r.kind = TK_CONV; r.name = prefix; r.srcs = srcs; r.out = out;
        c.u->tape.push_back(r);
    }
}

// Upsample(x2, nearest) + 3x3 (DD:89-93) as four 2x2 phase convs on the low-res tensor: 2.25x fewer MACs than
// convolving the up-sampled tensor.  The tape records the op in its reference form (the backward differentiates that).
static void upsample_conv(Ctx& c, const std::string& prefix, const SrcSpec& src_up, Tensor out) {
    ofd_unet* u = c.u;
    const ConvDesc& d = u->convs[u->cindex.at(prefix)];
    static int no_phase = -1;
    if (no_phase < 0) { const char* e = getenv("OFD_NO_PHASE_UPSAMPLE"); no_phase = (e && atoi(e)) ? 1 : 0; }
    if (d.phase_off < 0 || no_phase) { plain_conv(c, prefix, {src_up}, out); return; }
    if (c.rc != OFD_OK) return;
    const Tensor& lo = src_up.t;
    const double px = (double)c.B * out.H * out.W;
    c.begin(PC_CONVUP, 2.0 * px * d.Cout * (double)d.Cin * 9 /* algorithmic: as the reference executes it */, px * 2.0 * (d.Cout + d.Cin / 4.0),
            prefix + " " + std::to_string(d.Cin) + "->" + std::to_string(d.Cout) + " @" + std::to_string(out.H) + "x" + std::to_string(out.W) + " (4 phases)");
    static const int one_launch = getenv("OFD_PHASE_ONE_LAUNCH") ? atoi(getenv("OFD_PHASE_ONE_LAUNCH")) : 1;
    for (int ph = 0; ph < (one_launch ? 1 : 4); ++ph) {
        ofd_conv_args a{};
        a.B = c.B; a.H = lo.H; a.W = lo.W; a.ksize = 2; a.n_src = 1; a.Cout = d.Cout;
        a.src[0].src = lo.p; a.src[0].channels = lo.C; a.src[0].src_channels = lo.C;
        a.weight = u->d_wbuf + d.phase_off + (size_t)ph * 4 * d.Cin * d.Cout;
        a.bias = u->P(prefix + ".bias");
        a.out = out.p;
        a.up2_phase = one_launch ? 5 : ph + 1;          // 5: the four phases in one launch (phases of a tile on one XCD: conv_params.h)
        RUN(conv_forward_impl(&a, c.s));
    }
    c.end();
    if (c.train) {
        TapeRec r;
        r.kind = TK_CONV; r.name = prefix; r.srcs = {src_up}; r.out = out;
        u->tape.push_back(r);
    }
}

int run_forward(Ctx& c, const float* x, int Cx, const float* cond, int Cc, const int64_t* t, float* out, int H, int W, float* temb,
                float* temb_silu) {
    ofd_unet* u = c.u;
    const int B = c.B, dim = u->cfg.dim;
    u->taps.clear();
    u->last_B = B;
    if (c.train) u->tape.clear();
    // <= 8 input channels: 8-channel packing, the 7x7 pairs horizontally adjacent taps into one k-step (conv_igemm.hip Cfg::K7P).
    // Training keeps the same 8-channel tensor on the tape: its weight-gradient kernel reads either packing (conv_bwd.hip conv7_wgrad_kernel<8>;
    // r04: the 16-channel copy and its pack pass are gone).
    static const bool no_pack8 = getenv("OFD_NO_CONV7_PACK8") && atoi(getenv("OFD_NO_CONV7_PACK8"));
    const bool pack8 = !no_pack8 && u->convs[u->cindex.at("init_conv")].pack8_off >= 0;
    const int cpad = pack8 ? 8 : 16;
    Tensor xin = c.keep(cpad, H, W);
    c.begin(PC_MISC, 0, 0);
    RUN(k_pack_input(x, Cx, cond, cond ? Cc : 0, xin.p, B, H, W, c.s, cpad));
    if (!u->cfg.no_time) {
        RUN(k_time_mlp(t, u->P("time_mlp.1.weight"), u->P("time_mlp.1.bias"), u->P("time_mlp.3.weight"), u->P("time_mlp.3.bias"), temb, temb_silu, B, dim, c.s));
        RUN(k_block_mlp(temb_silu, u->d_mlp, (int)u->resblocks.size(), c.ss, B, dim * 4, u->ss_stride, c.s));
    } else {
        c.ss = nullptr;          // GroupNorm of block1 folds no scale/shift (DD:205-208 with scale_shift = None)
    }
    c.end();
    Tensor r = c.keep(dim, H, W);
    { SrcSpec s; s.t = xin; plain_conv(c, "init_conv", {s}, r); }
    u->taps["init_conv"] = r;

    Tensor xcur = r;
    std::vector<Tensor> hs;
    for (int i = 0; i < 4; ++i) {
        const std::string p = "downs." + std::to_string(i);
        c.reset_scratch();
        xcur = resblock(c, p + ".0", {xcur}, u->dims[i]);
        hs.push_back(xcur);
        u->taps[p + ".0"] = xcur;
        c.reset_scratch();
        Tensor x1 = resblock(c, p + ".1", {xcur}, u->dims[i]);
        u->taps[p + ".1"] = x1;
        c.reset_scratch();
        xcur = linattn(c, p + ".2", x1);
        hs.push_back(xcur);
        u->taps[p + ".2"] = xcur;
        if (i < 3) {
            Tensor d = c.keep(u->dims[i + 1], xcur.H / 2, xcur.W / 2);
            std::vector<SrcSpec> srcs;
            for (int sub = 0; sub < 4; ++sub) { SrcSpec s; s.t = xcur; s.unshuffle = 1; s.p1 = sub >> 1; s.p2 = sub & 1; srcs.push_back(s); }
            plain_conv(c, p + ".3.1", srcs, d);
            xcur = d;
        } else {
            Tensor d = c.keep(u->dims[i + 1], xcur.H, xcur.W);
            SrcSpec s; s.t = xcur;
            plain_conv(c, p + ".3", {s}, d);
            xcur = d;
        }
        u->taps[p + ".3"] = xcur;
    }
    c.reset_scratch();
    xcur = resblock(c, "mid_block1", {xcur}, u->dims[4]);
    u->taps["mid_block1"] = xcur;
    c.reset_scratch();
    xcur = midattn(c, xcur);
    u->taps["mid_attn"] = xcur;
    c.reset_scratch();
    xcur = resblock(c, "mid_block2", {xcur}, u->dims[4]);
    u->taps["mid_block2"] = xcur;
    for (int i = 0; i < 4; ++i) {
        const std::string p = "ups." + std::to_string(i);
        const int co = u->dims[4 - i], ci = u->dims[3 - i];
        c.reset_scratch();
        Tensor h = hs.back(); hs.pop_back();
        xcur = resblock(c, p + ".0", {xcur, h}, co);
        u->taps[p + ".0"] = xcur;
        c.reset_scratch();
        h = hs.back(); hs.pop_back();
        xcur = resblock(c, p + ".1", {xcur, h}, co);
        u->taps[p + ".1"] = xcur;
        c.reset_scratch();
        xcur = linattn(c, p + ".2", xcur);
        u->taps[p + ".2"] = xcur;
        if (i < 3) {
            Tensor d = c.keep(ci, xcur.H * 2, xcur.W * 2);
            SrcSpec s; s.t = xcur; s.upsample = 1;
            upsample_conv(c, p + ".3.1", s, d);
            xcur = d;
        } else {
            Tensor d = c.keep(ci, xcur.H, xcur.W);
            SrcSpec s; s.t = xcur;
            plain_conv(c, p + ".3", {s}, d);
            xcur = d;
        }
        u->taps[p + ".3"] = xcur;
    }
    c.reset_scratch();
    // inference: the final 1x1 conv (DD:361) rides on the tile of final_res_block's res_conv (conv1_wp.hip, FC): the block's 64-channel
    // output tensor is neither written nor read.  Off when the tensor itself is wanted (training tape, ofd_unet_set_debug_taps) or the
    // shape is not the streaming kernel's (out_dim 2, whole 128-pixel tiles)
    static const bool no_fc = (getenv("OFD_NO_FC_FUSE") && atoi(getenv("OFD_NO_FC_FUSE"))) || (getenv("OFD_CONV1_WP") && atoi(getenv("OFD_CONV1_WP")) == 0) ||
                              (getenv("OFD_CONV_DBG") && atoi(getenv("OFD_CONV_DBG")));
    const bool fuse_fc = !c.train && !u->debug_taps && !no_fc && u->cfg.out_dim == 2 && dim == 64 && ((long)H * W) % 128 == 0;
    if (fuse_fc) {
        const FcFuse fc{u->P("final_conv.weight"), u->P("final_conv.bias"), out};
        if (c.rc == OFD_OK && !c.dry) {
            if (hipMemsetAsync(out, 0, (size_t)B * 2 * H * W * sizeof(float), c.s) != hipSuccess) { set_error("unet_forward: hipMemsetAsync failed"); c.rc = OFD_ERR_HIP; }
            c.chain_ok = false;
        }
        xcur = resblock(c, "final_res_block", {xcur, r}, dim, true, &fc);
    } else {
        xcur = resblock(c, "final_res_block", {xcur, r}, dim);
        u->taps["final_res_block"] = xcur;
        c.begin(PC_MISC, 2.0 * B * H * W * dim * u->cfg.out_dim, (double)B * H * W * (dim * 2 + u->cfg.out_dim * 4));
        RUN(k_final_conv(xcur.p, u->P("final_conv.weight"), u->P("final_conv.bias"), out, B, H, W, dim, u->cfg.out_dim, c.s));
        c.end();
    }
    if (c.train) {
        u->ts.B = B; u->ts.H = H; u->ts.W = W; u->ts.t = t; u->ts.ss = c.ss; u->ts.temb = temb; u->ts.temb_silu = temb_silu;
        u->ts.xin = xin; u->ts.r = r; u->ts.xf = xcur; u->ts.persist_used = c.persist_used;
    }
    return c.rc;
}

size_t persist_bytes(const ofd_unet* u, int B, int H, int W) {
    // every kept tensor of the forward: r, 3 per down level, 1 resample per level, mid (3), 3+1 per up level, final
    auto T = [&](int C, int h, int w) { return ((size_t)B * h * w * C * 2 + 255) / 256 * 256; };
    size_t n = T(16, H, W) + T(u->dims[0], H, W);
    int h = H, w = W;
    for (int i = 0; i < 4; ++i) {
        n += 3 * T(u->dims[i], h, w);
        if (i < 3) { h /= 2; w /= 2; }
        n += T(u->dims[i + 1], h, w);
    }
    n += 3 * T(u->dims[4], h, w);
    for (int i = 0; i < 4; ++i) {
        const int co = u->dims[4 - i], ci = u->dims[3 - i];
        n += 3 * T(co, h, w);
        if (i < 3) { h *= 2; w *= 2; }
        n += T(ci, h, w);
    }
    n += T(u->dims[0], H, W);
    return n + 4096;
}

size_t scratch_bytes(const ofd_unet* u, int B, int H, int W) {
    // largest per-block need, taken at full resolution with C = dim (the 1/2, 1/4, 1/8 levels
    // have 2x channels on 1/4 of the pixels): h1 + h2 + out(tmp) + xn + qkv + ao + o2 + small
    const size_t px = (size_t)B * H * W;
    const int C = u->dims[0];
    size_t act = px * 2 * (size_t)(3 * C + C + 384 + 128 + C);
    // mid level widest: 512 ch at 1/64 of the pixels is far smaller; small buffers:
    size_t small = 4 * ofd_conv_gn_partial_count(B, H, W, u->dims[4]) * 4 + (size_t)B * 4 * ((size_t)(la_fwd_parts(B, H * W) > 256 ? la_fwd_parts(B, H * W) : 256) * 1088 + 1024) * 4 +
                   16 * (size_t)B * u->dims[4] * 4 + 64 * 1024;
    return act + small + 64 * 256;
}

size_t small_bytes(const ofd_unet* u, int B) {
    return ((size_t)B * u->ss_stride + 2 * (size_t)B * u->cfg.dim * 4) * 4 + 4096;
}

}  // namespace ofd

static void drop_graphs(ofd_unet* u);

static void upload_mlp_descs(ofd_unet* u) {
    if (u->cfg.no_time) return;
    std::vector<MlpDesc> descs;
    for (auto& name : u->resblocks) {
        MlpDesc d;
        d.weight = u->P(name + ".mlp.1.weight");
        d.bias = u->P(name + ".mlp.1.bias");
        d.n_out = (int)u->params[u->pindex.at(name + ".mlp.1.bias")].numel;
        d.offset = u->ss_offset.at(name);
        descs.push_back(d);
    }
    if (hipMemcpy(u->d_mlp, descs.data(), descs.size() * sizeof(MlpDesc), hipMemcpyHostToDevice) != hipSuccess)
        set_error("unet: uploading the time-projection descriptors failed");      // (surfaces at the first launch that reads them)
}

extern "C" int ofd_unet_bind_param_buffer(ofd_unet* u, float* dev_params, size_t floats) {
    OFD_CHECK_ARG(u && dev_params, "unet_bind_param_buffer: null argument");
    OFD_CHECK_ARG(floats >= u->n_param_floats, "unet_bind_param_buffer: %zu floats, need %zu", floats, u->n_param_floats);
    OFD_CHECK_ARG(((uintptr_t)dev_params & 15) == 0, "unet_bind_param_buffer: buffer must be 16-byte aligned");
    if (u->owns_params && u->d_params) (void)hipFree(u->d_params);
    drop_graphs(u);
    // the batched weight-preparation table caches u->P(...) pointers into the OLD buffer: rebuild it on the next prepare
    if (u->d_prep) { (void)hipFree(u->d_prep); u->d_prep = nullptr; u->n_prep = 0; u->prep_blocks = 0; }
    u->d_params = dev_params;
    u->owns_params = false;
    for (auto& p : u->params) p.set = true;          // the caller's buffer holds every parameter
    upload_mlp_descs(u);
    u->prepared = false;
    u->wt_prepared = false;
    return OFD_OK;
}

extern "C" int ofd_unet_create(const ofd_unet_config* cfg, ofd_unet** out) {
    OFD_CHECK_ARG(cfg && out, "unet_create: null argument");
    OFD_CHECK_ARG(cfg->dim == 64, "unet_create: dim=%d unsupported (the FlowDiffuser UNet is Unet(64), flow_diffuser.py:106)", cfg->dim);
    OFD_CHECK_ARG(cfg->channels >= 1 && cfg->channels <= 16, "unet_create: channels=%d (1..16)", cfg->channels);
    OFD_CHECK_ARG(cfg->out_dim >= 1 && cfg->out_dim <= 4, "unet_create: out_dim=%d (1..4)", cfg->out_dim);
    OFD_CHECK_ARG(cfg->eps_mode == 0 || cfg->eps_mode == 1, "unet_create: eps_mode=%d", cfg->eps_mode);
    OFD_CHECK_ARG(cfg->no_time == 0 || cfg->no_time == 1, "unet_create: no_time=%d", cfg->no_time);
    ofd_unet* u = new ofd_unet();
    u->cfg = *cfg;
    { const char* e = getenv("OFD_DETERMINISTIC"); u->deterministic = e && atoi(e) != 0; }      // default of ofd_unet_set_deterministic
    build_registry(u);
    if (hipMalloc(&u->d_params, u->n_param_floats * sizeof(float)) != hipSuccess ||
        hipMalloc(&u->d_wbuf, u->n_wbuf * sizeof(bf16_t)) != hipSuccess ||
        hipMalloc(&u->d_mlp, u->resblocks.size() * sizeof(MlpDesc)) != hipSuccess ||
        hipMalloc(&u->d_labuf, (u->n_labuf + 8) * sizeof(bf16_t)) != hipSuccess) {
        set_error("unet_create: hipMalloc failed");
        ofd_unet_destroy(u);
        return OFD_ERR_HIP;
    }
    if (hipMemset(u->d_params, 0, u->n_param_floats * sizeof(float)) != hipSuccess) {
        set_error("unet_create: hipMemset failed");
        ofd_unet_destroy(u);
        return OFD_ERR_HIP;
    }
    upload_mlp_descs(u);
    *out = u;
    return OFD_OK;
}

extern "C" void ofd_unet_destroy(ofd_unet* u) {
    if (!u) return;
    if (u->d_params && u->owns_params) (void)hipFree(u->d_params);
    if (u->d_wbuf) (void)hipFree(u->d_wbuf);
    if (u->d_mlp) (void)hipFree(u->d_mlp);
    if (u->d_labuf) (void)hipFree(u->d_labuf);
    if (u->d_wtbuf) (void)hipFree(u->d_wtbuf);
    if (u->d_wacc) (void)hipFree(u->d_wacc);
    if (u->d_fx) (void)hipFree(u->d_fx);
    if (u->d_det_miss) (void)hipFree(u->d_det_miss);
    if (u->d_prep) (void)hipFree(u->d_prep);
    if (u->d_tr) (void)hipFree(u->d_tr);
    drop_graphs(u);
    if (u->s2) {
        (void)hipStreamDestroy(u->s2);
        (void)hipEventDestroy(u->ev_fork); (void)hipEventDestroy(u->ev_join); (void)hipEventDestroy(u->ev_phase);
    }
    for (auto e : u->pool) (void)hipEventDestroy(e);
    delete u;
}

extern "C" int ofd_unet_num_params(const ofd_unet* u) { return u ? (int)u->params.size() : 0; }
extern "C" const char* ofd_unet_param_name(const ofd_unet* u, int i) {
    return (u && i >= 0 && i < (int)u->params.size()) ? u->params[i].name.c_str() : "";
}
extern "C" size_t ofd_unet_param_numel(const ofd_unet* u, int i) {
    return (u && i >= 0 && i < (int)u->params.size()) ? u->params[i].numel : 0;
}

extern "C" int ofd_unet_param_shape(const ofd_unet* u, int i, int* dims4) {
    if (!u || i < 0 || i >= (int)u->params.size() || !dims4) return 0;
    for (int k = 0; k < 4; ++k) dims4[k] = u->params[i].shape[k];
    return u->params[i].ndim;
}

extern "C" int ofd_unet_set_param(ofd_unet* u, int i, const float* dev_src, size_t numel, void* stream) {
    OFD_CHECK_ARG(u && dev_src && i >= 0 && i < (int)u->params.size(), "unet_set_param: bad argument");
    Param& p = u->params[i];
    OFD_CHECK_ARG(numel == p.numel, "unet_set_param: %s has %zu elements, got %zu", p.name.c_str(), p.numel, numel);
    OFD_HIP(hipMemcpyAsync(u->d_params + p.offset, dev_src, numel * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    p.set = true;
    u->prepared = false;
    u->wt_prepared = false;
    return OFD_OK;
}

extern "C" int ofd_unet_prepare(ofd_unet* u, void* stream) {
    OFD_CHECK_ARG(u, "unet_prepare: null handle");
    for (auto& p : u->params)
        if (!p.set) {
            set_error("unet_prepare: parameter %s was never set", p.name.c_str());
            return OFD_ERR_STATE;
        }
    // every conv's fp32 OIHW -> prepared bf16 weights in one launch (descriptor table built once per bound parameter buffer:
    // ofd_unet_bind_param_buffer drops it, ofd_unet_set_param writes into the same storage)
    if (!u->d_prep) {
        std::vector<ofd_weight_prep_desc> h;
        int blocks = 0;
        auto add = [&](const ConvDesc& c, size_t off, int cin_pad, int unshuffle) {
            ofd_weight_prep_desc d{u->P(c.wname), u->d_wbuf + off, c.Cout, c.Cin, cin_pad, c.ksize, c.ws_eps, unshuffle, blocks};
            h.push_back(d);
            blocks += c.Cout;
        };
        for (auto& c : u->convs) {
            add(c, c.w_off, c.Cin_pad, c.unshuffle);
            if (c.pack8_off >= 0) add(c, (size_t)c.pack8_off, 8, 0);
        }
        OFD_HIP(hipMalloc(&u->d_prep, h.size() * sizeof(ofd_weight_prep_desc)));
        OFD_HIP(hipMemcpy(u->d_prep, h.data(), h.size() * sizeof(ofd_weight_prep_desc), hipMemcpyHostToDevice));
        u->n_prep = (int)h.size();
        u->prep_blocks = blocks;
    }
    {
        int rc = k_conv_weight_prep_batched(u->d_prep, u->n_prep, u->prep_blocks, (hipStream_t)stream);
        if (rc != OFD_OK) return rc;
    }
    for (auto& c : u->convs) {
        if (c.phase_off >= 0) {
            int rc = ofd_conv_upsample_phase_weight_prep(u->P(c.wname), u->d_wbuf + c.phase_off, c.Cout, c.Cin, stream);
            if (rc != OFD_OK) return rc;
        }
    }
    for (auto& kv : u->la_fused) {
        const std::string& name = kv.first;
        const int C = kv.second.second;
        bf16_t* base = u->d_labuf + kv.second.first;
        int rc = k_la_weight_prep(u->P(name + ".fn.fn.to_qkv.weight"), u->P(name + ".fn.norm.g"), u->P(name + ".fn.fn.to_out.0.weight"),
                                  base, base + (size_t)128 * C, base + (size_t)384 * C, C, (hipStream_t)stream);
        if (rc != OFD_OK) return rc;
        if (C == 64) {
            rc = k_la_weight_prep(u->P(name + ".fn.fn.to_qkv.weight"), nullptr, nullptr, base + (size_t)512 * C, base + (size_t)640 * C, nullptr, C, (hipStream_t)stream);
            if (rc != OFD_OK) return rc;
        }
    }
    u->prepared = true;
    u->wt_prepared = false;
    return OFD_OK;
}

static size_t staging_bytes(const ofd_unet* u, int B, int H, int W) {
    // fixed homes of x | cond (together cfg.channels planes), t, out for graph replay
    auto al = [](size_t n) { return (n + 255) / 256 * 256; };
    return al((size_t)B * u->cfg.channels * H * W * 4) + al((size_t)B * 8) + al((size_t)B * u->cfg.out_dim * H * W * 4);
}

static size_t workspace_bytes_one(const ofd_unet* u, int B, int H, int W) {
    return persist_bytes(u, B, H, W) + scratch_bytes(u, B, H, W) + small_bytes(u, B) + 1024 + staging_bytes(u, B, H, W);
}

// two half-batch forwards on two streams (ofd_unet_set_split_streams): each half owns one half of the workspace
static size_t half_workspace_bytes(const ofd_unet* u, int B, int H, int W) { return (workspace_bytes_one(u, B / 2, H, W) + 255) / 256 * 256; }

extern "C" size_t ofd_unet_workspace_bytes(const ofd_unet* u, int B, int H, int W) {
    if (!u || B <= 0 || H <= 0 || W <= 0) return 0;
    const size_t one = workspace_bytes_one(u, B, H, W);
    if (B >= 2 && B % 2 == 0) { const size_t two = 2 * half_workspace_bytes(u, B, H, W); return two > one ? two : one; }
    return one;
}

static void drop_graphs(ofd_unet* u) {
    for (auto& g : u->graphs) {
        if (g.state == 2) { (void)hipGraphExecDestroy(g.exec); (void)hipGraphDestroy(g.graph); }
    }
    u->graphs.clear();
    if (u->cap_stream) { (void)hipStreamDestroy(u->cap_stream); u->cap_stream = nullptr; }
}

extern "C" int ofd_unet_set_graph(ofd_unet* u, int enabled) {
    OFD_CHECK_ARG(u, "unet_set_graph: null handle");
    u->graph_enabled = enabled != 0;
    if (!u->graph_enabled) drop_graphs(u);
    return OFD_OK;
}

extern "C" int ofd_unet_forward(ofd_unet* u, const float* x, int Cx, const float* cond, int Cc, const int64_t* t, float* out,
                                int B, int H, int W, void* workspace, size_t workspace_bytes, void* stream) {
    OFD_CHECK_ARG(u && x && (t || u->cfg.no_time) && out && workspace, "unet_forward: null argument");
    OFD_CHECK_ARG(B > 0 && H > 0 && W > 0 && H % 8 == 0 && W % 8 == 0, "unet_forward: H=%d W=%d must be positive multiples of 8 (three 2x down-samplings, DD:95-99)", H, W);
    OFD_CHECK_ARG(Cx + (cond ? Cc : 0) == u->cfg.channels, "unet_forward: %d + %d input channels, UNet has %d", Cx, cond ? Cc : 0, u->cfg.channels);
    if (!u->prepared) { set_error("unet_forward: call ofd_unet_prepare after setting parameters"); return OFD_ERR_STATE; }
    if (workspace_bytes < ofd_unet_workspace_bytes(u, B, H, W)) {
        set_error("unet_forward: workspace %zu < %zu", workspace_bytes, ofd_unet_workspace_bytes(u, B, H, W));
        return OFD_ERR_WORKSPACE;
    }
    OFD_CHECK_ARG(((uintptr_t)workspace & 255) == 0, "unet_forward: workspace must be 256-byte aligned");
    // lays a forward context of batch `b` over `bytes` of workspace at `w_`
    auto layout = [&](Ctx& c, int b, char* w_, size_t bytes, hipStream_t s_, float*& temb_, float*& temb_silu_) {
        c.u = u; c.s = s_; c.B = b;
        const size_t sb = (small_bytes(u, b) + 255) / 256 * 256;
        float* fsmall = (float*)w_;
        c.ss = fsmall;
        temb_ = fsmall + (size_t)b * u->ss_stride;
        temb_silu_ = temb_ + (size_t)b * u->cfg.dim * 4;
        c.persist = w_ + sb;
        c.persist_cap = persist_bytes(u, b, H, W);
        c.scratch = c.persist + (c.persist_cap + 255) / 256 * 256;
        c.scratch_cap = bytes - staging_bytes(u, b, H, W) - (size_t)(c.scratch - w_);
    };
    char* w = (char*)workspace;
    // ---- two half-batches on two streams (DESIGN section 4.1 "co-scheduling"): samples are independent in every kernel of the
    // network (GroupNorm, LinearAttention and attention are per sample), so half 1 runs the same launch sequence on a second stream,
    // started `split_offset` blocks behind half 0 -- the MFMA-bound 3x3 kernels of one half share the chip with the HBM-bound
    // 1x1 / normalisation / LinearAttention kernels of the other instead of running strictly in series
    const bool split_auto = u->split_streams < 0 && (long)B * H * W >= (1L << 21);
    if ((u->split_streams > 0 || split_auto) && B >= 2 && B % 2 == 0 && !u->profiling && !u->graph_enabled) {
        if (!u->s2) {
            OFD_HIP(hipStreamCreateWithFlags(&u->s2, hipStreamNonBlocking));
            OFD_HIP(hipEventCreateWithFlags(&u->ev_fork, hipEventDisableTiming));
            OFD_HIP(hipEventCreateWithFlags(&u->ev_join, hipEventDisableTiming));
            OFD_HIP(hipEventCreateWithFlags(&u->ev_phase, hipEventDisableTiming));
        }
        const int hb = B / 2;
        const size_t hbytes = half_workspace_bytes(u, B, H, W);
        hipStream_t s1 = (hipStream_t)stream;
        Ctx c0, c1;
        float *te0, *ts0, *te1, *ts1;
        layout(c0, hb, w, hbytes, s1, te0, ts0);
        layout(c1, hb, w + hbytes, hbytes, u->s2, te1, ts1);
        const size_t plane = (size_t)H * W;
        c0.signal_at = u->split_offset; c0.signal_ev = u->ev_phase;
        OFD_HIP(hipEventRecord(u->ev_fork, s1));                       // the inputs are ready on the caller's stream
        OFD_HIP(hipStreamWaitEvent(u->s2, u->ev_fork, 0));
        // From here on work may be queued on the library's own stream s2: EVERY path below reaches the join (record on s2, wait on the
        // caller's stream) before returning, so that s2 never holds work on the caller's workspace / output that the caller's stream is not
        // ordered behind -- also when a launch or an event call in between fails.  Half 1 is not launched when half 0 failed.
        int rc = run_forward(c0, x, Cx, cond, Cc, t, out, H, W, te0, ts0), rc1 = OFD_OK;
        hipError_t he = hipSuccess;
        if (rc == OFD_OK) {
            if (!c0.signalled) he = hipEventRecord(u->ev_phase, s1);
            u->taps_half0 = u->taps;
            if (he == hipSuccess) he = hipStreamWaitEvent(u->s2, u->ev_phase, 0);
            if (he == hipSuccess)
                rc1 = run_forward(c1, x + (size_t)hb * Cx * plane, Cx, cond ? cond + (size_t)hb * Cc * plane : nullptr, Cc, t ? t + hb : nullptr,
                                  out + (size_t)hb * u->cfg.out_dim * plane, H, W, te1, ts1);
        }
        hipError_t hj = hipEventRecord(u->ev_join, u->s2);
        if (hj == hipSuccess) hj = hipStreamWaitEvent(s1, u->ev_join, 0);          // the caller's stream owns the whole output again
        if (hj != hipSuccess) (void)hipStreamSynchronize(u->s2);                    // the join itself failed: drain s2 before handing control back
        if (he != hipSuccess || hj != hipSuccess) {
            set_error("unet_forward (split streams): %s", hipGetErrorString(he != hipSuccess ? he : hj));
            if (rc == OFD_OK && rc1 == OFD_OK) rc = OFD_ERR_HIP;
        }
        u->last_split = true;
        return rc != OFD_OK ? rc : rc1;
    }
    u->last_split = false;
    Ctx c;
    float *temb, *temb_silu;
    layout(c, B, w, workspace_bytes, (hipStream_t)stream, temb, temb_silu);
    const size_t stg = staging_bytes(u, B, H, W);
    if (!u->graph_enabled || u->profiling) return run_forward(c, x, Cx, cond, Cc, t, out, H, W, temb, temb_silu);

    // ---- graph replay: stage the inputs at fixed addresses, (capture once and) launch the whole forward as ONE graph
    auto al = [](size_t n) { return (n + 255) / 256 * 256; };
    char* sbase = w + workspace_bytes - stg;
    sbase = (char*)(((uintptr_t)sbase) & ~(uintptr_t)255);
    const size_t plane = (size_t)H * W * 4;
    float* sx = (float*)sbase;                                    // x planes then cond planes, per sample as given
    float* sc = sx + (size_t)B * Cx * H * W;
    int64_t* st = (int64_t*)(sbase + al((size_t)B * u->cfg.channels * plane));
    float* sout = (float*)((char*)st + al((size_t)B * 8));
    hipStream_t s_ = (hipStream_t)stream;
    OFD_HIP(hipMemcpyAsync(sx, x, (size_t)B * Cx * plane, hipMemcpyDeviceToDevice, s_));
    if (cond) OFD_HIP(hipMemcpyAsync(sc, cond, (size_t)B * Cc * plane, hipMemcpyDeviceToDevice, s_));
    if (t) OFD_HIP(hipMemcpyAsync(st, t, (size_t)B * 8, hipMemcpyDeviceToDevice, s_));
    ofd_unet::GraphEntry* e = nullptr;
    for (auto& g : u->graphs)
        if (g.workspace == workspace && g.B == B && g.H == H && g.W == W && g.Cx == Cx && g.Cc == (cond ? Cc : 0) && g.stream == s_) e = &g;
    int rc = OFD_OK;
    if (!e) {
        // first call for this configuration: run eagerly (also performs the one-time function-attribute calls that are
        // not allowed during capture); the next call captures
        if (u->graphs.size() >= 8) drop_graphs(u);
        u->graphs.push_back({workspace, B, H, W, Cx, cond ? Cc : 0, s_, 0, nullptr, nullptr});
        rc = run_forward(c, sx, Cx, cond ? sc : nullptr, Cc, st, sout, H, W, temb, temb_silu);
    } else {
        if (e->state == 0) {
            if (!u->cap_stream) OFD_HIP(hipStreamCreateWithFlags(&u->cap_stream, hipStreamNonBlocking));
            OFD_HIP(hipStreamBeginCapture(u->cap_stream, hipStreamCaptureModeThreadLocal));
            c.s = u->cap_stream;                         // every launch of the forward lands in the capture
            rc = run_forward(c, sx, Cx, cond ? sc : nullptr, Cc, st, sout, H, W, temb, temb_silu);
            hipGraph_t g = nullptr;
            const hipError_t er = hipStreamEndCapture(u->cap_stream, &g);
            if (rc != OFD_OK || er != hipSuccess || !g) {
                if (g) (void)hipGraphDestroy(g);
                if (rc == OFD_OK) { set_error("unet_forward: stream capture failed: %s", hipGetErrorString(er)); rc = OFD_ERR_HIP; }
                return rc;
            }
            hipGraphExec_t ex = nullptr;
            OFD_HIP(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
            e->graph = g; e->exec = ex; e->state = 2;
        }
        OFD_HIP(hipGraphLaunch(e->exec, s_));
    }
    if (rc != OFD_OK) return rc;
    OFD_HIP(hipMemcpyAsync(out, sout, (size_t)B * u->cfg.out_dim * plane, hipMemcpyDeviceToDevice, s_));
    return OFD_OK;
}

extern "C" int ofd_unet_set_split_streams(ofd_unet* u, int enabled, int offset_blocks) {
    OFD_CHECK_ARG(u, "unet_set_split_streams: null handle");
    u->split_streams = enabled < 0 ? -1 : (enabled != 0);        // (< 0: back to the size-based default)
    if (offset_blocks >= 0) u->split_offset = offset_blocks;
    return OFD_OK;
}

extern "C" int ofd_unet_set_debug_taps(ofd_unet* u, int enabled) {
    OFD_CHECK_ARG(u, "unet_set_debug_taps: null handle");
    u->debug_taps = enabled != 0;
    return OFD_OK;
}

extern "C" int ofd_unet_read_tap(ofd_unet* u, const char* name, float* dst, size_t numel, void* stream) {
    OFD_CHECK_ARG(u && name && dst, "unet_read_tap: null argument");
    auto it = u->taps.find(name);
    OFD_CHECK_ARG(it != u->taps.end(), "unet_read_tap: no tap named '%s' in the last forward%s", name,
                  std::string(name) == "final_res_block" ? " (that tensor is only materialised after ofd_unet_set_debug_taps(u, 1): the final 1x1 conv normally rides on its producer)" : "");
    const Tensor& t = it->second;
    const int halves = u->last_split ? 2 : 1;            // split forward: last_B is the half batch, u->taps belongs to half 1
    OFD_CHECK_ARG(numel == (size_t)halves * u->last_B * t.C * t.H * t.W, "unet_read_tap: %s has %zu elements, got %zu", name,
                  (size_t)halves * u->last_B * t.C * t.H * t.W, numel);
    if (u->last_split) {
        const Tensor& t0 = u->taps_half0.at(name);
        int rc = k_nhwc_to_nchw(t0.p, dst, u->last_B, t.H, t.W, t.C, (hipStream_t)stream);
        if (rc != OFD_OK) return rc;
        dst += (size_t)u->last_B * t.C * t.H * t.W;
    }
    return k_nhwc_to_nchw(t.p, dst, u->last_B, t.H, t.W, t.C, (hipStream_t)stream);
}

extern "C" int ofd_unet_set_profiling(ofd_unet* u, int enabled) {
    OFD_CHECK_ARG(u, "unet_set_profiling: null handle");
    u->profiling = enabled != 0;
    return OFD_OK;
}
extern "C" int ofd_unet_prof_dump_path(ofd_unet* u, const char* path) {
    OFD_CHECK_ARG(u, "unet_prof_dump_path: null handle");
    u->dump_path = path ? path : "";
    return OFD_OK;
}
namespace ofd {
// name of a profile class = the kernel that serves it under the switches this process runs with (the same environment variables, read the same
// way, as the dispatch in conv_igemm.hip:conv_forward_impl and conv_wp.hip:launch_conv3x3_wp), followed by the layer group in brackets
const char* prof_class_name(int cls) {
    static std::string names[PC_COUNT];
    static bool built = false;
    if (!built) {
        auto env = [](const char* n, int dflt) { const char* e = getenv(n); return e ? atoi(e) : dflt; };
        const int wp = env("OFD_CONV_WP", 7), wp16 = env("OFD_CONV_WP16", 1), bn256 = env("OFD_CONV_WP_BN256", 0);
        const bool phase_wp = env("OFD_PHASE_WP", 1) != 0;
        const bool pcw = env("OFD_CONV_PCW", 0) != 0;
        names[PC_CONV3] = std::string(!(wp & 1) ? "conv_igemm_kernel<3,128>" : (bn256 ? "conv3x3_wp_kernel<8,1>|<4,1>" : (pcw ? "conv3x3_pcw_kernel" : (wp16 ? "conv3x3_wp16_kernel" : "conv3x3_wp_kernel<4,1>")))) +
                          " [3x3, Cout a multiple of 128]";
        const bool pc = env("OFD_CONV_PC", 1) != 0;
        names[PC_CONV3_64] = std::string((wp & 2) ? (pc ? "conv3x3_pc_kernel" : "conv3x3_wp_kernel<2,2>") : "conv_igemm_kernel<3,64>") + " [3x3, Cin > 64 -> 64]";
        names[PC_CONV3_PP] = std::string((wp & 4) ? (pc ? "conv3x3_pc_kernel" : "conv3x3_wp_kernel<2,2>") : "conv3x3_c64_pingpong_kernel") + " [3x3, 64 -> 64]";
        names[PC_CONV1] = "conv1x1_wp_kernel | conv_igemm_kernel<1,BN> [1x1]";
        names[PC_CONV7] = std::string(env("OFD_CONV7_PERSIST", 1) ? "conv7x7_c8_persist_kernel" : "conv_igemm_kernel<8,64>") + " [7x7 init conv]";
        names[PC_GN] = "gn_finalize";
        names[PC_RESOUT] = "resblock_out";
        names[PC_LN] = "layernorm_c";
        names[PC_LINATTN] = "la_ctx_fused + la_out_fused | lc_* [LinearAttention]";
        names[PC_FLASH] = "flash_attn_d32 [mid attention]";
        names[PC_MISC] = "misc";
        names[PC_WGRAD3] = "conv_wgrad3 [3x3 weight gradients]";
        names[PC_WGRAD1] = "conv_wgrad1 [1x1 / 7x7 weight gradients]";
        names[PC_DGRAD3] = "conv3x3 data gradients [forward kernels on dY]";
        names[PC_DGRAD1] = "conv1x1 data gradients";
        names[PC_GNBWD] = "gn_silu_backward";
        names[PC_LABWD] = "linear_attention_backward";
        names[PC_FLASHBWD] = "flash_attention_backward";
        names[PC_CONVUP] = std::string(phase_wp ? "conv_up2_phases_wp_kernel" : "conv_igemm_kernel<2,BN>") + " [Upsample x2 + 3x3 as four 2x2 phase convs]";
        built = true;
    }
    return (cls >= 0 && cls < PC_COUNT) ? names[cls].c_str() : "";
}
}  // namespace ofd

extern "C" int ofd_unet_prof_count(const ofd_unet* u) { return u ? PC_COUNT : 0; }
extern "C" const char* ofd_unet_prof_name(const ofd_unet* u, int i) { return (u && i >= 0 && i < PC_COUNT) ? prof_class_name(i) : ""; }

static int prof_resolve(ofd_unet* u) {
    FILE* dump = u->dump_path.empty() ? nullptr : fopen(u->dump_path.c_str(), "a");
    for (auto& r : u->recs) {
        OFD_HIP(hipEventSynchronize(r.e1));
        float ms = 0.0f;
        OFD_HIP(hipEventElapsedTime(&ms, r.e0, r.e1));
        if (dump) fprintf(dump, "%s,%s,%.4f,%.6g,%.6g\n", prof_class_name(r.cls), r.label.c_str(), ms, r.flops, r.bytes);
        u->acc_ms[r.cls] += ms;
        u->acc_flops[r.cls] += r.flops;
        u->acc_bytes[r.cls] += r.bytes;
        u->acc_launch[r.cls] += 1;
    }
    if (dump) fclose(dump);
    u->recs.clear();
    u->pool_used = 0;
    return OFD_OK;
}

extern "C" int ofd_unet_prof_read(ofd_unet* u, int i, double* ms, long long* launches, double* flops, double* bytes) {
    OFD_CHECK_ARG(u && i >= 0 && i < PC_COUNT, "unet_prof_read: bad argument");
    int rc = prof_resolve(u);
    if (rc != OFD_OK) return rc;
    if (ms) *ms = u->acc_ms[i];
    if (launches) *launches = u->acc_launch[i];
    if (flops) *flops = u->acc_flops[i];
    if (bytes) *bytes = u->acc_bytes[i];
    return OFD_OK;
}

extern "C" int ofd_unet_prof_reset(ofd_unet* u) {
    OFD_CHECK_ARG(u, "unet_prof_reset: null handle");
    int rc = prof_resolve(u);
    for (int i = 0; i < PC_COUNT; ++i) { u->acc_ms[i] = 0; u->acc_flops[i] = 0; u->acc_bytes[i] = 0; u->acc_launch[i] = 0; }
    return rc;
}
