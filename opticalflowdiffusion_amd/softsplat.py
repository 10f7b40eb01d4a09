"""Forward splatting (mirror of the reference's softsplat_new.py interface) on the HIP kernels.

``softsplat(tenIn, tenFlow, tenMetric, strMode, scale, offset)`` keeps the reference signature
and modes (softsplat_new.py:278-333); the three CUDA kernels it JIT-compiled through CuPy are
replaced by ``ofd_splat_fwd`` / ``ofd_splat_bwd_in`` / ``ofd_splat_bwd_flow``.
"""
import torch

from . import _lib as L

L_mod = L
_lib_f32 = L.f32c

# scan radius (source pixels) of an output tile; FlowDiffuser clamps flow to +-flow_max = 20 px
# (flow_diffuser.py:141). Larger displacements stay correct through the far-corner list.
DEFAULT_RADIUS = 24

_ws_cache = {}


def _workspace(device, nbytes):
    key = (device.index,)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


def splat_forward(ten_in, ten_flow, scale=1, offset_x=0, offset_y=0, radius=DEFAULT_RADIUS):
    """softsplat_func.forward (softsplat_new.py:339-454) without autograd."""
    L.require_gpu(ten_in, ten_flow)
    ten_in, ten_flow = L.f32c(ten_in), L.f32c(ten_flow)
    B, C, H, W = ten_in.shape
    if ten_flow.shape != (B, 2, H, W):
        raise L.OfdError(f"flow must be (B,2,H,W) = {(B, 2, H, W)}, got {tuple(ten_flow.shape)}")
    out = torch.empty(B, C, H // scale, W // scale, dtype=torch.float32, device=ten_in.device)
    nbytes = L.lib().ofd_splat_workspace_bytes(B, H, W)
    ws = _workspace(ten_in.device, nbytes)
    L.check(L.lib().ofd_splat_fwd(L.ptr(ten_in), L.ptr(ten_flow), L.ptr(out), B, C, H, W, scale, offset_x, offset_y,
                                  radius, L.ptr(ws), ws.numel(), L.stream()))
    return out


def splat_corners(ten_flow, scale=1, offset_x=0, offset_y=0):
    """int32 (B,H,W,2) north-west corner (x0,y0) of softsplat_out per source pixel (parity tests)."""
    L.require_gpu(ten_flow)
    ten_flow = L.f32c(ten_flow)
    B, _, H, W = ten_flow.shape
    out = torch.empty(B, H, W, 2, dtype=torch.int32, device=ten_flow.device)
    L.check(L.lib().ofd_splat_corners(L.ptr(ten_flow), L.ptr(out), B, H, W, scale, offset_x, offset_y, L.stream()))
    return out


class softsplat_func(torch.autograd.Function):
    """softsplat_new.py:339-730.  Inputs are cast to fp32 as the reference's custom_fwd does."""

    @staticmethod
    def forward(ctx, tenIn, tenFlow, scale, offset_x, offset_y):
        tenIn, tenFlow = L.f32c(tenIn), L.f32c(tenFlow)
        out = splat_forward(tenIn, tenFlow, scale, offset_x, offset_y)
        ctx.save_for_backward(tenIn, tenFlow)
        ctx.geom = (scale, offset_x, offset_y)
        return out

    @staticmethod
    def backward(ctx, tenOutgrad):
        tenIn, tenFlow = ctx.saved_tensors
        scale, ox, oy = ctx.geom
        g = L.f32c(tenOutgrad)
        B, C, H, W = tenIn.shape
        ingrad = flowgrad = None
        if ctx.needs_input_grad[0]:
            ingrad = torch.empty_like(tenIn)
            L.check(L.lib().ofd_splat_bwd_in(L.ptr(tenFlow), L.ptr(g), L.ptr(ingrad), B, C, H, W, scale, ox, oy, L.stream()))
        if ctx.needs_input_grad[1]:
            flowgrad = torch.empty_like(tenFlow)
            L.check(L.lib().ofd_splat_bwd_flow(L.ptr(tenIn), L.ptr(tenFlow), L.ptr(g), L.ptr(flowgrad), B, C, H, W,
                                               scale, ox, oy, L.stream()))
        return ingrad, flowgrad, None, None, None


def softsplat(tenIn, tenFlow, tenMetric, strMode, scale=1, offset=(0, 0)):
    """softsplat_new.py:278-333 (same modes, same assertions)."""
    base = strMode.split("-")[0]
    assert base in ["sum", "avg", "linear", "soft", "linear_unn"]
    if strMode in ("sum", "avg"):
        assert tenMetric is None
    if base in ("linear", "linear_unn", "soft"):
        assert tenMetric is not None

    if strMode == "avg":
        tenIn = torch.cat([tenIn, tenIn.new_ones([tenIn.shape[0], 1, tenIn.shape[2], tenIn.shape[3]])], 1)
    elif base in ("linear", "linear_unn"):
        tenIn = torch.cat([tenIn * tenMetric, tenMetric], 1)
    elif base == "soft":
        tenIn = torch.cat([tenIn * tenMetric.exp(), tenMetric.exp()], 1)

    tenOut = softsplat_func.apply(tenIn, tenFlow, scale, offset[0], offset[1])

    if base in ["avg", "linear", "soft"]:
        tenNormalize = tenOut[:, -1:, :, :]
        parts = strMode.split("-")
        if len(parts) == 1 or parts[1] == "addeps":
            tenNormalize = tenNormalize + 0.0000001
        elif parts[1] == "zeroeps":
            tenNormalize = tenNormalize.clone()
            tenNormalize[tenNormalize == 0.0] = 1.0
        elif parts[1] == "clipeps":
            tenNormalize = tenNormalize.clip(0.0000001, None)
        return torch.cat((tenOut[:, :-1, :, :] / tenNormalize, tenOut[:, -1, None, :, :]), dim=1)
    return tenOut


# ---- all L*L offsets of a scale-L splat in one call (the FlowLearner pyramid, flow_learner.py:159-206) -------------------------
class _PyramidSplat(torch.autograd.Function):
    """T[n, c, L*cy + b, L*cx + a] = softsplat_func(tenIn, tenFlow, L, a, b)[n, c, cy, cx] for every offset (a, b)."""

    @staticmethod
    def forward(ctx, tenIn, tenFlow, L):
        L_ = int(L)
        tenIn, tenFlow = _lib_f32(tenIn), _lib_f32(tenFlow)
        B, C, H, W = tenIn.shape
        out = torch.empty(B, C, L_ * (H // L_), L_ * (W // L_), dtype=torch.float32, device=tenIn.device)
        lib = L_mod.lib()
        ws = _workspace(tenIn.device, lib.ofd_splat_pyramid_workspace_bytes(B, C, H, W))
        L_mod.check(lib.ofd_splat_pyramid_fwd(L_mod.ptr(tenIn), L_mod.ptr(tenFlow), L_mod.ptr(out), B, C, H, W, L_, DEFAULT_RADIUS,
                                              L_mod.ptr(ws), ws.numel(), L_mod.stream()))
        ctx.save_for_backward(tenIn, tenFlow)
        ctx.level = L_
        return out

    @staticmethod
    def backward(ctx, dT):
        tenIn, tenFlow = ctx.saved_tensors
        B, C, H, W = tenIn.shape
        dT = _lib_f32(dT)
        need_in, need_flow = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        ingrad = torch.empty_like(tenIn) if need_in else None
        flowgrad = torch.empty_like(tenFlow) if need_flow else None
        if need_in or need_flow:
            lib = L_mod.lib()
            ws = _workspace(tenIn.device, lib.ofd_splat_pyramid_workspace_bytes(B, C, H, W))
            L_mod.check(lib.ofd_splat_pyramid_bwd(L_mod.ptr(tenIn), L_mod.ptr(tenFlow), L_mod.ptr(dT), L_mod.ptr(ingrad), L_mod.ptr(flowgrad),
                                                  B, C, H, W, ctx.level, L_mod.ptr(ws), ws.numel(), L_mod.stream()))
        return ingrad, flowgrad, None


def splat_pyramid(tenIn, tenFlow, level):
    """(B, C, L*(H//L), L*(W//L)): every offset of the scale-`level` splat, interleaved (see `pyramid_offsets`)."""
    L_mod.require_gpu(tenIn, tenFlow)
    return _PyramidSplat.apply(tenIn, tenFlow, level)


def pyramid_offsets(T, level):
    """view of splat_pyramid's result as (a, b, B, C, H//L, W//L): [a, b] is the splat with offset=[a, b]."""
    B, C, Ht, Wt = T.shape
    return T.view(B, C, Ht // level, level, Wt // level, level).permute(5, 3, 0, 1, 2, 4)


def softsplat_pyramid(tenIn, tenFlow, tenMetric, strMode, scale):
    """`softsplat(tenIn, tenFlow, tenMetric, strMode, scale, offset=[a, b])` for ALL offsets at once, interleaved as
    splat_pyramid does (softsplat_new.py:278-333: same modes and normalisation)."""
    base = strMode.split("-")[0]
    assert base in ["sum", "avg", "linear", "soft", "linear_unn"]
    if strMode in ("sum", "avg"):
        assert tenMetric is None
    if base in ("linear", "linear_unn", "soft"):
        assert tenMetric is not None
    if strMode == "avg":
        tenIn = torch.cat([tenIn, tenIn.new_ones([tenIn.shape[0], 1, tenIn.shape[2], tenIn.shape[3]])], 1)
    elif base in ("linear", "linear_unn"):
        tenIn = torch.cat([tenIn * tenMetric, tenMetric], 1)
    elif base == "soft":
        tenIn = torch.cat([tenIn * tenMetric.exp(), tenMetric.exp()], 1)
    tenOut = splat_pyramid(tenIn, tenFlow, scale)
    if base in ["avg", "linear", "soft"]:
        tenNormalize = tenOut[:, -1:, :, :]
        parts = strMode.split("-")
        if len(parts) == 1 or parts[1] == "addeps":
            tenNormalize = tenNormalize + 0.0000001
        elif parts[1] == "zeroeps":
            tenNormalize = tenNormalize.clone()
            tenNormalize[tenNormalize == 0.0] = 1.0
        elif parts[1] == "clipeps":
            tenNormalize = tenNormalize.clip(0.0000001, None)
        return torch.cat((tenOut[:, :-1, :, :] / tenNormalize, tenOut[:, -1, None, :, :]), dim=1)
    return tenOut


class _PyramidCharbonnier(torch.autograd.Function):
    """mean over the L*L offsets of nan_charbonnier(target slice, filled input slice) on two un-normalised "soft" pyramid splats
    (B, C+1, Ht, Wt): one reduction kernel forward, one elementwise kernel backward (gradient w.r.t. the input pyramid only)."""

    @staticmethod
    def forward(ctx, Tin, Ttg, L):
        L_ = int(L)
        Tin, Ttg = _lib_f32(Tin), _lib_f32(Ttg)
        B, C1, Ht, Wt = Tin.shape
        acc = torch.empty(2, L_, L_, dtype=torch.float64, device=Tin.device)
        L_mod.check(L_mod.lib().ofd_pyramid_charbonnier_fwd(L_mod.ptr(Tin), L_mod.ptr(Ttg), L_mod.ptr(acc[0]), L_mod.ptr(acc[1]), B, C1 - 1, Ht, Wt,
                                                            L_, L_mod.stream()))
        ctx.save_for_backward(Tin, Ttg, acc)
        ctx.level = L_
        return (acc[0] / acc[1]).mean().to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        Tin, Ttg, acc = ctx.saved_tensors
        B, C1, Ht, Wt = Tin.shape
        dT = torch.empty_like(Tin)
        gs = g.reshape(1).to(torch.float32).contiguous()
        L_mod.check(L_mod.lib().ofd_pyramid_charbonnier_bwd(L_mod.ptr(Tin), L_mod.ptr(Ttg), L_mod.ptr(acc[1]), L_mod.ptr(gs), L_mod.ptr(dT), B, C1 - 1,
                                                            Ht, Wt, ctx.level, L_mod.stream()))
        return dT, None, None


def pyramid_charbonnier(Tin, Ttg, level):
    """level loss of flow_learner.py:176-191 from the raw (un-normalised) soft pyramid splats of input and target"""
    L_mod.require_gpu(Tin, Ttg)
    return _PyramidCharbonnier.apply(Tin, Ttg, level)
