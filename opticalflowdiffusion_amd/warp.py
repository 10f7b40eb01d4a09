"""Warp layer with the reference's interface (algorithms/diffusion_animation/warp.py, rep='flow').

``warp(first, second, flow, rep='flow', mode='backward'|'forward', **kw)`` -- WP:83-93.
  forward  -> warp_forward_flow  (WP:121-156): NaN-aware forward splat through the HIP splat kernels
  backward -> warp_backward_flow (WP:95-119) : bilinear grid_sample gather + validity mask
rep='filter' (FlowLearner / MatrixFlow only) is outside the FlowDiffuser path and raises.
"""
import torch

from . import _lib as L
from .softsplat import softsplat_func, splat_forward


class _WarpForward(torch.autograd.Function):
    """warp_forward_flow with warp_style in {"sum","linear"}: prep -> splat -> holes, three HIP
    launches; gradients flow to `first` and `flow` through the reference's backward kernels."""

    @staticmethod
    def forward(ctx, first, flow, scale, ox, oy, set_nans, linear, square):
        first, flow = L.f32c(first), L.f32c(flow)
        B, C, H, W = first.shape
        lib = L.lib()
        ten_in = torch.empty(B, C + 1, H, W, dtype=torch.float32, device=first.device)
        L.check(lib.ofd_warp_prep(L.ptr(first), L.ptr(ten_in), B, C, H, W, int(square), L.stream()))
        ten_out = splat_forward(ten_in, flow, scale, ox, oy)
        Ho, Wo = ten_out.shape[-2:]
        img = torch.empty(B, C, Ho, Wo, dtype=torch.float32, device=first.device)
        L.check(lib.ofd_warp_holes(L.ptr(ten_out), L.ptr(img), B, C, Ho, Wo, int(linear), int(set_nans), L.stream()))
        ctx.save_for_backward(ten_in, flow, ten_out)
        ctx.cfg = (scale, ox, oy, set_nans, linear, square)
        return img

    @staticmethod
    def backward(ctx, g_img):
        ten_in, flow, ten_out = ctx.saved_tensors
        scale, ox, oy, set_nans, linear, square = ctx.cfg
        if square:
            raise L.OfdError("get_variance=True is not differentiable in this build")
        B, C1, H, W = ten_in.shape
        C = C1 - 1
        g_img = L.f32c(g_img)
        w = ten_out[:, -1:]
        if set_nans:                                  # torch.where(weights > 0, img, nan) (WP:154-155)
            g_img = torch.where(w > 0, g_img, torch.zeros_like(g_img))
        g_out = torch.zeros_like(ten_out)
        if linear:                                    # img = sum / (w + 1e-7)
            denom = w + 0.0000001
            g_out[:, :C] = g_img / denom
            g_out[:, C:] = -(g_img * ten_out[:, :C] / (denom * denom)).sum(1, keepdim=True)
        else:
            g_out[:, :C] = g_img
        lib = L.lib()
        g_first = g_flow = None
        if ctx.needs_input_grad[0]:
            g_in = torch.empty_like(ten_in)
            L.check(lib.ofd_splat_bwd_in(L.ptr(flow), L.ptr(g_out), L.ptr(g_in), B, C1, H, W, scale, ox, oy, L.stream()))
            # ten_in[:, :C] = nan_to_zero(first) * w_in: gradient reaches `first` where w_in == 1
            g_first = g_in[:, :C] * ten_in[:, C:]
        if ctx.needs_input_grad[1]:
            g_flow = torch.empty_like(flow)
            L.check(lib.ofd_splat_bwd_flow(L.ptr(ten_in), L.ptr(flow), L.ptr(g_out), L.ptr(g_flow), B, C1, H, W,
                                           scale, ox, oy, L.stream()))
        return g_first, g_flow, None, None, None, None, None, None


def warp_forward_flow(first, second, flow, scale=1, set_nans=True, get_variance=False, offset=[0, 0], warp_style="sum"):
    """WP:121-156."""
    L.require_gpu(first, flow)
    offset = [o % scale for o in offset]
    linear = warp_style != "sum"
    img = _WarpForward.apply(first, flow, scale, offset[0], offset[1], bool(set_nans) and not get_variance, linear, False)
    if get_variance:                                  # WP:142-152: splat(x^2) - splat(x)^2
        var = _WarpForward.apply(first, flow, scale, offset[0], offset[1], False, False, True)
        img = var - torch.square(img)
        if set_nans:                                  # holes of the NaN-aware weights (WP:154-155)
            valid = (~torch.isnan(first).any(dim=1, keepdim=True)).float()
            w = _WarpForward.apply(valid, flow, scale, offset[0], offset[1], False, False, False)
            img = torch.where(w > 0, img, torch.full_like(img, float("nan")))
    return img


class _GridWarp(torch.autograd.Function):
    """warp_backward_flow (WP:95-119) with the gradients autograd derives for it: d/d second is the bilinear scatter of the
    incoming gradient (the splat kernel on grid_sample's coordinates), d/d flow ATen's grid gradient; the thresholded mask
    (WP:116-117 overwrites every element) carries none."""

    @staticmethod
    def forward(ctx, second, flow):
        second, flow = L.f32c(second), L.f32c(flow)
        B, C, H, W = second.shape
        out = torch.empty_like(second)
        mask = torch.empty_like(second)
        L.check(L.lib().ofd_grid_warp_fwd(L.ptr(second), L.ptr(flow), L.ptr(out), L.ptr(mask), B, C, H, W, L.stream()))
        ctx.save_for_backward(second, flow)
        ctx.mark_non_differentiable(mask)
        return out, mask

    @staticmethod
    def backward(ctx, g_out, _g_mask):
        from .softsplat import DEFAULT_RADIUS, _workspace
        second, flow = ctx.saved_tensors
        B, C, H, W = second.shape
        g_out = L.f32c(g_out)
        need_second, need_flow = ctx.needs_input_grad
        g_second = torch.empty_like(second) if need_second else None
        g_flow = torch.empty_like(flow) if need_flow else None
        if need_second or need_flow:
            lib = L.lib()
            ws = _workspace(second.device, lib.ofd_splat_workspace_bytes(B, H, W)) if need_second else None
            L.check(lib.ofd_grid_warp_bwd(L.ptr(second), L.ptr(flow), L.ptr(g_out), L.ptr(g_second) if need_second else None,
                                          L.ptr(g_flow) if need_flow else None, B, C, H, W, DEFAULT_RADIUS,
                                          L.ptr(ws) if need_second else None, ws.numel() if need_second else 0, L.stream()))
        return g_second, g_flow


def warp_backward_flow(first, second, flow):
    """WP:95-119: returns (output, mask)."""
    L.require_gpu(second, flow)
    if flow.shape != (second.shape[0], 2, second.shape[2], second.shape[3]):
        raise L.OfdError(f"flow must be (B,2,H,W) for second {tuple(second.shape)}, got {tuple(flow.shape)}")
    return _GridWarp.apply(second, flow)


def grid_warp_corners(flow):
    L.require_gpu(flow)
    flow = L.f32c(flow)
    B, _, H, W = flow.shape
    out = torch.empty(B, H, W, 2, dtype=torch.int32, device=flow.device)
    L.check(L.lib().ofd_grid_warp_corners(L.ptr(flow), L.ptr(out), B, H, W, L.stream()))
    return out


def warp(first, second, flow, rep="flow", mode="backward", **kwargs):
    """WP:83-93."""
    if rep != "flow":
        raise NotImplementedError("rep='filter' belongs to FlowLearner/MatrixFlow, outside the FlowDiffuser path")
    if mode == "backward":
        return warp_backward_flow(first, second, flow, **kwargs)
    elif mode == "forward":
        return warp_forward_flow(first, second, flow, **kwargs)
    raise ValueError(f"unknown warp mode {mode!r}")


class _NanMseMean(torch.autograd.Function):
    """nanmean of the masked squared error and its gradient w.r.t. the prediction (the training loss,
    DD:908,973)."""

    @staticmethod
    def forward(ctx, pred, target):
        p, t = L.f32c(pred).reshape(-1), L.f32c(target).reshape(-1)
        res = torch.empty(L.lib().ofd_nan_mse_result_doubles(), dtype=torch.float64, device=p.device)     # [0] sum, [1] count, scratch
        L.check(L.lib().ofd_nan_mse_sum(L.ptr(p), L.ptr(t), p.numel(), L.ptr(res), L.stream()))
        ctx.save_for_backward(p, t, res)
        ctx.shape = pred.shape
        return (res[0] / res[1]).float()

    @staticmethod
    def backward(ctx, gout):
        p, t, res = ctx.saved_tensors
        g = L.f32c(gout).reshape(1)
        dp = torch.empty_like(p)
        L.check(L.lib().ofd_nan_mse_grad(L.ptr(p), L.ptr(t), p.numel(), L.ptr(res), L.ptr(g), L.ptr(dp), L.stream()))
        return dp.view(ctx.shape), None


class _NanSqSum(torch.autograd.Function):
    """(sum, count) of the squared error over the entries where neither side is NaN; the sum is differentiable
    w.r.t. the prediction.  Building block of the pyramid loss (DD:902-973), whose `nanmean` over the
    concatenation of all levels is sum_L L^4 S_L / sum_L N_L."""

    @staticmethod
    def forward(ctx, pred, target):
        p, t = L.f32c(pred).reshape(-1), L.f32c(target).reshape(-1)
        res = torch.empty(L.lib().ofd_nan_mse_result_doubles(), dtype=torch.float64, device=p.device)     # [0] sum, [1] count, scratch
        L.check(L.lib().ofd_nan_mse_sum(L.ptr(p), L.ptr(t), p.numel(), L.ptr(res), L.stream()))
        ctx.save_for_backward(p, t)
        ctx.shape = pred.shape
        ctx.mark_non_differentiable(res)
        return res[0].float(), res

    @staticmethod
    def backward(ctx, gsum, _gres):
        p, t = ctx.saved_tensors
        g = L.f32c(gsum).reshape(1)
        unit = torch.tensor([0.0, 1.0], dtype=torch.float64, device=p.device)     # count 1: the kernel divides by it
        dp = torch.empty_like(p)
        L.check(L.lib().ofd_nan_mse_grad(L.ptr(p), L.ptr(t), p.numel(), L.ptr(unit), L.ptr(g), L.ptr(dp), L.stream()))
        return dp.view(ctx.shape), None


def nan_sq_sum(pred, target):
    """returns (sum of squared errors over non-NaN pairs [differentiable], count [fp64 0-dim])."""
    L.require_gpu(pred, target)
    s, res = _NanSqSum.apply(pred, target)
    return s, res[1]


def nan_mse(pred, target, reduction="mean"):
    """WP:260-271.  reduction='mean' runs the fused HIP reduction (differentiable w.r.t. pred);
    'none' returns the compacted squared errors (dynamic shape, as the reference)."""
    if reduction == "mean" and not target.requires_grad:
        L.require_gpu(pred, target)
        return _NanMseMean.apply(pred, target)
    pred, target = pred.flatten(), target.flatten()
    ok = torch.logical_not(torch.logical_or(torch.isnan(target), torch.isnan(pred)))
    sq = torch.square(pred[ok] - target[ok])
    return torch.nanmean(sq) if reduction == "mean" else sq


def scale(img, up=None, down=None):
    """WP:234-243."""
    if up is not None and down is not None:
        raise ValueError("one of up or down")
    if up is not None:
        return torch.nn.functional.interpolate(img, scale_factor=up, mode="bilinear")
    if down is not None:
        b, c, h, w = img.shape
        p = img.reshape(b, c, h // down, down, w // down, down)
        return torch.mean(torch.mean(p, dim=-1), dim=-2)
    return img


# ---- photometric-loss helpers of the FlowLearner path (WP:273-303) ---------------------------------------------------
def fill_holes_nan(img, weights):
    """WP:273-276: NaN wherever nothing was splatted (weight <= 0)."""
    return torch.where(weights > 0, img, torch.full_like(img, float("nan")))


def charbonnier(x, alpha=0.5, eps=1e-3):
    """WP:278-279."""
    return torch.pow(torch.square(x) + eps ** 2, alpha)


def nan_charbonnier(pred, target):
    """WP:281-287: mean Charbonnier penalty over the positions where neither side is NaN."""
    pred, target = pred.flatten(), target.flatten()
    ok = torch.logical_not(torch.logical_or(torch.isnan(target), torch.isnan(pred)))
    return torch.mean(charbonnier(pred[ok] - target[ok]))


def edgeaware_smoothness1(image, flow, edge_weight=30):
    """WP:289-303: first-order flow smoothness, down-weighted across image edges."""
    image_grad_y = image[:, :, 1:, :] - image[:, :, :-1, :]
    image_grad_x = image[:, :, :, 1:] - image[:, :, :, :-1]
    flow_grad_y = flow[:, :, 1:, :] - flow[:, :, :-1, :]
    flow_grad_x = flow[:, :, :, 1:] - flow[:, :, :, :-1]
    y_weights = torch.exp(-edge_weight * torch.mean(image_grad_y ** 2, dim=1, keepdim=True))
    x_weights = torch.exp(-edge_weight * torch.mean(image_grad_x ** 2, dim=1, keepdim=True))
    loss = torch.mean(x_weights * charbonnier(flow_grad_x)) + torch.mean(y_weights * charbonnier(flow_grad_y))
    return loss / 2
