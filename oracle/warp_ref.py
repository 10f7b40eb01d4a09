"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the warp layer (WP = warp.py, SS = softsplat_new.py).

The three splat kernels live in ``splat_ref.c`` (scalar C, one loop iteration per reference
CUDA thread); this file restates the Python wrappers around them and the ``grid_sample``
backward warp, which calls the very torch CPU op the reference calls.
"""
import ctypes
import os
import subprocess

import numpy as np
import torch
import torch.nn.functional as F

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "_build", "libofd_oracle.so")
        if not os.path.exists(path):
            build()
        _LIB = ctypes.CDLL(path)
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _f32(t):
    return np.ascontiguousarray(t.detach().cpu().numpy().astype(np.float32, copy=False))


def splat_out(ten_in, ten_flow, scale=1, ox=0, oy=0, return_corners=False):
    """SS:339-454 forward: zero-filled (B,C,H//s,W//s) output, then softsplat_out."""
    a, f = _f32(ten_in), _f32(ten_flow)
    B, C, H, W = a.shape
    assert f.shape == (B, 2, H, W)
    out = np.zeros((B, C, H // scale, W // scale), np.float32)
    corners = np.zeros((B, H, W, 2), np.int32) if return_corners else None
    _lib().ofd_ref_splat_out(_p(a), _p(f), _p(out), _p(corners) if return_corners else None,
                             B, C, H, W, scale, ox, oy)
    if return_corners:
        return torch.from_numpy(out), torch.from_numpy(corners)
    return torch.from_numpy(out)


def splat_ingrad(ten_flow, outgrad, in_shape, scale=1, ox=0, oy=0):
    """SS:489-565."""
    f, g = _f32(ten_flow), _f32(outgrad)
    B, C, H, W = in_shape
    ing = np.zeros((B, C, H, W), np.float32)
    _lib().ofd_ref_splat_ingrad(_p(f), _p(g), _p(ing), B, C, H, W, scale, ox, oy)
    return torch.from_numpy(ing)


def splat_flowgrad(ten_in, ten_flow, outgrad, scale=1, ox=0, oy=0):
    """SS:600-700."""
    a, f, g = _f32(ten_in), _f32(ten_flow), _f32(outgrad)
    B, C, H, W = a.shape
    fg = np.zeros((B, 2, H, W), np.float32)
    _lib().ofd_ref_splat_flowgrad(_p(a), _p(f), _p(g), _p(fg), B, C, H, W, scale, ox, oy)
    return torch.from_numpy(fg)


def softsplat(ten_in, ten_flow, ten_metric, mode, scale=1, offset=(0, 0)):
    """SS:278-333: the Python-level modes around the kernel."""
    base = mode.split("-")[0]
    assert base in ("sum", "avg", "linear", "soft", "linear_unn")
    if base in ("sum", "avg"):
        assert ten_metric is None
    else:
        assert ten_metric is not None
    if mode == "avg":
        ten_in = torch.cat([ten_in, ten_in.new_ones(ten_in.shape[0], 1, *ten_in.shape[2:])], 1)
    elif base in ("linear", "linear_unn"):
        ten_in = torch.cat([ten_in * ten_metric, ten_metric], 1)
    elif base == "soft":
        ten_in = torch.cat([ten_in * ten_metric.exp(), ten_metric.exp()], 1)
    out = splat_out(ten_in, ten_flow, scale, offset[0], offset[1])
    if base in ("avg", "linear", "soft"):
        norm = out[:, -1:]
        parts = mode.split("-")
        if len(parts) == 1 or parts[1] == "addeps":
            norm = norm + 0.0000001
        elif parts[1] == "zeroeps":
            norm = norm.clone()
            norm[norm == 0.0] = 1.0
        elif parts[1] == "clipeps":
            norm = norm.clip(0.0000001, None)
        return torch.cat((out[:, :-1] / norm, out[:, -1:]), dim=1)
    return out


def warp_forward_flow(first, flow, scale=1, set_nans=True, get_variance=False, offset=(0, 0), warp_style="sum"):
    """WP:121-156."""
    first = first.clone()
    weights = torch.ones_like(first[:, 0])
    nans = torch.isnan(first)
    first[nans] = 0.0
    weights[torch.any(nans, dim=1)] = 0.0
    offset = [o % scale for o in offset]
    mode = "linear_unn" if warp_style == "sum" else "linear"
    ret = softsplat(first, flow, weights[:, None], mode, scale, offset)
    img = ret[:, :-1]
    w = ret[:, -1:].repeat(1, img.shape[1], 1, 1)
    if get_variance:
        var = softsplat(torch.square(first), flow, weights[:, None], "linear_unn", scale, offset)
        img = var[:, :-1] - torch.square(img)
    if set_nans:
        img = torch.where(w > 0, img, torch.full_like(img, float("nan")))
    return img


def warp_backward_flow(second, flow):
    """WP:95-119: mesh grid + flow.flip(1), normalise to [-1,1], grid_sample(align_corners=True) twice."""
    B, C, H, W = second.shape
    xx = torch.arange(0, W).view(1, 1, 1, W).expand(B, 1, H, W)
    yy = torch.arange(0, H).view(1, 1, H, 1).expand(B, 1, H, W)
    grid = torch.cat((xx, yy), 1).float()
    vgrid = grid + flow.flip(1)
    vx = 2.0 * vgrid[:, 0] / max(W - 1, 1) - 1.0
    vy = 2.0 * vgrid[:, 1] / max(H - 1, 1) - 1.0
    vgrid = torch.stack((vx, vy), dim=-1)
    output = F.grid_sample(second, vgrid, align_corners=True)
    mask = F.grid_sample(torch.ones_like(second), vgrid, align_corners=True)
    mask = mask.clone()
    mask[mask < 0.999] = 0
    mask[mask > 0] = 1
    return output, mask


def warp(first, second, flow, rep="flow", mode="backward", **kw):
    """WP:83-93 (rep='flow' only)."""
    assert rep == "flow"
    if mode == "backward":
        return warp_backward_flow(second, flow)
    return warp_forward_flow(first, flow, **kw)


def scale_down(img, down):
    """WP:234-243 (down branch): box mean over down x down patches."""
    b, c, h, w = img.shape
    p = img.reshape(b, c, h // down, down, w // down, down)
    return p.mean(dim=-1).mean(dim=-2)
