"""Flow visualisation used by the validation diagnostics (flow_diffuser.py:283-364 calls `torchvision.utils.flow_to_image`).

torchvision is not a dependency of this package; the Middlebury colour wheel (Baker et al., "A Database and Evaluation
Methodology for Optical Flow", ICCV 2007: 55 hues in six segments RY 15, YG 6, GC 4, CB 11, BM 13, MR 6) is small enough to
restate.  Semantics follow the call the reference makes: flow (N, 2, H, W) float -> uint8 image (N, 3, H, W), flows normalised by
the largest magnitude of the batch, saturation growing with the magnitude.
"""
import math

import torch


def _color_wheel(device):
    segs = (15, 6, 4, 11, 13, 6)                                   # RY, YG, GC, CB, BM, MR
    wheel = torch.zeros(sum(segs), 3, device=device)
    col = 0
    ramps = [(0, 1, +1), (1, 0, -1), (1, 2, +1), (2, 1, -1), (2, 0, +1), (0, 2, -1)]   # (full channel, ramping channel, direction)
    for n, (full, ramp, direction) in zip(segs, ramps):
        r = torch.floor(255.0 * torch.arange(n, device=device) / n)
        wheel[col:col + n, full] = 255.0
        wheel[col:col + n, ramp] = r if direction > 0 else 255.0 - r
        col += n
    return wheel


def flow_to_image(flow):
    """(N, 2, H, W) or (2, H, W) float flow -> uint8 RGB image of the same spatial size."""
    single = flow.dim() == 3
    if single:
        flow = flow[None]
    if flow.dim() != 4 or flow.shape[1] != 2:
        raise ValueError(f"flow must be (N, 2, H, W) or (2, H, W), got {tuple(flow.shape)}")
    flow = flow.float()
    norm = torch.sqrt(torch.sum(flow ** 2, dim=1))
    eps = torch.finfo(flow.dtype).eps
    flow = flow / (norm.max() + eps)
    u, v = flow[:, 0], flow[:, 1]
    rad = torch.sqrt(u ** 2 + v ** 2)
    wheel = _color_wheel(flow.device)
    ncols = wheel.shape[0]
    a = torch.atan2(-v, -u) / math.pi
    fk = (a + 1) / 2 * (ncols - 1)
    k0 = torch.floor(fk).long()
    k1 = torch.where(k0 + 1 == ncols, torch.zeros_like(k0), k0 + 1)
    f = fk - k0
    img = torch.zeros(flow.shape[0], 3, *flow.shape[2:], dtype=torch.uint8, device=flow.device)
    for c in range(3):
        col0, col1 = wheel[k0, c] / 255.0, wheel[k1, c] / 255.0
        col = (1 - f) * col0 + f * col1
        col = 1 - rad * (1 - col)
        img[:, c] = torch.floor(255.0 * col).to(torch.uint8)
    return img[0] if single else img
