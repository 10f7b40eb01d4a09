"""Oracle of the warp layer: grid_sample warp pinned to reference goldens; splat checked
against warp_test.py's known answer / properties and analytic cases (PARITY UNPINNED, see
oracle/__init__.py)."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_l2
from oracle import diffusion_ref as D
from oracle import warp_ref as W


def test_backward_warp_matches_reference():
    g = load_golden("warp_backward")
    for tag, img_k in (("rand", "rand.img"), ("int", "rand.img"), ("wideint", "wide.img")):
        out, mask = W.warp(None, g[img_k], g[f"{tag}.flow"], mode="backward")
        assert torch.equal(out, g[f"{tag}.out"]), tag
        assert torch.equal(mask, g[f"{tag}.mask"]), tag
    out, mask = W.warp(None, g["wide.img"], torch.zeros(1, 2, 6, 1024), mode="backward")
    assert torch.equal(out, g["wide.out"]) and torch.equal(mask, g["wide.mask"])


def test_nan_mse_and_scale():
    g = load_golden("warp_backward")
    assert torch.equal(D.nan_mse_none(g["nan_mse.a"], g["nan_mse.b"]), g["nan_mse.none"])
    assert float(torch.nanmean(D.nan_mse_none(g["nan_mse.a"], g["nan_mse.b"]))) == pytest.approx(float(g["nan_mse.mean"]))
    assert torch.equal(W.scale_down(g["scale.x"], 2), g["scale.down2"])


# ---- splat: known answers -------------------------------------------------------------------
def test_splat_known_answer_from_warp_test():
    """warp_test.py:22-27 (commented case): a single 1.0 at (row 1, col 2) with x-flow 0.5
    splits 0.5/0.5 across cols 2 and 3."""
    src = torch.zeros(1, 1, 2, 4)
    src[0, 0, 1, 2] = 1.0
    flow = torch.zeros(1, 2, 2, 4)
    flow[0, 0] = 0.5
    out = W.splat_out(src, flow)
    exp = torch.zeros(1, 1, 2, 4)
    exp[0, 0, 1, 2] = 0.5
    exp[0, 0, 1, 3] = 0.5
    assert torch.equal(out, exp)


def test_splat_identity_shift_and_holes():
    torch.manual_seed(0)
    img = torch.rand(2, 3, 6, 9)
    z = torch.zeros(2, 2, 6, 9)
    assert torch.equal(W.warp(img, None, z, mode="forward"), img)           # zero flow == identity
    f = z.clone()
    f[:, 0] = 2.0                                                           # ch0 = x displacement (SS:368)
    f[:, 1] = -1.0
    out = W.warp(img, None, f, mode="forward")
    assert torch.equal(out[:, :, :5, 2:], img[:, :, 1:, :7])
    assert torch.isnan(out[:, :, 5, :]).all() and torch.isnan(out[:, :, :, :2]).all()   # holes -> NaN (WP:154)
    img2 = img.clone()
    img2[0, 1, 2, 3] = float("nan")                                         # NaN input -> weight 0 (WP:124-126)
    out = W.warp(img2, None, z, mode="forward")
    assert torch.isnan(out[0, :, 2, 3]).all() and torch.equal(out[1], img[1])
    f = z.clone()
    f[0, 0, 1, 1] = float("inf")                                            # non-finite flow skipped (SS:371)
    out = W.splat_out(img, f)
    assert (out[0, :, 1, 1] == 0).all()


def test_splat_scale_zero_flow_is_box_sum():
    torch.manual_seed(1)
    img = torch.rand(1, 2, 8, 12)
    z = torch.zeros(1, 2, 8, 12)
    for L in (2, 4):
        out = W.splat_out(img, z, scale=L)
        # a source pixel at x lands at x/L: corners floor(x/L) and +1 with bilinear weights
        exp = torch.zeros(1, 2, 8 // L, 12 // L)
        for y in range(8):
            for x in range(12):
                fy, fx = y / L, x / L
                y0, x0 = int(np.floor(fy)), int(np.floor(fx))
                for (cy, cx, w) in ((y0, x0, (x0 + 1 - fx) * (y0 + 1 - fy)), (y0, x0 + 1, (fx - x0) * (y0 + 1 - fy)),
                                    (y0 + 1, x0, (x0 + 1 - fx) * (fy - y0)), (y0 + 1, x0 + 1, (fx - x0) * (fy - y0))):
                    if 0 <= cy < 8 // L and 0 <= cx < 12 // L:
                        exp[0, :, cy, cx] += img[0, :, y, x] * np.float32(w)
        assert rel_l2(out, exp) < 1e-6


def test_splat_property_p1_two_stage_equals_direct():
    """warp_test.py:59-75: warp(src, flow, scale=L, offset)/L^2 == warp(warp(src, flow), 0, scale=L, offset)/L^2
    within 1e-4 max-abs, for flows mixing integer and real values in [-2, 2]."""
    torch.manual_seed(2)
    L = 2
    src = torch.rand(2, 3, 32, 32)
    flow = torch.where(torch.rand(2, 2, 32, 32) < 0.5, torch.randint(-2, 3, (2, 2, 32, 32)).float(),
                       torch.rand(2, 2, 32, 32) * 4 - 2)
    for off in ((0, 0), (1, 0), (1, 1)):
        direct = W.warp(src, None, flow, mode="forward", scale=L, offset=list(off), set_nans=False) / L ** 2
        stage1 = W.warp(src, None, flow, mode="forward", set_nans=False)
        two = W.warp(stage1, None, torch.zeros_like(flow), mode="forward", scale=L, offset=list(off), set_nans=False) / L ** 2
        inner = (slice(None), slice(None), slice(2, -2), slice(2, -2))       # borders differ by the edge remap
        assert float((direct[inner] - two[inner]).abs().max()) < 1e-4, off


def test_splat_gradients_match_finite_differences_scale1():
    """ingrad is the exact adjoint of the splat; flowgrad matches d(out)/d(flow) for in-range
    samples -- with the reference's crossed dflt factors (SS:664-672) both are 1 at scale 1."""
    torch.manual_seed(3)
    img = torch.rand(1, 2, 6, 7, dtype=torch.float32)
    flow = (torch.rand(1, 2, 6, 7) - 0.5) * 1.5
    gout = torch.rand(1, 2, 6, 7)
    ing = W.splat_ingrad(flow, gout, img.shape)
    # adjoint test: <splat(img), gout> == <img, ingrad>
    lhs = float((W.splat_out(img, flow).double() * gout.double()).sum())
    rhs = float((img.double() * ing.double()).sum())
    assert lhs == pytest.approx(rhs, rel=1e-5)
    fg = W.splat_flowgrad(img, flow, gout)
    eps = 1e-2
    for (c, y, x) in ((0, 2, 3), (1, 3, 2), (0, 4, 4)):
        fp, fm = flow.clone(), flow.clone()
        fp[0, c, y, x] += eps
        fm[0, c, y, x] -= eps
        num = float(((W.splat_out(img, fp).double() - W.splat_out(img, fm).double()) * gout.double()).sum() / (2 * eps))
        assert float(fg[0, c, y, x]) == pytest.approx(num, rel=2e-2, abs=1e-3)


def test_splat_flowgrad_frozen_outside_and_crossed():
    img = torch.ones(1, 1, 4, 4)
    gout = torch.rand(1, 1, 4, 4)
    flow = torch.zeros(1, 2, 4, 4)
    flow[0, 0, :, 3] = 0.5            # x target 3.5 >= W-1 -> dfltXX = 0 (SS:628-631)
    fg = W.splat_flowgrad(img, flow, gout)
    assert (fg[0, 1, :, 3] == 0).all()      # channel 1 is scaled by dfltXX (crossed, SS:671-672)
    assert (fg[0, 0, :3, 3] != 0).any()     # channel 0 is scaled by dfltYY, still live
