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


# ---- splat: an executable third-party pin ---------------------------------------------------
def _aten_grid(flow, dtype=torch.float64):
    """Normalised sampling grid whose ATen un-normalisation ((g+1)/2*(size-1), align_corners=True) lands on x + flow."""
    B, _, H, Wd = flow.shape
    # SS:368-369: fltOutputX = x + flow[:, 0] in the flow's own precision (fp32 in the reference), everything after in `dtype`
    xs = (torch.arange(Wd, dtype=flow.dtype).view(1, 1, Wd) + flow[:, 0]).to(dtype)
    ys = (torch.arange(H, dtype=flow.dtype).view(1, H, 1) + flow[:, 1]).to(dtype)
    return torch.stack((2 * xs / (Wd - 1) - 1, 2 * ys / (H - 1) - 1), dim=-1)


@pytest.mark.parametrize("shape,mag", [((2, 3, 17, 23), 3.0), ((1, 4, 40, 56), 12.0), ((1, 1, 9, 64), 30.0), ((1, 2, 100, 120), 20.0)])
def test_splat_scale1_is_the_adjoint_of_aten_grid_sample(shape, mag):
    """At scale 1 the reference's splat (SS:352-423) is the exact transpose of ATen's bilinear grid_sample
    (align_corners=True, zeros padding) evaluated at the un-normalised coordinates x + flow: torch's CPU autograd of
    that op yields the splat, its input gradient and (for targets inside the plain branch, SS:626-647) its flow gradient
    without going through oracle/splat_ref.c.  This is the third-party pin of the C restatement."""
    B, C, H, Wd = shape
    g = torch.Generator().manual_seed(41)
    img = torch.rand(B, C, H, Wd, generator=g)
    flow = (torch.rand(B, 2, H, Wd, generator=g) * 2 - 1) * mag
    flow[0, :, 1, 1] = 0.0                                                 # an exact integer target
    flow[0, 0, 2, 2] = 1.0
    gout = torch.rand(B, C, H, Wd, generator=g)

    # forward: d/dG <grid_sample(G, grid), img> = splat(img)
    G = torch.zeros(B, C, H, Wd, dtype=torch.float64, requires_grad=True)
    grid = _aten_grid(flow)
    (torch.nn.functional.grid_sample(G, grid, mode="bilinear", padding_mode="zeros", align_corners=True) * img.double()).sum().backward()
    ref = W.splat_out(img, flow)
    assert float((ref.double() - G.grad).abs().max()) <= 1e-6 * max(1.0, float(G.grad.abs().max()))
    assert rel_l2(ref, G.grad) < 2e-7

    # ingrad (SS:489-565) is the gather itself
    gather = torch.nn.functional.grid_sample(gout.double(), grid, mode="bilinear", padding_mode="zeros", align_corners=True)
    ing = W.splat_ingrad(flow, gout, img.shape)
    assert float((ing.double() - gather).abs().max()) <= 2e-6

    # flowgrad (SS:600-700): ATen's grid gradient chained through x + flow, for targets in the plain branch on both axes
    fl = flow.clone().requires_grad_(True)
    s = (torch.nn.functional.grid_sample(gout.double(), _aten_grid(fl), mode="bilinear", padding_mode="zeros",
                                         align_corners=True) * img.double()).sum()
    s.backward()
    fg = W.splat_flowgrad(img, flow, gout)
    xs = torch.arange(Wd).view(1, 1, Wd) + flow[:, 0]
    ys = torch.arange(H).view(1, H, 1) + flow[:, 1]
    frac = lambda v: (v - v.floor())
    plain = ((xs >= 0) & (xs < Wd - 1) & (ys >= 0) & (ys < H - 1) &
             (frac(xs) > 1e-3) & (frac(xs) < 1 - 1e-3) & (frac(ys) > 1e-3) & (frac(ys) < 1 - 1e-3))   # off the kinks
    plain = plain.unsqueeze(1).expand_as(fg)
    assert int(plain.sum()) > 0.3 * plain.numel() / max(1.0, mag / 6)
    assert float((fg.double() - fl.grad)[plain].abs().max()) <= 2e-5 * max(1.0, float(fl.grad.abs().max()))


def test_splat_property_p2_direct_and_two_stage_values_and_gradients():
    """warp_test.py:77-102 (`are_they_equal`): method_a = warp(src, flow, scale=L, offset)/L^2 against
    method_b = warp(warp(src, flow, set_nans=True), 0, scale=L, offset, set_nans=False)/L^2 -- values and the gradient of an
    MSE against a random image, here carried on to `src` through the restated backward kernels (the reference compares
    the gradient at the method output, which is equal iff the values are)."""
    g = torch.Generator().manual_seed(42)
    L = 2
    B, C, H, Wd = 1, 1, 64, 64
    src = torch.rand(B, C, H, Wd, generator=g)
    flow = torch.where(torch.rand(B, 2, H, Wd, generator=g) < 0.5,
                       torch.round(4 * torch.rand(B, 2, H, Wd, generator=g) - 2.0), 4 * torch.rand(B, 2, H, Wd, generator=g) - 2.0)
    comp = torch.rand(B, C, H // L, Wd // L, generator=g)
    zero = torch.zeros_like(flow)
    inner = (slice(None), slice(None), slice(2, -2), slice(2, -2))
    for off in ((0, 0), (1, 0), (0, 1), (1, 1)):
        a = W.warp(src, None, flow, mode="forward", scale=L, set_nans=False, offset=list(off)) / L ** 2
        high = W.warp(src, None, flow, mode="forward", scale=1, set_nans=True, offset=[0, 0])
        b = W.warp(high, None, zero, mode="forward", scale=L, set_nans=False, offset=list(off)) / L ** 2
        assert float((a - b)[inner].abs().max()) < 1e-4, off
        # -d mse / d method (warp_test.py:93-96), restricted to the interior the two methods share
        ga = torch.zeros_like(a)
        gb = torch.zeros_like(b)
        ga[inner] = -2 * (a - comp)[inner] / a[inner].numel()
        gb[inner] = -2 * (b - comp)[inner] / b[inner].numel()
        assert float((ga - gb).abs().max()) < 1e-4 * float(ga.abs().max())
        # ... and pulled back to src: scale-L ingrad against scale-1 ingrad of the zero-flow scale-L ingrad
        da = W.splat_ingrad(flow, ga / L ** 2, src.shape, L, off[0], off[1])
        valid = (~torch.isnan(high)).float()
        db = W.splat_ingrad(flow, W.splat_ingrad(zero, gb / L ** 2, src.shape, L, off[0], off[1]) * valid, src.shape)
        core = (slice(None), slice(None), slice(8, -8), slice(8, -8))
        assert float((da - db)[core].abs().max()) < 1e-4 * float(da.abs().max()), off
