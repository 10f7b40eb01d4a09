"""GPU parity of the flow-warp kernels (through the C-ABI) against the CPU oracle and the
reference goldens.  Corner indices must be bit-exact, values within fp32 summation-order noise."""
import pytest
import torch

from conftest import load_golden, rel_l2
from oracle import warp_ref as WR

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ofd():
    import opticalflowdiffusion_amd as m
    from opticalflowdiffusion_amd import _lib
    _lib.lib()          # fails loudly if the HIP library is missing
    return m


def _flows(B, H, W, seed, mag):
    g = torch.Generator().manual_seed(seed)
    real = (torch.rand(B, 2, H, W, generator=g) * 2 - 1) * mag
    integer = torch.randint(-int(mag), int(mag) + 1, (B, 2, H, W), generator=g).float()
    mixed = torch.where(torch.rand(B, 2, H, W, generator=g) < 0.5, integer, real)
    return {"real": real, "int": integer, "mixed": mixed}


@pytest.mark.parametrize("shape", [(2, 4, 40, 72), (1, 3, 130, 200), (3, 1, 64, 64), (1, 7, 33, 47)])
@pytest.mark.parametrize("kind", ["real", "int", "mixed"])
def test_splat_forward_scale1(ofd, shape, kind):
    from opticalflowdiffusion_amd.softsplat import splat_forward, splat_corners
    B, C, H, W = shape
    torch.manual_seed(1)
    img = torch.rand(B, C, H, W)
    flow = _flows(B, H, W, 2, 9.0)[kind]
    ref, ref_c = WR.splat_out(img, flow, return_corners=True)
    out = splat_forward(img.cuda(), flow.cuda()).cpu()
    corners = splat_corners(flow.cuda()).cpu()
    assert torch.equal(corners, ref_c)                       # integer grid indexing: bit-exact
    assert float((out - ref).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max()))
    assert rel_l2(out, ref) < 1e-6


@pytest.mark.parametrize("scale,off", [(2, (0, 0)), (2, (1, 1)), (4, (3, 1)), (8, (5, 2)), (16, (0, 7))])
def test_splat_forward_pyramid_scales(ofd, scale, off):
    from opticalflowdiffusion_amd.softsplat import splat_forward, splat_corners
    torch.manual_seed(3)
    B, C, H, W = 2, 4, 96, 144
    img = torch.rand(B, C, H, W)
    flow = _flows(B, H, W, 4, 6.0)["mixed"]
    ref, ref_c = WR.splat_out(img, flow, scale, off[0], off[1], return_corners=True)
    out = splat_forward(img.cuda(), flow.cuda(), scale, off[0], off[1]).cpu()
    assert torch.equal(splat_corners(flow.cuda(), scale, off[0], off[1]).cpu(), ref_c)
    assert rel_l2(out, ref) < 1e-6


def test_splat_far_displacements_and_nonfinite(ofd):
    """|flow| far beyond the tile radius goes through the far-corner list; inf/NaN flow is skipped."""
    from opticalflowdiffusion_amd.softsplat import splat_forward
    torch.manual_seed(5)
    B, C, H, W = 2, 3, 150, 260
    img = torch.rand(B, C, H, W)
    flow = (torch.rand(B, 2, H, W) * 2 - 1) * 120.0
    flow[0, 0, 3, 5] = float("inf")
    flow[1, 1, 9, 9] = float("nan")
    ref = WR.splat_out(img, flow)
    for radius in (0, 8, 24, 400):
        out = splat_forward(img.cuda(), flow.cuda(), radius=radius).cpu()
        assert rel_l2(out, ref) < 1e-6, radius


def test_splat_converging_flows_and_wide_dynamic_range(ofd):
    """The scale-1 forward splat keeps its accumulators as 64-bit fixed point, 44 fraction bits below the largest finite |in| of the
    (sample, channel) plane (warp.hip: splat_tile_fast_kernel).  (1) flows converging on one tile overflow its survivor list: the
    overflow entries are accumulated in place; (2) a plane whose values span six decades keeps 1e-6 relative accuracy in its small
    region (the documented limit: relative error = 2^-44 x plane maximum / value); (3) non-finite inputs raise the IEEE result at
    their corners only.  Against oracle/splat_ref.c."""
    from opticalflowdiffusion_amd.softsplat import splat_forward
    torch.manual_seed(6)
    B, C, H, W = 2, 4, 192, 256
    img = torch.rand(B, C, H, W) + 0.1
    yy, xx = torch.meshgrid(torch.arange(H).float(), torch.arange(W).float(), indexing="ij")
    flow = torch.zeros(B, 2, H, W)
    # sample 0: two bundles of pixels converge on the tile x 64..127, y 64..127: more than 6144 survivors for that tile
    near = ((xx - 100).abs() <= 24) & ((yy - 96).abs() <= 24)
    flow[0, 0] = torch.where(near, 100.3 - xx, torch.rand(H, W) * 4 - 2)
    flow[0, 1] = torch.where(near, 96.6 - yy, torch.rand(H, W) * 4 - 2)
    near2 = ((xx - 70).abs() <= 30) & ((yy - 110).abs() <= 14)
    flow[0, 0] = torch.where(near2 & ~near, 75.5 - xx + (xx % 7) * 0.37, flow[0, 0])
    flow[0, 1] = torch.where(near2 & ~near, 100.25 - yy + (yy % 5) * 0.61, flow[0, 1])
    flow[1] = (torch.rand(2, H, W) * 2 - 1) * 9.0
    img[1, :, :, :128] *= 1e-3
    img[1, :, :, 128:] *= 1e3
    img[1, 2, 40, 200] = float("inf")
    img[1, 1, 41, 30] = float("nan")
    ref = WR.splat_out(img, flow)
    out = splat_forward(img.cuda(), flow.cuda()).cpu()
    fin = torch.isfinite(ref)
    assert torch.equal(torch.isnan(out), torch.isnan(ref)) and torch.equal(torch.isinf(out), torch.isinf(ref))
    assert rel_l2(out[0], ref[0]) < 1e-6
    small, big = (slice(None), slice(None), slice(0, 100)), (slice(None), slice(None), slice(160, 256))
    o1, r1, f1 = out[1], ref[1], fin[1]
    assert rel_l2(torch.where(f1, o1, torch.zeros_like(o1))[small], torch.where(f1, r1, torch.zeros_like(r1))[small]) < 1e-6
    assert rel_l2(torch.where(f1, o1, torch.zeros_like(o1))[big], torch.where(f1, r1, torch.zeros_like(r1))[big]) < 1e-6
    assert 0 < float(r1[0, :, :100].abs().max()) < 1e-1 and float(r1[0, :, 160:].abs().max()) > 1e2      # the fixture does span the decades


def test_splat_backward_kernels(ofd):
    from opticalflowdiffusion_amd._lib import lib, check, ptr, stream
    torch.manual_seed(6)
    B, C, H, W = 2, 4, 48, 80
    img = torch.rand(B, C, H, W)
    for scale, off, mag in ((1, (0, 0), 5.0), (2, (1, 0), 5.0), (4, (2, 3), 30.0)):
        flow = _flows(B, H, W, 7, mag)["mixed"]
        gout = torch.rand(B, C, H // scale, W // scale)
        ref_in = WR.splat_ingrad(flow, gout, img.shape, scale, off[0], off[1])
        ref_fl = WR.splat_flowgrad(img, flow, gout, scale, off[0], off[1])
        d_img, d_flow, d_g = img.cuda(), flow.cuda(), gout.cuda()
        g_in = torch.empty_like(d_img)
        g_fl = torch.empty_like(d_flow)
        check(lib().ofd_splat_bwd_in(ptr(d_flow), ptr(d_g), ptr(g_in), B, C, H, W, scale, off[0], off[1], stream()))
        check(lib().ofd_splat_bwd_flow(ptr(d_img), ptr(d_flow), ptr(d_g), ptr(g_fl), B, C, H, W, scale, off[0], off[1], stream()))
        assert rel_l2(g_in.cpu(), ref_in) < 1e-6, scale
        assert rel_l2(g_fl.cpu(), ref_fl) < 1e-5, scale


def test_softsplat_modes_and_autograd(ofd):
    torch.manual_seed(8)
    B, C, H, W = 1, 3, 24, 40
    img = torch.rand(B, C, H, W)
    flow = _flows(B, H, W, 9, 3.0)["real"]
    metric = torch.rand(B, 1, H, W)
    for mode, m in (("sum", None), ("avg", None), ("linear", metric), ("linear_unn", metric), ("soft", metric),
                    ("linear-zeroeps", metric), ("linear-clipeps", metric)):
        ref = WR.softsplat(img, flow, m, mode)
        out = ofd.softsplat(img.cuda(), flow.cuda(), None if m is None else m.cuda(), mode).cpu()
        assert rel_l2(out, ref) < 1e-6, mode
    a = img.cuda().requires_grad_(True)
    f = flow.cuda().requires_grad_(True)
    out = ofd.softsplat(a, f, None, "sum")
    gout = torch.rand_like(out)
    out.backward(gout)
    assert rel_l2(a.grad.cpu(), WR.splat_ingrad(flow, gout.cpu(), img.shape)) < 1e-6
    assert rel_l2(f.grad.cpu(), WR.splat_flowgrad(img, flow, gout.cpu())) < 1e-5


def test_warp_forward_wrapper(ofd):
    """warp(mode='forward') incl. NaN inputs, holes, scale/offset and warp_style (WP:121-156)."""
    torch.manual_seed(10)
    B, C, H, W = 2, 3, 64, 96
    img = torch.rand(B, C, H, W) * 2 - 1
    img[0, 1, 5, 7] = float("nan")
    img[1, :, 20, 30] = float("nan")
    flow = _flows(B, H, W, 11, 12.0)["mixed"]
    for kw in ({}, {"scale": 2}, {"scale": 4, "offset": [5, 2]}, {"set_nans": False}, {"warp_style": "linear"},
               {"get_variance": True}):
        ref = WR.warp(img, None, flow, mode="forward", **kw)
        out = ofd.warp(img.cuda(), None, flow.cuda(), mode="forward", **kw).cpu()
        assert torch.equal(torch.isnan(out), torch.isnan(ref)), kw
        ok = ~torch.isnan(ref)
        assert float((out[ok] - ref[ok]).abs().max()) < 2e-5, kw


def test_warp_test_known_answer_and_property_p1(ofd):
    """warp_test.py:22-27 known answer, warp_test.py:59-75 property P1 -- on the GPU path."""
    src = torch.zeros(1, 1, 2, 4)
    src[0, 0, 1, 2] = 1.0
    flow = torch.zeros(1, 2, 2, 4)
    flow[0, 0] = 0.5
    out = ofd.warp(src.cuda(), None, flow.cuda(), mode="forward", set_nans=False).cpu()
    exp = torch.zeros(1, 1, 2, 4)
    exp[0, 0, 1, 2] = exp[0, 0, 1, 3] = 0.5
    assert torch.equal(out, exp)
    torch.manual_seed(12)
    L = 2
    s = torch.rand(2, 3, 128, 128).cuda()
    f = _flows(2, 128, 128, 13, 2.0)["mixed"].cuda()
    for off in ([0, 0], [1, 0], [1, 1]):
        direct = ofd.warp(s, None, f, mode="forward", scale=L, offset=off, set_nans=False) / L ** 2
        two = ofd.warp(ofd.warp(s, None, f, mode="forward", set_nans=False), None, torch.zeros_like(f), mode="forward",
                       scale=L, offset=off, set_nans=False) / L ** 2
        assert float((direct - two)[:, :, 2:-2, 2:-2].abs().max()) < 1e-4


def test_grid_sample_warp_against_reference_goldens(ofd):
    from opticalflowdiffusion_amd.warp import grid_warp_corners
    g = load_golden("warp_backward")
    cases = [("rand", g["rand.img"], g["rand.flow"]), ("int", g["rand.img"], g["int.flow"]),
             ("wideint", g["wide.img"], g["wideint.flow"]), ("wide", g["wide.img"], torch.zeros(1, 2, 6, 1024))]
    for tag, img, flow in cases:
        out, mask = ofd.warp(None, img.cuda(), flow.cuda(), mode="backward")
        assert torch.equal(mask.cpu(), g[f"{tag}.mask"]), tag
        assert float((out.cpu() - g[f"{tag}.out"]).abs().max()) < 2e-6, tag
        # integer grid indexing: the exact fp32 op order of WP:108-109 + ATen's un-normalise
        B, _, H, W = flow.shape
        xx = torch.arange(W).view(1, 1, W).float() + flow[:, 1]
        yy = torch.arange(H).view(1, H, 1).float() + flow[:, 0]
        vx = 2.0 * xx / max(W - 1, 1) - 1.0
        vy = 2.0 * yy / max(H - 1, 1) - 1.0
        ix = torch.floor(((vx + 1) / 2) * (W - 1)).int()
        iy = torch.floor(((vy + 1) / 2) * (H - 1)).int()
        c = grid_warp_corners(flow.cuda()).cpu()
        assert torch.equal(c[..., 0], ix) and torch.equal(c[..., 1], iy), tag


def test_grid_sample_warp_gradients_against_oracle_autograd(ofd):
    """warp(mode='backward') is differentiable like the reference's (autograd through WP:95-119): d/d second (bilinear scatter)
    and d/d flow (ATen's grid gradient) against CPU autograd of the oracle restatement, incl. targets outside the image, the
    border rows/columns and a wide tile-crossing case."""
    from oracle import warp_ref as WR
    torch.manual_seed(21)
    for (B, C, H, W, amp) in [(2, 3, 20, 32, 3.0), (1, 2, 37, 64, 12.0), (1, 3, 70, 132, 30.0), (2, 1, 9, 8, 6.0)]:
        img = torch.rand(B, C, H, W)
        flow = (torch.rand(B, 2, H, W) * 2 - 1) * amp
        flow[0, :, 0, 0] = 0.25                                   # an interior sub-pixel case at the corner
        gout = torch.randn(B, C, H, W)
        a, b = img.clone().requires_grad_(True), flow.clone().requires_grad_(True)
        ro, rm = WR.warp_backward_flow(a, b)
        (ro * gout).sum().backward()
        x, f = img.cuda().requires_grad_(True), flow.cuda().requires_grad_(True)
        o, m = ofd.warp(None, x, f, mode="backward")
        assert not m.requires_grad
        (o * gout.cuda()).sum().backward()
        scale_s = float(a.grad.abs().max()) + 1e-6
        scale_f = float(b.grad.abs().max()) + 1e-6
        assert float((x.grad.cpu() - a.grad).abs().max()) < 2e-5 * scale_s + 1e-6, (B, C, H, W)
        assert float((f.grad.cpu() - b.grad).abs().max()) < 2e-5 * scale_f + 1e-6, (B, C, H, W)
    # only one of the two gradients requested
    x = img.cuda().requires_grad_(True)
    o, _ = ofd.warp(None, x, flow.cuda(), mode="backward")
    o.sum().backward()
    assert x.grad is not None
    f = flow.cuda().requires_grad_(True)
    o, _ = ofd.warp(None, img.cuda(), f, mode="backward")
    o.sum().backward()
    assert f.grad is not None and bool(torch.isfinite(f.grad).all())


def test_grid_sample_warp_full_size_properties(ofd):
    """BASELINE size (16,3,440,1024): zero flow is the identity up to the fp32 round trip, and a
    constant integer shift equals a slice (size-independent properties; the oracle is too slow here)."""
    torch.manual_seed(14)
    img = torch.rand(16, 3, 440, 1024, device="cuda")
    z = torch.zeros(16, 2, 440, 1024, device="cuda")
    out, mask = ofd.warp(None, img, z, mode="backward")
    assert float((out - img).abs().max()) < 2e-4 and bool((mask == 1).all())
    f = z.clone()
    f[:, 1] = 3.0        # channel 1 displaces x after the flip (WP:105)
    f[:, 0] = -2.0
    out, mask = ofd.warp(None, img, f, mode="backward")
    assert float((out[:, :, 2:, :-3] - img[:, :, :-2, 3:]).abs().max()) < 2e-4
    assert bool((mask[:, :, :2] == 0).all()) and bool((mask[:, :, :, -3:] == 0).all())
    # adjoint identity at full size: <warp(img), g> == <img, warp^T(g)> for the scatter kernel behind d/d second
    fr = (torch.rand(16, 2, 440, 1024, device="cuda") * 2 - 1) * 20.0
    g = torch.randn(16, 3, 440, 1024, device="cuda")
    x = img.clone().requires_grad_(True)
    o, _ = ofd.warp(None, x, fr, mode="backward")
    lhs = float((o.detach().double() * g.double()).sum())
    (o * g).sum().backward()
    rhs = float((img.double() * x.grad.double()).sum())
    assert abs(lhs - rhs) < 1e-6 * max(abs(lhs), 1.0) + 1e-2


@pytest.mark.parametrize("B,H,W", [(16, 440, 1024), (3, 100, 1000), (1, 64, 64), (2, 40, 132), (1, 97, 260)])
def test_grid_sample_warp_ring_kernel_against_the_oracle_at_size(ofd, B, H, W):
    """The C = 3 grid_sample warp at sizes: the BAND kernel (warp.hip: grid_warp_band_kernel, r04: a workgroup slides down a 128-column band
    over a 72-row LDS ring, counted waits, look-ahead window groups) where W >= 128 and H >= 32, else the 64 x 64 tile kernel.  At the
    BASELINE size every workgroup walks a 224-row segment of a band; (3, 100, 1000) has a partial band and a partial last step; (2, 40, 132)
    a 4-column band; (1, 97, 260) an odd height; one sample carries displacements beyond the staged window (the global-load fallback) and
    non-finite flows.  Against the CPU oracle (the torch op the reference calls, WP:95-119): mask bit-equal, values within 2e-6; the
    mask-less entry point (mask = NULL) returns the same image to the bit."""
    from opticalflowdiffusion_amd._lib import lib, check, ptr, stream
    g = torch.Generator().manual_seed(B * 1000 + H)
    img = torch.rand(B, 3, H, W, generator=g)
    flow = torch.nn.functional.avg_pool2d(torch.randn(B, 2, H, W, generator=g) * 8 * 9, 9, 1, 4).clamp(-20, 20)      # SURVEY 8d flow
    flow[-1] = (torch.rand(2, H, W, generator=g) * 2 - 1) * 45.0                                                    # beyond the window
    flow[0, :, H // 3, W // 2] = float("nan")
    flow[0, 0, H // 2, W // 3] = float("inf")
    flow[0, 1, 5, 7] = -float("inf")
    ro, rm = WR.warp_backward_flow(img, flow)
    img_d, flow_d = img.cuda(), flow.cuda()                   # (kept alive: the C call below takes raw pointers)
    o, m = ofd.warp(None, img_d, flow_d, mode="backward")
    torch.cuda.synchronize()
    ok = torch.isfinite(ro)                                   # (non-finite grid positions: ATen returns NaN there or 0, see below)
    assert torch.equal(m.cpu()[ok], rm[ok])
    assert float((o.cpu()[ok] - ro[ok]).abs().max()) < 2e-6
    bad = ~torch.isfinite(flow).all(dim=1, keepdim=True).expand_as(ro)
    assert bool((o.cpu()[bad] == 0).all()) and bool((m.cpu()[bad] == 0).all())      # non-finite target: no corner in bounds (as ATen's zeros padding)
    o2 = torch.empty_like(o)
    check(lib().ofd_grid_warp_fwd(ptr(img_d), ptr(flow_d), ptr(o2), None, B, 3, H, W, stream()))
    torch.cuda.synchronize()
    assert torch.equal(o2, o)


@pytest.mark.parametrize("B,H,W", [(2, 40, 132), (1, 97, 260), (3, 100, 1000), (4, 440, 1024)])
def test_grid_sample_warp_band_kernel_equals_the_tile_kernel(ofd, B, H, W):
    """grid_warp_band_kernel (the default for C = 3 at W >= 128) against grid_warp_tile_kernel (OFD_GW_BAND=0) on the same inputs: same
    arithmetic in the same order, so image and mask are equal to the bit -- also where corners leave the staged window (|flow| up to 45 px),
    at the image border, with NaN / inf flows and at a segment seam."""
    import os
    g = torch.Generator().manual_seed(7 * B + H)
    img = torch.rand(B, 3, H, W, generator=g).cuda()
    flow = torch.nn.functional.avg_pool2d(torch.randn(B, 2, H, W, generator=g) * 8 * 9, 9, 1, 4).clamp(-20, 20)
    flow[-1] = (torch.rand(2, H, W, generator=g) * 2 - 1) * 45.0
    flow[0, :, H // 3, W // 2] = float("nan")
    flow[0, 0, H // 2, W // 3] = float("inf")
    flow = flow.cuda()
    old = os.environ.get("OFD_GW_BAND")
    try:
        os.environ["OFD_GW_BAND"] = "1"
        o1, m1 = ofd.warp(None, img, flow, mode="backward")
        os.environ["OFD_GW_BAND"] = "0"
        o0, m0 = ofd.warp(None, img, flow, mode="backward")
    finally:
        if old is None:
            os.environ.pop("OFD_GW_BAND", None)
        else:
            os.environ["OFD_GW_BAND"] = old
    torch.cuda.synchronize()
    assert torch.equal(m1, m0)
    assert torch.equal(o1, o0), float((o1 - o0).abs().max())


def test_splat_full_size_properties(ofd):
    """BASELINE size (16,4,440,1024): mass conservation (sum of the splat == sum of the inputs
    whose four corners stay inside) and zero flow identity."""
    from opticalflowdiffusion_amd.softsplat import splat_forward
    torch.manual_seed(15)
    img = torch.rand(16, 4, 440, 1024, device="cuda")
    z = torch.zeros(16, 2, 440, 1024, device="cuda")
    assert torch.equal(splat_forward(img, z), img)
    f = (torch.rand(16, 2, 440, 1024, device="cuda") * 2 - 1) * 20.0
    out = splat_forward(img, f)
    xs = torch.arange(1024, device="cuda").view(1, 1, 1024) + f[:, 0]
    ys = torch.arange(440, device="cuda").view(1, 440, 1) + f[:, 1]
    inside = ((xs >= 0) & (xs <= 1023) & (ys >= 0) & (ys <= 439)).unsqueeze(1)
    lo = float((img * inside).double().sum())
    hi = float(img.double().sum())
    total = float(out.double().sum())
    assert lo * (1 - 1e-5) <= total <= hi * (1 + 1e-5)


def test_c5_warp_kernels_at_8x3x1080x1920(ofd):
    """BASELINE configs[4] (1080p, bs 8): the splat conserves mass and reproduces the image under zero flow; grid_sample warp is
    the identity under zero flow (up to the fp32 round trip of WP:108-109) and a slice under an integer shift; the two are adjoint
    (<gather(x), g> == <x, scatter(g)>: grid_sample's d/d second is the splat kernel on the gather's coordinates)."""
    from opticalflowdiffusion_amd.softsplat import splat_forward
    B, C, H, W = 8, 3, 1080, 1920
    g = torch.Generator(device="cuda").manual_seed(17)
    img = torch.rand(B, C + 1, H, W, device="cuda", generator=g)
    z = torch.zeros(B, 2, H, W, device="cuda")
    assert torch.equal(splat_forward(img, z), img)
    f = (torch.rand(B, 2, H, W, device="cuda", generator=g) * 2 - 1) * 20.0
    out = splat_forward(img, f)
    xs = torch.arange(W, device="cuda").view(1, 1, W) + f[:, 0]
    ys = torch.arange(H, device="cuda").view(1, H, 1) + f[:, 1]
    inside = ((xs >= 0) & (xs <= W - 1) & (ys >= 0) & (ys <= H - 1)).unsqueeze(1)
    lo, hi, total = float((img * inside).double().sum()), float(img.double().sum()), float(out.double().sum())
    assert lo * (1 - 1e-5) <= total <= hi * (1 + 1e-5)
    # an integer shift moves whole pixels: exact
    fi = z.clone()
    fi[:, 0], fi[:, 1] = 7.0, -3.0
    sh = splat_forward(img, fi)
    assert torch.equal(sh[:, :, :-3, 7:], img[:, :, 3:, :-7]) and bool((sh[:, :, -3:] == 0).all()) and bool((sh[:, :, :, :7] == 0).all())
    del out, sh, inside
    im3 = img[:, :3].contiguous()
    o, m = ofd.warp(None, im3, z, mode="backward")
    assert float((o - im3).abs().max()) < 4e-4 and bool((m == 1).all())
    fb = z.clone()
    fb[:, 1], fb[:, 0] = 3.0, -2.0                       # channel 1 displaces x after the flip (WP:105)
    o, m = ofd.warp(None, im3, fb, mode="backward")
    assert float((o[:, :, 2:, :-3] - im3[:, :, :-2, 3:]).abs().max()) < 4e-4
    assert bool((m[:, :, :2] == 0).all()) and bool((m[:, :, :, -3:] == 0).all())
    gg = torch.randn(B, 3, H, W, device="cuda", generator=g)
    x = im3.clone().requires_grad_(True)
    o, _ = ofd.warp(None, x, f, mode="backward")
    lhs = float((o.detach().double() * gg.double()).sum())
    (o * gg).sum().backward()
    rhs = float((im3.double() * x.grad.double()).sum())
    assert abs(lhs - rhs) < 1e-6 * max(abs(lhs), 1.0) + 2e-2


def test_hip_splat_against_aten_grid_sample_adjoint(ofd):
    """The third-party pin of tests/test_oracle_warp.py applied to the HIP kernels themselves: at scale 1 `ofd_splat_fwd` is the
    transpose of ATen's bilinear grid_sample at x + flow, `ofd_splat_bwd_in` is that gather and `ofd_splat_bwd_flow` its grid
    gradient (plain branch) -- torch CPU autograd in float64 is the checker, oracle/splat_ref.c is not involved."""
    from opticalflowdiffusion_amd.softsplat import splat_forward
    from opticalflowdiffusion_amd._lib import lib, check, ptr, stream
    from test_oracle_warp import _aten_grid
    GS = lambda a, grid: torch.nn.functional.grid_sample(a, grid, mode="bilinear", padding_mode="zeros", align_corners=True)
    for (B, C, H, W, mag) in [(2, 4, 130, 200, 20.0), (1, 3, 70, 96, 4.0)]:
        g = torch.Generator().manual_seed(43)
        img = torch.rand(B, C, H, W, generator=g)
        flow = (torch.rand(B, 2, H, W, generator=g) * 2 - 1) * mag
        gout = torch.rand(B, C, H, W, generator=g)
        G = torch.zeros(B, C, H, W, dtype=torch.float64, requires_grad=True)
        (GS(G, _aten_grid(flow)) * img.double()).sum().backward()
        out = splat_forward(img.cuda(), flow.cuda()).cpu()
        assert float((out.double() - G.grad).abs().max()) <= 1e-6 * max(1.0, float(G.grad.abs().max()))
        assert rel_l2(out, G.grad) < 2e-7
        fl = flow.clone().requires_grad_(True)
        gather = GS(gout.double(), _aten_grid(fl))
        (gather * img.double()).sum().backward()
        d_img, d_flow, d_g = img.cuda(), flow.cuda(), gout.cuda()
        g_in, g_fl = torch.empty_like(d_img), torch.empty_like(d_flow)
        check(lib().ofd_splat_bwd_in(ptr(d_flow), ptr(d_g), ptr(g_in), B, C, H, W, 1, 0, 0, stream()))
        check(lib().ofd_splat_bwd_flow(ptr(d_img), ptr(d_flow), ptr(d_g), ptr(g_fl), B, C, H, W, 1, 0, 0, stream()))
        assert float((g_in.cpu().double() - gather.detach()).abs().max()) <= 2e-6
        xs = torch.arange(W).view(1, 1, W) + flow[:, 0]
        ys = torch.arange(H).view(1, H, 1) + flow[:, 1]
        fr = lambda v: v - v.floor()
        plain = ((xs >= 0) & (xs < W - 1) & (ys >= 0) & (ys < H - 1) & (fr(xs) > 1e-3) & (fr(xs) < 1 - 1e-3) &
                 (fr(ys) > 1e-3) & (fr(ys) < 1 - 1e-3)).unsqueeze(1).expand(B, 2, H, W)
        assert float((g_fl.cpu().double() - fl.grad)[plain].abs().max()) <= 2e-5 * max(1.0, float(fl.grad.abs().max()))


def test_warp_test_property_p2_values_and_gradients(ofd):
    """warp_test.py:77-102 on the GPU path: method_a (direct scale-L splat) against method_b (scale-1 splat with NaN holes, then a
    zero-flow scale-L splat), values and the gradient of an MSE against a random image -- carried through to `src` by autograd
    over the HIP backward kernels, which is where the two methods use different kernels (scale-L ingrad vs scale-1 o scale-L)."""
    g = torch.Generator().manual_seed(44)
    L = 2
    B, C, H, W = 1, 1, 128, 128                                                  # warp_test.py:14
    flow = torch.where(torch.rand(B, 2, H, W, generator=g) < 0.5, torch.round(4 * torch.rand(B, 2, H, W, generator=g) - 2.0),
                       4 * torch.rand(B, 2, H, W, generator=g) - 2.0).cuda()
    src0 = torch.rand(B, C, H, W, generator=g).cuda()
    comp = torch.rand(B, C, H // L, W // L, generator=g).cuda()
    zero = torch.zeros_like(flow)
    inner = (slice(None), slice(None), slice(2, -2), slice(2, -2))
    for off in ([0, 0], [1, 0], [0, 1], [1, 1]):
        grads, vals = [], []
        for method in ("a", "b"):
            src = src0.clone().requires_grad_(True)
            if method == "a":
                m = ofd.warp(src, None, flow, scale=L, set_nans=False, mode="forward", offset=off) / L ** 2
            else:
                high = ofd.warp(src, None, flow, scale=1, set_nans=True, mode="forward", offset=[0, 0])
                m = ofd.warp(high, None, zero, scale=L, set_nans=False, mode="forward", offset=off) / L ** 2
            torch.nn.functional.mse_loss(m[inner], comp[inner]).backward()
            vals.append(m.detach())
            grads.append(src.grad)
        assert float((vals[0] - vals[1])[inner].abs().max()) < 1e-4, off
        core = (slice(None), slice(None), slice(8, -8), slice(8, -8))
        assert float((grads[0] - grads[1])[core].abs().max()) < 1e-4 * float(grads[0].abs().max()), off


def test_diffusion_elementwise_kernels(ofd):
    from opticalflowdiffusion_amd._lib import lib, check, ptr, stream
    from oracle import diffusion_ref as D
    S = D.make_schedule(1000)
    torch.manual_seed(16)
    B, C, H, Wd = 3, 2, 24, 40
    x0 = torch.rand(B, C, H, Wd) * 2 - 1
    nz = torch.randn(B, C, H, Wd)
    mo = torch.randn(B, C, H, Wd) * 1.5
    t = torch.tensor([0, 499, 999])
    n = C * H * Wd
    keep = []

    def d(v):                       # device copy that stays alive until the test ends
        keep.append(v.cuda().contiguous())
        return ptr(keep[-1])

    out = torch.empty(B, C, H, Wd, device="cuda")
    xs = torch.empty(B, C, H, Wd, device="cuda")
    check(lib().ofd_q_sample(d(x0), d(nz), d(S["sqrt_alphas_cumprod"][t]), d(S["sqrt_one_minus_alphas_cumprod"][t]),
                             ptr(out), B, n, stream()))
    assert rel_l2(out.cpu(), D.q_sample(S, x0, t, nz)) < 1e-6
    for ti in (999, 1, 0):
        tt = torch.full((B,), ti)
        sigma = (0.5 * S["posterior_log_variance_clipped"][tt]).exp() if ti > 0 else torch.zeros(B)
        check(lib().ofd_ddpm_update(d(x0), d(mo), d(nz), d(S["posterior_mean_coef1"][tt]), d(S["posterior_mean_coef2"][tt]),
                                    d(sigma), ptr(out), ptr(xs), B, n, stream()))
        ref, ref_xs = D.p_sample_update(S, x0, ti, mo, nz)
        assert rel_l2(out.cpu(), ref) < 1e-6 and torch.equal(xs.cpu(), ref_xs), ti
    for time, time_next in ((999, 979), (19, -1)):
        tt = torch.full((B,), time)
        an = S["alphas_cumprod"][max(time_next, 0)]
        rep = lambda v: d(v.reshape(1).repeat(B))
        check(lib().ofd_ddim_update(d(x0), d(mo), None, d(S["sqrt_recip_alphas_cumprod"][tt]),
                                    d(S["sqrt_recipm1_alphas_cumprod"][tt]), rep(an.sqrt()), rep((1 - an).sqrt()), None,
                                    int(time_next < 0), ptr(out), ptr(xs), B, n, stream()))
        ref, ref_xs = D.ddim_update(S, x0, time, time_next, mo, torch.zeros_like(x0))
        assert rel_l2(out.cpu(), ref) < 1e-6, time
    a = torch.randn(5000)
    b = torch.randn(5000)
    a[17] = float("nan")
    b[4000] = float("nan")
    assert float(ofd.nan_mse(a.cuda(), b.cuda())) == pytest.approx(float(torch.nanmean(D.nan_mse_none(a, b))), rel=1e-6)
