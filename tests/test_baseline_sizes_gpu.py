"""BASELINE.json's configurations at their FULL sizes, through the plugin surface, checked by size-independent properties
(the CPU oracle needs minutes per sample here):

  C2  one training step at 16 x 440 x 1024, bf16, T = 1000  (FD:218-235, DD:823-891)
  C3  50-step DDIM at bs = 64 + forward-splat reconstruction  (DD:731-774, FD:189-215)
  C5  splat / grid_sample warp at 8 x 3 x 1080 x 1920         (SS:339-454, WP:95-119)   [UNet at 1080p: test_plugin_gpu.py]

C4 (bs 128 over 8 GPUs) needs hardware this box does not have; its data path is C2 per rank + tests/test_distributed_cpu.py.
"""
import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu

H, W = 440, 1024


def _batch(B, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    img = torch.rand(B, 3, H, W, device="cuda", generator=g)
    tgt = torch.rand(B, 3, H, W, device="cuda", generator=g)
    flow = torch.clamp(torch.randn(B, 2, H, W, device="cuda", generator=g) * 8.0, -20, 20)
    flow = torch.nn.functional.avg_pool2d(flow, 9, stride=1, padding=4)          # Sintel-like smooth flow
    return img, tgt, flow


def test_c2_training_step_at_16x440x1024_is_the_mean_of_its_per_sample_steps():
    """Samples are independent in the loss (per-sample t and noise, GroupNorm per sample, DD:985-993), so the gradient of the
    batch-16 step must equal the mean of the sixteen batch-1 gradients of the same samples with the same t and noise: checks
    every kernel of the forward / backward at the BASELINE size against itself at B = 1 (whose small-shape twin is checked
    against oracle autograd in test_backward_gpu.py), plus finiteness of the loss and of all 276 gradients."""
    from opticalflowdiffusion_amd import FlowDiffuser
    torch.manual_seed(0)
    B = 16
    cfg = dict(target="flow", image_size=[H, W], timesteps=1000, flow_max=20, zero_init=False)
    fd = FlowDiffuser(cfg).cuda()
    fd.train()
    img, tgt, flow = _batch(B, 1)
    tgt_, cond, flow_ = fd.preprocess((img, tgt, flow), aug=False)
    g = torch.Generator(device="cuda").manual_seed(2)
    t = torch.randint(0, 1000, (B,), device="cuda", generator=g)
    noise = torch.randn(B, 2, H, W, device="cuda", generator=g)
    params = [p for p in fd.model.parameters()]
    assert len(list(fd.unet.state_dict())) == 276

    loss = fd.model.p_losses(tgt_, t, noise=noise, external_cond=cond)
    loss.backward()
    assert torch.isfinite(loss) and 0 < float(loss.detach()) < 10
    flat16 = torch.cat([p.grad.reshape(-1) for p in params]).clone()
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in params)
    assert float(flat16.abs().max()) > 0
    loss16 = float(loss.detach())

    acc = torch.zeros_like(flat16, dtype=torch.float64)
    losses = []
    for i in range(B):
        for p in params:
            p.grad = None
        li = fd.model.p_losses(tgt_[i:i + 1], t[i:i + 1], noise=noise[i:i + 1], external_cond=cond[i:i + 1])
        li.backward()
        losses.append(float(li.detach()))
        acc += torch.cat([p.grad.reshape(-1) for p in params]).double()
    mean = (acc / B).float()
    assert loss16 == pytest.approx(sum(losses) / B, rel=1e-5)
    err = rel_l2(flat16, mean)
    print(f"\n  C2: loss {loss16:.5f}; batch-16 gradient vs mean of 16 batch-1 gradients rel-L2 {err:.2e}, "
          f"max-abs {float((flat16 - mean).abs().max()):.2e} of {float(mean.abs().max()):.2e}")
    assert err < 1e-3
    # per-parameter: no tensor hides behind the big ones
    off = 0
    worst = ("", 0.0)
    names = [n for n, _ in fd.model.named_parameters()]
    for n, p in zip(names, params):
        k = p.numel()
        a, b = flat16[off:off + k], mean[off:off + k]
        off += k
        e = float((a - b).norm() / (b.norm() + 1e-12 * (1 + float(mean.abs().max()))))
        if float(b.norm()) > 1e-6 * float(mean.norm()) and e > worst[1]:
            worst = (n, e)
    print(f"  worst parameter: {worst[0]} {worst[1]:.2e}")
    assert worst[1] < 5e-3, worst


def test_c3_ddim_50_steps_bs64_and_splat_reconstruction():
    """50-step DDIM (eta = 0: deterministic given x_T) at bs = 64, 440 x 1024, then FD:200-202's forward-splat reconstruction.
    Chains are independent: (1) with the batch order reversed every chain must come out bit for bit the same (all kernels of
    the denoise step are deterministic and per-sample); (2) a bs-2 run of the first two chains from the same x_T must agree to
    bf16 noise -- not to the bit: the executor sizes the LinearAttention partial sums and the conv workgroup width from the
    grid, which depends on B, so fp32 summation order differs and 50 steps carry that to ~5e-3; outputs finite; holes of
    the reconstruction consistent across channels."""
    from opticalflowdiffusion_amd import FlowDiffuser, warp
    torch.manual_seed(0)
    B = 64
    cfg = dict(target="flow", image_size=[H, W], timesteps=1000, sampling_timesteps=50, flow_max=20, zero_init=False)
    fd = FlowDiffuser(cfg).cuda()
    fd.eval()
    g = torch.Generator(device="cuda").manual_seed(5)
    cond = torch.rand(B, 3, H, W, device="cuda", generator=g) * 2 - 1
    x_T = torch.randn(B, 2, H, W, device="cuda", generator=g)
    assert fd.model.is_ddim_sampling and fd.model.sampling_timesteps == 50
    with torch.no_grad():
        flow64 = fd.model.ddim_sample((B, 2, H, W), external_cond=cond, x_T=x_T)
        rev = fd.model.ddim_sample((B, 2, H, W), external_cond=cond.flip(0).contiguous(), x_T=x_T.flip(0).contiguous())
        flow2 = fd.model.ddim_sample((2, 2, H, W), external_cond=cond[:2], x_T=x_T[:2])
    assert flow64.shape == (B, 2, H, W) and bool(torch.isfinite(flow64).all())
    assert torch.equal(rev.flip(0), flow64)
    del rev
    for i in range(2):
        e = rel_l2(flow64[i].cpu(), flow2[i].cpu())
        print(f"\n  C3 chain {i}: bs-64 vs bs-2 rel-L2 {e:.2e}")
        assert e < 2e-2
    assert not torch.equal(flow64[0], flow64[1])
    with torch.no_grad():
        rec = warp(cond, None, flow64, mode="forward")                                  # FD:200-202 (flow as the net emits it)
    holes = torch.isnan(rec)
    assert rec.shape == (B, 3, H, W) and bool(torch.isfinite(rec[~holes]).all())
    assert bool((holes.all(dim=1) == holes.any(dim=1)).all())                           # a hole is a hole in every channel (WP:154)
    assert float(holes.float().mean()) < 0.5


@pytest.mark.parametrize("cin,pro", [(64, True), (64, False), (128, False)])
def test_full_size_3x3_producer_consumer_kernel_against_the_wave_private_kernel(cin, pro):
    """The 64-channel-block 3x3 layers of the benchmark shape (16 x 440 x 1024: every persistent workgroup of conv3x3_pc_kernel walks 56 tiles,
    chunk stream across tile and sample seams, the last tile row half empty) against conv3x3_wp_kernel<2,2> (OFD_CONV_PC=0) on the same
    inputs: two kernels, one operation -- values within the per-op tolerance of each other, GroupNorm partial sums equal per (sample, group)
    to 1e-3, and the sums equal to those of the values as stored."""
    import ctypes, math, os
    from opticalflowdiffusion_amd import _lib as L
    lib = L.lib()
    B, Cout = 16, 64
    g = torch.Generator(device="cuda").manual_seed(5 + cin)
    x = torch.randn(B, H, W, cin, device="cuda", generator=g).to(torch.bfloat16)
    w = torch.randn(Cout, cin, 3, 3, device="cuda", generator=g) / math.sqrt(cin * 9)
    wp = torch.empty(lib.ofd_conv_weight_elems(Cout, cin, 3), dtype=torch.bfloat16, device="cuda")
    L.check(lib.ofd_conv_weight_prep(L.ptr(w), L.ptr(wp), Cout, cin, cin, 3, -1.0, 0, L.stream()))
    bias = torch.randn(Cout, device="cuda", generator=g) * 0.1
    sc = torch.rand(B, cin, device="cuda", generator=g) + 0.5
    sh = torch.randn(B, cin, device="cuda", generator=g) * 0.3
    outs = {}
    old = os.environ.get("OFD_CONV_PC")
    try:
        for pc in ("1", "0"):
            os.environ["OFD_CONV_PC"] = pc
            out = torch.empty(B, H, W, Cout, dtype=torch.bfloat16, device="cuda")
            gn = torch.full((lib.ofd_conv_gn_partial_count(B, H, W, Cout),), float("nan"), device="cuda")
            a = L.ConvArgs()
            a.B, a.H, a.W, a.ksize, a.n_src, a.Cout = B, H, W, 3, 1, Cout
            a.src[0].src = x.data_ptr(); a.src[0].channels = cin; a.src[0].src_channels = cin
            a.weight = wp.data_ptr(); a.bias = bias.data_ptr(); a.out = out.data_ptr(); a.gn_partial = gn.data_ptr()
            if pro:
                a.in_scale = sc.data_ptr(); a.in_shift = sh.data_ptr()
            L.check(lib.ofd_conv_forward(ctypes.byref(a), L.stream()))
            torch.cuda.synchronize()
            outs[pc] = (out.float(), gn)
    finally:
        if old is None:
            os.environ.pop("OFD_CONV_PC", None)
        else:
            os.environ["OFD_CONV_PC"] = old
    o1, o0 = outs["1"][0], outs["0"][0]
    assert bool(torch.isfinite(o1).all())
    assert float((o1 - o0).norm() / o0.norm()) < 2e-3
    tiles = math.ceil(H / 8) * math.ceil(W / 32)
    p1 = outs["1"][1].reshape(B, 8, tiles * 4, Cout // 64, 2).sum(2).reshape(B, Cout // 8, 2)     # (gn_partial_index, csrc/conv_params.h)
    p0 = outs["0"][1].reshape(B, 8, tiles * 4, Cout // 64, 2).sum(2).reshape(B, Cout // 8, 2)
    assert bool(torch.isfinite(p1).all())
    assert torch.allclose(p1, p0, rtol=1e-3, atol=2.0)
    oc = o1.reshape(B, H, W, Cout // 8, 8)
    assert torch.allclose(p1[..., 0], oc.sum(dim=(1, 2, 4)), rtol=1e-4, atol=1.0)
    assert torch.allclose(p1[..., 1], (oc * oc).sum(dim=(1, 2, 4)), rtol=1e-4, atol=1.0)
