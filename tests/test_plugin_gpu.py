"""GPU tests of the plugin surface (FlowDiffuser / ConditionalDiffusion / UnetWithWarp) against
oracle loops driven with the same injected noise, plus large-shape runs checked by properties."""
import types

import pytest
import torch

from conftest import rel_l2
from oracle import diffusion_ref as D
from oracle import unet_ref as R
from oracle import warp_ref as WR
from test_unet_gpu import default_init_params, make_unet

pytestmark = pytest.mark.gpu


def _oracle_model(P):
    def fn(x, cond, t):
        return R.unet_forward(P, x, cond, t, mode="bf16c")
    return fn


def test_ddpm_and_ddim_loops_follow_the_oracle():
    """DD:700-729 and DD:731-774 with T=6: every step's network call and fused update on the GPU,
    the oracle loop fed with the SAME noise tensors (torch RNG is plumbing, not under test)."""
    from opticalflowdiffusion_amd import ConditionalDiffusion
    torch.manual_seed(0)
    P = default_init_params(5, seed=3)
    B, H, W, T = 2, 32, 48, 6
    unet = make_unet(5, P)
    cond = torch.rand(B, 3, H, W) * 2 - 1
    S = D.make_schedule(T)
    model = _oracle_model(P)

    diff = ConditionalDiffusion(unet, (H, W), objective="pred_x0", channels=2, auto_normalize=False, timesteps=T,
                                min_snr_loss_weight=True).cuda()
    img = torch.randn(B, 2, H, W)
    ref, got = img.clone(), img.cuda()
    for t in reversed(range(T)):
        z = torch.randn(B, 2, H, W)
        with torch.no_grad():
            out = model(ref, cond, torch.full((B,), t))
            ref, _ = D.p_sample_update(S, ref, t, out, z)
            got, _, _ = diff.p_sample(got, t, None, external_cond=cond.cuda(), noise=z.cuda())
        assert rel_l2(got.cpu(), ref) < 3e-2, t
    # DDIM, 3 sampling steps, eta = 0 (deterministic given x_T)
    diff2 = ConditionalDiffusion(unet, (H, W), objective="pred_x0", channels=2, auto_normalize=False, timesteps=T,
                                 sampling_timesteps=3, min_snr_loss_weight=True).cuda()
    torch.manual_seed(11)
    x_T = torch.randn(B, 2, H, W, device="cuda")
    torch.manual_seed(11)
    traj = diff2.sample(batch_size=B, return_all_timesteps=True, external_cond=cond.cuda())     # sample() reaches DDIM
    assert traj.shape == (B, 4, 2, H, W) and torch.equal(traj[:, 0], x_T)
    ref = x_T.cpu()
    for time, time_next in D.ddim_times(T, 3):
        with torch.no_grad():
            out = model(ref, cond, torch.full((B,), time))
        ref, _ = D.ddim_update(S, ref, time, time_next, out, torch.zeros_like(ref))
    assert rel_l2(traj[:, -1].cpu(), ref) < 3e-2


def test_flow_diffuser_sample_validation_and_joint_model():
    """FD:189-215 sample() (flow target: DDPM trajectory + forward-splat reconstruction),
    FD:237-281 validation loss, FD:20-63 UnetWithWarp forward for target='joint'."""
    from opticalflowdiffusion_amd import FlowDiffuser
    torch.manual_seed(1)
    H, W, B = 32, 40, 2
    cfg = dict(target="flow", image_size=[H, W], timesteps=4, flow_max=20, zero_init=True)
    fd = FlowDiffuser(cfg).cuda()
    img, tgt = torch.rand(B, 3, H, W).cuda(), torch.rand(B, 3, H, W).cuda()
    flow = ((torch.rand(B, 2, H, W) * 2 - 1) * 10).cuda()
    with torch.no_grad():
        tgt_, cond, flow_ = fd.preprocess((img, tgt, flow), aug=False)
        assert torch.equal(tgt_, flow_) and float(flow_.abs().max()) <= 1.0          # FD:141,163
        samples, traj = fd.sample(cond, flow_)
        assert traj.shape == (B, 5, 2, H, W) and samples.shape == (B, 3, H, W)
        ref = WR.warp(cond.cpu(), None, traj[:, -1].cpu(), mode="forward")             # FD:201-202
        ok = ~torch.isnan(ref)
        assert torch.equal(torch.isnan(samples.cpu()), ~ok) and float((samples.cpu()[ok] - ref[ok]).abs().max()) < 1e-5
        loss = fd.validation_step((img, tgt, flow), 0)
        assert torch.isfinite(loss) and "val/loss" in fd.logged
    loss_t = fd.training_step((img, tgt, flow), 0)         # FD:218-235: differentiable through the HIP executor
    assert loss_t.requires_grad and torch.isfinite(loss_t)

    cfgj = dict(target="joint", image_size=[H, W], timesteps=4, flow_max=20, zero_init=False)
    fj = FlowDiffuser(cfgj).cuda()
    x = torch.randn(B, 5, H, W).cuda()
    x[0, 1, 3, 4] = float("nan")
    t = torch.tensor([1, 3]).cuda()
    with torch.no_grad():
        out = fj._model(x, cond, t)                       # UnetWithWarp: cat(warped cond, flow)
        assert out.shape == (B, 5, H, W)
        xin = x.clone()
        nan_mask = torch.isnan(xin).any(dim=1, keepdim=True).float()
        xin[torch.isnan(xin)] = 0.0
        flow_pred = fj.unet(torch.cat((xin, nan_mask), dim=1), cond, t)
        assert torch.equal(out[:, 3:], flow_pred)
        ref = WR.warp(cond.cpu(), None, (flow_pred * 20).cpu(), mode="forward")
        ok = ~torch.isnan(ref)
        assert float((out[:, :3].cpu()[ok] - ref[ok]).abs().max()) < 1e-4


@pytest.mark.parametrize("B,H,W", [(1, 1080, 1920), (2, 440, 1024)])
def test_large_shapes_properties(B, H, W):
    """BASELINE configs[4] (1080p, mid attention over 32 400 tokens) and configs[1] size: finite
    output, batch-permutation equivariance (samples are independent: what data parallelism
    relies on), and translation of the result by 32 px when input+cond are rolled by 32 px is NOT
    expected (zero padding) -- so only the interior statistics are compared."""
    torch.manual_seed(2)
    P = default_init_params(5, seed=5)
    unet = make_unet(5, P)
    x = torch.randn(B, 2, H, W, device="cuda")
    cond = torch.rand(B, 3, H, W, device="cuda") * 2 - 1
    t = torch.randint(0, 1000, (B,), device="cuda")
    with torch.no_grad():
        y = unet(x, cond, t)
        assert y.shape == (B, 2, H, W) and torch.isfinite(y).all()
        if B > 1:
            y2 = unet(x.flip(0), cond.flip(0), t.flip(0))
            assert torch.equal(y2.flip(0), y)            # deterministic kernels: bit-identical per sample
        y3 = unet(x, cond, t)
        assert torch.equal(y3, y)                        # run-to-run reproducible (no atomics in the forward)


def test_fused_adam_matches_torch_adam_with_clipping():
    """FD:131-134 optimiser + exp_base.py:205 clipping: same trajectory as
    clip_grad_norm_ + torch.optim.Adam(weight_decay) over several steps."""
    from opticalflowdiffusion_amd.optim import FusedAdam
    torch.manual_seed(0)
    shapes = [(64, 64, 3, 3), (128,), (300, 257), (1, 64, 1, 1), (70001,)]
    ref_p = [torch.nn.Parameter(torch.randn(s, device="cuda")) for s in shapes]
    my_p = [torch.nn.Parameter(p.detach().clone()) for p in ref_p]
    ref = torch.optim.Adam(ref_p, lr=1e-3, weight_decay=1e-2)
    mine = FusedAdam(my_p, lr=1e-3, weight_decay=1e-2, max_grad_norm=5.0)
    for it in range(4):
        for a, b in zip(ref_p, my_p):
            g = torch.randn_like(a) * (3.0 if it % 2 == 0 else 0.01)
            a.grad = g.clone()
            b.grad = g.clone()
        total = torch.nn.utils.clip_grad_norm_(ref_p, 5.0)
        ref.step()
        mine.step()
        assert float(mine.last_grad_norm) == pytest.approx(float(total), rel=1e-5)
        for a, b in zip(ref_p, my_p):
            assert rel_l2(b.detach().cpu(), a.detach().cpu()) < 1e-6, it
    sd = mine.state_dict()
    assert set(sd["state"][0]) >= {"step", "exp_avg", "exp_avg_sq"}


def test_flow_diffuser_training_steps_reduce_the_loss():
    """FD:218-235 + FD:131-134: training_step -> backward through the HIP executor -> FusedAdam.step, the
    loop pl.Trainer.fit drives (exp_base.py:209-214).  Overfits one synthetic batch: the loss must fall."""
    from opticalflowdiffusion_amd import FlowDiffuser
    torch.manual_seed(0)
    H, W, B = 32, 64, 4
    fd = FlowDiffuser(dict(target="flow", image_size=[H, W], timesteps=50, flow_max=20, zero_init=False, lr=2e-4, weight_decay=0.0)).cuda()
    fd.log_dict = lambda *a, **k: None
    opt = fd.configure_optimizers()
    img = torch.rand(B, 3, H, W, device="cuda")
    flow = torch.clamp(torch.randn(B, 2, H, W, device="cuda") * 8, -20, 20)
    losses = []
    for it in range(16):
        torch.manual_seed(100)                              # same t and noise each step: a clean overfitting signal
        loss = fd.training_step((img, img, flow), it)
        assert loss.requires_grad and torch.isfinite(loss)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert losses[-1] < 0.7 * losses[0] and max(losses) < 1.5 * losses[0], losses
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in fd.unet.parameters())


def test_flow_diffuser_training_is_bit_reproducible_in_deterministic_mode():
    """two training runs from the same seed -- GPU augmentation (fixed-point gray means), q_sample, UNet training forward, nan_mse (ordered
    partial sums), backward in deterministic mode (csrc/det.h: fixed-point gradient accumulation), gradient-norm clip (ordered partial sums)
    and Adam -- give the same losses and the same parameters after six steps, BIT for bit; no accumulation missed its shadow."""
    from opticalflowdiffusion_amd import FlowDiffuser
    H, W, B = 40, 72, 3

    def run():
        torch.manual_seed(0)
        fd = FlowDiffuser(dict(target="flow", image_size=[H, W], timesteps=50, flow_max=20, zero_init=False, lr=2e-4, weight_decay=1e-4,
                               gradient_clip_val=0.5)).cuda()
        fd.log_dict = lambda *a, **k: None
        fd.unet.set_deterministic(True)
        opt = fd.configure_optimizers()
        g = torch.Generator(device="cuda").manual_seed(5)
        img = torch.rand(B, 3, H, W, device="cuda", generator=g)
        tgt = torch.rand(B, 3, H, W, device="cuda", generator=g)
        flow = torch.clamp(torch.randn(B, 2, H, W, device="cuda", generator=g) * 8, -20, 20)
        losses = []
        for it in range(6):
            loss = fd.training_step((img, tgt, flow), it)
            opt.zero_grad()
            loss.backward()
            opt.step()
            losses.append(loss.detach().clone())
        torch.cuda.synchronize()
        assert fd.unet.deterministic_misses() == 0
        return torch.stack(losses), torch.cat([p.detach().flatten() for p in fd.unet.parameters()]).clone()

    l1, p1 = run()
    l2, p2 = run()
    assert torch.isfinite(l1).all() and torch.isfinite(p1).all()
    assert torch.equal(l1, l2), (l1, l2)
    assert torch.equal(p1, p2), f"{int((p1 != p2).sum())} of {p1.numel()} parameters differ after six steps"


def test_training_step_logged_statistics_and_the_augmentation_table():
    """`ofd_batch_stats` (the eight scalars FD:218-235 logs, one pass per tensor) against torch.min / max / mean / mean(std(x, dim=0)), and
    `ofd_augment_table` (Augmentor.draw on the device: one launch) against the tensor-op form of the same table from the same uniforms."""
    from opticalflowdiffusion_amd import FlowDiffuser
    from opticalflowdiffusion_amd.augmentation import Augmentor
    torch.manual_seed(3)
    for shape in [(4, 3, 40, 72), (3, 2, 17, 33), (2, 1, 300, 500)]:
        x = torch.randn(*shape, device="cuda") * 3 + 0.5
        mn, mx, mean, sd = FlowDiffuser._batch_stats(x)
        assert float(mn) == float(torch.min(x)) and float(mx) == float(torch.max(x))
        assert float(mean) == pytest.approx(float(torch.mean(x)), rel=1e-5, abs=1e-6)
        assert float(sd) == pytest.approx(float(torch.mean(torch.std(x, dim=0))), rel=1e-5)
    aug = Augmentor(seed=11)
    B = 257
    P = aug.draw(B, torch.device("cuda"))                   # the kernel
    aug2 = Augmentor(seed=11)
    u = aug2._rand((B, 14), torch.device("cuda"))           # the same uniforms (same seed, same first draw)
    aug2._rand = lambda shape, device: u.cpu()              # ... through the tensor-op form on the CPU
    P_ref = aug2.draw(B, torch.device("cpu"))
    assert P.shape == (B, 16) and torch.equal(P[:, [0, 4, 5, 7, 8, 9, 14, 15]].cpu(), P_ref[:, [0, 4, 5, 7, 8, 9, 14, 15]])
    assert torch.allclose(P.cpu(), P_ref, rtol=1e-6, atol=1e-7)


def test_trajectory_stride_keeps_strided_frames_only():
    """optional `trajectory_stride`: x_T, every k-th step and the final sample; the kept frames equal those of the full trajectory"""
    from opticalflowdiffusion_amd import FlowDiffuser
    H, W, B = 16, 24, 2
    fd = FlowDiffuser(dict(target="flow", image_size=[H, W], timesteps=10, flow_max=20, zero_init=False)).cuda()
    cond = torch.rand(B, 3, H, W, device="cuda") * 2 - 1
    torch.manual_seed(5)
    full = fd.model.sample(batch_size=B, external_cond=cond, return_all_timesteps=True)
    fd.model.trajectory_stride = 4
    torch.manual_seed(5)
    part = fd.model.sample(batch_size=B, external_cond=cond, return_all_timesteps=True)
    assert full.shape[1] == 11 and part.shape[1] == 4                       # x_T, after steps 4 and 8, final
    assert torch.equal(part, full[:, [0, 4, 8, 10]])
    fd2 = FlowDiffuser(dict(target="flow", image_size=[H, W], timesteps=10, trajectory_stride=5)).cuda()
    assert fd2.model.trajectory_stride == 5


def test_flow_diffuser_regression_mode_is_diffusion_false():
    """FD:106-111 / 128-129 / 176-186 / 204-213 with is_diffusion=False, target=flow: Unet(64, channels=3, out_dim=2, time_in=False)
    regresses the flow from the condition image, loss = mse; training reduces it; sample = forward splat of the condition."""
    from opticalflowdiffusion_amd import FlowDiffuser, Unet, warp
    torch.manual_seed(0)
    H, W, B = 32, 64, 4
    fd = FlowDiffuser(dict(target="flow", is_diffusion=False, image_size=[H, W], flow_max=20, zero_init=False, lr=2e-4, weight_decay=0.0)).cuda()
    assert isinstance(fd.model, Unet) and fd.model is fd.unet and not fd.unet.time_in and fd.unet.channels == 3
    assert not any(n.startswith("time_mlp") or ".mlp." in n for n, _ in fd.named_parameters())
    fd.log_dict = lambda *a, **k: None
    opt = fd.configure_optimizers()
    img = torch.rand(B, 3, H, W, device="cuda")
    flow = torch.clamp(torch.randn(B, 2, H, W, device="cuda") * 8, -20, 20)
    tgt_, cond, flow_ = fd.preprocess((img, img, flow), aug=False)
    with torch.no_grad():
        direct = torch.nn.functional.mse_loss(fd.unet(cond), flow_)
        assert float(fd.loss(tgt_, cond, flow_)) == pytest.approx(float(direct), rel=1e-6)
        assert float(fd.loss(tgt_, cond, flow_, override=flow_)) == 0.0
    losses = []
    for it in range(12):
        loss = fd.training_step((img, img, flow), it)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert losses[-1] < 0.8 * losses[0], losses
    with torch.no_grad():
        samples, f = fd.sample(cond, flow_)
        assert f.shape == (B, 2, H, W) and samples.shape == (B, 3, H, W)
        again = warp(cond[:, :3], None, f, mode="forward")
        assert torch.equal(torch.isnan(samples), torch.isnan(again)) and torch.allclose(torch.nan_to_num(samples), torch.nan_to_num(again))
        fd.validation_step((img, img, flow), 0)


def test_backward_reports_every_gradient_range_once():
    """the ranges handed to the data-parallel hook cover the flat gradient buffer exactly once, last layers first"""
    from opticalflowdiffusion_amd import Unet
    from opticalflowdiffusion_amd.warp import nan_mse
    torch.manual_seed(0)
    net = Unet(64, channels=5, out_dim=2).cuda()

    class Recorder:
        def __init__(self):
            self.ranges = []
        def begin(self, flat):
            self.ranges = []
        def on_range(self, flat, b, e):
            self.ranges.append((b, e))
        def finish(self, flat):
            pass

    rec = net.grad_sync = Recorder()
    out = net(torch.randn(1, 2, 16, 16, device="cuda"), external_cond=torch.randn(1, 3, 16, 16, device="cuda"), time=torch.tensor([3], device="cuda"))
    nan_mse(out, torch.zeros_like(out)).backward()
    n = net.flat_grads().numel()
    pos = 0
    for b, e in sorted(rec.ranges):
        assert b == pos, (b, pos)
        pos = e
    assert pos == n
    offs = dict(zip(net._names, net._goffsets))
    assert rec.ranges[0][0] == offs["final_conv.weight"]            # backward order: the head first ...
    assert rec.ranges[-1][0] == offs["time_mlp.1.weight"]           # ... the time MLP (fed by every block) last
    with pytest.raises(Exception, match="older forward|tape"):
        nan_mse(out, torch.zeros_like(out)).backward()              # the tape is single-use


def test_joint_pyramid_loss_value_and_gradient():
    """DD:893-983 (the shipped config is target='joint', flow_diffuser.yaml:15): loss value against the oracle
    pyramid (CPU C splat), and its gradient w.r.t. the predicted flow -- through the splat backward kernel at
    five scales -- against central differences of the same loss."""
    from opticalflowdiffusion_amd import FlowDiffuser
    torch.manual_seed(0)
    B, H, W = 2, 32, 64
    fd = FlowDiffuser(dict(target="joint", image_size=[H, W], timesteps=8, flow_max=20, zero_init=False)).cuda()
    cond = torch.rand(B, 3, H, W) * 2 - 1
    smooth = lambda t: torch.nn.functional.avg_pool2d(t, 9, 1, 4)
    flow_gt = smooth(torch.randn(B, 2, H, W) * 2).clamp(-1, 1)                 # normalised flow (x flow_max = px)
    flow_out = (flow_gt + smooth(torch.randn(B, 2, H, W)) * 0.1)
    target = WR.warp(cond, None, flow_gt * 20, mode="forward")                # x_start[:, :3] of preprocess (FD:160)
    fo = flow_out.cuda().requires_grad_(True)
    image_out = fd._model._warp(cond.cuda(), fo)                              # FD:50: what UnetWithWarp returns
    loss = fd.model._loss(image_out, target.cuda(), None, flow_gt.cuda(), cond.cuda(), fo, 0.0)
    ref = D.pyramid_loss(WR.warp(cond, None, flow_out * 20, mode="forward"), target, cond, flow_out, 20.0)
    assert abs(float(loss) - float(ref)) < 1e-4 * abs(float(ref)), (float(loss), float(ref))
    loss.backward()
    g = fo.grad.clone()
    assert torch.isfinite(g).all() and float(g.abs().max()) > 0
    # the reference's backward IS its softsplat_flowgrad kernel (SS:600-700, not the exact derivative at scale > 1:
    # DESIGN "frozen / crossed factors"), so the gradient is checked against the oracle's restatement of that kernel
    # chained by hand: d loss / d splat-output on the valid pixels of every level -> flowgrad -> * flow_max
    flow_px = flow_out * 20
    ten_in = torch.cat((cond, torch.ones(B, 1, H, W)), 1)                       # linear_unn: [image * w, w], w = 1 (no NaNs in cond)
    gouts, n_valid = [], 0
    for level in (1, 2, 4, 8, 16):
        ret = WR.splat_out(ten_in, flow_px, level)
        img = torch.where(ret[:, -1:] > 0, ret[:, :-1], torch.full_like(ret[:, :-1], float("nan")))
        tgt = target if level == 1 else WR.warp(target, None, torch.zeros_like(flow_px), mode="forward", scale=level)
        ok = ~(torch.isnan(img) | torch.isnan(tgt))
        n_valid += int(ok.sum())
        gi = torch.where(ok, 2.0 * (torch.nan_to_num(img) - torch.nan_to_num(tgt)) * level ** 4, torch.zeros_like(img))
        gouts.append((level, torch.cat((gi, torch.zeros_like(ret[:, -1:])), 1)))
    want = sum(WR.splat_flowgrad(ten_in, flow_px, go / n_valid, scale=level) for level, go in gouts) * 20.0
    assert rel_l2(g.cpu(), want) < 1e-3, rel_l2(g.cpu(), want)


def test_joint_training_step_reaches_every_parameter():
    """FD:218-235 with the shipped target='joint': UnetWithWarp (NaN-safe 9-channel UNet + splat) -> pyramid loss ->
    backward through splat and UNet -> FusedAdam; the loss falls on a fixed batch."""
    from opticalflowdiffusion_amd import FlowDiffuser
    torch.manual_seed(0)
    B, H, W = 2, 32, 64
    fd = FlowDiffuser(dict(target="joint", image_size=[H, W], timesteps=50, flow_max=20, zero_init=False, lr=2e-4, weight_decay=0.0)).cuda()
    fd.log_dict = lambda *a, **k: None
    opt = fd.configure_optimizers()
    img = torch.rand(B, 3, H, W, device="cuda")
    flow = torch.clamp(torch.nn.functional.avg_pool2d(torch.randn(B, 2, H, W, device="cuda") * 30, 9, 1, 4), -20, 20)
    losses = []
    for it in range(10):
        torch.manual_seed(100)
        loss = fd.training_step((img, img, flow), it)
        assert torch.isfinite(loss)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in fd.unet.parameters())
    assert losses[-1] < losses[0], losses


def test_gradient_accumulation_and_zero_grad_in_place():
    """two backward passes without clearing add up (accumulate_grad_batches > 1, XB:203), and
    zero_grad(set_to_none=False) followed by a backward gives the plain gradient again"""
    from opticalflowdiffusion_amd import Unet
    from opticalflowdiffusion_amd.warp import nan_mse
    torch.manual_seed(0)
    net = Unet(64, channels=5, out_dim=2).cuda()
    x, c, t = torch.randn(1, 2, 16, 16, device="cuda"), torch.randn(1, 3, 16, 16, device="cuda"), torch.tensor([3], device="cuda")
    tgt = torch.zeros(1, 2, 16, 16, device="cuda")

    def backward():
        nan_mse(net(x, external_cond=c, time=t), tgt).backward()

    backward()
    g1 = [p.grad.clone() for p in net.parameters()]
    backward()
    for p, g in zip(net.parameters(), g1):
        assert torch.allclose(p.grad, 2 * g, rtol=1e-5, atol=1e-8)
    net.zero_grad(set_to_none=False)
    backward()
    for p, g in zip(net.parameters(), g1):
        assert torch.allclose(p.grad, g, rtol=1e-5, atol=1e-8)


def test_graph_replay_is_bit_identical_and_survives_weight_updates():
    """ofd_unet_set_graph: eager call, captured call and replays give the same bits; new weights (same buffers,
    re-prepared) are picked up by the existing graph; another shape gets its own graph."""
    from opticalflowdiffusion_amd import Unet
    torch.manual_seed(0)
    net = Unet(64, channels=5, out_dim=2).cuda()
    x, c, t = torch.randn(2, 2, 32, 48, device="cuda"), torch.randn(2, 3, 32, 48, device="cuda"), torch.tensor([3, 900], device="cuda")
    with torch.no_grad():
        ref = net(x, external_cond=c, time=t).clone()
        net.set_graph(True)
        outs = [net(x.clone(), external_cond=c.clone(), time=t.clone()).clone() for _ in range(4)]     # eager, capture, replay, replay
        assert all(torch.equal(o, ref) for o in outs)
        x2 = torch.randn_like(x)
        assert torch.equal(net(x2, external_cond=c, time=t), _eager(net, x2, c, t))
        for p in net.parameters():                       # an optimizer-style in-place update
            p.mul_(1.01)
        got = net(x, external_cond=c, time=t).clone()
        assert not torch.equal(got, ref) and torch.equal(got, _eager(net, x, c, t))
        xb = torch.randn(1, 2, 16, 16, device="cuda")
        cb, tb = torch.randn(1, 3, 16, 16, device="cuda"), torch.tensor([5], device="cuda")
        o1 = [net(xb, external_cond=cb, time=tb).clone() for _ in range(3)]
        assert torch.equal(o1[0], o1[2]) and torch.equal(o1[0], _eager(net, xb, cb, tb))
        net.set_graph(False)


def _eager(net, x, c, t):
    net.set_graph(False)
    out = net(x, external_cond=c, time=t).clone()
    net.set_graph(True)
    return out


@pytest.mark.parametrize("target", ["joint", "target"])
def test_validation_step_diagnostics(target):
    """FD:237-364 for the two warped targets: the reference's 19 `val/*` scalars (incl. `val/mse`, `val/ideal_loss` through the
    `override=` path FD:256-259), `val/last_step`, every image key it logs, the mid-trajectory strips (`[:, ::50]`) and the
    `grad_flow` image from `_loss` backward through the splat kernels (FD:351-364)."""
    from opticalflowdiffusion_amd import FlowDiffuser
    torch.manual_seed(3)
    H, W, B, T = 32, 48, 2, 100
    fd = FlowDiffuser(dict(target=target, image_size=[H, W], timesteps=T, flow_max=20, zero_init=False)).cuda()
    img, tgt = torch.rand(B, 3, H, W).cuda(), torch.rand(B, 3, H, W).cuda()
    flow = ((torch.rand(B, 2, H, W) * 2 - 1) * 6).cuda()
    loss = fd.validation_step((img, tgt, flow), 0)
    want = ["val/loss", "val/mse", "val/ideal_loss", "val/last_step"] + [f"val/{a}_{b}" for a in ("cond", "flow", "samples", "p_flow")
                                                                          for b in ("min", "max", "mean", "std")]
    assert len(want) == 20
    for k in want:
        assert k in fd.logged and bool(torch.isfinite(fd.logged[k])), k
    assert float(loss) == pytest.approx(float(fd.logged["val/loss"]))
    # the ground-truth flow pushed through the override path must beat the untrained net's own prediction on the pyramid loss
    assert float(fd.logged["val/ideal_loss"]) < float(fd.logged["val/loss"])
    imgs = fd.logged_images
    for k in ("original", "target", "diffusion_tgt", "original_warped", "gt_flow", "target_p", "concat", "difference", "samples",
              "mid_samples", "mid_flows", "last_step", "grad_flow"):
        assert k in imgs and len(imgs[k]) == B, k
    nstrip = len(range(0, T + 1, 50))                                   # samples[:, ::50] of a (B, T+1, ...) trajectory
    nflow = nstrip if target == "joint" else len(range(1, T + 1, 50))   # FD:249-250: the "target" mode slices a list that starts with None
    assert imgs["mid_samples"][0].shape == (1, 3, H, W * nstrip) and imgs["mid_flows"][0].shape == (1, 3, H, W * nflow)
    assert imgs["concat"][0].shape == (1, 3, H, 2 * W) and imgs["grad_flow"][0].shape == (1, 3, H, W)
    g = torch.cat(imgs["grad_flow"])
    assert bool(torch.isfinite(g).all()) and float(g.std()) > 0         # a non-trivial descent direction
    # gradient-norm statistics of a training step (FD:367-388)
    fd.training_step((img, tgt, flow), 0).backward()
    fd.log_grad_norm_stat()
    assert all(bool(torch.isfinite(fd.logged[f"train/grad_norm/{b}"])) for b in ("min", "max", "std", "mean", "median"))
    assert all(f"train/gpr/{b}" in fd.logged for b in ("min", "max", "std", "mean", "median"))      # (a zero-norm parameter makes the ratio inf, as in the reference)
    assert bool(torch.isfinite(fd.logged["train/gpr/min"])) and bool(torch.isfinite(fd.logged["train/gpr/median"]))


def test_fused_augmentation_kernel_against_its_tensor_op_restatement():
    """`ofd_augment` (one pass over the batch) against `Augmentor.apply_torch` on the SAME table of per-sample decisions: random
    draws on a non-square batch, then every transform forced on at once (jitter + grayscale + blur + both flips + crop), the
    reference's own flow arithmetic (swapped flip channels, multiplying crop scaling), and the wiring through FlowDiffuser.preprocess(aug=True)."""
    from opticalflowdiffusion_amd import FlowDiffuser
    from opticalflowdiffusion_amd.augmentation import Augmentor
    torch.manual_seed(4)
    B, H, W = 12, 40, 72
    img, tgt = torch.rand(B, 3, H, W, device="cuda"), torch.rand(B, 3, H, W, device="cuda")
    flow = torch.randn(B, 2, H, W, device="cuda") * 5
    for ref_flip in (False, True):
        a = Augmentor(reference_semantics=ref_flip)
        for trial in range(3):
            P = a.draw(B, img.device)
            if trial == 2:                                  # everything at once, for every sample
                P[:, 0] = P[:, 4] = P[:, 5] = P[:, 7] = P[:, 8] = P[:, 9] = 1.0
                P[:, 4] = (torch.arange(B, device="cuda") % 2).float()
                P[:, 12], P[:, 13], P[:, 10], P[:, 11] = 0.85, 0.93, 0.1, 0.05
            got = a.apply_hip(img, tgt, flow, P)
            want = a.apply_torch(img, tgt, flow, P)
            for g_, w_, name in zip(got, want, ("img", "tgt", "flow")):
                assert g_.shape == w_.shape and float((g_ - w_).abs().max()) < 2e-5 * max(1.0, float(w_.abs().max())), (name, trial, ref_flip)
    ident = torch.zeros(B, Augmentor.NP, device="cuda")
    ident[:, 1:4], ident[:, 6], ident[:, 12:14] = 1.0, 0.25, 1.0
    got = Augmentor().apply_hip(img, tgt, flow, ident)
    assert torch.equal(got[0], img) and torch.equal(got[1], tgt) and torch.equal(got[2], flow)      # no transform: a copy
    fd = FlowDiffuser(dict(target="flow", image_size=[H, W], timesteps=10, flow_max=20)).cuda()
    torch.manual_seed(9)
    t1 = fd.preprocess((img, tgt, flow), aug=True)
    torch.manual_seed(9)
    t2 = fd.preprocess((img, tgt, flow), aug=True)
    plain = fd.preprocess((img, tgt, flow), aug=False)
    assert torch.equal(t1[1], t2[1]) and not torch.equal(t1[1], plain[1]) and t1[0].shape == plain[0].shape
    fd_off = FlowDiffuser(dict(target="flow", image_size=[H, W], timesteps=10, flow_max=20, augment=False)).cuda()
    assert torch.equal(fd_off.preprocess((img, tgt, flow), aug=True)[1], plain[1])
