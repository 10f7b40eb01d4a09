"""FlowLearner (flow_learner.py): the photometric splat pyramid on the HIP kernels against the same formula
evaluated with the CPU oracle's splat, and the training loop."""
import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu


def oracle_pyramid_loss(input_img, flow_pred, warp_weights, tgt, levels):
    """FL:159-206 with oracle/warp_ref.softsplat (the C restatement of the reference kernels) and torch CPU ops."""
    from oracle import warp_ref as WR
    photo = []
    for level in levels:
        per = []
        for a in range(level):
            for b in range(level):
                sw = WR.softsplat(input_img, flow_pred, warp_weights, "soft", scale=level, offset=(a, b))
                w = sw[:, -1:].repeat(1, 3, 1, 1)
                filled = torch.where(w > 0, sw[:, :-1], torch.full_like(sw[:, :-1], float("nan")))
                dt = WR.softsplat(tgt, torch.zeros_like(flow_pred), torch.ones_like(warp_weights), "soft", scale=level, offset=(a, b))[:, :-1]
                p, t = filled.flatten(), dt.flatten()
                ok = ~(torch.isnan(p) | torch.isnan(t))
                per.append(torch.mean(torch.pow(torch.square(t[ok] - p[ok]) + 1e-6, 0.5)))
        photo.append(sum(per) / len(per))
    return sum(photo) / len(photo)


@pytest.mark.parametrize("B,H,W,levels", [(2, 24, 40, (1, 2, 4, 5)), (1, 33, 47, (1, 7, 8))])
def test_photometric_pyramid_loss_against_oracle(B, H, W, levels):
    from opticalflowdiffusion_amd.flow_learner import photometric_pyramid_loss
    torch.manual_seed(31)
    img = torch.rand(B, 3, H, W) * 2 - 1
    tgt = torch.rand(B, 3, H, W) * 2 - 1
    flow = (torch.rand(B, 2, H, W) * 2 - 1) * 6.0
    wts = torch.randn(B, 1, H, W) * 0.5
    ref = oracle_pyramid_loss(img, flow, wts, tgt, levels)
    got = photometric_pyramid_loss(img.cuda(), flow.cuda(), wts.cuda(), tgt.cuda(), levels)
    assert float(got) == pytest.approx(float(ref), rel=2e-5)


def test_flow_learner_training_reduces_the_loss_and_samples():
    from opticalflowdiffusion_amd import FlowLearner
    torch.manual_seed(0)
    B, H, W = 2, 32, 48
    fl = FlowLearner(dict(image_size=[H, W], flow_max=20, zero_init=False, lr=2e-4, weight_decay=0.0, levels=[1, 2, 4])).cuda()
    names = [n for n, _ in fl.named_parameters()]
    assert all(n.startswith("unet.model.") for n in names) and not any("time_mlp" in n or ".mlp." in n for n in names)
    assert fl.unet.model.channels == 6 and fl.unet.model.out_dim == 3
    fl.log_dict = lambda *a, **k: None
    fl.log = lambda *a, **k: None
    opt = fl.configure_optimizers()
    img = torch.rand(B, 3, H, W, device="cuda")
    true_flow = torch.zeros(B, 2, H, W, device="cuda")
    true_flow[:, 0] = 3.0
    from opticalflowdiffusion_amd import warp
    tgt = torch.nan_to_num(warp(img, None, true_flow, mode="forward"), nan=0.5)
    losses = []
    for it in range(10):
        loss = fl.training_step((img, tgt, true_flow), it)
        assert torch.isfinite(loss)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert losses[-1] < losses[0], losses
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in fl.parameters())
    with torch.no_grad():
        samples, flow, wts = fl.sample(torch.cat((2 * img - 1, 2 * tgt - 1), dim=1), true_flow)
        assert samples.shape == (B, 3, H, W) and flow.shape == (B, 2, H, W) and wts.shape == (B, 1, H, W)
        fl.logged = {}
        fl.log_dict = lambda d, **k: fl.logged.update(d)
        fl.validation_step((img, tgt, true_flow), 0)
        assert {"val/loss", "val/ideal_loss", "val/mse", "val/flow_mse"} <= set(fl.logged)
        # the ground-truth flow explains the pair better than the untrained prediction would at initialisation
        assert torch.isfinite(fl.logged["val/ideal_loss"])
