"""Pins oracle/diffusion_ref.py to vectors captured from the reference's ConditionalDiffusion."""
import pytest
import torch

from conftest import load_golden, rel_l2
from oracle import diffusion_ref as D
from oracle import unet_ref as R


@pytest.mark.parametrize("T", [4, 1000])
def test_schedule_buffers_bit_exact(T):
    g = load_golden(f"schedule_T{T}")
    S = D.make_schedule(T)
    assert set(D.BUFFER_NAMES) == set(g)
    for k in D.BUFFER_NAMES:
        assert torch.equal(S[k], g[k]), k


def test_schedule_known_answers():
    S = D.make_schedule(1000)
    assert float(S["betas"][0]) == pytest.approx(3.00279e-4, rel=1e-5)
    assert float(S["betas"][499]) == pytest.approx(3.30342e-3, rel=1e-5)
    assert float(S["betas"][999]) == pytest.approx(0.999)
    assert float(S["alphas_cumprod"][499]) == pytest.approx(0.5, rel=1e-5)


def _model(P):
    def fn(x, cond, t):
        return R.unet_forward(P, x, cond, t, mode="fp32")
    return fn


def test_q_sample_and_p_sample_steps():
    g = load_golden("diffusion_T4")
    S = D.make_schedule(4)
    P = R.closed_form_params(R.unet_param_shapes(64, 5, 2))
    model = _model(P)
    x_t = D.q_sample(S, g["x0"], g["t"], g["noise"])
    assert torch.equal(x_t, g["q_sample"])
    with torch.no_grad():
        out = model(x_t, g["cond"], g["t"])
    assert rel_l2(out, g["pred_x_start"]) < 1e-5            # pred_x0, no clip (DD:653-656)
    assert rel_l2(D.predict_noise_from_start(S, x_t, g["t"], g["pred_x_start"]), g["pred_noise"]) < 1e-6
    for ti in (2, 0):
        t = torch.full((2,), ti, dtype=torch.long)
        with torch.no_grad():
            out = model(x_t, g["cond"], t)
        img, xs = D.p_sample_update(S, x_t, ti, out, g[f"p_sample_t{ti}.z"])
        assert rel_l2(xs, g[f"p_sample_t{ti}.x_start"]) < 1e-5
        assert rel_l2(img, g[f"p_sample_t{ti}.img"]) < 1e-5


def test_p_sample_loop_and_ddim_trajectories():
    g = load_golden("diffusion_T4")
    S = D.make_schedule(4)
    P = R.closed_form_params(R.unet_param_shapes(64, 5, 2))
    model = _model(P)
    cond = g["cond"]
    # DDPM: DD:700-729 -- same generator consumption order as the reference
    torch.manual_seed(int(g["p_sample_loop.seed"]))
    img = torch.randn(2, 2, 16, 16)
    traj = [img]
    for t in reversed(range(4)):
        tt = torch.full((2,), t, dtype=torch.long)
        with torch.no_grad():
            out = model(img, cond, tt)
        z = torch.randn_like(img) if t > 0 else None
        img, _ = D.p_sample_update(S, img, t, out, z)
        traj.append(img)
    assert rel_l2(torch.stack(traj, 1), g["p_sample_loop.traj"]) < 1e-4
    # DDIM: DD:731-774
    torch.manual_seed(int(g["ddim.seed"]))
    img = torch.randn(2, 2, 16, 16)
    traj = [img]
    for time, time_next in D.ddim_times(4, 2):
        tt = torch.full((2,), time, dtype=torch.long)
        with torch.no_grad():
            out = model(img, cond, tt)
        z = torch.randn_like(img) if time_next >= 0 else None
        img, _ = D.ddim_update(S, img, time, time_next, out, z)
        traj.append(img)
    assert rel_l2(torch.stack(traj, 1), g["ddim.traj"]) < 1e-4


def test_p_losses_value_and_gradients():
    g = load_golden("diffusion_T4")
    S = D.make_schedule(4)
    P = R.closed_form_params(R.unet_param_shapes(64, 5, 2))
    for v in P.values():
        v.requires_grad_(True)
    loss = D.p_losses_flow(S, _model(P), g["x0"], g["cond"], g["p_losses.t"], g["p_losses.noise"])
    assert float(loss) == pytest.approx(float(g["p_losses.loss"]), rel=1e-5)
    loss.backward()
    assert rel_l2(P["final_conv.weight"].grad, g["p_losses.grad_final_conv_w"]) < 1e-4
    assert float(P["init_conv.weight"].grad.norm()) == pytest.approx(float(g["p_losses.grad_init_conv_w_norm"]), rel=1e-3)
    assert float(P["mid_attn.fn.fn.to_qkv.weight"].grad.norm()) == pytest.approx(float(g["p_losses.grad_mid_qkv_norm"]), rel=1e-3)
