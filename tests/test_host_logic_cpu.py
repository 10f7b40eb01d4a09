"""Host-side logic of the plugin surface that does not need a GPU."""
import types

import pytest
import torch

from oracle import diffusion_ref as D


class _DummyModel(torch.nn.Module):
    self_condition = False


@pytest.mark.parametrize("T", [4, 1000])
def test_conditional_diffusion_buffers_match_oracle(T):
    from opticalflowdiffusion_amd import ConditionalDiffusion
    cd = ConditionalDiffusion(_DummyModel(), (40, 64), objective="pred_x0", channels=2, auto_normalize=False,
                              timesteps=T, min_snr_loss_weight=True)
    S = D.make_schedule(T)
    sd = cd.state_dict()
    assert set(D.BUFFER_NAMES) <= set(sd)
    for k in D.BUFFER_NAMES:
        assert torch.equal(sd[k], S[k]), k
    assert cd._hw() == (40, 64) and not cd.is_ddim_sampling
    cd2 = ConditionalDiffusion(_DummyModel(), 32, objective="pred_x0", channels=2, auto_normalize=False,
                               timesteps=1000, sampling_timesteps=50)
    assert cd2.is_ddim_sampling and cd2._hw() == (32, 32)


def test_unsupported_configurations_raise():
    from opticalflowdiffusion_amd import ConditionalDiffusion, Unet
    with pytest.raises(NotImplementedError):
        ConditionalDiffusion(_DummyModel(), 32, objective="pred_v")
    with pytest.raises(NotImplementedError):
        ConditionalDiffusion(_DummyModel(), 32, objective="pred_x0", auto_normalize=False, noise_space="flow")
    with pytest.raises(NotImplementedError):
        Unet(32, channels=5)
    with pytest.raises(NotImplementedError):
        Unet(64, channels=5, self_condition=True)


def test_cfg_adapter_accepts_dict_namespace_and_defaults():
    from opticalflowdiffusion_amd.flow_diffuser import _Cfg
    c = _Cfg({"target": "flow", "image_size": [440, 1024], "timesteps": 50})
    assert c.target == "flow" and c.flow_max == 20 and c.image_size == [440, 1024] and "lr" in c
    ns = types.SimpleNamespace(target="joint", flow_max=10, zero_init=False)
    c = _Cfg(ns)
    assert c.target == "joint" and c.flow_max == 10 and c.zero_init is False and c.timesteps == 1000
    with pytest.raises(AttributeError):
        c.nonexistent


def test_ops_refuse_cpu_tensors():
    import opticalflowdiffusion_amd as m
    from opticalflowdiffusion_amd._lib import OfdError
    img, flow = torch.rand(1, 3, 8, 8), torch.zeros(1, 2, 8, 8)
    with pytest.raises(OfdError):
        m.warp(img, None, flow, mode="forward")
    with pytest.raises(OfdError):
        m.warp(None, img, flow, mode="backward")
    with pytest.raises(NotImplementedError):
        m.warp(img, None, flow, rep="filter", mode="forward")
    with pytest.raises(AssertionError):
        m.softsplat(img, flow, None, "linear")          # linear needs a metric (SS:285-286)


def test_batch_sharding_and_rates():
    from opticalflowdiffusion_amd import parallel as P
    for gb, world in ((128, 8), (8, 8), (10, 4), (3, 4)):
        spans = [P.shard_batch(gb, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == gb
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1
    assert len({P.rank_seed(0, r) for r in range(8)}) == 8
    assert P.whole_job_rate(10, 8, 2.0) == 40.0


def test_photometric_loss_helpers_match_reference_goldens():
    """WP:273-303 (host-side torch functions of the FlowLearner loss): values captured from the reference's warp module."""
    from conftest import load_golden
    from opticalflowdiffusion_amd.warp import charbonnier, edgeaware_smoothness1, fill_holes_nan, nan_charbonnier
    g = load_golden("loss_helpers")
    c = charbonnier(g["a"])
    assert torch.equal(torch.isnan(c), torch.isnan(g["charbonnier"])) and torch.allclose(torch.nan_to_num(c), torch.nan_to_num(g["charbonnier"]), rtol=1e-6)
    assert float(nan_charbonnier(g["a"], g["b"])) == pytest.approx(float(g["nan_charbonnier"]), rel=1e-6)
    f = fill_holes_nan(g["img"], g["w"])
    assert torch.equal(torch.isnan(f), torch.isnan(g["fill_holes_nan"])) and torch.equal(torch.nan_to_num(f), torch.nan_to_num(g["fill_holes_nan"]))
    assert float(edgeaware_smoothness1(g["img"], g["flow"])) == pytest.approx(float(g["edgeaware_smoothness1"]), rel=1e-6)


def test_flow_to_image_colour_wheel():
    """visualization.flow_to_image (stand-in for torchvision.utils.flow_to_image, FD:289): zero flow is white, the hue follows the
    direction, the saturation the magnitude relative to the batch maximum, opposite directions get complementary hues."""
    from opticalflowdiffusion_amd.visualization import flow_to_image
    z = flow_to_image(torch.zeros(1, 2, 3, 4))
    assert z.dtype == torch.uint8 and z.shape == (1, 3, 3, 4) and bool((z == 255).all())
    f = torch.zeros(4, 2, 1, 1)
    f[0, 0], f[1, 0], f[2, 1], f[3, 1] = 1.0, -1.0, 1.0, -1.0            # +x, -x, +y, -y at full magnitude
    im = flow_to_image(f)[:, :, 0, 0].int()
    assert im[0].tolist() == [255, 0, 0]                                 # +x: red (start of the wheel)
    assert im[1, 0] == 0 and im[1, 1] > 200 and im[1, 2] > 200           # -x: cyan
    assert im[2, 0] > 200 and im[2, 1] > 200 and im[2, 2] == 0           # +y: yellow
    assert im[3, 0] < 160 and im[3, 1] == 0 and im[3, 2] == 255          # -y: blue-violet
    half = torch.zeros(2, 2, 1, 1)
    half[0, 0], half[1, 0] = 1.0, 0.5                                    # same hue, half the magnitude -> half-way to white
    h = flow_to_image(half)[:, :, 0, 0].int()
    assert h[1].tolist() == [255, 127, 127]
    assert flow_to_image(torch.randn(2, 5, 7)).shape == (3, 5, 7)
    with pytest.raises(ValueError):
        flow_to_image(torch.zeros(1, 3, 4, 4))
