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


def _write_config_tree(root):
    """a `configurations/` tree with the reference's layout and keys (configurations/config.yaml, experiment/base.yaml,
    experiment/matrix_flow.yaml, algorithm/flow_diffuser.yaml, dataset/sintel.yaml)"""
    import os
    files = {
        "config.yaml": "defaults:\n  - experiment: matrix_flow\n  - dataset: sintel\n  - algorithm: pwc_learner\n\nwandb:\n  entity: e\n  project: p\n  mode: dryrun\n  resume: null\n",
        "experiment/base.yaml": "tasks: [train]\nepochs: -1\ntraining:\n  precision: 32\n  data:\n    batch_size: 64\n    num_workers: 16\n    shuffle: True\n"
                                "  optim:\n    accumulate_grad_batches: 1\n  checkpointing:\n    every_n_train_steps: 5000\n"
                                "validation:\n  check_interval: 400\n  check_epoch: 1\n  limit_batch: 1\n  data:\n    batch_size: 8\n",
        "experiment/matrix_flow.yaml": "defaults:\n  - base\n\nname: matrix_flow\n",
        "algorithm/pwc_learner.yaml": "name: pwc_learner\nlr: 1e-4\n",
        "algorithm/flow_diffuser.yaml": "name: flow_diffuser\n\nimage_size: 128\nlatent_dim: 16\nflow_max: 20\nlatent_max: 2\n\nlr: 1e-5\nflow_weight: 0.0\n"
                                        "weight_decay: 1e-6\nis_diffusion: true\nlatent: false\ntimesteps: 1000\n\ntarget: joint\nae: px8q8g0m\nnoiser: image\n\nzero_init: true\n",
        "dataset/sintel.yaml": "name: sintel\n\nimage_size: 512,256\n",
    }
    for rel, text in files.items():
        path = os.path.join(root, rel)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            f.write(text)


def test_compose_reads_a_reference_style_configuration_tree(tmp_path):
    """compat.compose = what @hydra.main(config_path="configurations", config_name="config") hands main.py:30, for the override
    grammar the reference is launched with; the result answers the probes the reference makes on its DictConfig."""
    from opticalflowdiffusion_amd.compat import Config, compose
    _write_config_tree(str(tmp_path))
    cfg = compose(str(tmp_path))
    assert cfg.algorithm.name == "pwc_learner" and cfg.experiment.name == "matrix_flow" and cfg.dataset.name == "sintel"   # config.yaml:1-4
    assert cfg.experiment.training.data.batch_size == 64 and cfg.experiment.tasks == ["train"]           # base.yaml through `defaults: [base]`
    assert cfg.wandb.mode == "dryrun" and cfg.wandb.get("resume", None) is None                           # main.py:46-47
    cfg = compose(str(tmp_path), ["algorithm=flow_diffuser", "algorithm.target=flow", "experiment.training.data.batch_size=16",
                                  "+experiment.training.clipping=100", "algorithm.lr=3e-5", "~algorithm.ae"])
    assert cfg.algorithm.name == "flow_diffuser" and cfg.algorithm.target == "flow" and cfg.algorithm.timesteps == 1000
    assert cfg.algorithm.lr == pytest.approx(3e-5) and cfg.algorithm.weight_decay == pytest.approx(1e-6) and "ae" not in cfg.algorithm
    assert cfg.experiment.training.data.batch_size == 16
    assert "clipping" in dir(cfg.experiment.training) and cfg.experiment.training.clipping == 100        # exp_base.py:191
    assert isinstance(cfg.experiment.training, Config) and {**cfg.experiment.training.checkpointing} == {"every_n_train_steps": 5000}   # main.py:20
    assert "zero_init" in cfg.algorithm and cfg.algorithm.zero_init is True                               # `in` probing, flow_learner.py:71-73
    with pytest.raises(FileNotFoundError):
        compose(str(tmp_path), ["algorithm=does_not_exist"])
    with pytest.raises(AttributeError):
        cfg.algorithm.no_such_key
    # the plugin accepts the composed node as its cfg (attribute and `in` access, FD:70-129)
    from opticalflowdiffusion_amd.flow_diffuser import _Cfg
    c = _Cfg(cfg.algorithm)
    assert c.target == "flow" and c.flow_max == 20 and "sampling_timesteps" in c and c.sampling_timesteps is None


def test_utils_shim_and_import_alias(tmp_path):
    """the `utils` package main.py:9 imports (absent from the reference repository) and the `algorithms.diffusion_animation` alias"""
    import importlib
    import os
    import sys
    shims = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "opticalflowdiffusion_amd", "compat", "shims")
    sys.path.insert(0, shims)
    try:
        for m in [k for k in sys.modules if k == "utils" or k.startswith("utils.") or k == "algorithms" or k.startswith("algorithms.")]:
            del sys.modules[m]
        wu = importlib.import_module("utils.wandb_utils")
        importlib.import_module("utils.video_prediction.visualization").log_video()
        with pytest.raises(RuntimeError):
            wu.download_latest_checkpoint("entity/project/run", tmp_path)
        ck = tmp_path / "a.ckpt"
        torch.save({"state_dict": {"unet.final_conv.weight": torch.ones(2), "betas": torch.zeros(3)}, "global_step": 7}, ck)
        out = torch.load(wu.rewrite_checkpoint_for_compatibility(str(ck)), weights_only=False)
        assert set(out["state_dict"]) == {"unet.final_conv.weight", "_model.final_conv.weight", "model.model.final_conv.weight", "betas"}
        assert out["global_step"] == 7
        with pytest.raises(FileExistsError):                      # an existing output is not silently replaced
            wu.rewrite_checkpoint_for_compatibility(str(ck))
        wu.rewrite_checkpoint_for_compatibility(str(ck), overwrite=True)
        import argparse                                           # a checkpoint that needs the full unpickler is refused by default
        bad = tmp_path / "b.ckpt"
        torch.save({"state_dict": {"unet.final_conv.weight": torch.ones(2)}, "hyper_parameters": argparse.Namespace(lr=1.0)}, bad)
        with pytest.raises(RuntimeError, match="weights_only"):
            wu.rewrite_checkpoint_for_compatibility(str(bad))
        assert wu.rewrite_checkpoint_for_compatibility(str(bad), allow_pickle=True).endswith("b.compat.ckpt")
        da = importlib.import_module("algorithms.diffusion_animation")
        import opticalflowdiffusion_amd as m
        assert da.FlowDiffuser is m.FlowDiffuser and da.Unet is m.Unet and da.ConditionalDiffusion is m.ConditionalDiffusion
    finally:
        sys.path.remove(shims)
        for m in [k for k in sys.modules if k == "utils" or k.startswith("utils.") or k == "algorithms" or k.startswith("algorithms.")]:
            del sys.modules[m]


def test_bench_self_launch_command(monkeypatch):
    """`python bench.py --gpus N` without a launcher re-runs itself under torch.distributed.run as a CHILD process (never an exec:
    the parent has not touched the GPU, and must not), rendezvous on 127.0.0.1, and returns the child's exit code."""
    import argparse
    import importlib
    import subprocess
    import sys
    bench = importlib.import_module("bench")
    seen = {}

    def fake_call(cmd, **kw):
        seen["cmd"] = cmd
        return 7

    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    rc = bench.self_launch(argparse.Namespace(gpus=4))
    cmd = seen["cmd"]
    assert rc == 7
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-4:] == ["--gpus", "4", "--steps", "3"] and cmd[-5].endswith("bench.py")
