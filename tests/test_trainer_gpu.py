"""train.py (the Lightning-free stand-in for experiments/exp_base.py:177-214): checkpoint layout with the
reference's key prefixes and a resume that continues where the run stopped."""
import json
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SETS = ["algorithm.target=flow", "algorithm.image_size=[32,48]", "algorithm.timesteps=20", "algorithm.lr=1e-4", "algorithm.zero_init=false",
        "experiment.training.data.batch_size=2", "experiment.training.checkpointing.every_n_train_steps=3"]


def test_checkpoint_layout_and_resume(tmp_path, capsys):
    import train
    d = str(tmp_path / "ck")
    fd, logs = train.main(["--steps", "6", "--log-every", "1", "--ckpt-dir", d, "--set", *SETS])
    assert os.path.exists(os.path.join(d, "step=3.ckpt")) and os.path.exists(os.path.join(d, "last.ckpt"))
    ck = torch.load(os.path.join(d, "step=3.ckpt"), map_location="cpu", weights_only=False)
    assert {"state_dict", "optimizer_states", "global_step", "epoch"} <= set(ck) and ck["global_step"] == 3
    keys = set(ck["state_dict"])
    # the same UNet under the three names the reference registers it with (FD:106-127) + the 13 schedule buffers (DD:532-578)
    for k in ("unet.downs.0.0.block1.proj.weight", "_model.downs.0.0.block1.proj.weight", "model.model.downs.0.0.block1.proj.weight",
              "unet.final_conv.weight", "model.model.mid_attn.fn.fn.to_qkv.weight"):
        assert k in keys, k
    buffers = [k for k in keys if k.startswith("model.") and not k.startswith("model.model.")]
    assert len(buffers) == 13 and "model.betas" in keys and "model.loss_weight" in keys
    assert sum(1 for k in keys if k.startswith("unet.")) == 276
    # resume from step 3: steps 4..6 follow the uninterrupted run (same RNG state, same Adam moments; the weight-gradient
    # kernels add partial sums with fp32 atomics, so the match is to rounding, not to the bit)
    fd2, logs2 = train.main(["--steps", "6", "--log-every", "1", "--resume", os.path.join(d, "step=3.ckpt"), "--set", *SETS])
    want = {r["step"]: r["loss"] for r in logs if r["step"] > 3}
    got = {r["step"]: r["loss"] for r in logs2}
    assert set(got) == set(want) == {4, 5, 6}
    assert got[4] == want[4]                                        # first resumed step: identical inputs and weights
    for k in want:
        assert abs(got[k] - want[k]) < 2e-3 * abs(want[k]), (got, want)
    for (n1, p1), (n2, p2) in zip(fd.named_parameters(), fd2.named_parameters()):
        assert n1 == n2 and torch.allclose(p1, p2, rtol=1e-3, atol=2e-5), n1
    out = capsys.readouterr().out
    assert all(json.loads(ln)["world"] == 1 for ln in out.strip().splitlines())
