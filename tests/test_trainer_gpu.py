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


def test_trainer_runs_flow_learner_and_grad_sync_attaches_to_its_unet(tmp_path):
    """`algorithm.name=flow_learner` (exp_99.py:22-28): the regression UNet + photometric pyramid through the same trainer; the
    checkpoint carries the reference's key prefixes (unet.model.* / model.model.*, FL:93-99); attach_grad_sync finds the engine
    UNet inside UnetWithWarp."""
    import train
    from opticalflowdiffusion_amd import FlowLearner, Unet, parallel
    d = str(tmp_path / "ck")
    fl, logs = train.main(["--steps", "3", "--log-every", "1", "--ckpt-dir", d, "--set", "algorithm.name=flow_learner",
                           "algorithm.image_size=[32,48]", "algorithm.levels=[1,2,4]", "algorithm.zero_init=false",
                           "experiment.training.data.batch_size=2"])
    assert isinstance(fl, FlowLearner) and len(logs) == 3 and all(r["loss"] == r["loss"] for r in logs)
    ck = torch.load(os.path.join(d, "last.ckpt"), map_location="cpu", weights_only=False)
    keys = set(ck["state_dict"])
    assert "unet.model.final_conv.weight" in keys and "model.model.final_conv.weight" in keys and not any("time_mlp" in k for k in keys)
    sync = parallel.attach_grad_sync(fl)
    assert isinstance(fl.unet.model, Unet) and fl.unet.model.grad_sync is sync


def _dp_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      OFD_DIST_BACKEND="gloo")
    import torch.distributed as dist
    from opticalflowdiffusion_amd import Unet, parallel
    from opticalflowdiffusion_amd.warp import nan_mse
    torch.cuda.set_device(0)
    parallel.init()
    torch.manual_seed(0)                                   # same weights on every rank
    net = Unet(64, channels=5, out_dim=2).cuda()
    g = torch.Generator().manual_seed(100 + rank)          # different data per rank
    x, c = torch.randn(1, 2, 16, 32, generator=g).cuda(), torch.randn(1, 3, 16, 32, generator=g).cuda()
    t, tgt = torch.tensor([5 + 7 * rank]).cuda(), torch.randn(1, 2, 16, 32, generator=g).cuda()

    def grads():
        net.zero_grad(set_to_none=True)
        nan_mse(net(x, external_cond=c, time=t), tgt).backward()
        return torch.cat([p.grad.flatten() for p in net.parameters()])

    local = grads()                                        # no grad sync attached: this rank's own gradient
    parallel.attach_grad_sync(net, bucket_bytes=8 << 20)   # several buckets over the 143 MB buffer
    synced = grads()
    others = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(others, local)
    want = sum(others) / world
    err = float((synced - want).norm() / want.norm())
    q.put((rank, err, float((synced - local).norm() / local.norm())))
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce_on_the_gpu():
    """the data-parallel exchange step with real executor gradients: two ranks (sharing this GPU, gloo as the transport:
    RCCL needs one device per rank) end up with the average of their individual gradients."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, err, moved in res:
        assert err < 1e-3, (rank, err)          # (fp32 atomics in the weight-gradient kernels: two runs differ in the last bits)
        assert moved > 1e-2                     # and it really is a different vector than the rank's own gradient


def test_trainer_composes_a_reference_style_configuration_tree(tmp_path):
    """`python train.py --config-dir <configurations> algorithm=flow_diffuser ...` = main.py's Hydra composition (main.py:25-30)
    followed by exp_base.py's train(): group selection, `defaults: [base]` inheritance, leaf overrides and the keys the trainer
    reads (batch size, lr, clipping, checkpoint interval)."""
    import train
    from test_host_logic_cpu import _write_config_tree
    _write_config_tree(str(tmp_path / "configurations"))
    fd, logs = train.main(["--steps", "2", "--log-every", "1", "--config-dir", str(tmp_path / "configurations"),
                           "algorithm=flow_diffuser", "algorithm.target=flow", "algorithm.image_size=[32,48]", "algorithm.timesteps=20",
                           "algorithm.zero_init=false", "experiment.training.data.batch_size=2", "+experiment.training.clipping=50",
                           "dataset.name=synthetic"])
    from opticalflowdiffusion_amd import FlowDiffuser
    assert isinstance(fd, FlowDiffuser) and fd.cfg.target == "flow" and fd.cfg.lr == pytest.approx(1e-5) and fd.cfg.clip == 50.0
    assert fd.optimizers.param_groups[0]["max_grad_norm"] == 50.0 and len(logs) == 2 and logs[-1]["global_batch"] == 2


def _torch_ddp_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      OFD_DIST_BACKEND="gloo")
    import torch.distributed as dist
    from opticalflowdiffusion_amd import FlowDiffuser, parallel
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    parallel.init()
    torch.manual_seed(0)                                   # same weights on every rank (DDP broadcasts rank 0's anyway)
    fd = FlowDiffuser(dict(target="flow", image_size=[16, 32], timesteps=10, flow_max=20, zero_init=False)).to(dev)
    fd.log_dict = lambda *a, **k: None
    g = torch.Generator().manual_seed(100 + rank)          # different data per rank
    batch = (torch.rand(1, 3, 16, 32, generator=g).to(dev), torch.rand(1, 3, 16, 32, generator=g).to(dev),
             ((torch.rand(1, 2, 16, 32, generator=g) * 2 - 1) * 8).to(dev))
    params = list(fd.model.parameters())

    def grads(module):
        for p in params:
            p.grad = None
        torch.manual_seed(7 + rank)                        # the same t and noise in both passes
        module.training_step(batch, 0).backward()
        return torch.cat([p.grad.flatten() for p in params])

    local = grads(fd)                                      # this rank's own gradient
    synced = grads(parallel.TorchDDP(fd, dev))             # through torch.nn.parallel.DistributedDataParallel (exp_base.py:198)
    others = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(others, local)
    want = sum(others) / world
    q.put((rank, float((synced - want).norm() / want.norm()), float((synced - local).norm() / local.norm())))
    dist.destroy_process_group()


def test_plugin_trains_under_torch_distributed_data_parallel():
    """What Lightning's DDPStrategy(find_unused_parameters=False) does to the plugin (exp_base.py:193-206): wrapped in
    torch.nn.parallel.DistributedDataParallel, two ranks end up with the mean of their gradients -- the engine's flat-buffer
    parameters and custom autograd Function are transparent to DDP's reducer."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_torch_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, err, moved in res:
        assert err < 1e-3, (rank, err)
        assert moved > 1e-2
