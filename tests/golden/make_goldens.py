"""Generates tests/golden/*.npz by RUNNING THE REFERENCE's own modules on CPU.

Run once in the build container (it needs /root/reference, which never travels):
    python tests/golden/make_goldens.py
Only data (inputs, expected outputs) is written; no reference source is stored.
The reference package __init__ imports plugins whose third-party deps are absent, so the
two modules on the hot path are imported directly with empty stubs for the imports they
never use on this path.  The splat cannot be captured: its kernels
are CUDA-only (softsplat_new.py:444).
"""
import importlib
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from oracle.unet_ref import closed_form_params, random_params  # noqa: E402  (weight fills shared with tests)


def import_reference():
    def stub(name, **a):
        m = types.ModuleType(name)
        m.__dict__.update(a)
        sys.modules[name] = m
        return m
    tv = stub("torchvision")
    tv.transforms = stub("torchvision.transforms")
    tv.utils = stub("torchvision.utils")
    stub("ema_pytorch", EMA=object)
    stub("pytorch_fid")
    stub("pytorch_fid.inception", InceptionV3=object)
    stub("pytorch_fid.fid_score", calculate_frechet_distance=None)
    stub("cupy", memoize=lambda **k: (lambda f: f), int32=int, float32=float,
         ndarray=type("ndarray", (), {}))
    pkg = types.ModuleType("refda")
    pkg.__path__ = ["/root/reference/algorithms/diffusion_animation"]
    sys.modules["refda"] = pkg
    dd = importlib.import_module("refda.denoising_diffusion")
    wp = importlib.import_module("refda.warp")
    return dd, wp


def fill(module, amp=None):
    sd = module.state_dict()
    shapes = {k: tuple(v.shape) for k, v in sd.items()}
    P = closed_form_params(shapes, amp)
    module.load_state_dict(P)
    return P


def sin_tensor(shape, k, scale=1.0):
    n = int(np.prod(shape))
    i = torch.arange(n, dtype=torch.float64)
    return (scale * torch.sin(0.61 * i + 0.13 * i * i / n + k)).to(torch.float32).reshape(shape)


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = v
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.1f} KiB")


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    dd, wp = import_reference()

    # ---- (1) schedule buffers -------------------------------------------------------------
    for T in (4, 1000):
        u = dd.Unet(64, channels=5, out_dim=2)
        cd = dd.ConditionalDiffusion(u, 16, objective="pred_x0", channels=2, auto_normalize=False,
                                     noise_space="image", timesteps=T, min_snr_loss_weight=True)
        save(f"schedule_T{T}", **{k: v for k, v in cd.state_dict().items() if not k.startswith("model.")})

    # ---- (2) per-module goldens -----------------------------------------------------------
    mods = {}
    rb = dd.ResnetBlock(128, 64, time_emb_dim=256)
    fill(rb)
    x = sin_tensor((2, 128, 8, 12), 1)
    temb = sin_tensor((2, 256), 2)
    mods["resblock_128_64"] = dict(x=x, temb=temb, y=rb(x, temb))
    rb2 = dd.ResnetBlock(64, 64, time_emb_dim=256)
    fill(rb2)
    x = sin_tensor((1, 64, 8, 8), 3)
    temb = sin_tensor((1, 256), 4)
    mods["resblock_64_64"] = dict(x=x, temb=temb, y=rb2(x, temb))
    la = dd.Residual(dd.PreNorm(64, dd.LinearAttention(64)))
    fill(la)
    x = sin_tensor((2, 64, 8, 12), 5, 2.0)
    mods["linattn_64"] = dict(x=x, y=la(x))
    at = dd.Residual(dd.PreNorm(64, dd.Attention(64)))
    fill(at)
    x = sin_tensor((2, 64, 6, 8), 6, 2.0)
    mods["attn_64"] = dict(x=x, y=at(x))
    dn = dd.Downsample(64, 128)
    fill(dn)
    x = sin_tensor((1, 64, 8, 12), 7)
    mods["downsample_64_128"] = dict(x=x, y=dn(x))
    up = dd.Upsample(128, 64)
    fill(up)
    x = sin_tensor((1, 128, 4, 6), 8)
    mods["upsample_128_64"] = dict(x=x, y=up(x))
    ws = dd.WeightStandardizedConv2d(64, 64, 3, padding=1)
    fill(ws)
    x = sin_tensor((1, 64, 8, 8), 9)
    mods["wsconv_64_64"] = dict(x=x, y=ws(x))
    flat = {}
    for m, d in mods.items():
        for k, v in d.items():
            flat[f"{m}.{k}"] = v
    with torch.no_grad():
        save("modules_fp32", **flat)

    # ---- (3) UNet end to end --------------------------------------------------------------
    taps_wanted = (["init_conv"] + [f"downs.{i}.{j}" for i in range(4) for j in (0, 2, 3)] +
                   ["mid_block1", "mid_attn", "mid_block2"] +
                   [f"ups.{i}.{j}" for i in range(4) for j in (2, 3)] + ["final_res_block"])
    for tag, ch, hw in (("c5_32x32", 5, (32, 32)), ("c9_32x48", 9, (32, 48))):
        u = dd.Unet(64, channels=ch, out_dim=2)
        fill(u)
        u.eval()
        B = 2
        xin = sin_tensor((B, ch - 3, *hw), 11, 1.0)
        cond = sin_tensor((B, 3, *hw), 12, 1.0)
        t = torch.tensor([3, 700])
        rec = {}
        hooks = []
        for name, mod in u.named_modules():
            if name in taps_wanted:
                hooks.append(mod.register_forward_hook(
                    lambda m, i, o, name=name: rec.__setitem__(name, o.detach().float().clone())))
        with torch.no_grad():
            y = u(xin, cond, t)
        for h in hooks:
            h.remove()
        arrays = dict(x=xin, cond=cond, t=t, y=y)
        for name, v in rec.items():
            arrays[f"tap.{name}.mean"] = v.mean()
            arrays[f"tap.{name}.std"] = v.std()
            arrays[f"tap.{name}.slice"] = v[:, :4, :6, :6]
        save(f"unet_fp32_{tag}", **arrays)

        # same input under bf16 autocast, plus the eps each WS-conv / LayerNorm saw (DD:107,122)
        eps_seen = {}
        hooks = []
        for name, mod in u.named_modules():
            if isinstance(mod, (dd.WeightStandardizedConv2d, dd.LayerNorm)):
                hooks.append(mod.register_forward_pre_hook(
                    lambda m, i, name=name: eps_seen.__setitem__(
                        name, 1e-5 if i[0].dtype == torch.float32 else 1e-3)))
        with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
            yb = u(xin, cond, t)
        for h in hooks:
            h.remove()
        save(f"unet_autocast_{tag}", y=yb.float())
        with open(os.path.join(HERE, f"unet_autocast_eps_{tag}.json"), "w") as f:
            json.dump(eps_seen, f, indent=0, sort_keys=True)

    # ---- (4) diffusion steps (target=flow wiring of FD:118-127) ---------------------------
    u = dd.Unet(64, channels=5, out_dim=2)
    fill(u)
    u.eval()
    cd = dd.ConditionalDiffusion(u, 16, objective="pred_x0", channels=2, auto_normalize=False,
                                 noise_space="image", timesteps=4, min_snr_loss_weight=True)
    B = 2
    cond = sin_tensor((B, 3, 16, 16), 21)
    x0 = sin_tensor((B, 2, 16, 16), 22, 0.8)
    noise = sin_tensor((B, 2, 16, 16), 23, 1.3)
    t = torch.tensor([1, 3])
    arrays = dict(cond=cond, x0=x0, noise=noise, t=t, q_sample=cd.q_sample(x0, t, noise))
    x_t = arrays["q_sample"]
    with torch.no_grad():
        mp = cd.model_predictions(x_t, t, external_cond=cond)
    arrays["pred_noise"] = mp.pred_noise
    arrays["pred_x_start"] = mp.pred_x_start
    for ti in (2, 0):
        torch.manual_seed(100 + ti)
        z = torch.randn_like(x_t)
        torch.manual_seed(100 + ti)
        img, xs, _ = cd.p_sample(x_t, ti, external_cond=cond)
        arrays[f"p_sample_t{ti}.z"] = z
        arrays[f"p_sample_t{ti}.img"] = img
        arrays[f"p_sample_t{ti}.x_start"] = xs
    # full DDPM loop, 4 steps
    torch.manual_seed(7)
    traj = cd.p_sample_loop((B, 2, 16, 16), return_all_timesteps=True, external_cond=cond)
    arrays["p_sample_loop.seed"] = np.int64(7)
    arrays["p_sample_loop.traj"] = traj
    # DDIM (direct call, D4 in SURVEY.md: sample() cannot reach it as shipped)
    cd2 = dd.ConditionalDiffusion(u, 16, objective="pred_x0", channels=2, auto_normalize=False,
                                  noise_space="image", timesteps=4, sampling_timesteps=2,
                                  min_snr_loss_weight=True)
    torch.manual_seed(9)
    traj2 = cd2.ddim_sample((B, 2, 16, 16), return_all_timesteps=True, external_cond=cond)
    arrays["ddim.seed"] = np.int64(9)
    arrays["ddim.traj"] = traj2
    # training loss + gradient norms
    u.train()
    torch.manual_seed(5)
    tt = torch.randint(0, 4, (B,)).long()
    nz = torch.randn_like(x0)
    torch.manual_seed(5)
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        loss = cd(x0, external_cond=cond)
    loss.backward()
    arrays["p_losses.t"] = tt
    arrays["p_losses.noise"] = nz
    arrays["p_losses.loss"] = loss.detach()
    arrays["p_losses.grad_final_conv_w"] = u.final_conv.weight.grad
    arrays["p_losses.grad_init_conv_w_norm"] = u.init_conv.weight.grad.norm()
    arrays["p_losses.grad_mid_qkv_norm"] = u.mid_attn.fn.fn.to_qkv.weight.grad.norm()
    torch.autograd.set_detect_anomaly(False)
    save("diffusion_T4", **arrays)

    # ---- (5) grid_sample backward warp + helpers ------------------------------------------
    arrays = {}
    img = sin_tensor((2, 3, 40, 64), 31) * 0.5 + 0.5
    flow = sin_tensor((2, 2, 40, 64), 32, 6.0)
    out, mask = wp.warp(None, img, flow, mode="backward")
    arrays.update({"rand.img": img, "rand.flow": flow, "rand.out": out, "rand.mask": mask})
    iflow = torch.round(sin_tensor((2, 2, 40, 64), 33, 5.0))
    out, mask = wp.warp(None, img, iflow, mode="backward")
    arrays.update({"int.flow": iflow, "int.out": out, "int.mask": mask})
    # wide case: integer targets where the fp32 normalise/un-normalise round trip floors to x-1
    img = sin_tensor((1, 2, 6, 1024), 34) * 0.5 + 0.5
    zflow = torch.zeros(1, 2, 6, 1024)
    out, mask = wp.warp(None, img, zflow, mode="backward")
    arrays.update({"wide.img": img, "wide.out": out, "wide.mask": mask})
    iflow = torch.round(sin_tensor((1, 2, 6, 1024), 35, 3.0))
    out, mask = wp.warp(None, img, iflow, mode="backward")
    arrays.update({"wideint.flow": iflow, "wideint.out": out, "wideint.mask": mask})
    a = sin_tensor((2, 3, 8, 8), 36)
    b = sin_tensor((2, 3, 8, 8), 37)
    a[0, 1, 2, 3] = float("nan")
    b[1, 0, 5, 5] = float("nan")
    arrays.update({"nan_mse.a": a, "nan_mse.b": b, "nan_mse.none": wp.nan_mse(a, b, reduction="none"),
                   "nan_mse.mean": wp.nan_mse(a, b, reduction="mean"),
                   "scale.x": img[:, :, :, :64], "scale.down2": wp.scale(img[:, :, :, :64], down=2)})
    save("warp_backward", **arrays)


def regression_unet():
    """(6) Unet(time_in=False): FlowDiffuser with is_diffusion=False (FD:106-111) and FlowLearner's flow+weight regressor
    (flow_learner.py:93-98): Unet(64, channels=6, out_dim=3, time_in=False), forward, bf16-autocast forward and the gradient
    of a fixed linear functional with respect to a few parameters."""
    torch.manual_seed(0)
    torch.set_num_threads(8)
    dd, _ = import_reference()
    u = dd.Unet(64, channels=6, out_dim=3, time_in=False)
    fill(u)
    u.eval()
    B, hw = 2, (32, 40)
    x = sin_tensor((B, 6, *hw), 41, 1.0)
    gy = sin_tensor((B, 3, *hw), 42, 1.0)
    y = u(x)
    (y * gy).sum().backward()
    arrays = dict(x=x, gy=gy, y=y.detach())
    P = dict(u.named_parameters())
    for name in ("final_conv.weight", "final_conv.bias", "final_res_block.block1.norm.weight", "ups.3.0.res_conv.weight",
                 "mid_block1.block2.proj.bias", "downs.0.0.block1.proj.weight", "init_conv.bias"):
        arrays[f"grad.{name}"] = P[name].grad
    arrays["n_params"] = np.int64(len(u.state_dict()))
    with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
        yb = u(x)
    arrays["y_autocast"] = yb.float()
    try:
        u(x, None, torch.tensor([1, 2]))
        arrays["time_rejected"] = np.int64(0)
    except ValueError:
        arrays["time_rejected"] = np.int64(1)           # DD:382-383
    save("unet_notime_c6_32x40", **arrays)


def loss_helpers():
    """(7) the photometric-loss helpers of the FlowLearner path (WP:273-303), run from the reference's warp module."""
    _, wp = import_reference()
    img = sin_tensor((2, 3, 12, 20), 51)
    flow = sin_tensor((2, 2, 12, 20), 52, 4.0)
    a = sin_tensor((2, 3, 12, 20), 53)
    b = sin_tensor((2, 3, 12, 20), 54)
    a[0, 1, 2, 3] = float("nan")
    b[1, 0, 5, 5] = float("nan")
    w = sin_tensor((2, 1, 12, 20), 55)
    save("loss_helpers", img=img, flow=flow, a=a, b=b, w=w, charbonnier=wp.charbonnier(a), nan_charbonnier=wp.nan_charbonnier(a, b),
         fill_holes_nan=wp.fill_holes_nan(img, w), edgeaware_smoothness1=wp.edgeaware_smoothness1(img, flow))


TAPS = (["init_conv"] + [f"downs.{i}.{j}" for i in range(4) for j in (0, 2, 3)] + ["mid_block1", "mid_attn", "mid_block2"] +
        [f"ups.{i}.{j}" for i in range(4) for j in (2, 3)] + ["final_res_block"])


def random_weight_unet():
    """(8) the reference Unet under WELL-CONDITIONED weights (oracle.unet_ref.random_params: default-init-like statistics, the
    two dtype-dependent eps sites made immaterial), in fp32 AND under bf16 autocast, output and 19 taps each.  Under these
    weights the reference's own bf16-vs-fp32 distance is rounding noise (~1.5e-2), so it is the floor the HIP engine's bf16
    output is asserted against DIRECTLY (tests/test_unet_gpu.py::test_hip_unet_against_the_reference_module_outputs)."""
    torch.set_num_threads(8)
    dd, _ = import_reference()
    for tag, ch, hw in (("c5_64x96", 5, (64, 96)), ("c9_32x48", 9, (32, 48))):
        u = dd.Unet(64, channels=ch, out_dim=2)
        P = random_params({k: tuple(v.shape) for k, v in u.state_dict().items()}, seed=1)
        u.load_state_dict(P)
        u.eval()
        g = torch.Generator().manual_seed(3)
        x = torch.randn(2, ch - 3, *hw, generator=g)
        cond = torch.rand(2, 3, *hw, generator=g) * 2 - 1
        t = torch.tensor([3, 700])
        arrays = dict(x=x, cond=cond, t=t, seed=np.int64(1))
        for mode in ("fp32", "autocast"):
            rec = {}
            hooks = [mod.register_forward_hook(lambda m, i, o, name=name: rec.__setitem__(name, o.detach().float().clone()))
                     for name, mod in u.named_modules() if name in TAPS]
            with torch.no_grad():
                if mode == "fp32":
                    y = u(x, cond, t)
                else:
                    with torch.autocast("cpu", dtype=torch.bfloat16):
                        y = u(x, cond, t)
            for h in hooks:
                h.remove()
            arrays[f"y.{mode}"] = y.float()
            for name, v in rec.items():
                arrays[f"tap.{name}.{mode}"] = v[:, :8, :8, :8]
                arrays[f"tapstat.{name}.{mode}"] = torch.stack((v.mean(), v.std()))
                arrays[f"tapshape.{name}"] = np.asarray(v.shape, dtype=np.int64)
        save(f"unet_rand_{tag}", **arrays)


GRAD_FULL = ("init_conv.weight", "time_mlp.1.weight", "downs.0.0.block1.proj.weight", "downs.0.0.mlp.1.weight",
             "downs.1.2.fn.fn.to_qkv.weight", "downs.2.0.block2.norm.weight", "downs.3.2.fn.fn.to_out.1.g", "mid_attn.fn.fn.to_out.bias",
             "mid_block1.block1.proj.bias", "ups.1.2.fn.norm.g", "ups.3.0.res_conv.weight", "ups.3.1.block2.proj.weight",
             "final_res_block.block1.norm.bias", "final_conv.weight", "final_conv.bias")


def random_weight_unet_gradients():
    """(9) gradients of the DIFFUSION Unet (time_in=True, channels=9) from the reference module itself: (y * gy).sum().backward() under
    the well-conditioned weights of (8), in fp32 and under bf16 autocast.  Stored: the norm of every one of the 276 parameter
    gradients in both modes, the per-tensor relative distance between the two (the reference's own bf16-vs-fp32 gradient floor), and
    fifteen full fp32 gradient tensors spread over the levels.  The HIP backward is asserted against these DIRECTLY
    (tests/test_backward_gpu.py::test_hip_unet_gradients_against_the_reference_module)."""
    torch.set_num_threads(8)
    dd, _ = import_reference()
    ch, hw = 9, (32, 48)
    u = dd.Unet(64, channels=ch, out_dim=2)
    P = random_params({k: tuple(v.shape) for k, v in u.state_dict().items()}, seed=1)
    u.load_state_dict(P)
    u.train()                      # (no dropout / batch statistics in this UNet: train() == eval(), DD:272-361)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, ch - 3, *hw, generator=g)
    cond = torch.rand(2, 3, *hw, generator=g) * 2 - 1
    gy = torch.randn(2, 2, *hw, generator=g)
    t = torch.tensor([3, 700])
    arrays = dict(x=x, cond=cond, t=t, gy=gy, seed=np.int64(1))
    names = [n for n, _ in u.named_parameters()]
    grads = {}
    for mode in ("fp32", "autocast"):
        u.zero_grad(set_to_none=True)
        if mode == "fp32":
            y = u(x, cond, t)
        else:
            with torch.autocast("cpu", dtype=torch.bfloat16):
                y = u(x, cond, t)
        (y.float() * gy).sum().backward()
        arrays[f"y.{mode}"] = y.detach().float()
        grads[mode] = {n: p.grad.detach().float().clone() for n, p in u.named_parameters()}
    arrays["names"] = np.asarray(names)
    arrays["norm.fp32"] = np.asarray([float(grads["fp32"][n].double().norm()) for n in names])
    arrays["norm.autocast"] = np.asarray([float(grads["autocast"][n].double().norm()) for n in names])
    arrays["floor"] = np.asarray([float((grads["autocast"][n].double() - grads["fp32"][n].double()).norm() / (grads["fp32"][n].double().norm() + 1e-30))
                                  for n in names])
    for n in GRAD_FULL:
        arrays[f"grad.{n}"] = grads["fp32"][n]
    fl = arrays["floor"]
    print(f"276-gradient floor (reference bf16 autocast vs its fp32): median {np.median(fl):.3e}  max {fl.max():.3e} ({names[int(fl.argmax())]})")
    save("unet_rand_grads_c9_32x48", **arrays)


if __name__ == "__main__":
    if "--only-loss-helpers" in sys.argv:
        loss_helpers()
    elif "--only-regression-unet" in sys.argv:
        regression_unet()
    elif "--only-random-unet" in sys.argv:
        random_weight_unet()
    elif "--only-random-unet-gradients" in sys.argv:
        random_weight_unet_gradients()
    else:
        main()
        regression_unet()
        loss_helpers()
        random_weight_unet()
        random_weight_unet_gradients()
