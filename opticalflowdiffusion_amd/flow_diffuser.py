"""`FlowDiffuser` / `UnetWithWarp` with the reference's plugin surface
(algorithms/diffusion_animation/flow_diffuser.py, "FD") on the HIP engine.

The reference class is a `pl.LightningModule`; Lightning, Hydra and W&B are optional here: when
`pytorch_lightning` is importable it is used as the base class, otherwise a minimal stand-in
with the hooks `experiments/exp_base.py` drives (`log_dict`, `configure_optimizers`,
`training_step`, `validation_step`).  `cfg` may be a DictConfig, a dict or any attribute object
with the keys of configurations/algorithm/flow_diffuser.yaml (+ optional `image_size: [H, W]`,
`sampling_timesteps`, `precision`).
"""
import torch

from .denoising_diffusion import Unet, ConditionalDiffusion
from .warp import warp
from . import _lib as L

try:                                                   # pragma: no cover - not installed in this image
    import pytorch_lightning as pl
    _Base = pl.LightningModule
except Exception:                                      # noqa: BLE001
    class _Base(torch.nn.Module):
        """What FlowDiffuser needs from LightningModule when Lightning is absent."""

        def __init__(self):
            super().__init__()
            self.logged = {}

        def log_dict(self, d, **kw):
            self.logged.update({k: (v.detach() if torch.is_tensor(v) else v) for k, v in d.items()})

        def log(self, k, v, **kw):
            self.log_dict({k: v})


class _Cfg:
    """attribute + `in` access over a dict / DictConfig / namespace"""

    _DEFAULTS = dict(name="flow_diffuser", image_size=128, latent_dim=16, flow_max=20, latent_max=2, lr=1e-5,
                     flow_weight=0.0, weight_decay=1e-6, is_diffusion=True, latent=False, timesteps=1000,
                     target="joint", ae="px8q8g0m", noiser="image", zero_init=True,
                     sampling_timesteps=None, precision="bf16", augment=True)

    def __init__(self, cfg):
        self._d = dict(self._DEFAULTS)
        if isinstance(cfg, dict):
            self._d.update(cfg)
        else:
            for k in list(self._DEFAULTS) + [k for k in dir(cfg) if not k.startswith("_")]:
                try:
                    v = cfg[k] if hasattr(cfg, "__getitem__") else getattr(cfg, k)
                except Exception:                      # noqa: BLE001
                    continue
                if not callable(v):
                    self._d[k] = v

    def __getattr__(self, k):
        try:
            return self.__dict__["_d"][k]
        except KeyError:
            raise AttributeError(k)

    def __contains__(self, k):
        return k in self._d


class UnetWithWarp(torch.nn.Module):
    """FD:20-63: NaN-safe UNet call followed by a forward splat of the condition image."""

    def __init__(self, cfg, unet, full_output, nan_safe=True):
        super().__init__()
        self.cfg = cfg
        self.flow_max = cfg.flow_max
        self.dim = cfg.latent_dim if cfg.latent else 3
        self.model = unet
        self.full_output = full_output
        self.nan_safe = nan_safe
        if cfg.zero_init:                                                   # FD:31-33
            self.model.final_conv.weight.data = torch.zeros_like(self.model.final_conv.weight.data)
            self.model.final_conv.bias.data = torch.zeros_like(self.model.final_conv.bias.data)

    @property
    def self_condition(self):
        return False

    def _warp(self, image, flow, **kwargs):                                 # FD:35-36
        return warp(image[:, :self.dim], None, flow * self.flow_max, mode="forward", **kwargs)

    def forward(self, x, external_cond=None, t=None, self_cond=None, additional_out=False):
        if self.nan_safe:                                                   # FD:39-45
            x = x.clone()
            where_nans = torch.isnan(x)
            x[where_nans] = 0.0
            where_nans = torch.any(where_nans, dim=1)[:, None]
            flow = self.model(torch.cat((x, where_nans.to(x.dtype)), dim=1), external_cond, t, self_cond)
        else:
            flow = self.model(x, external_cond, t, self_cond)
        if external_cond is not None:
            warped = self._warp(external_cond, flow[:, :2])
        else:
            warped = self._warp(x[:, :self.dim], flow[:, :2])
        out = warped
        if self.full_output:
            out = torch.cat((out, flow), dim=1)
        if not additional_out:
            return out
        return torch.cat((out, flow), dim=1)


class FlowDiffuser(_Base):
    """FD:65-388 (latent=False): `training_step` is differentiable through the HIP training executor
    (`ofd_unet_train_forward` / `ofd_unet_backward` behind `denoising_diffusion._UnetTrain`)."""

    def __init__(self, cfg):
        super().__init__()
        cfg = cfg if isinstance(cfg, _Cfg) else _Cfg(cfg)
        self.cfg = cfg
        self.flow_max = cfg.flow_max
        self.latent_max = cfg.latent_max
        self.is_diffusion = cfg.is_diffusion
        self.latent = cfg.latent
        self.target = cfg.target
        if self.latent:
            raise NotImplementedError("latent=True needs the reference's W&B autoencoder checkpoint (FD:84-90): network fetch")
        self.dim = 3
        if self.target == "target":                                         # FD:98-104
            unet_dims = self.dim + 1
        elif self.target == "joint":
            unet_dims = self.dim + 3
        else:
            unet_dims = 2
        self.unet = Unet(64, channels=self.dim + unet_dims * int(self.is_diffusion), out_dim=2, time_in=bool(self.is_diffusion),
                         precision=cfg.precision)                           # FD:106-111
        if cfg.target in ["target", "joint"]:
            self._model = UnetWithWarp(cfg, self.unet, full_output=cfg.target == "joint")
        else:
            self._model = self.unet
        if not self.is_diffusion:                                           # FD:128-129: plain regression cond -> flow
            self.model = self._model
            return
        self.model = ConditionalDiffusion(                                  # FD:118-127
            self._model, cfg.image_size, objective="pred_x0",
            channels=2 + 1 * int(cfg.target == "target") + 3 * int(cfg.target == "joint"),
            auto_normalize=False, noise_space="image" if cfg.noiser == "image" else "flow",
            timesteps=cfg.timesteps, sampling_timesteps=cfg.sampling_timesteps, min_snr_loss_weight=True)
        if "trajectory_stride" in cfg:                                      # optional key, default = every frame as the reference
            self.model.trajectory_stride = cfg.trajectory_stride

    def configure_optimizers(self):                                         # FD:131-134
        """Adam(lr, weight_decay) as the reference; the HIP multi-tensor step (optim.FusedAdam) has
        torch.optim.Adam's update rule and state-dict keys.  `cfg.clip` (optional) folds the trainer's
        gradient_clip_val (exp_base.py:205) into the same launches."""
        from .optim import FusedAdam
        clip = getattr(self.cfg, "clip", 0.0) if "clip" in self.cfg else 0.0
        self.optimizers = FusedAdam(self.model.parameters(), lr=self.cfg.lr, weight_decay=self.cfg.weight_decay, max_grad_norm=clip)
        return self.optimizers

    def preprocess(self, batch, aug=True):
        """FD:136-168.  `aug=True` (what training_step passes, FD:219) runs the batched GPU `Augmentor` (augmentation.py of this
        package: same structure and probabilities as the reference's torchvision pipeline, non-square aware); the optional cfg key
        `augment: false` switches it off."""
        if aug and self.cfg.augment:
            if getattr(self, "augmentor", None) is None:
                from .augmentation import Augmentor
                self.augmentor = Augmentor()
            with torch.no_grad():
                batch = self.augmentor(batch)
        img, tgt, flow = batch
        flow = torch.clamp(flow / self.flow_max, -1.0, 1.0)
        img = 2 * img - 1.0
        tgt = 2 * tgt - 1.0
        ret = []
        if self.target == "target":
            ret.append(warp(img, None, flow * self.flow_max, mode="forward"))
        elif self.target == "joint":
            ret.append(torch.cat((warp(img, None, flow * self.flow_max, mode="forward"), flow), dim=1))
        else:
            ret.append(flow)
        ret.append(img)
        ret.append(flow)
        return tuple(ret)

    def loss(self, tgt, cond, flow, override=None):                         # FD:170-187
        if not self.is_diffusion:
            # FD:176-186.  (The reference tests `override is not None` the wrong way round, FD:177-180: it calls the model when an
            # override IS given and uses None otherwise; the evident intent is implemented.)
            out = override if override is not None else self.model(cond, additional_out=self.cfg.target == "target")
            mse = torch.nn.functional.mse_loss
            if self.cfg.target in ["target", "joint"]:
                return mse(out[:, :self.dim], tgt) + self.cfg.flow_weight * mse(out[:, self.dim:], flow)
            return mse(out, flow)
        if self.cfg.target == "target":
            return self.model(tgt, external_cond=cond, additional_tgt=flow, additional_weight=self.cfg.flow_weight,
                              model_out_override=override)
        return self.model(tgt, external_cond=cond, model_out_override=override)

    def sample(self, cond, flow):                                           # FD:189-215
        bsz = flow.shape[0]
        if not self.is_diffusion:                                           # FD:204-213
            if self.cfg.target in ["target", "joint"]:
                samples = self.model(cond, additional_out=True) if self.cfg.target == "target" else self.model(cond)
                return samples[:, :self.dim], samples[:, -2:]
            flow = self.model(cond)
            return warp(cond[:, :self.dim], None, flow, mode="forward"), flow
        if self.cfg.target == "target":
            samples, flow = self.model.sample(batch_size=bsz, external_cond=cond, additional_tgt=flow, return_all_timesteps=True)
        elif self.cfg.target == "joint":
            joint = self.model.sample(batch_size=bsz, external_cond=cond, return_all_timesteps=True)
            samples = joint[:, :, :self.dim]
            flow = joint[:, :, self.dim:]
        else:
            flow = self.model.sample(batch_size=bsz, external_cond=cond, return_all_timesteps=True)
            img = cond[:, :self.dim]
            samples = warp(img, None, flow[:, -1], mode="forward")
        return samples, flow

    @staticmethod
    def _batch_stats(x):
        """(min, max, mean, mean(std(x, dim=0))) of a (B, ...) tensor as 0-dim views of one 4-float result: what FD:218-235 logs, in one pass
        over x (`ofd_batch_stats`) instead of five reductions"""
        L.require_gpu(x)
        x = L.f32c(x)
        ws = torch.empty(L.lib().ofd_batch_stats_ws_doubles(), dtype=torch.float64, device=x.device)
        out = torch.empty(4, dtype=torch.float32, device=x.device)
        L.check(L.lib().ofd_batch_stats(L.ptr(x), x.shape[0], x[0].numel(), L.ptr(ws), L.ptr(out), L.stream()))
        return out[0], out[1], out[2], out[3]

    def training_step(self, batch, batch_idx):                              # FD:218-235
        batch = self.preprocess(batch)
        loss = self.loss(*batch)
        tgt, cond, flow = batch
        with torch.no_grad():
            c_min, c_max, c_mean, c_std = self._batch_stats(cond)
            f_min, f_max, f_mean, f_std = self._batch_stats(flow)
        self.log_dict({
            "train/loss": loss,
            "train/cond_min": c_min, "train/cond_max": c_max, "train/cond_mean": c_mean, "train/cond_std": c_std,
            "train/flow_min": f_min, "train/flow_max": f_max, "train/flow_mean": f_mean, "train/flow_std": f_std,
        })
        return loss

    def _log_image(self, key, images):
        """`self.logger.log_image(key=, images=, step=)` when the trainer attached a logger with that method (W&B in the
        reference, FD:296-347); kept in `self.logged_images` otherwise."""
        logger = getattr(self, "logger", None)
        if logger is not None and hasattr(logger, "log_image"):
            logger.log_image(key=key, images=images, step=getattr(self, "global_step", 0))
        else:
            if not hasattr(self, "logged_images"):
                self.logged_images = {}
            self.logged_images[key] = images

    def validation_step(self, batch, batch_idx):
        """FD:237-364: validation loss, a full sampling run, 19 logged scalars incl. `val/mse` and `val/ideal_loss` (the loss
        of the ground-truth flow pushed through `model_out_override`, FD:256-259), flow colour images, mid-trajectory strips
        (`[:, ::50]`), `val/last_step` and the `grad_flow` image (the loss gradient w.r.t. the sampled flow, through the splat
        backward kernels, FD:351-364).  The reference is only coherent for target in {target, joint} (with target == flow its
        `ideal_loss` is undefined and the call raises NameError); here the flow target logs the scalars that exist."""
        from .visualization import flow_to_image
        img, tgt, flow = batch
        tgt_, cond, flow_ = self.preprocess(batch, aug=False)
        bsz = img.shape[0]
        warped_target = self.target in ("target", "joint")

        with torch.no_grad():
            loss = self.loss(tgt_, cond, flow_)
            samples, p_flows = self.sample(cond, flow_)
            mid_samples = mid_flows = None
            if self.is_diffusion and warped_target:
                mid_samples = samples[:, ::50]                               # FD:246
                samples = samples[:, -1]
                if self.target == "target":
                    p_flows = [None] + [p * self.flow_max for p in p_flows[1:]]
                    mid_flows = p_flows[1::50]
                    p_flows = p_flows[-1]
                else:
                    mid_flows = p_flows[:, ::50] * self.flow_max
                    p_flows = p_flows[:, -1] * self.flow_max
            elif self.is_diffusion:                                          # flow target: trajectory of flows, one reconstruction
                p_flows = p_flows[:, -1] * self.flow_max
            else:
                p_flows = p_flows * self.flow_max if not warped_target else p_flows
            # samples / tgt live in [-1, 1] / [0, 1] exactly as in the reference's comparison (FD:255)
            mse = torch.nn.functional.mse_loss(torch.nan_to_num(samples), tgt)
            scalars = {
                "val/loss": loss, "val/mse": mse,
                "val/cond_min": torch.min(cond), "val/cond_max": torch.max(cond), "val/cond_mean": torch.mean(cond),
                "val/cond_std": torch.mean(torch.std(cond, dim=0)),
                "val/flow_min": torch.min(flow), "val/flow_max": torch.max(flow), "val/flow_mean": torch.mean(flow),
                "val/flow_std": torch.mean(torch.std(flow, dim=0)),
                "val/samples_min": torch.min(torch.nan_to_num(samples)), "val/samples_max": torch.max(torch.nan_to_num(samples)),
                "val/samples_mean": torch.nanmean(samples), "val/samples_std": torch.mean(torch.std(torch.nan_to_num(samples), dim=0)),
                "val/p_flow_min": torch.min(p_flows), "val/p_flow_max": torch.max(p_flows), "val/p_flow_mean": torch.mean(p_flows),
                "val/p_flow_std": torch.mean(torch.std(p_flows, dim=0)),
                "val/flow_mse": torch.nn.functional.mse_loss(p_flows / self.flow_max, flow_),
            }
            if self.is_diffusion and warped_target:                          # FD:256-259
                ideal_img = warp(cond[:, :self.dim], None, flow_ * self.flow_max, mode="forward")
                if self.target == "target":
                    scalars["val/ideal_loss"] = self.loss(tgt_, cond, flow_, override=(ideal_img, flow_))
                else:
                    scalars["val/ideal_loss"] = self.loss(tgt_, cond, flow_, override=(torch.cat((ideal_img, flow_), dim=1), None))
            self.log_dict(scalars, sync_dist=True)

            def chunk(x):
                x = x.clone()
                x[:, 0, 0, 0] = x[:, 0, 0, 0] * 0.95                        # FD:285: not completely white
                return list(torch.chunk(x, bsz))

            flos = flow_to_image(torch.cat((flow, p_flows, flow - p_flows), dim=0)) / 255.0      # FD:289-292
            gt_flow, sample_flow, diff_flow = flos[:bsz], flos[bsz:2 * bsz], flos[2 * bsz:]
            self._log_image("original", chunk(img))
            self._log_image("target", chunk(tgt))
            self._log_image("diffusion_tgt", chunk((tgt_[:, :self.dim] + 1.0) * 0.5) if tgt_.shape[1] >= self.dim else chunk(tgt))
            self._log_image("original_warped", chunk(warp(img, None, flow, mode="forward")))
            self._log_image("gt_flow", chunk(gt_flow))
            self._log_image("target_p", chunk(sample_flow))
            self._log_image("concat", chunk(torch.cat((gt_flow, sample_flow), dim=3)))
            self._log_image("difference", chunk(diff_flow))
            self._log_image("samples", chunk(samples))

            if self.is_diffusion and warped_target:                          # FD:317-338: strips of every 50th step
                strip = torch.cat(torch.chunk(mid_samples, mid_samples.shape[1], dim=1), dim=-1)[:, 0]
                strip = torch.clamp(torch.nan_to_num(strip), -1.0, 1.0)
                if self.target == "target":
                    fstrip = torch.cat([flow_to_image(m) / 255.0 for m in mid_flows], dim=-1)
                else:
                    shp = list(mid_flows.shape)
                    fl = flow_to_image(mid_flows.reshape(-1, 2, shp[-2], shp[-1])) / 255.0
                    shp[2] = 3
                    fstrip = torch.cat(torch.chunk(fl.reshape(shp), shp[1], dim=1), dim=-1)[:, 0]
                self._log_image("mid_samples", chunk(strip))
                self._log_image("mid_flows", chunk(fstrip))
                # FD:341-349: what the network answers at t = 0 when shown the clean target
                last_step = self.model.model(tgt_, cond, torch.zeros((bsz,), device=tgt_.device, dtype=torch.long), None, additional_out=True)
                last_step = last_step[:, -2:]
                self.log_dict({"val/last_step": torch.nn.functional.mse_loss(last_step, flow_)}, sync_dist=True)
                fl2 = flow_to_image(torch.cat((flow_, last_step), dim=0)) / 255.0
                self._log_image("last_step", chunk(torch.cat((fl2[:bsz], fl2[bsz:]), dim=-1)))

        if self.is_diffusion and warped_target:                              # FD:351-364: descent direction of the pyramid loss w.r.t. the flow
            with torch.set_grad_enabled(True):
                pf = p_flows.detach().clone().requires_grad_(True)
                gl = self.model._loss(warp(cond, None, pf, mode="forward"), tgt_[:, :self.dim], None, flow_, cond, pf / self.flow_max, 0.0)
                gl.backward()
                grad_flow = -pf.grad.clone()
            self._log_image("grad_flow", list(torch.chunk(flow_to_image(grad_flow) / 255.0, bsz, dim=0)))
        return loss

    def log_grad_norm_stat(self):
        """FD:367-388: gradient-norm and gradient-to-parameter-ratio statistics over the parameters that have a gradient."""
        with torch.no_grad():
            gn, gpr = [], []
            for _name, p in self.named_parameters():
                if p.grad is not None:
                    gn.append(torch.norm(p.grad))
                    gpr.append(torch.norm(p.grad) / torch.norm(p))
            gn, gpr = torch.stack(gn), torch.stack(gpr)
            self.log_dict({
                "train/grad_norm/min": gn.min(), "train/grad_norm/max": gn.max(), "train/grad_norm/std": gn.std(),
                "train/grad_norm/mean": gn.mean(), "train/grad_norm/median": torch.median(gn),
                "train/gpr/min": gpr.min(), "train/gpr/max": gpr.max(), "train/gpr/std": gpr.std(), "train/gpr/mean": gpr.mean(),
                "train/gpr/median": torch.median(gpr),
            })
