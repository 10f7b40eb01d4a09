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
                     sampling_timesteps=None, precision="bf16")

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
        """FD:136-168.  The reference's Augmentor is torchvision-on-CPU and assumes square inputs
        (augmentation.py:45-49); it is outside this path, so aug=True is accepted and ignored."""
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

    def training_step(self, batch, batch_idx):                              # FD:218-235
        batch = self.preprocess(batch)
        loss = self.loss(*batch)
        tgt, cond, flow = batch
        self.log_dict({
            "train/loss": loss,
            "train/cond_min": torch.min(cond), "train/cond_max": torch.max(cond), "train/cond_mean": torch.mean(cond),
            "train/cond_std": torch.mean(torch.std(cond, dim=0)),
            "train/flow_min": torch.min(flow), "train/flow_max": torch.max(flow), "train/flow_mean": torch.mean(flow),
            "train/flow_std": torch.mean(torch.std(flow, dim=0)),
        })
        return loss

    @torch.no_grad()
    def validation_step(self, batch, batch_idx):
        """FD:237-281 without the W&B image logging (FD:283-364, out of scope)."""
        img, tgt, flow = batch
        tgt_, cond, flow_ = self.preprocess(batch, aug=False)
        loss = self.loss(tgt_, cond, flow_)
        samples, flow_pred = self.sample(cond, flow_)
        final_flow = flow_pred[:, -1] if flow_pred.dim() == 5 else flow_pred
        self.log_dict({"val/loss": loss,
                       "val/flow_mse": torch.nn.functional.mse_loss(final_flow[:, -2:], flow_)}, sync_dist=True)
        return loss
