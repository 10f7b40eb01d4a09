"""`FlowLearner` (algorithms/diffusion_animation/flow_learner.py, "FL") on the HIP engine: the second consumer of the
UNet executor and of the splat kernels.

A `Unet(64, channels=6, out_dim=3, time_in=False)` (FL:93-98) regresses flow + a splat-weight map from an image pair and
is trained by a photometric pyramid: for each of 10 levels L and each of its L*L sub-pixel offsets the first image is
soft-splatted with the predicted flow at scale L and compared with the equally down-sampled second image (FL:159-206) --
1052 splat pairs per step, a splat-bound workload.  Only the flow representation is built (`radius` selects the reference's
filter representation, FL:75-81, which needs ConvToFilter / filter_to_flow: outside this path).
"""
import torch

from .denoising_diffusion import Unet
from .flow_diffuser import UnetWithWarp, _Base, _Cfg
from .softsplat import pyramid_charbonnier, pyramid_offsets, softsplat, softsplat_pyramid, splat_pyramid
from .warp import charbonnier, edgeaware_smoothness1, fill_holes_nan, nan_charbonnier

LEVELS = (1, 2, 4, 5, 7, 8, 10, 11, 14, 16)          # FL:163


class _LearnerCfg(_Cfg):
    _DEFAULTS = dict(name="flow_learner", image_size=128, flow_max=20, latent=False, zero_init=True, c2f=False, lr=8e-5,
                     weight_decay=1e-6, sparsity_weight=0.0, occlusion_mask=True, train_aug=True, latent_dim=16, precision="bf16")


def photometric_pyramid_loss(input_img, flow_pred, warp_weights, tgt, levels=LEVELS):
    """FL:159-206: mean over levels of the mean over the L*L offsets of nan_charbonnier(down-sampled target, splatted input).
    The per-offset reference structure: two soft splats per (level, offset)."""
    zero_flow = torch.zeros_like(flow_pred)
    ones = torch.ones_like(warp_weights)
    photo = []
    for level in levels:
        per_offset = []
        for a in range(level):
            for b in range(level):
                sw = softsplat(input_img, flow_pred, warp_weights, "soft", scale=level, offset=[a, b])
                filled = fill_holes_nan(sw[:, :-1], sw[:, -1:])
                dt = softsplat(tgt, zero_flow, ones, "soft", scale=level, offset=[a, b])[:, :-1]
                per_offset.append(nan_charbonnier(dt, filled))
        photo.append(sum(per_offset) / len(per_offset))
    return sum(photo) / len(photo)


def photometric_pyramid_loss_fused(input_img, flow_pred, warp_weights, tgt, levels=LEVELS):
    """The same loss with ONE pyramid splat per level and image (softsplat.splat_pyramid: scale-1 splat + tent filters + border
    scatter) instead of L*L splats, and ONE reduction kernel per level (softsplat.pyramid_charbonnier: soft-mode normalisation,
    hole filling, Charbonnier penalty and the per-offset means)."""
    zero_flow = torch.zeros_like(flow_pred)
    e_in = warp_weights.exp()
    in4 = torch.cat([input_img * e_in, e_in], 1)                               # softsplat "soft": cat(x e^m, e^m)
    e_tg = torch.ones_like(warp_weights).exp()
    tg4 = torch.cat([tgt * e_tg, e_tg], 1)
    photo = [pyramid_charbonnier(splat_pyramid(in4, flow_pred, level), splat_pyramid(tg4, zero_flow, level), level) for level in levels]
    return sum(photo) / len(photo)


def photometric_pyramid_loss_fused_torch(input_img, flow_pred, warp_weights, tgt, levels=LEVELS):
    """photometric_pyramid_loss_fused with the per-level loss spelled out in torch ops on the interleaved layout (test reference
    of the reduction kernel)."""
    zero_flow = torch.zeros_like(flow_pred)
    ones = torch.ones_like(warp_weights)
    photo = []
    for level in levels:
        sw = pyramid_offsets(softsplat_pyramid(input_img, flow_pred, warp_weights, "soft", level), level)      # (a, b, B, 4, Ho, Wo)
        filled = fill_holes_nan(sw[:, :, :, :-1], sw[:, :, :, -1:])
        dt = pyramid_offsets(softsplat_pyramid(tgt, zero_flow, ones, "soft", level), level)[:, :, :, :-1]
        ok = torch.logical_not(torch.logical_or(torch.isnan(dt), torch.isnan(filled)))
        diff = torch.where(ok, dt - filled, torch.zeros_like(dt))
        pen = torch.where(ok, charbonnier(diff), torch.zeros_like(dt))
        per_offset = pen.sum(dim=(2, 3, 4, 5)) / ok.sum(dim=(2, 3, 4, 5))        # nan_charbonnier of every (a, b)
        photo.append(per_offset.mean())
    return sum(photo) / len(photo)


class FlowLearner(_Base):
    """FL:62-424, flow representation."""

    def __init__(self, cfg):
        super().__init__()
        cfg = cfg if isinstance(cfg, _Cfg) else _LearnerCfg(cfg)
        self.cfg = cfg
        if "radius" in cfg:
            raise NotImplementedError("the filter representation (cfg.radius, FL:75-81) is outside this path")
        self.radius = None
        self.flow_max = cfg.flow_max
        self.rep = "flow"
        self.levels = tuple(cfg.levels) if "levels" in cfg else LEVELS
        # pyramid: "fused" (one pyramid splat per level) or "loop" (the reference's L*L splats per level)
        self.pyramid = cfg.pyramid if "pyramid" in cfg else "fused"
        # 3 outputs: optical flow + the splat weight map (FL:93-98)
        self.unet = UnetWithWarp(cfg, Unet(64, channels=6, out_dim=3, time_in=False, precision=cfg.precision), False, nan_safe=False)
        self.model = self.unet

    def configure_optimizers(self):                                         # FL:104-107
        from .optim import FusedAdam
        clip = getattr(self.cfg, "clip", 0.0) if "clip" in self.cfg else 0.0
        self.optimizers = FusedAdam(self.model.parameters(), lr=self.cfg.lr, weight_decay=self.cfg.weight_decay, max_grad_norm=clip)
        return self.optimizers

    def preprocess(self, batch, aug=True):
        """FL:114-130 (the torchvision Augmentor is outside this path: aug is accepted and ignored)."""
        img, tgt, flow = batch
        flow = torch.clamp(flow / self.flow_max, -1.0, 1.0)
        img = 2 * img - 1.0
        tgt = 2 * tgt - 1.0
        return tgt, torch.cat((img, tgt), dim=1), flow

    def loss(self, tgt, cond, flow_, override_flow=None):                   # FL:141-233
        if override_flow is None:
            out = self.model(cond, additional_out=True)
            flow_weight_pred = out[:, -3:]
            flow_pred = flow_weight_pred[:, :2] * self.flow_max
            warp_weights = flow_weight_pred[:, 2:]
        else:
            flow_pred = override_flow * self.flow_max
            warp_weights = torch.ones_like(flow_pred[:, :1])
        input_img = cond[:, :3]
        pyr = photometric_pyramid_loss_fused if self.pyramid == "fused" else photometric_pyramid_loss
        loss = pyr(input_img, flow_pred, warp_weights, tgt, self.levels)
        return loss + edgeaware_smoothness1(input_img, flow_pred) * 0.01    # FL:208-209

    def sample(self, cond, flo, log_additional=False):                      # FL:235-248
        out = self.model(cond, additional_out=True)
        flow_weight_pred = out[:, -3:]
        flow = flow_weight_pred[:, :2] * self.flow_max
        warp_weights = flow_weight_pred[:, 2:]
        sw = softsplat(cond[:, :3], flow, warp_weights, "soft", scale=1, offset=[0, 0])
        return fill_holes_nan(sw[:, :-1], sw[:, -1:]), flow, warp_weights

    def training_step(self, batch, batch_idx):                              # FL:288-309
        tgt, cond, flow = self.preprocess(batch, aug=self.cfg.train_aug)
        loss = self.loss(tgt, cond, flow)
        self.log_dict({
            "train/loss": loss,
            "train/cond_min": torch.min(cond), "train/cond_max": torch.max(cond), "train/cond_mean": torch.mean(cond),
            "train/cond_std": torch.mean(torch.std(cond, dim=0)),
            "train/flow_min": torch.min(flow), "train/flow_max": torch.max(flow), "train/flow_mean": torch.mean(flow),
            "train/flow_std": torch.mean(torch.std(flow, dim=0)),
        })
        self.log("loss", loss, prog_bar=True)
        return loss

    @torch.no_grad()
    def validation_step(self, batch, batch_idx):
        """FL:311-345 without the W&B image logging."""
        img, tgt, flow = batch
        tgt_, cond, flow_ = self.preprocess(batch, aug=False)
        loss = self.loss(tgt_, cond, flow_)
        ideal_loss = self.loss(tgt_, cond, flow_, override_flow=flow_)
        samples, p_flows, _ = self.sample(cond, flow_)
        samples = torch.where(torch.isnan(samples), torch.zeros_like(samples), samples)
        self.log_dict({"val/loss": loss, "val/ideal_loss": ideal_loss,
                       "val/mse": torch.nn.functional.mse_loss(samples, tgt),
                       "val/flow_mse": torch.nn.functional.mse_loss(flow_, p_flows / self.flow_max)}, sync_dist=True)
        return loss
