"""GPU-side batch augmentation with the call contract of the reference's `Augmentor` (augmentation.py:6-76):
`(img, tgt, flow) -> (img, tgt, flow)`, photometric jitter on the two frames, then geometric transforms on the whole stack with the
flow kept consistent.

The reference runs torchvision transforms per SAMPLE on the host (`itemize`, augmentation.py:58-64) and assumes square inputs
(`image_size = batch.shape[-1]`, the crop is resized to `(image_size, image_size)`, 45-49).  Here every transform is a batched tensor
op on the device (torchvision is not a dependency), the random decisions are drawn per sample from one `torch.Generator`, and height
and width are treated separately.  Same structure and probabilities:
  image_augs : colour jitter p = 0.4 (brightness / contrast / saturation factors 1 +- 0.1, hue omitted: its +-0.1 offset collapses
               to a ~0.01-wide band in the reference's parameterisation), grayscale p = 0.1, 3x3 Gaussian blur p = 0.2 (sigma <= 0.5);
               the SAME draw is applied to img and tgt (the reference applies one transform object to both halves, 9-13);
  whole_augs : horizontal flip p = 0.3, vertical flip p = 0.3, random resized crop p = 0.15 (scale 0.8-1.0, ratio 0.9-1.1) resized back
               to (H, W) with the flow vectors rescaled by the crop's zoom per axis.
Flips negate the flow component along the flipped axis (channel 0 = x, channel 1 = y, SS:368).  [The reference negates the OTHER
channel (augmentation.py:37-45: `batch[:, -1]` on a horizontal flip), which breaks the img -> tgt correspondence; pass
`reference_flip_channels=True` to reproduce it.]
"""
import torch
import torch.nn.functional as F


class Augmentor:
    def __init__(self, seed=None, reference_flip_channels=False):
        self.gen = None
        self.seed = seed
        self.ref_flip = reference_flip_channels

    def _rand(self, n, device):
        """per-sample uniforms: from the device's default generator (so `torch.manual_seed` and a checkpointed RNG state reproduce the
        augmentations), or from a private generator when a seed was given"""
        if self.seed is None:
            return torch.rand(n, device=device)
        if self.gen is None or self.gen.device != device:
            self.gen = torch.Generator(device=device)
            self.gen.manual_seed(self.seed)
        return torch.rand(n, device=device, generator=self.gen)

    # ---- photometric --------------------------------------------------------------------------------------------------------
    @staticmethod
    def _gray(x):
        return (0.299 * x[:, 0:1] + 0.587 * x[:, 1:2] + 0.114 * x[:, 2:3])

    def _image_augs(self, img, tgt):
        B, dev = img.shape[0], img.device
        pair = torch.stack((img, tgt), dim=1)                                  # (B, 2, 3, H, W): one draw per sample for both frames
        sel = lambda p: (self._rand(B, dev) < p).view(B, 1, 1, 1, 1)
        fac = lambda: (1.0 + (self._rand(B, dev) - 0.5) * 0.2).view(B, 1, 1, 1, 1)
        # colour jitter: brightness, contrast (about the mean gray level), saturation (about the gray image)
        jit = pair * fac()
        g = self._gray(jit.flatten(0, 1)).unflatten(0, (B, 2))
        jit = (jit - g.mean(dim=(2, 3, 4), keepdim=True)) * fac() + g.mean(dim=(2, 3, 4), keepdim=True)
        g = self._gray(jit.flatten(0, 1)).unflatten(0, (B, 2))
        jit = ((jit - g) * fac() + g).clamp(0.0, 1.0)
        pair = torch.where(sel(0.4), jit, pair)
        gray = self._gray(pair.flatten(0, 1)).unflatten(0, (B, 2)).expand(-1, -1, 3, -1, -1)
        pair = torch.where(sel(0.1), gray, pair)
        sigma = (self._rand(B, dev) * 0.5).clamp_min(1e-3)
        k1 = torch.exp(-0.5 * (torch.arange(-1, 2, device=dev).float()[None] / sigma[:, None]) ** 2)
        k1 = k1 / k1.sum(dim=1, keepdim=True)                                  # (B, 3) per-sample 1-D kernels
        k2 = (k1[:, :, None] * k1[:, None, :])                                 # (B, 3, 3)
        flat = pair.reshape(1, B * 6, *pair.shape[-2:])
        w = k2.repeat_interleave(6, dim=0)[:, None]                            # one depth-wise kernel per (sample, frame, channel)
        blur = F.conv2d(F.pad(flat, (1, 1, 1, 1), mode="reflect"), w, groups=B * 6).reshape(pair.shape)
        pair = torch.where(sel(0.2), blur, pair)
        return pair[:, 0], pair[:, 1]

    # ---- geometric ----------------------------------------------------------------------------------------------------------
    def _whole_augs(self, img, tgt, flow):
        B, _, H, W = img.shape
        dev = img.device
        stack = torch.cat((img, tgt, flow), dim=1)                            # (B, 8, H, W); channels 6, 7 = flow x, y
        cx, cy = (7, 6) if self.ref_flip else (6, 7)
        hf = (self._rand(B, dev) < 0.3).view(B, 1, 1, 1)
        fl = stack.flip(-1).clone()
        fl[:, cx] = -fl[:, cx]
        stack = torch.where(hf, fl, stack)
        vf = (self._rand(B, dev) < 0.3).view(B, 1, 1, 1)
        fl = stack.flip(-2).clone()
        fl[:, cy] = -fl[:, cy]
        stack = torch.where(vf, fl, stack)
        # random resized crop: area fraction in [0.8, 1], aspect change in [0.9, 1.1] (log-uniform), position uniform
        crop = self._rand(B, dev) < 0.15
        area = 0.8 + 0.2 * self._rand(B, dev)
        logr = (self._rand(B, dev) * 2 - 1) * torch.log(torch.tensor(1.1, device=dev))
        ratio = torch.exp(logr)
        ch = (torch.sqrt(area / ratio)).clamp(max=1.0)                        # crop height / width as fractions of the image
        cw = (torch.sqrt(area * ratio)).clamp(max=1.0)
        oy = self._rand(B, dev) * (1 - ch)
        ox = self._rand(B, dev) * (1 - cw)
        ch, cw = torch.where(crop, ch, torch.ones_like(ch)), torch.where(crop, cw, torch.ones_like(cw))
        oy, ox = torch.where(crop, oy, torch.zeros_like(oy)), torch.where(crop, ox, torch.zeros_like(ox))
        # sampling grid of the crop window, resized back to (H, W) (bilinear, pixel centres: align_corners=False)
        ys = (torch.arange(H, device=dev).float() + 0.5) / H
        xs = (torch.arange(W, device=dev).float() + 0.5) / W
        gy = (oy[:, None] + ch[:, None] * ys[None]) * 2 - 1                    # (B, H)
        gx = (ox[:, None] + cw[:, None] * xs[None]) * 2 - 1                    # (B, W)
        grid = torch.stack((gx[:, None, :].expand(B, H, W), gy[:, :, None].expand(B, H, W)), dim=-1)
        out = F.grid_sample(stack, grid, mode="bilinear", padding_mode="border", align_corners=False)
        out = torch.where(crop.view(B, 1, 1, 1), out, stack)
        scale = torch.ones(B, 8, 1, 1, device=dev)
        scale[:, 6, 0, 0] = 1.0 / cw                                           # a crop of width fraction cw is zoomed by 1 / cw along x
        scale[:, 7, 0, 0] = 1.0 / ch
        out = out * scale
        return out[:, :3], out[:, 3:6], out[:, 6:]

    def __call__(self, batch):
        img, tgt, flow = batch
        img, tgt = self._image_augs(img, tgt)
        return self._whole_augs(img, tgt, flow)
