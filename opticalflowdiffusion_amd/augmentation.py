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
Deviations from the reference, all on purpose and all listed in INTEGRATION.md:
  * flips negate the flow component along the flipped axis (channel 0 = x, channel 1 = y, SS:368); the reference negates the OTHER
    channel (augmentation.py:37-45: `batch[:, -1]` on a horizontal flip), which breaks the img -> tgt correspondence;
  * the crop divides each flow component by its axis' window fraction (a window of width fraction cw is zoomed by 1 / cw); the
    reference MULTIPLIES, and pairs channel -2 with the height and channel -1 with the width (augmentation.py:47-48);
  * hue jitter is dropped (above); the random factors are drawn per SAMPLE (the reference draws ColorJitter's factors once per call of
    the transform and applies the transform sample by sample, 58-64 -- per sample as well, but from torchvision's own RNG stream).
`reference_semantics=True` reproduces the first two (the flow arithmetic); nothing reproduces torchvision's RNG stream.
"""
import torch
import torch.nn.functional as F


class Augmentor:
    """`Augmentor()(img, tgt, flow)`.  CUDA tensors go through the one-pass HIP kernel (`ofd_augment`); the same table of per-sample
    decisions drives `apply_torch`, the tensor-op restatement used on CPU tensors and as the kernel's checker."""

    NP = 16     # columns of the parameter table (include/ofd.h: ofd_augment)

    def __init__(self, seed=None, reference_semantics=False, reference_flip_channels=None):
        self.gen = None
        self.seed = seed
        if reference_flip_channels is not None:          # old name of the flag (it covered the flips only)
            reference_semantics = reference_flip_channels
        self.ref_flip = bool(reference_semantics)        # the reference's flow arithmetic: swapped flip channels AND its crop scaling

    def _rand(self, shape, device):
        """uniforms from the device's default generator (so `torch.manual_seed` and a checkpointed RNG state reproduce the
        augmentations), or from a private generator when a seed was given"""
        if self.seed is None:
            return torch.rand(shape, device=device)
        if self.gen is None or self.gen.device != device:
            self.gen = torch.Generator(device=device)
            self.gen.manual_seed(self.seed)
        return torch.rand(shape, device=device, generator=self.gen)

    def draw(self, B, device):
        """(B, 16) table: 0 jitter on, 1 brightness, 2 contrast, 3 saturation, 4 grayscale on, 5 blur on, 6 sigma, 7 h-flip, 8 v-flip,
        9 crop on, 10 oy, 11 ox, 12 ch, 13 cw -- ONE random draw per call, no host synchronisation"""
        u = self._rand((B, 14), device)
        if u.is_cuda:                                       # the same table in one launch (ofd_augment_table) instead of ~50 tensor ops
            from . import _lib as L
            P = torch.empty(B, self.NP, device=device)
            L.check(L.lib().ofd_augment_table(L.ptr(u), L.ptr(P), B, L.stream()))
            return P
        P = torch.zeros(B, self.NP, device=device)
        P[:, 0] = (u[:, 0] < 0.4).float()
        P[:, 1:4] = 1.0 + (u[:, 1:4] - 0.5) * 0.2                              # brightness, contrast, saturation in 1 +- 0.1
        P[:, 4] = (u[:, 4] < 0.1).float()
        P[:, 5] = (u[:, 5] < 0.2).float()
        P[:, 6] = (u[:, 6] * 0.5).clamp_min(0.05)                              # Gaussian sigma (GaussianBlur(3, sigma), augmentation.py:26-27)
        P[:, 7] = (u[:, 7] < 0.3).float()
        P[:, 8] = (u[:, 8] < 0.3).float()
        crop = u[:, 9] < 0.15
        area = 0.8 + 0.2 * u[:, 10]                                            # RandomResizedCrop scale (0.8, 1.0), ratio (0.9, 1.1)
        ratio = torch.exp((u[:, 11] * 2 - 1) * 0.09531018)                     # log-uniform in [1 / 1.1, 1.1]
        ch = torch.sqrt(area / ratio).clamp(max=1.0)
        cw = torch.sqrt(area * ratio).clamp(max=1.0)
        one, zero = torch.ones_like(ch), torch.zeros_like(ch)
        P[:, 9] = crop.float()
        P[:, 12] = torch.where(crop, ch, one)
        P[:, 13] = torch.where(crop, cw, one)
        P[:, 10] = torch.where(crop, u[:, 12] * (1 - ch), zero)
        P[:, 11] = torch.where(crop, u[:, 13] * (1 - cw), zero)
        return P

    # ---- tensor-op restatement ----------------------------------------------------------------------------------------------
    @staticmethod
    def _gray(x):
        return (0.299 * x[:, 0:1] + 0.587 * x[:, 1:2] + 0.114 * x[:, 2:3])

    def apply_torch(self, img, tgt, flow, P):
        B, _, H, W = img.shape
        dev = img.device
        v5 = lambda c: P[:, c].view(B, 1, 1, 1, 1)
        pair = torch.stack((img, tgt), dim=1).float()                          # (B, 2, 3, H, W): the same draw for both frames
        # colour jitter: brightness, contrast about the mean gray level, saturation about the gray image
        jit = pair * v5(1)
        g = self._gray(jit.flatten(0, 1)).unflatten(0, (B, 2))
        m = g.mean(dim=(2, 3, 4), keepdim=True)
        jit = (jit - m) * v5(2) + m
        g = self._gray(jit.flatten(0, 1)).unflatten(0, (B, 2))
        jit = ((jit - g) * v5(3) + g).clamp(0.0, 1.0)
        pair = torch.where(v5(0) != 0, jit, pair)
        gray = self._gray(pair.flatten(0, 1)).unflatten(0, (B, 2)).expand(-1, -1, 3, -1, -1)
        pair = torch.where(v5(4) != 0, gray, pair)
        e = torch.exp(-0.5 / (P[:, 6] * P[:, 6]))
        k1 = torch.stack((e, torch.ones_like(e), e), dim=1) / (1.0 + 2.0 * e)[:, None]      # (B, 3) per-sample 1-D kernels
        k2 = k1[:, :, None] * k1[:, None, :]
        flat = pair.reshape(1, B * 6, H, W)
        blur = F.conv2d(F.pad(flat, (1, 1, 1, 1), mode="reflect"), k2.repeat_interleave(6, dim=0)[:, None], groups=B * 6).reshape(pair.shape)
        pair = torch.where(v5(5) != 0, blur, pair)
        # geometric: flips (the flow component along the flipped axis changes sign), then the crop window resized back to (H, W)
        stack = torch.cat((pair[:, 0], pair[:, 1], flow.float()), dim=1)      # (B, 8, H, W); channels 6, 7 = flow x, y
        cx, cy = (7, 6) if self.ref_flip else (6, 7)
        v4 = lambda c: P[:, c].view(B, 1, 1, 1)
        fl = stack.flip(-1).clone()
        fl[:, cx] = -fl[:, cx]
        stack = torch.where(v4(7) != 0, fl, stack)
        fl = stack.flip(-2).clone()
        fl[:, cy] = -fl[:, cy]
        stack = torch.where(v4(8) != 0, fl, stack)
        ys = (torch.arange(H, device=dev).float() + 0.5) / H
        xs = (torch.arange(W, device=dev).float() + 0.5) / W
        gy = (P[:, 10, None] + P[:, 12, None] * ys[None]) * 2 - 1              # (B, H): pixel centres, align_corners=False
        gx = (P[:, 11, None] + P[:, 13, None] * xs[None]) * 2 - 1
        grid = torch.stack((gx[:, None, :].expand(B, H, W), gy[:, :, None].expand(B, H, W)), dim=-1)
        out = F.grid_sample(stack, grid, mode="bilinear", padding_mode="border", align_corners=False)
        out = torch.where(v4(9) != 0, out, stack)
        if self.ref_flip:          # augmentation.py:47-48: channel -2 times the height fraction, channel -1 times the width fraction
            out = torch.cat((out[:, :6], out[:, 6:7] * v4(12), out[:, 7:8] * v4(13)), dim=1)
        else:                      # a window of width fraction cw is zoomed by 1 / cw
            out = torch.cat((out[:, :6], out[:, 6:7] / v4(13), out[:, 7:8] / v4(12)), dim=1)
        return out[:, :3], out[:, 3:6], out[:, 6:]

    def apply_hip(self, img, tgt, flow, P):
        from . import _lib as L
        img, tgt, flow, P = L.f32c(img), L.f32c(tgt), L.f32c(flow), L.f32c(P)
        B, _, H, W = img.shape
        o_img, o_tgt, o_flow = torch.empty_like(img), torch.empty_like(tgt), torch.empty_like(flow)
        means = torch.empty(B * 2, dtype=torch.float64, device=img.device)
        L.check(L.lib().ofd_augment(L.ptr(img), L.ptr(tgt), L.ptr(flow), L.ptr(P), L.ptr(means), L.ptr(o_img), L.ptr(o_tgt), L.ptr(o_flow),
                                    B, H, W, int(self.ref_flip), L.stream()))
        return o_img, o_tgt, o_flow

    def __call__(self, batch):
        img, tgt, flow = batch
        P = self.draw(img.shape[0], img.device)
        return self.apply_hip(img, tgt, flow, P) if img.is_cuda else self.apply_torch(img, tgt, flow, P)
