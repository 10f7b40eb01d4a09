"""Input side of the FlowDiffuser path: the Middlebury `.flo` codec the reference's Sintel
loader uses (datasets/animation/sintel.py:59-65) and an offline procedural dataset with the reference's
batch contract `(img in [0,1] (3,H,W), tgt (3,H,W), flow px (2,H,W); flow channel 0 = x)`.

Host-side file / tensor plumbing only -- no arithmetic of the hot path lives here.
"""
import struct

import numpy as np
import torch

FLO_MAGIC = 202021.25          # 'PIEH' as little-endian float32


def read_flo(path):
    """sintel.py:59-65: float32 magic, int32 width, int32 height, then h*w*2 float32 (x, y interleaved).
    Returns (h, w, 2) float32.  Unlike the reference the magic and the payload size are checked."""
    with open(path, "rb") as f:
        head = f.read(12)
        if len(head) != 12:
            raise ValueError(f"{path}: truncated .flo header")
        magic, w, h = struct.unpack("<fii", head)
        if magic != FLO_MAGIC:
            raise ValueError(f"{path}: bad .flo magic {magic!r} (expected {FLO_MAGIC})")
        if w <= 0 or h <= 0 or w * h > (1 << 28):
            raise ValueError(f"{path}: implausible .flo size {w}x{h}")
        data = np.fromfile(f, np.float32, count=h * w * 2)
    if data.size != h * w * 2:
        raise ValueError(f"{path}: truncated .flo payload ({data.size} of {h * w * 2} floats)")
    return data.reshape(h, w, 2)


def write_flo(path, flow_hw2):
    a = np.ascontiguousarray(np.asarray(flow_hw2, dtype=np.float32))
    if a.ndim != 3 or a.shape[2] != 2:
        raise ValueError("write_flo expects (h, w, 2)")
    with open(path, "wb") as f:
        f.write(struct.pack("<fii", FLO_MAGIC, a.shape[1], a.shape[0]))
        a.tofile(f)


def flo_to_tensor(flow_hw2, size=None):
    """(h, w, 2) -> (2, H, W) tensor, channel 0 = x displacement.  With `size=(H, W)` the field is resized
    bilinearly AND rescaled (the reference resizes without rescaling the vectors, sintel.py:80)."""
    t = torch.from_numpy(np.ascontiguousarray(flow_hw2)).permute(2, 0, 1).float()
    if size is not None and tuple(size) != tuple(t.shape[-2:]):
        h, w = t.shape[-2:]
        t = torch.nn.functional.interpolate(t[None], size=tuple(size), mode="bilinear", align_corners=False)[0]
        t[0] *= size[1] / w
        t[1] *= size[0] / h
    return t


class SyntheticFlowPairs(torch.utils.data.Dataset):
    """Offline stand-in for the Sintel frame pairs (the files are absent, sintel.py:19-21): smooth random colour
    images and Sintel-like flow magnitudes (N(0, sigma^2) px box-filtered 9x9, clamped to +-flow_max), SURVEY 8d.
    Deterministic per (seed, index); every rank / worker can draw its own disjoint indices."""

    def __init__(self, length, height, width, flow_sigma=8.0, flow_max=20.0, seed=0):
        self.length, self.h, self.w = int(length), int(height), int(width)
        self.sigma, self.fmax, self.seed = float(flow_sigma), float(flow_max), int(seed)

    def __len__(self):
        return self.length

    def sample(self, i, device="cpu"):
        """sample i generated ON `device` (deterministic per (seed, index, device type): the CPU and GPU generators differ)."""
        if not 0 <= i < self.length:
            raise IndexError(i)
        g = torch.Generator(device=device).manual_seed(self.seed * 1000003 + i)
        box = lambda t, k: torch.nn.functional.avg_pool2d(t[None], k, 1, k // 2)[0]
        img = box(torch.rand(3, self.h, self.w, generator=g, device=device), 5)
        img = (img - img.amin()) / (img.amax() - img.amin() + 1e-8)
        flow = box(torch.randn(2, self.h, self.w, generator=g, device=device) * self.sigma * 9.0, 9).clamp(-self.fmax, self.fmax)
        return img, img.clone(), flow

    def __getitem__(self, i):
        return self.sample(i, "cpu")

    def batch(self, first, count, device):
        """`count` consecutive samples (wrapping around) stacked on `device` -- the input pipeline of train.py: generating on
        the GPU keeps the host out of the step (the CPU box filters cost 50 ms per sample)."""
        return tuple(torch.stack(x) for x in zip(*(self.sample((first + k) % self.length, device) for k in range(count))))


class SintelPairs(torch.utils.data.Dataset):
    """MPI-Sintel frame pairs in the reference's batch contract.  Directory layout of the public archive:
    `<root>/training/<render>/<scene>/frame_NNNN.png` and `<root>/training/flow/<scene>/frame_NNNN.flo` (flow of frame N -> N+1).
    Item i = (frame N, frame N+1, flow N) as `(img in [0,1] (3,H,W), tgt (3,H,W), flow px (2,H,W), channel 0 = x)` -- the
    `(frame2, frame3, flow)` triple the reference's loader builds (datasets/animation/sintel.py:22-50, 68-104), without its
    hard-coded paths and ImageNet normalisation (FlowDiffuser.preprocess expects [0,1], FD:150).  `image_size=(H, W)` resizes
    bilinearly and rescales the flow vectors (flo_to_tensor); `pad_to=8` pads H, W up to multiples of 8 by edge replication for the
    images and zeros for the flow (436 -> 440: three 2x down-samplings in the UNet, DD:95-99)."""

    def __init__(self, root, render="clean", scenes=None, image_size=None, pad_to=8):
        import os
        self.image_size, self.pad_to = (tuple(image_size) if image_size else None), int(pad_to)
        frames_root = os.path.join(root, "training", render)
        flow_root = os.path.join(root, "training", "flow")
        if not os.path.isdir(frames_root) or not os.path.isdir(flow_root):
            raise FileNotFoundError(f"{root}: expected training/{render}/<scene>/frame_NNNN.png and training/flow/<scene>/frame_NNNN.flo")
        self.items = []
        for scene in sorted(os.listdir(flow_root)):
            if scenes is not None and scene not in scenes:
                continue
            for fn in sorted(os.listdir(os.path.join(flow_root, scene))):
                if not fn.endswith(".flo"):
                    continue
                n = int(fn[len("frame_"):-len(".flo")])
                a = os.path.join(frames_root, scene, f"frame_{n:04d}.png")
                b = os.path.join(frames_root, scene, f"frame_{n + 1:04d}.png")
                if os.path.exists(a) and os.path.exists(b):
                    self.items.append((a, b, os.path.join(flow_root, scene, fn)))
        if not self.items:
            raise FileNotFoundError(f"{root}: no (frame, next frame, flow) triples found")

    def __len__(self):
        return len(self.items)

    @staticmethod
    def _png(path):
        from PIL import Image
        with Image.open(path) as im:
            a = np.asarray(im.convert("RGB"), dtype=np.float32) / 255.0
        return torch.from_numpy(a).permute(2, 0, 1).contiguous()

    def __getitem__(self, i):
        pa, pb, pf = self.items[i]
        img, tgt = self._png(pa), self._png(pb)
        flow_hw2 = read_flo(pf)
        if tuple(flow_hw2.shape[:2]) != tuple(img.shape[-2:]):
            raise ValueError(f"{pf}: flow {flow_hw2.shape[:2]} does not match frame {tuple(img.shape[-2:])}")
        if self.image_size is not None:
            img = torch.nn.functional.interpolate(img[None], size=self.image_size, mode="bilinear", align_corners=False)[0]
            tgt = torch.nn.functional.interpolate(tgt[None], size=self.image_size, mode="bilinear", align_corners=False)[0]
        flow = flo_to_tensor(flow_hw2, self.image_size)
        if self.pad_to > 1:
            h, w = img.shape[-2:]
            ph, pw = (-h) % self.pad_to, (-w) % self.pad_to
            if ph or pw:
                img = torch.nn.functional.pad(img[None], (0, pw, 0, ph), mode="replicate")[0]
                tgt = torch.nn.functional.pad(tgt[None], (0, pw, 0, ph), mode="replicate")[0]
                flow = torch.nn.functional.pad(flow, (0, pw, 0, ph))
        return img, tgt, flow
