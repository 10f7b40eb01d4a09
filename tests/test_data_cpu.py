"""Input side: the Middlebury .flo codec against a hand-built byte stream
(datasets/animation/sintel.py:59-65) and the procedural dataset's batch contract."""
import struct

import numpy as np
import pytest
import torch

from opticalflowdiffusion_amd.data import FLO_MAGIC, SyntheticFlowPairs, flo_to_tensor, read_flo, write_flo


def test_flo_known_bytes(tmp_path):
    # 3 wide, 2 high; (x, y) interleaved row-major, exactly what np.fromfile(...).resize((h, w, 2)) decodes in the reference
    vals = [1.5, -2.0, 0.0, 0.25, 7.0, 8.0, -1.0, -1.0, 3.0, 4.0, 100.0, -100.0]
    raw = b"PIEH" + struct.pack("<ii", 3, 2) + struct.pack("<12f", *vals)
    p = tmp_path / "a.flo"
    p.write_bytes(raw)
    f = read_flo(str(p))
    assert f.shape == (2, 3, 2) and f.dtype == np.float32
    assert f[0, 0].tolist() == [1.5, -2.0] and f[1, 2].tolist() == [100.0, -100.0]
    assert struct.unpack("<f", b"PIEH")[0] == FLO_MAGIC
    t = flo_to_tensor(f)
    assert t.shape == (2, 2, 3) and float(t[0, 0, 0]) == 1.5 and float(t[1, 0, 0]) == -2.0      # channel 0 = x
    q = tmp_path / "b.flo"
    write_flo(str(q), f)
    assert q.read_bytes() == raw


def test_flo_errors(tmp_path):
    p = tmp_path / "bad.flo"
    p.write_bytes(struct.pack("<fii", 1.0, 2, 2) + b"\0" * 32)
    with pytest.raises(ValueError, match="magic"):
        read_flo(str(p))
    p.write_bytes(b"PIEH" + struct.pack("<ii", 4, 4) + b"\0" * 8)
    with pytest.raises(ValueError, match="truncated"):
        read_flo(str(p))
    p.write_bytes(b"PI")
    with pytest.raises(ValueError, match="header"):
        read_flo(str(p))


def test_flo_resize_rescales_vectors():
    f = np.zeros((4, 8, 2), np.float32)
    f[..., 0], f[..., 1] = 2.0, -1.0
    t = flo_to_tensor(f, size=(8, 4))           # height x2, width /2
    assert t.shape == (2, 8, 4)
    assert torch.allclose(t[0], torch.full((8, 4), 1.0)) and torch.allclose(t[1], torch.full((8, 4), -2.0))


def test_synthetic_pairs_contract():
    ds = SyntheticFlowPairs(10, 32, 48, flow_max=20, seed=3)
    img, tgt, flow = ds[4]
    assert img.shape == (3, 32, 48) and tgt.shape == (3, 32, 48) and flow.shape == (2, 32, 48)
    assert 0.0 <= float(img.min()) and float(img.max()) <= 1.0 and float(flow.abs().max()) <= 20.0
    img2, _, flow2 = ds[4]
    assert torch.equal(img, img2) and torch.equal(flow, flow2)            # deterministic per index
    assert not torch.equal(flow, ds[5][2])
    with pytest.raises(IndexError):
        ds[10]


def test_sintel_pairs_adapter(tmp_path):
    """a miniature MPI-Sintel tree: (frame N, frame N+1, flow N) triples in the reference's batch contract, padded 436 -> 440
    style (here 10 -> 16), resized with rescaled vectors on request"""
    from PIL import Image
    from opticalflowdiffusion_amd.data import SintelPairs, write_flo
    root = tmp_path / "MPI_Sintel"
    rng = np.random.default_rng(3)
    frames = {}
    for scene, n in (("alley_1", 3), ("bamboo_2", 2)):
        (root / "training" / "clean" / scene).mkdir(parents=True)
        (root / "training" / "flow" / scene).mkdir(parents=True)
        for k in range(1, n + 1):
            a = rng.integers(0, 256, size=(10, 12, 3), dtype=np.uint8)
            frames[(scene, k)] = a
            Image.fromarray(a).save(root / "training" / "clean" / scene / f"frame_{k:04d}.png")
        for k in range(1, n):
            write_flo(root / "training" / "flow" / scene / f"frame_{k:04d}.flo", np.full((10, 12, 2), [k, -k], np.float32))
    ds = SintelPairs(str(root))
    assert len(ds) == 3                                           # alley_1: frames (1,2), (2,3); bamboo_2: (1,2)
    img, tgt, flow = ds[1]
    assert img.shape == tgt.shape == (3, 16, 16) and flow.shape == (2, 16, 16)
    assert torch.equal(img[:, :10, :12], torch.from_numpy(frames[("alley_1", 2)]).permute(2, 0, 1).float() / 255.0)
    assert torch.equal(tgt[:, :10, :12], torch.from_numpy(frames[("alley_1", 3)]).permute(2, 0, 1).float() / 255.0)
    assert torch.equal(img[:, 10:, :12], img[:, 9:10, :12].expand(-1, 6, -1))          # edge replication
    assert float(flow[0, 0, 0]) == 2.0 and float(flow[1, 0, 0]) == -2.0 and float(flow[:, 10:].abs().max()) == 0.0
    small = SintelPairs(str(root), scenes=["bamboo_2"], image_size=(20, 24), pad_to=1)
    i2, _, f2 = small[0]
    assert len(small) == 1 and i2.shape == (3, 20, 24) and f2.shape == (2, 20, 24)
    assert float(f2[0, 5, 5]) == pytest.approx(2.0) and float(f2[1, 5, 5]) == pytest.approx(-2.0)   # vectors rescaled by 24/12, 20/10
    with pytest.raises(FileNotFoundError):
        SintelPairs(str(tmp_path / "nothing"))


def test_gpu_augmentor_keeps_the_flow_consistent():
    """augmentation.Augmentor (the batched stand-in for the reference's torchvision pipeline, augmentation.py:6-76) on non-square
    inputs: shapes, determinism under the global seed, the flip negates the flow component ALONG the flipped axis so that a
    translated pair stays a translated pair, the crop rescales the vectors by its zoom per axis, grayscale / identity branches."""
    import torch
    from opticalflowdiffusion_amd.augmentation import Augmentor
    B, H, W = 64, 24, 40
    g = torch.Generator().manual_seed(0)
    img, tgt = torch.rand(B, 3, H, W, generator=g), torch.rand(B, 3, H, W, generator=g)
    flow = torch.zeros(B, 2, H, W)
    flow[:, 0], flow[:, 1] = 3.0, -2.0                                  # constant flow: every geometric transform keeps it constant
    torch.manual_seed(5)
    a = Augmentor()
    i1, t1, f1 = a((img, tgt, flow))
    torch.manual_seed(5)
    i2, t2, f2 = Augmentor()((img, tgt, flow))
    assert i1.shape == img.shape and t1.shape == tgt.shape and f1.shape == flow.shape
    assert torch.equal(i1, i2) and torch.equal(f1, f2)                  # reproducible from the global RNG state
    assert float(i1.min()) >= 0.0 and float(i1.max()) <= 1.0 and bool(torch.isfinite(f1).all())
    fx, fy = f1[:, 0, H // 2, W // 2], f1[:, 1, H // 2, W // 2]
    # per sample the flow stays constant over the image, and is +-3 / +-2 times the crop zoom (1 .. 1 / 0.8 / 0.9)
    assert float((f1[:, 0] - fx.view(B, 1, 1)).abs().max()) < 1e-4 and float((f1[:, 1] - fy.view(B, 1, 1)).abs().max()) < 1e-4
    zx, zy = fx.abs() / 3.0, fy.abs() / 2.0
    assert float(zx.min()) >= 1.0 - 1e-5 and float(zx.max()) < 1.45 and float(zy.min()) >= 1.0 - 1e-5 and float(zy.max()) < 1.45
    nflip_x, nflip_y, ncrop = int((fx < 0).sum()), int((fy > 0).sum()), int(((zx > 1 + 1e-4) | (zy > 1 + 1e-4)).sum())
    assert 8 <= nflip_x <= 32 and 8 <= nflip_y <= 32 and 2 <= ncrop <= 22      # p = 0.3, 0.3, 0.15 of 64
    # a horizontally flipped translated pair is a translated pair with the x displacement negated (channel 0 = x, SS:368)
    class OnlyFlipX(Augmentor):
        def draw(self, B, device):
            P = torch.zeros(B, self.NP, device=device)
            P[:, 1:4], P[:, 6], P[:, 7], P[:, 12:14] = 1.0, 0.25, 1.0, 1.0     # h-flip only
            return P
    base = torch.rand(1, 3, H, W + 8, generator=g)
    im, tg = base[..., 4:W + 4], base[..., 1:W + 1]                      # tg(x) = im(x - 3): flow = (+3, 0)
    fl = torch.zeros(1, 2, H, W)
    fl[:, 0] = 3.0
    i3, t3, f3 = OnlyFlipX()((im, tg, fl))
    assert torch.equal(i3, im.flip(-1)) and float(f3[:, 0].mean()) == -3.0 and float(f3[:, 1].abs().max()) == 0.0
    assert torch.allclose(t3[..., :-3], i3[..., 3:])                     # t3(x) = i3(x + 3), i.e. flow x = -3
    r = OnlyFlipX(reference_semantics=True)((im, tg, fl))[2]
    assert float(r[:, 0].mean()) == 3.0                                 # the reference's own channel choice leaves x untouched (augmentation.py:37-39)
    # crop scaling: ours divides each component by its axis' window fraction; reference_semantics multiplies channel 0 by the HEIGHT
    # fraction and channel 1 by the WIDTH fraction (augmentation.py:47-48: batch[:, -2:] / image_size * (h, w))
    class OnlyCrop(Augmentor):
        def draw(self, B, device):
            P = torch.zeros(B, self.NP, device=device)
            P[:, 1:4], P[:, 6], P[:, 9] = 1.0, 0.25, 1.0
            P[:, 10], P[:, 11], P[:, 12], P[:, 13] = 0.05, 0.1, 0.8, 0.9       # oy, ox, ch, cw
            return P
    fl2 = torch.zeros(1, 2, H, W)
    fl2[:, 0], fl2[:, 1] = 3.0, -2.0
    ours, ref = OnlyCrop()((im, tg, fl2))[2], OnlyCrop(reference_semantics=True)((im, tg, fl2))[2]
    assert torch.allclose(ours[:, 0], torch.full_like(ours[:, 0], 3.0 / 0.9)) and torch.allclose(ours[:, 1], torch.full_like(ours[:, 1], -2.0 / 0.8))
    assert torch.allclose(ref[:, 0], torch.full_like(ref[:, 0], 3.0 * 0.8)) and torch.allclose(ref[:, 1], torch.full_like(ref[:, 1], -2.0 * 0.9))
