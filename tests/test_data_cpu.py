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
