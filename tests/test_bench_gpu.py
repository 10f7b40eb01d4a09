"""bench.py's contract on the GPU box: one JSON line carrying the headline metric, `roofline`, `warp` and `train`; and
`python bench.py --gpus N` starting its own N ranks as a child process (the reference gets its ranks from Lightning's
devices="auto", experiments/exp_base.py:193-206).  Small shapes: the numbers themselves are the driver's business."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

SMALL = ["--batch", "2", "--height", "64", "--width", "96", "--steps", "2", "--warmup", "1", "--profile-steps", "1",
         "--train-steps", "1", "--train-warmup", "1", "--no-cpu-baseline"]


def _run(extra, env=None):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + SMALL + extra, cwd=ROOT, env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return lines[0]


def test_bench_line_carries_roofline_warp_and_train():
    d = _run([])
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["unit"] == "denoise_steps/s" and d["value"] > 0
    assert d["hip_events_in_timed_region"] is False
    assert d["config"]["workload"].startswith("custom shape") and "(2,2,64,96)" in d["config"]["workload"]
    assert d["roofline"]["bound"] == "mfma" and d["roofline"]["launches"] > 0 and d["roofline"]["achieved"] > 0
    for k in ("splat_fwd", "grid_warp_fwd"):
        assert d["warp"][k]["ms"] > 0 and 0 < d["warp"][k]["frac_of_hbm_peak"] < 1
    assert d["warp"]["shape"] == [2, 64, 96]
    assert d["train"]["value"] > 0 and "augmentation" in d["train"]["includes"]


def test_bench_gpus_2_starts_its_own_ranks():
    """two ranks share this box's one GPU (OFD_FORCE_DEVICE) and talk over gloo (RCCL wants a device per rank)"""
    d = _run(["--gpus", "2"], env={"OFD_DIST_BACKEND": "gloo", "OFD_FORCE_DEVICE": "0"})
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 4 and d["config"]["parallelism"] == "replicas x2"
    assert d["train"]["global_batch"] == 4 and "gloo" in d["train"]["grad_sync"]
