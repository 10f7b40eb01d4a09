"""`from utils.wandb_utils import download_latest_checkpoint, rewrite_checkpoint_for_compatibility` (main.py:9, 67-76).
Checkpoints live on W&B servers; this box has no network, so the download raises with the local alternative spelled out,
and the compatibility rewrite (Lightning checkpoint -> the triple-aliased key layout unet.* / _model.* / model.* the plugin's
state dict has) works on local files."""
import os

import torch


def download_latest_checkpoint(run_path, download_dir):
    raise RuntimeError(f"download_latest_checkpoint({run_path!r}): W&B is unreachable here; pass a local Lightning-layout checkpoint "
                       f"(train.py --resume <file>, or build_experiment(cfg, logger, ckpt_path))")


def rewrite_checkpoint_for_compatibility(path, allow_pickle=None, overwrite=False):
    """returns a path whose state dict carries every alias of the UNet's parameters (a checkpoint written from `.unet` alone
    loads into `FlowDiffuser`, whose modules `unet`, `_model` and `model.model` share them).

    The file is read with `torch.load(weights_only=True)` (tensors and plain containers only).  A checkpoint that needs the full
    unpickler (arbitrary objects = arbitrary code of whoever wrote the file) is refused unless the caller opts in with
    `allow_pickle=True` or OFD_TRUST_CHECKPOINTS=1.  An existing `<name>.compat.ckpt` is not overwritten unless `overwrite=True`."""
    if allow_pickle is None:
        allow_pickle = os.environ.get("OFD_TRUST_CHECKPOINTS", "0") not in ("", "0")
    try:
        ck = torch.load(path, map_location="cpu", weights_only=True)
    except Exception as e:
        if not allow_pickle:
            raise RuntimeError(f"{path}: not loadable with weights_only=True ({type(e).__name__}: {e}); it would need the full "
                               "unpickler, which executes code from the file -- pass allow_pickle=True (or OFD_TRUST_CHECKPOINTS=1) "
                               "only for checkpoints you trust") from e
        ck = torch.load(path, map_location="cpu", weights_only=False)
    sd = ck.get("state_dict", ck)
    unet = {k[len("unet."):]: v for k, v in sd.items() if k.startswith("unet.")}
    for k, v in unet.items():
        for prefix in ("_model.", "model.model."):
            sd.setdefault(prefix + k, v)
    out = os.path.splitext(str(path))[0] + ".compat.ckpt"
    if os.path.exists(out) and not overwrite:
        raise FileExistsError(f"{out} exists; pass overwrite=True to replace it")
    torch.save(ck, out)
    return out
