"""`from utils.image_prediction.logging import log_photos` (diffusion_animation.py:7; outside the FlowDiffuser path)."""


def log_photos(*args, **kwargs):
    return None
