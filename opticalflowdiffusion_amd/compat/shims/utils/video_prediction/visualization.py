"""`from utils.video_prediction.visualization import log_video` (flow_diffuser.py:12; only called from commented-out lines)."""


def log_video(*args, **kwargs):
    return None
