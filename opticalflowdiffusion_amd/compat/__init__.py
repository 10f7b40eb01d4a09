"""What the reference's entry point needs around the FlowDiffuser plugin when Hydra / OmegaConf / Lightning / W&B are absent
(SURVEY 8f next-1): a composer for its `configurations/` tree (main.py:25-30), the `utils` package it imports but does not ship
(main.py:9, flow_diffuser.py:12-13) and an import alias so that `from algorithms.diffusion_animation import FlowDiffuser`
(experiments/exp_99.py:14) resolves to this engine.  See INTEGRATION.md."""
from .config import Config, compose  # noqa: F401
