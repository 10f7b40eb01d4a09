"""Import alias: with `opticalflowdiffusion_amd/compat/shims` ahead of the reference on sys.path,
`from algorithms.diffusion_animation import FlowDiffuser, FlowLearner` (experiments/exp_99.py:14) and
`from algorithms.diffusion_animation import Unet, ConditionalDiffusion` (__init__.py:1) resolve to the MI355X engine."""
from opticalflowdiffusion_amd import ConditionalDiffusion, FlowDiffuser, FlowLearner, Unet, UnetWithWarp, nan_mse, softsplat, warp  # noqa: F401
