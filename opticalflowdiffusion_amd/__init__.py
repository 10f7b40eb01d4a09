"""MI355X-native FlowDiffuser hot path (gfx950): hand-written HIP kernels behind the reference's
`algorithms.diffusion_animation` plugin surface.  See DESIGN.md / INTEGRATION.md."""
from .warp import warp, nan_mse, scale, warp_forward_flow, warp_backward_flow  # noqa: F401
from .softsplat import softsplat  # noqa: F401
from .denoising_diffusion import Unet, ConditionalDiffusion  # noqa: F401,E402
from .flow_diffuser import FlowDiffuser, UnetWithWarp  # noqa: F401,E402
from .flow_learner import FlowLearner  # noqa: F401,E402
