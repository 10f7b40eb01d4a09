"""TEST INFRASTRUCTURE ONLY -- CPU restatement ("oracle") of the FlowDiffuser hot path.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker.  The shipped path
(``opticalflowdiffusion_amd``) never imports this package and raises when its
HIP library is missing.

Every function cites the reference file:line it restates
(``DD`` = algorithms/diffusion_animation/denoising_diffusion.py,
``FD`` = flow_diffuser.py, ``WP`` = warp.py, ``SS`` = softsplat_new.py).

Pinning status (see DESIGN.md "Oracle"):
  * UNet / diffusion / grid_sample warp / nan_mse: pinned to golden vectors
    produced by importing the reference's own modules on CPU
    (tests/golden/make_goldens.py, fixtures in tests/golden/*.npz).
  * forward splat (softsplat_out / _ingrad / _flowgrad): the reference kernels
    are CUDA-only strings (SS:444 ``assert False`` on CPU), so there is nothing
    executable to pin against: PARITY UNPINNED beyond the known-answer case and
    the two properties of warp_test.py plus analytic cases (tests/test_oracle_splat.py).
"""
