"""manigaussian_amd -- MI355X-native (gfx950) Gaussian-splatting hot path of ManiGaussian.

Scope: the differentiable tile rasterizer (RGB + feature channels, fwd+bwd) behind the
reference's GaussianRasterizer / GaussianRasterizationSettings API, and the deformation-field per-Gaussian
apply.  Native code: manigaussian_amd/csrc (HIP) -> libmgsplat.so, C ABI in include/mgsplat.h.
"""
from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer, rasterize_gaussians  # noqa: F401
from .views import GaussianRasterizerBatch  # noqa: F401
from .regressor import gaussian_epilogue  # noqa: F401
from .voxel import point_latent_pe  # noqa: F401
from . import camera  # noqa: F401
from ._state import (check_status, forward_mode, overflow_policy, set_forward_mode, set_headroom,  # noqa: F401
                     set_overflow_policy, set_safe_workspace)

__version__ = "0.1.0"
