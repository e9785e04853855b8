"""Drop-in module name: `from diff_gaussian_rasterization import GaussianRasterizationSettings,
GaussianRasterizer` (agents/manigaussian_bc/gaussian_renderer/__init__.py:14 in the reference) resolves
here when this repository is on sys.path.  Everything is implemented in manigaussian_amd (HIP, gfx950)."""
from manigaussian_amd import _C  # noqa: F401  (the reference exposes the native module under this name)
from manigaussian_amd.rasterizer import (GaussianRasterizationSettings, GaussianRasterizer,  # noqa: F401
                                         _RasterizeGaussians, rasterize_gaussians)

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians"]
