"""Drop-in module for the reference's `from diff_gaussian_rasterization import
GaussianRasterizationSettings, GaussianRasterizer` (gaussian_renderer/__init__.py:14).
Everything is implemented in binocular3dgs_amd (HIP, gfx950); this file only re-exports."""
from binocular3dgs_amd.rasterizer import (GaussianRasterizationSettings, GaussianRasterizer,  # noqa: F401
                                          rasterize_gaussians, _RasterizeGaussians, _C)

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians"]
