"""binocular3dgs_amd -- MI355X-native differentiable Gaussian rasterizer (hot path of
hanl2010/Binocular3DGS: gaussian_renderer.render() -> diff_gaussian_rasterization).

Layout:
  csrc/                 hand-written HIP kernels + the C ABI (libb3gs_raster.so, include/b3gs_raster.h)
  _lib.py               ctypes binding of the C ABI (fails loudly when the library is missing)
  rasterizer.py         mirror of the `diff_gaussian_rasterization` Python surface (+ `_C` functions)
  render.py             mirror of gaussian_renderer.render()
  camera.py             camera matrices exactly as scene/cameras.py builds them, binocular shift
  gaussian_model.py     parameter store + the accessors render() reads (scene/gaussian_model.py:95-118)
  synth.py              deterministic synthetic scenes of BASELINE.md section 3
  step.py               view-sharded training step (loss block of train.py:123-149, RCCL all-reduce)
"""
from .rasterizer import (GaussianRasterizationSettings, GaussianRasterizer, rasterize_gaussians,  # noqa: F401
                         _C, _RasterizeGaussians)

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians", "_C",
           "_RasterizeGaussians"]
__version__ = "0.1.0"
