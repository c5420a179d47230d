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
import os as _os
import sysconfig as _sysconfig

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians", "_C",
           "_RasterizeGaussians"]
__version__ = "0.1.0"

_EXT = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "_C" + (_sysconfig.get_config_var("EXT_SUFFIX") or ".so"))
if _os.path.exists(_EXT):
    from .rasterizer import (GaussianRasterizationSettings, GaussianRasterizer, rasterize_gaussians,  # noqa: F401
                             _C, _RasterizeGaussians)
else:
    # A tree that has not been built yet: `python -m binocular3dgs_amd.build` (and __graft_entry__.build()) must be able to
    # import the package to reach build.py; every product name fails loudly until the compiled module exists -- there is no
    # python or CPU stand-in for it.
    def __getattr__(name):
        if name not in __all__:
            raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
        if not _os.path.exists(_EXT):
            raise ImportError(f"{_EXT} has not been built: run `python -m binocular3dgs_amd.build` "
                              "(hipcc for gfx950 + g++ against the torch headers); there is no fallback")
        import importlib
        if name == "_C":
            return importlib.import_module("._C", __name__)
        return getattr(importlib.import_module(".rasterizer", __name__), name)
