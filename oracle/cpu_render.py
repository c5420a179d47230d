"""TEST INFRASTRUCTURE: an oracle-backed stand-in for the rasterizer so that host logic
(render(), the view-sharded step, the reference's own render()) can be exercised on CPU, and so that bench.py's
`cpu_baseline` can time the reference-shaped CPU render path (render() with the *_python sub-steps on PyTorch-CPU around
oracle/tile_ref.c).  Never imported by the product (tests/test_abi.py checks)."""
import math

import numpy as np
import torch

from oracle import tile_ref


class OracleRasterize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3D, rs):
        def n(t):
            return None if t is None or t.numel() == 0 else t.detach().cpu().numpy()
        st = tile_ref.forward(means3D=n(means3D), opacities=n(opacities), shs=n(sh), colors_precomp=n(colors_precomp),
                              scales=n(scales), rotations=n(rotations), cov3D_precomp=n(cov3D),
                              viewmatrix=n(rs.viewmatrix), projmatrix=n(rs.projmatrix), campos=n(rs.campos), bg=n(rs.bg),
                              W=rs.image_width, H=rs.image_height, tanfovx=rs.tanfovx, tanfovy=rs.tanfovy,
                              sh_degree=rs.sh_degree, scale_modifier=rs.scale_modifier, prefiltered=rs.prefiltered)
        ctx.st = st
        ctx.flags = [t is not None and t.numel() != 0 for t in (sh, colors_precomp, scales, rotations, cov3D)]
        radii = torch.from_numpy(st.radii.copy())
        ctx.mark_non_differentiable(radii)
        return (torch.from_numpy(st.color.copy()), radii, torch.from_numpy(st.depth.copy()),
                torch.from_numpy(st.alpha.copy()))

    @staticmethod
    def backward(ctx, gc, gr, gd, ga):
        st = ctx.st
        H, W = st.H, st.W
        z3 = np.zeros((3, H, W), np.float32)
        g = tile_ref.backward(st, z3 if gc is None else gc.numpy(), None if gd is None else gd.numpy(),
                              None if ga is None else ga.numpy())
        t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a))  # noqa: E731
        has_sh, has_col, has_s, has_r, has_cov = ctx.flags
        return (t(g["dL_dmeans3D"]), t(g["dL_dmeans2D"]), t(g["dL_dsh"]) if has_sh else None,
                t(g["dL_dcolors"]) if has_col else None, t(g["dL_dopacity"]),
                t(g["dL_dscales"]) if has_s else None, t(g["dL_drotations"]) if has_r else None,
                t(g["dL_dcov3D"]) if has_cov else None, None)


class OracleRasterizer(torch.nn.Module):
    """Same call surface as GaussianRasterizer (gaussian_renderer/__init__.py:51,85-93)."""

    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        e = torch.empty(0)
        f = lambda t: e if t is None else t  # noqa: E731
        return OracleRasterize.apply(means3D, means2D, f(shs), f(colors_precomp), opacities, f(scales), f(rotations),
                                     f(cov3D_precomp), self.raster_settings)
