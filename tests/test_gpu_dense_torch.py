"""GPU: the HIP path compared DIRECTLY with the second, independent restatement (oracle/dense_torch.py: dense
O(P*H*W) PyTorch, float64, gradients by autograd) -- not only through oracle/tile_ref.c.  All four input modes
(SH or precomputed colours x scale/rotation or precomputed covariance), colour + depth + alpha losses.
The integer tile rects of the HIP forward are imposed on the dense renderer (they are bit-exact with tile_ref,
tests/test_gpu_parity.py), so only differentiable arithmetic is compared.
Tolerances: images 3e-5 * (1 + |x|) (fp32 kernels vs float64), gradients 2e-4 relative L2 per tensor."""
import numpy as np
import pytest
import torch

from helpers import rel_l2, small_scene

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode", ["sh_sr", "col_sr", "sh_cov", "col_cov"])
@pytest.mark.parametrize("seed", [0, 1])
def test_hip_vs_dense_autograd(mode, seed):
    from oracle import dense_torch
    from binocular3dgs_amd import _C
    from binocular3dgs_amd.gaussian_model import covariance_from_scaling_rotation
    from test_gpu_parity import _run_hip_forward
    P, W, H = 1500, 64, 64
    d, _ = small_scene(P=P, W=W, H=H, seed=40 + seed, near_frac=0.05, scale_mu=0.08)
    g = torch.Generator().manual_seed(3 + seed)
    d["colors_precomp"] = torch.rand(P, 3, generator=g)
    d["cov3D_precomp"] = covariance_from_scaling_rotation(d["scales"], 1.0, d["rotations"])
    out = _run_hip_forward(d, mode)
    # integer decisions of the HIP forward: radius > 0 and the tile rect (derived exactly as the kernel does)
    radii = out["radii"].cpu()
    rec = out["views"]["records"].cpu()
    rad = radii.float()
    gx, gy = (W + 15) // 16, (H + 15) // 16
    cl = lambda v, hi: torch.clamp(v, 0, hi).to(torch.int64)  # noqa: E731
    rect = torch.stack([cl((rec[:, 0] - rad) / 16, gx), cl((rec[:, 1] - rad) / 16, gy),
                        cl((rec[:, 0] + rad + 15) / 16, gx), cl((rec[:, 1] + rad + 15) / 16, gy)], 1)
    rect[radii <= 0] = 0
    names = ["means3D", "opacities"] + (["colors_precomp"] if "col" in mode else ["shs"]) + \
        (["cov3D_precomp"] if "cov" in mode else ["scales", "rotations"])
    leaf = {k: d[k].double().clone().requires_grad_(True) for k in names}
    dd = {k: v for k, v in d.items() if k not in ("means3D", "opacities", "shs", "colors_precomp", "scales", "rotations",
                                                  "cov3D_precomp", "scale_modifier")}
    off = torch.zeros(P, 2, dtype=torch.float64, requires_grad=True)
    ref = dense_torch.render_dense(**dd, **leaf, rect=rect, pix_offset=off)
    assert np.array_equal(ref["radii"].numpy(), radii.numpy())
    for k in ("color", "depth", "alpha"):
        r = ref[k].detach().numpy()
        err = np.abs(out[k].cpu().numpy() - r) / (1 + np.abs(r))
        assert err.max() < 3e-5, (k, float(err.max()))
    gc, gd, ga = (torch.randn(c, H, W, generator=g, dtype=torch.float64) for c in (3, 1, 1))
    ((ref["color"] * gc).sum() + (ref["depth"] * gd).sum() + (ref["alpha"] * ga).sum()).backward()
    G, sh, colors, scales, rots, cov = out["inputs"]
    res = _C.rasterize_gaussians_backward(
        G["bg"], G["means3D"], out["radii"], colors, scales, rots, 1.0, cov, G["viewmatrix"], G["projmatrix"],
        d["tanfovx"], d["tanfovy"], gc.float().cuda(), gd.float().cuda(), ga.float().cuda(), sh, d["sh_degree"],
        G["campos"], out["geom"], out["n"], out["binning"], out["img"], out["alpha"], False)
    got = dict(zip(("dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales",
                    "dL_drotations"), res))
    want = {"dL_dmeans3D": leaf["means3D"].grad, "dL_dopacity": leaf["opacities"].grad.reshape(-1, 1)}
    if "col" in mode:
        want["dL_dcolors"] = leaf["colors_precomp"].grad
    else:
        want["dL_dsh"] = leaf["shs"].grad
    if "cov" in mode:
        # the extension reports the gradient of the 6 stored entries; off-diagonal entries appear twice in Sigma
        want["dL_dcov3D"] = leaf["cov3D_precomp"].grad
    else:
        want["dL_dscales"], want["dL_drotations"] = leaf["scales"].grad, leaf["rotations"].grad
    for k, r in want.items():
        e = rel_l2(got[k].cpu().numpy(), r.numpy())
        assert e <= 2e-4, f"{mode} {k}: rel L2 {e:.3e}"
    assert rel_l2(got["dL_dmeans2D"][:, :2].cpu().numpy(), off.grad.numpy() * np.array([0.5 * W, 0.5 * H])) <= 2e-4
