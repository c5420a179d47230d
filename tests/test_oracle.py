"""CPU: trust in the oracle itself.  The reference ships no rasterizer source and no tests, so the
oracle is pinned by (i) analytic known-answer cases KA1-KA8 of SURVEY.md 8c, (ii) an independent
dense PyTorch restatement (oracle/dense_torch.py) for images AND autograd gradients (KA10),
(iii) fp64 finite differences through the dense restatement (KA9)."""
import math

import numpy as np
import pytest
import torch

from helpers import oracle_kwargs, rel_l2, small_scene
from oracle import dense_torch, tile_ref

from binocular3dgs_amd.camera import Camera


def _cam(W=64, H=64, fov=60.0):
    f = math.radians(fov)
    return Camera(np.eye(3), np.zeros(3), f, f, W, H)


def _one(cam, means, scales, opac, colors, rot=None, bg=(0.2, 0.3, 0.4), **kw):
    P = len(means)
    rot = np.tile(np.array([1, 0, 0, 0], np.float32), (P, 1)) if rot is None else rot
    return tile_ref.forward(means3D=np.asarray(means, np.float32), opacities=np.asarray(opac, np.float32),
                            colors_precomp=np.asarray(colors, np.float32), scales=np.asarray(scales, np.float32),
                            rotations=rot, viewmatrix=cam.world_view_transform.numpy(),
                            projmatrix=cam.full_proj_transform.numpy(), campos=cam.camera_center.numpy(), bg=bg,
                            W=cam.image_width, H=cam.image_height, tanfovx=math.tan(cam.FoVx / 2),
                            tanfovy=math.tan(cam.FoVy / 2), **kw)


def test_ka1_single_isotropic_gaussian_on_axis():
    cam = _cam(65, 65)      # odd size: pixel (32,32) is exactly on the optical axis
    s, z, op = 0.05, 4.0, 0.6
    st = _one(cam, [[0, 0, z]], [[s, s, s]], [op], [[1.0, 0.5, 0.25]])
    f = 65 / (2 * math.tan(math.radians(30)))
    var = (f * s / z) ** 2 + 0.3
    # lambda_max = mid + sqrt(max(0.1, mid^2 - det)); isotropic -> the 0.1 floor is active
    assert st.radii[0] == math.ceil(3 * math.sqrt(var + math.sqrt(0.1)))
    np.testing.assert_allclose(st.means2D[0], [32.0, 32.0], atol=1e-4)
    np.testing.assert_allclose(st.conic_opacity[0], [1 / var, 0, 1 / var, op], rtol=1e-5, atol=1e-7)
    a = op
    np.testing.assert_allclose(st.color[:, 32, 32], a * np.array([1.0, 0.5, 0.25]) + (1 - a) * np.array([0.2, 0.3, 0.4]), rtol=1e-5)
    np.testing.assert_allclose(st.depth[0, 32, 32], a * z, rtol=1e-5)
    np.testing.assert_allclose(st.alpha[0, 32, 32], a, rtol=1e-5)
    # one pixel to the right: alpha = op * exp(-0.5/var)
    np.testing.assert_allclose(st.alpha[0, 32, 33], op * math.exp(-0.5 / var), rtol=1e-5)
    # opacity above the cap is clamped to 0.99
    st2 = _one(cam, [[0, 0, z]], [[s, s, s]], [1.0], [[1, 1, 1]])
    np.testing.assert_allclose(st2.alpha[0, 32, 32], 0.99, rtol=1e-6)


def test_ka2_depth_order_and_swap():
    cam = _cam(65, 65)
    col = [[1, 0, 0], [0, 1, 0]]
    near_first = _one(cam, [[0, 0, 3.0], [0, 0, 5.0]], [[0.2] * 3] * 2, [0.5, 0.5], col, bg=(0, 0, 0))
    swapped = _one(cam, [[0, 0, 5.0], [0, 0, 3.0]], [[0.2] * 3] * 2, [0.5, 0.5], col, bg=(0, 0, 0))
    np.testing.assert_allclose(near_first.color[:, 32, 32], [0.5, 0.25, 0], rtol=1e-5)
    np.testing.assert_allclose(swapped.color[:, 32, 32], [0.25, 0.5, 0], rtol=1e-5)
    assert list(near_first.point_list[:2]) != list(swapped.point_list[:2]) or near_first.N == 0


def test_ka3_near_plane_cull():
    cam = _cam()
    st = _one(cam, [[0, 0, 0.2], [0, 0, 0.2001], [0, 0, -1.0]], [[0.01] * 3] * 3, [0.5] * 3, [[1, 1, 1]] * 3)
    assert st.radii[0] == 0 and st.radii[1] > 0 and st.radii[2] == 0


def test_ka4_alpha_below_1_over_255_contributes_nothing():
    cam = _cam(65, 65)
    st = _one(cam, [[0, 0, 4.0]], [[0.05] * 3], [1.0 / 256.0], [[1, 1, 1]], bg=(0, 0, 0))
    assert st.radii[0] > 0 and float(np.abs(st.color).max()) == 0.0 and int(st.n_contrib.max()) == 0


def test_ka5_saturation_terminates_and_records_n_contrib():
    cam = _cam(65, 65)
    n = 12
    means = [[0, 0, 2.0 + 0.1 * i] for i in range(n)]
    st = _one(cam, means, [[0.5] * 3] * n, [0.95] * n, [[1, 1, 1]] * n, bg=(0, 0, 0))
    # T after k blends = 0.05^k (alpha = 0.95 at the centre); stop before the first k with T*(1-a) < 1e-4:
    # 0.05^3 = 1.25e-4 >= 1e-4 (blend 3), 0.05^4 = 6.25e-6 < 1e-4 -> 3 contributors
    assert st.n_contrib[32, 32] == 3
    np.testing.assert_allclose(st.final_T[32, 32], 0.05 ** 3, rtol=1e-4)
    np.testing.assert_allclose(st.alpha[0, 32, 32], 1 - 0.05 ** 3, rtol=1e-5)


def test_ka6_gaussian_straddling_four_tiles():
    cam = _cam(64, 64)
    # project to pixel (15.5+16, 15.5+16) = corner shared by tiles (1,1),(2,1),(1,2),(2,2) with a small radius
    f = 64 / (2 * math.tan(math.radians(30)))
    z = 4.0
    x = (31.5 - 31.5) * z / f
    st = _one(cam, [[x, x, z]], [[0.01] * 3], [0.8], [[1, 1, 1]])
    assert st.radii[0] >= 2 and st.tiles_touched[0] == 4
    ranges_nonempty = int(((st.ranges[:, 1] - st.ranges[:, 0]) > 0).sum())
    assert ranges_nonempty == 4


def test_ka7_offscreen_gaussian_has_radius_zero():
    cam = _cam()
    st = _one(cam, [[50.0, 0, 4.0]], [[0.05] * 3], [0.5], [[1, 1, 1]])
    assert st.radii[0] == 0 and st.N == 0
    np.testing.assert_allclose(st.color, np.broadcast_to(np.array([0.2, 0.3, 0.4], np.float32).reshape(3, 1, 1), st.color.shape))


def test_ka8_equal_depth_ties_keep_index_order():
    cam = _cam()
    P = 9
    means = [[0.01 * (i % 3), 0.0, 3.0] for i in range(P)]    # identical depth bits
    st = _one(cam, means, [[0.05] * 3] * P, [0.3] * P, [[1, 1, 1]] * P)
    for t in range(st.ranges.shape[0]):
        r0, r1 = st.ranges[t]
        ids = st.point_list[r0:r1]
        assert list(ids) == sorted(ids), "stable sort: equal keys must stay in Gaussian-index order"
    keys = st.keys
    assert np.all(keys[:-1] <= keys[1:])


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_ka10_tile_oracle_vs_dense_oracle_images_and_gradients(seed):
    P, W, H = 500, 64, 64
    d, _ = small_scene(P=P, W=W, H=H, seed=seed, near_frac=0.05)
    st = tile_ref.forward(**oracle_kwargs(d))
    leaf = {k: d[k].double().clone().requires_grad_(True) for k in ("means3D", "opacities", "scales", "rotations", "shs")}
    off = torch.zeros(P, 2, dtype=torch.float64, requires_grad=True)
    out = dense_torch.render_dense(**{**d, **leaf}, rect=torch.from_numpy(st.rect), pix_offset=off)
    assert np.array_equal(out["radii"].numpy(), st.radii)
    for k in ("color", "depth", "alpha"):
        err = np.abs(out[k].detach().numpy() - getattr(st, k)) / (1 + np.abs(getattr(st, k)))
        assert err.max() < 3e-5, k
    g = torch.Generator().manual_seed(5 + seed)
    gc, gd, ga = (torch.randn(c, H, W, generator=g, dtype=torch.float64) for c in (3, 1, 1))
    ((out["color"] * gc).sum() + (out["depth"] * gd).sum() + (out["alpha"] * ga).sum()).backward()
    gr = tile_ref.backward(st, gc.numpy(), gd.numpy(), ga.numpy())
    pairs = [("dL_dmeans3D", leaf["means3D"].grad), ("dL_dopacity", leaf["opacities"].grad.reshape(-1, 1)),
             ("dL_dscales", leaf["scales"].grad), ("dL_drotations", leaf["rotations"].grad), ("dL_dsh", leaf["shs"].grad)]
    for name, ref in pairs:
        assert rel_l2(gr[name], ref.numpy()) < 5e-5, name
    assert rel_l2(gr["dL_dmeans2D"][:, :2], off.grad.numpy() * np.array([0.5 * W, 0.5 * H])) < 5e-5


def test_ka10b_precomputed_colour_and_covariance_gradients():
    from binocular3dgs_amd.gaussian_model import covariance_from_scaling_rotation
    P, W, H = 300, 48, 48
    d, _ = small_scene(P=P, W=W, H=H, seed=7)
    g = torch.Generator().manual_seed(2)
    col = torch.rand(P, 3, generator=g)
    cov = covariance_from_scaling_rotation(d["scales"], 1.0, d["rotations"])
    kw = oracle_kwargs(d, shs=None, scales=None, rotations=None, colors_precomp=col.numpy(), cov3D_precomp=cov.numpy())
    st = tile_ref.forward(**kw)
    colr, covr = col.double().requires_grad_(True), cov.double().requires_grad_(True)
    dd = {k: v for k, v in d.items() if k not in ("shs", "scales", "rotations")}
    out = dense_torch.render_dense(**dd, colors_precomp=colr, cov3D_precomp=covr, rect=torch.from_numpy(st.rect))
    gc = torch.randn(3, H, W, generator=g, dtype=torch.float64)
    (out["color"] * gc).sum().backward()
    gr = tile_ref.backward(st, gc.numpy())
    assert rel_l2(gr["dL_dcolors"], colr.grad.numpy()) < 5e-5
    assert rel_l2(gr["dL_dcov3D"], covr.grad.numpy()) < 5e-5


def test_ka9_finite_differences_fp64():
    """Central differences through the dense restatement (its own autograd is what KA10 trusts)."""
    P, W, H = 40, 32, 32
    d, _ = small_scene(P=P, W=W, H=H, seed=3, scale_mu=0.12)
    st = tile_ref.forward(**oracle_kwargs(d))
    rect = torch.from_numpy(st.rect)
    g = torch.Generator().manual_seed(9)
    gc, gd, ga = (torch.randn(c, H, W, generator=g, dtype=torch.float64) for c in (3, 1, 1))

    def loss_of(**over):
        o = dense_torch.render_dense(**{**d, **over}, rect=rect)
        return (o["color"] * gc).sum() + (o["depth"] * gd).sum() + (o["alpha"] * ga).sum()

    for name in ("means3D", "scales", "rotations", "opacities", "shs"):
        x = d[name].double().clone().requires_grad_(True)
        loss_of(**{name: x}).backward()
        ana = x.grad.reshape(-1)
        flat = x.detach().reshape(-1)
        rng = np.random.default_rng(1)
        idxs = rng.choice(flat.numel(), size=12, replace=False)
        for i in idxs:
            eps = 1e-6 * max(1.0, abs(float(flat[i])))
            xp, xm = flat.clone(), flat.clone()
            xp[i] += eps
            xm[i] -= eps
            num = (loss_of(**{name: xp.reshape(x.shape)}) - loss_of(**{name: xm.reshape(x.shape)})) / (2 * eps)
            assert abs(float(num) - float(ana[i])) <= 1e-4 * max(1.0, abs(float(ana[i]))) + 1e-6, (name, int(i))


def test_empty_inputs():
    cam = _cam(40, 24)
    st = tile_ref.forward(means3D=np.zeros((0, 3), np.float32), opacities=np.zeros((0,), np.float32),
                          colors_precomp=np.zeros((0, 3), np.float32), scales=np.zeros((0, 3), np.float32),
                          rotations=np.zeros((0, 4), np.float32), viewmatrix=cam.world_view_transform.numpy(),
                          projmatrix=cam.full_proj_transform.numpy(), campos=cam.camera_center.numpy(), bg=[1, 0, 0],
                          W=40, H=24, tanfovx=0.5, tanfovy=0.3)
    assert st.N == 0 and float(st.color[0].min()) == 1.0 and float(st.alpha.max()) == 0.0
