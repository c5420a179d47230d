"""CPU: the host-side mirrors and the oracle against the golden vectors generated from the
reference's own Python (tests/golden/make_golden.py).  These pin every stage of the hot path
that exists in the reference tree."""
import math
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return np.load(os.path.join(GOLD, name))


def test_g1_sh_eval_host_and_oracle():
    from binocular3dgs_amd.render import eval_sh
    from oracle import tile_ref
    import ctypes as C
    g = load("sh_eval.npz")
    sh, dirs = torch.from_numpy(g["sh"]), torch.from_numpy(g["dirs"])
    for deg in range(4):
        np.testing.assert_allclose(eval_sh(deg, sh, dirs).numpy(), g[f"rgb_deg{deg}"], rtol=1e-5, atol=1e-6)
    # the oracle's SH stage, driven through its preprocess: place Gaussians along `dirs` from the camera
    P = sh.shape[0]
    means = (dirs * 5.0 + torch.tensor([0.0, 0.0, 20.0])).numpy()   # all in front of an identity camera
    from binocular3dgs_amd.camera import Camera
    cam = Camera(np.eye(3), np.zeros(3), math.radians(90), math.radians(90), 64, 64)
    campos = np.array([0.0, 0.0, 20.0], np.float32)                 # SH direction = means - campos = 5*dirs
    for deg in range(4):
        st = tile_ref.forward(means3D=means, opacities=np.full(P, 0.5, np.float32),
                              shs=np.ascontiguousarray(g["sh"].transpose(0, 2, 1)), scales=np.full((P, 3), 0.5, np.float32),
                              rotations=np.tile(np.array([1, 0, 0, 0], np.float32), (P, 1)),
                              viewmatrix=cam.world_view_transform.numpy(), projmatrix=cam.full_proj_transform.numpy(),
                              campos=campos, bg=[0, 0, 0], W=64, H=64, tanfovx=1.0, tanfovy=1.0, sh_degree=deg)
        vis = st.radii > 0
        assert vis.sum() > 50
        ref = np.maximum(g[f"rgb_deg{deg}"] + 0.5, 0.0)
        np.testing.assert_allclose(st.rgb[vis], ref[vis], rtol=2e-5, atol=2e-6)
        np.testing.assert_array_equal(st.clamped[vis].astype(bool), (g[f"rgb_deg{deg}"] + 0.5 < 0)[vis])


def test_g2_activations_and_covariance():
    from binocular3dgs_amd.gaussian_model import GaussianModel
    from oracle import tile_ref
    g = load("covariance.npz")
    m = GaussianModel.from_tensors(torch.zeros(300, 3), torch.from_numpy(g["features_dc"]), torch.from_numpy(g["features_rest"]),
                                   torch.from_numpy(g["scaling_raw"]), torch.from_numpy(g["rotation_raw"]),
                                   torch.from_numpy(g["opacity_raw"]), sh_degree=1, requires_grad=False)
    np.testing.assert_allclose(m.get_scaling.numpy(), g["get_scaling"], rtol=1e-6)
    np.testing.assert_allclose(m.get_rotation.numpy(), g["get_rotation"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(m.get_opacity.numpy(), g["get_opacity"], rtol=1e-6)
    np.testing.assert_array_equal(m.get_features.numpy(), g["get_features"])
    np.testing.assert_allclose(m.get_covariance(1.0).numpy(), g["cov_mod1"], rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(m.get_covariance(0.7).numpy(), g["cov_mod07"], rtol=1e-5, atol=1e-8)
    # oracle: scale/rotation path (normalised quaternion in) == reference covariance
    from binocular3dgs_amd.camera import Camera
    cam = Camera(np.eye(3), np.zeros(3), math.radians(90), math.radians(90), 32, 32)
    for mod, key in ((1.0, "cov_mod1"), (0.7, "cov_mod07")):
        st = tile_ref.forward(means3D=np.tile(np.array([0, 0, 5.0], np.float32), (300, 1)), opacities=np.full(300, 0.5, np.float32),
                              colors_precomp=np.zeros((300, 3), np.float32), scales=g["get_scaling"], rotations=g["get_rotation"],
                              viewmatrix=cam.world_view_transform.numpy(), projmatrix=cam.full_proj_transform.numpy(),
                              campos=np.zeros(3, np.float32), bg=[0, 0, 0], W=32, H=32, tanfovx=1.0, tanfovy=1.0,
                              scale_modifier=mod)
        np.testing.assert_allclose(st.cov3D, g[key], rtol=2e-5, atol=1e-8)


def test_g3_g4_cameras_and_binocular_shift():
    from binocular3dgs_amd.camera import Camera
    g = load("cameras.npz")
    for i in range(int(g["n"])):
        fx, fy, w, h = g[f"fov{i}"]
        cam = Camera(g[f"R{i}"], g[f"T{i}"], fx, fy, int(w), int(h))
        np.testing.assert_array_equal(cam.world_view_transform.numpy(), g[f"wvt{i}"])
        np.testing.assert_array_equal(cam.projection_matrix.numpy(), g[f"proj{i}"])
        np.testing.assert_array_equal(cam.full_proj_transform.numpy(), g[f"full{i}"])
        # inverse of a contiguous copy vs the reference's inverse of a transposed view: 1-ulp LAPACK noise
        np.testing.assert_allclose(cam.camera_center.numpy(), g[f"center{i}"], rtol=0, atol=5e-7)
        np.testing.assert_allclose(np.array(cam.get_focal()), g[f"focal{i}"], rtol=1e-12)
        for j in range(4):
            t = float(g[f"shift{i}_{j}_t"])
            sc = cam.shifted(t)
            # closed form vs the reference's two host-side matrix inversions: fp32 round-off only
            np.testing.assert_allclose(sc.world_view_transform.numpy(), g[f"shift{i}_{j}_wvt"], rtol=0, atol=3e-6)
            np.testing.assert_allclose(sc.full_proj_transform.numpy(), g[f"shift{i}_{j}_full"], rtol=0, atol=1e-5)
            np.testing.assert_allclose(sc.camera_center.numpy(), g[f"shift{i}_{j}_center"], rtol=0, atol=1e-5)
            assert sc.image_width == cam.image_width and sc.FoVx == cam.FoVx


def test_g5_render_kwargs_routing(monkeypatch):
    import binocular3dgs_amd.render as R
    from binocular3dgs_amd.camera import Camera
    from binocular3dgs_amd.gaussian_model import GaussianModel
    g, c = load("render_kwargs.npz"), load("covariance.npz")
    rec = {}

    class Recorder:
        def __init__(self, raster_settings):
            self.rs = raster_settings

        def __call__(self, **kw):
            rec["rs"], rec["kw"] = self.rs, kw
            P = kw["means3D"].shape[0]
            H, W = self.rs.image_height, self.rs.image_width
            return torch.zeros(3, H, W), torch.zeros(P, dtype=torch.int32), torch.zeros(1, H, W), torch.zeros(1, H, W)

    monkeypatch.setattr(R, "GaussianRasterizer", Recorder)
    m = GaussianModel.from_tensors(torch.from_numpy(g["xyz"]), torch.from_numpy(c["features_dc"]),
                                   torch.from_numpy(c["features_rest"]), torch.from_numpy(c["scaling_raw"]),
                                   torch.from_numpy(c["rotation_raw"]), torch.from_numpy(c["opacity_raw"]), sh_degree=1,
                                   active_sh_degree=1)
    fx, fy, w, h = g["camfov"]
    cam = Camera(g["camR"], g["camT"], fx, fy, int(w), int(h))
    names = ("means3D", "means2D", "shs", "colors_precomp", "opacities", "scales", "rotations", "cov3D_precomp")
    for a in (False, True):
        for b in (False, True):
            tag = f"sh{int(a)}_cov{int(b)}"
            pkg = R.render(cam, m, R.PipelineParams(a, b), torch.tensor([0.1, 0.2, 0.3]), scaling_modifier=0.9)
            kw, rs = rec["kw"], rec["rs"]
            np.testing.assert_array_equal(np.array([kw[k] is not None for k in names]), g[f"{tag}_present"])
            for k in names[2:]:
                if kw[k] is not None:
                    np.testing.assert_allclose(kw[k].detach().numpy(), g[f"{tag}_{k}"], rtol=2e-5, atol=1e-6, err_msg=f"{tag} {k}")
            s = g[f"{tag}_settings"]
            assert (rs.image_height, rs.image_width, rs.sh_degree) == (int(s[0]), int(s[1]), int(s[5]))
            np.testing.assert_allclose([rs.tanfovx, rs.tanfovy, rs.scale_modifier], s[2:5], rtol=1e-12)
            assert rs.prefiltered is False and rs.debug is False
            np.testing.assert_array_equal(rs.viewmatrix.numpy(), g[f"{tag}_viewmatrix"])
            np.testing.assert_array_equal(rs.projmatrix.numpy(), g[f"{tag}_projmatrix"])
            np.testing.assert_allclose(rs.campos.numpy(), g[f"{tag}_campos"], rtol=0, atol=5e-7)
            assert set(pkg) == {"render", "viewspace_points", "visibility_filter", "radii", "rendered_depth", "rendered_alpha"}
            assert pkg["viewspace_points"].requires_grad and pkg["viewspace_points"].shape == (300, 3)


def test_g6_loss_block_values_and_pixel_gradients():
    from binocular3dgs_amd.loss import binocular_loss
    g = load("loss_block.npz")
    t = lambda k: torch.from_numpy(g[k]).clone().requires_grad_(True)  # noqa: E731
    image, depth, alpha, shifted = t("image"), t("depth"), t("alpha"), t("shifted")
    focal_x, trans, lam = [float(x) for x in g["scalars"]]
    total, parts = binocular_loss(image, depth, alpha, torch.from_numpy(g["gt"]), lambda_dssim=lam, shifted_image=shifted,
                                  focal_x=focal_x, trans_dist=trans, gt_alpha_mask=torch.from_numpy(g["gt_alpha_mask"]))
    total.backward()
    np.testing.assert_allclose(parts["warped"].detach().numpy(), g["warped"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(parts["shift_mask"].detach().numpy(), g["shift_mask"], rtol=1e-5, atol=1e-6)
    for k in ("l1_masked", "smooth", "alpha_loss", "Ll1", "ssim"):
        np.testing.assert_allclose(parts[k].detach().numpy(), g[k], rtol=2e-5, atol=1e-7, err_msg=k)
    np.testing.assert_allclose(total.detach().numpy(), g["total"], rtol=2e-5)
    for ten, key in ((image, "g_image"), (depth, "g_depth"), (alpha, "g_alpha"), (shifted, "g_shifted")):
        np.testing.assert_allclose(ten.grad.numpy(), g[key], rtol=2e-4, atol=2e-9, err_msg=key)
        assert np.abs(g[key]).max() > 0


def test_g7_misc():
    from binocular3dgs_amd.gaussian_model import inverse_sigmoid
    from binocular3dgs_amd.loss import expon_lr, psnr
    g = load("misc.npz")
    for s, v in zip(g["lr_steps"], g["lr_vals"]):
        assert expon_lr(int(s), 1.6e-4, 1.6e-6, lr_delay_mult=0.01, max_steps=30000) == pytest.approx(float(v), rel=1e-12)
    np.testing.assert_allclose(inverse_sigmoid(torch.from_numpy(g["isig_x"])).numpy(), g["isig_y"], rtol=1e-6)
    np.testing.assert_allclose(psnr(torch.from_numpy(g["psnr_a"]), torch.from_numpy(g["psnr_b"])).numpy(), g["psnr"], rtol=1e-6)


@pytest.mark.parametrize("case", ["a", "b"])
def test_g8_densify_and_prune_restatement(case):
    """The formulation the HIP densify kernels implement (per-Gaussian decisions + one scatter, tests/densify_ref.py)
    reproduces what the reference's densify_and_prune did to parameters and Adam state."""
    import densify_ref
    g = load("densify.npz")
    names = ("xyz", "f_dc", "f_rest", "scaling", "rotation", "opacity")
    t = lambda k: torch.from_numpy(g[k]).clone()  # noqa: E731
    params = {n: t(f"{case}_in_{n}") for n in names}
    m = {n: t(f"{case}_in_{n}_exp_avg") for n in names}
    v = {n: t(f"{case}_in_{n}_exp_avg_sq") for n in names}
    thr, min_op, extent, pd, size = [float(x) for x in g[f"{case}_scalars"]]
    P = params["xyz"].shape[0]
    grads = torch.nan_to_num(t(f"{case}_accum")[:, 0] / t(f"{case}_denom")[:, 0], nan=0.0)
    sel = torch.nonzero((grads >= thr) & (torch.exp(params["scaling"]).max(1).values > pd * extent))[:, 0]
    nz = t(f"{case}_noise")
    noise = torch.zeros(2, P, 3)
    noise[0, sel], noise[1, sel] = nz[:len(sel)], nz[len(sel):]
    op, om, ov = densify_ref.densify_and_prune(params, m, v, t(f"{case}_accum"), t(f"{case}_denom"), thr, min_op, extent,
                                               None if size < 0 else size, pd, noise)
    for n in names:
        ref = g[f"{case}_out_{n}"]
        assert tuple(op[n].shape) == ref.shape, n
        np.testing.assert_allclose(op[n].numpy(), ref, rtol=1e-6, atol=1e-7, err_msg=n)
        np.testing.assert_array_equal(om[n].numpy(), g[f"{case}_out_{n}_exp_avg"], err_msg=n)
        np.testing.assert_array_equal(ov[n].numpy(), g[f"{case}_out_{n}_exp_avg_sq"], err_msg=n)
    np.testing.assert_allclose(torch.log(torch.sigmoid(t("decay_in")) * 0.995 / (1 - torch.sigmoid(t("decay_in")) * 0.995)).numpy(),
                               g["decay_out"], rtol=1e-5, atol=1e-6)


@pytest.mark.skipif(not os.path.isdir("/root/reference/gaussian_renderer"), reason="reference tree only exists in the build container")
def test_config1_reference_render_runs_on_top_of_the_boundary():
    """BASELINE config 1 (plumbing): the REFERENCE's own render() (imported read-only under the CPU shim),
    with this repo's `diff_gaussian_rasterization` surface injected and the oracle standing in for
    the GPU, 10k Gaussians @ 256x256 forward.  Proves the drop-in import/keyword/return contract."""
    import subprocess
    import sys
    code = r'''
import sys, types, math, torch, numpy as np
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/reference")
for n in ("plyfile", "simple_knn", "simple_knn._C", "imageio", "skimage", "skimage.transform"):
    sys.modules[n] = types.ModuleType(n)
sys.modules["plyfile"].PlyData = object; sys.modules["plyfile"].PlyElement = object
sys.modules["simple_knn._C"].distCUDA2 = None
import binocular3dgs_amd.rasterizer as ours
from cpu_render import OracleRasterizer
shim = types.ModuleType("diff_gaussian_rasterization")
shim.GaussianRasterizationSettings = ours.GaussianRasterizationSettings   # OUR settings type
shim.GaussianRasterizer = OracleRasterizer                                # oracle instead of the GPU
sys.modules["diff_gaussian_rasterization"] = shim
class M(torch.overrides.TorchFunctionMode):
    def __torch_function__(self, f, t, a=(), k=None):
        k = dict(k or {})
        if "device" in k and k["device"] is not None and "cuda" in str(k["device"]): k["device"] = "cpu"
        if f is torch.Tensor.cuda: return a[0]
        if f is torch.Tensor.to and len(a) > 1 and isinstance(a[1], (str, torch.device)) and "cuda" in str(a[1]): return a[0]
        return f(*a, **k)
from binocular3dgs_amd import synth
with M():
    from gaussian_renderer import render
    from scene.gaussian_model import GaussianModel
    from scene.cameras import Camera
    p = synth.synth_gaussians(10000, 0, 256, 256)
    gm = GaussianModel(1)
    gm._xyz, gm._features_dc, gm._features_rest = p["xyz"], p["features_dc"], p["features_rest"]
    gm._scaling, gm._rotation, gm._opacity = p["scaling"], p["rotation"], p["opacity"]
    gm.active_sh_degree = 1
    fov = math.radians(60)
    cam = Camera(colmap_id=0, R=np.eye(3), T=np.zeros(3), FoVx=fov, FoVy=fov, image=torch.zeros(3, 256, 256), gt_alpha_mask=None, image_name="c", uid=0)
    class Pipe: convert_SHs_python = False; compute_cov3D_python = False; debug = False
    pkg = render(cam, gm, Pipe(), torch.zeros(3))
    assert pkg["render"].shape == (3, 256, 256) and pkg["rendered_depth"].shape == (1, 256, 256)
    assert int(pkg["visibility_filter"].sum()) > 5000 and float(pkg["rendered_alpha"].mean()) > 0.5
    print("CONFIG1_OK", int(pkg["visibility_filter"].sum()))
'''
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1"))
    assert "CONFIG1_OK" in r.stdout, r.stdout + r.stderr
