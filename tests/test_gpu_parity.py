"""GPU parity: the HIP path (through the C ABI) against oracle/tile_ref.c on seeded inputs.

Tolerances (fp32; stated here as the contract):
  * radii, tiles_touched, point_list (tile-major depth-sorted Gaussian ids), tile ranges: BIT-EXACT
  * colour / depth / alpha: |err| <= 2e-5 * (1 + |x|)   (GPU exp2-based exp vs libm expf)
  * n_contrib: identical except at borderline pixels (alpha ~ 1/255 or T ~ 1e-4): <= 0.1 % of pixels
  * gradients: relative L2 error per tensor <= 2e-4 vs the oracle's double-accumulated sums
"""
import math

import numpy as np
import pytest
import torch

from helpers import oracle_kwargs, rel_l2, small_scene

pytestmark = pytest.mark.gpu


def _gpu(d):
    return {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in d.items()}


def _run_hip_forward(d, mode="sh_sr"):
    from binocular3dgs_amd import _C
    from binocular3dgs_amd.debug import state_views
    g = _gpu(d)
    e = torch.empty(0, device="cuda")
    sh, colors, scales, rots, cov = g["shs"], e, g["scales"], g["rotations"], e
    if "col" in mode:
        sh, colors = e, g["colors_precomp"]
    if "cov" in mode:
        scales, rots, cov = e, e, g["cov3D_precomp"]
    n, color, depth, alpha, radii, geom, binning, img = _C.rasterize_gaussians(
        g["bg"], g["means3D"], colors, g["opacities"], scales, rots, d.get("scale_modifier", 1.0), cov,
        g["viewmatrix"], g["projmatrix"], d["tanfovx"], d["tanfovy"], d["H"], d["W"], sh, d["sh_degree"], g["campos"],
        False, True)
    views = state_views(g["means3D"].shape[0], d["W"], d["H"], n, geom, binning, img)
    return dict(n=n, color=color, depth=depth, alpha=alpha, radii=radii, geom=geom, binning=binning, img=img,
                views=views, inputs=(g, sh, colors, scales, rots, cov))


def _check_forward(d, st, out, px_tol=1e-3, flip_frac=0.0):
    """flip_frac: fraction of pixels allowed above the 2e-5 band (by at most one minimal contribution, 1/255):
    a Gaussian whose alpha sits within an ulp of the 1/255 skip rule can be taken by one implementation and
    skipped by the other (GPU exp2-based exp vs libm); over millions of pixels a handful of such pixels exist."""
    P = st.P
    assert out["n"] == st.N
    np.testing.assert_array_equal(out["radii"].cpu().numpy(), st.radii)
    v = out["views"]
    np.testing.assert_array_equal(v["tiles_touched"].cpu().numpy().astype(np.uint32), st.tiles_touched)
    if st.N:
        np.testing.assert_array_equal(v["point_list"].cpu().numpy().astype(np.uint32), st.point_list)
        np.testing.assert_array_equal(v["tile_ids"].cpu().numpy().astype(np.uint64), st.keys >> np.uint64(32))
    np.testing.assert_array_equal(v["ranges"].cpu().numpy().astype(np.uint32), st.ranges)
    vis = st.radii > 0
    rec = v["records"].cpu().numpy()
    np.testing.assert_array_equal(rec[vis, 0:2], st.means2D[vis])          # pixel positions: same IEEE ops
    np.testing.assert_array_equal(rec[vis][:, [2, 3, 4, 5]], st.conic_opacity[vis])
    np.testing.assert_array_equal(rec[vis][:, [6, 7, 8]], st.rgb[vis])
    np.testing.assert_array_equal(rec[vis, 9], st.depths[vis])
    # word 12 of the record: the SH clamp bits the chain rule reads (bit c: colour channel c clamped at 0)
    bits = st.clamped[:, 0].astype(np.uint32) | (st.clamped[:, 1].astype(np.uint32) << 1) | (st.clamped[:, 2].astype(np.uint32) << 2)
    np.testing.assert_array_equal(np.ascontiguousarray(rec[:, 12]).view(np.uint32)[vis], bits[vis])
    for name, ref in (("color", st.color), ("depth", st.depth), ("alpha", st.alpha)):
        got = out[name].cpu().numpy()
        err = np.abs(got - ref) / (1 + np.abs(ref))
        if flip_frac > 0:
            assert float((err > 2e-5).mean()) <= flip_frac and err.max() <= 1.0 / 255, f"{name}: {float((err > 2e-5).mean()):.2e}"
        else:
            assert err.max() <= 2e-5, f"{name}: max err {err.max():.3e}"
    nc = v["n_contrib"].cpu().numpy().astype(np.uint32)
    frac = float((nc != st.n_contrib).mean())
    assert frac <= px_tol, f"n_contrib differs at {frac:.4%} of pixels"
    fT = v["final_T"].cpu().numpy()
    if flip_frac > 0:
        bad = np.abs(fT - st.final_T) > 2e-4 * np.abs(st.final_T) + 2e-6
        assert float(bad.mean()) <= flip_frac and np.abs(fT - st.final_T).max() <= 1.0 / 255
    else:
        np.testing.assert_allclose(fT, st.final_T, rtol=2e-4, atol=2e-6)


@pytest.mark.parametrize("seed,P,W,H", [(0, 500, 64, 48), (1, 3000, 200, 120), (2, 20000, 320, 240), (3, 257, 33, 17)])
def test_forward_parity_sh_scale_rot(seed, P, W, H):
    from oracle import tile_ref
    d, _ = small_scene(P=P, W=W, H=H, seed=seed, near_frac=0.05)
    st = tile_ref.forward(**oracle_kwargs(d))
    out = _run_hip_forward(d)
    _check_forward(d, st, out)


@pytest.mark.parametrize("mode", ["col_sr", "sh_cov", "col_cov"])
def test_forward_parity_precomputed_inputs(mode):
    from oracle import tile_ref
    from binocular3dgs_amd.gaussian_model import covariance_from_scaling_rotation
    d, _ = small_scene(P=1500, W=160, H=96, seed=11)
    g = torch.Generator().manual_seed(3)
    d["colors_precomp"] = torch.rand(1500, 3, generator=g)
    d["cov3D_precomp"] = covariance_from_scaling_rotation(d["scales"], 1.0, d["rotations"])
    kw = oracle_kwargs(d)
    if "col" in mode:
        kw["shs"] = None
    else:
        kw["colors_precomp"] = None
    if "cov" in mode:
        kw["scales"] = kw["rotations"] = None
    else:
        kw["cov3D_precomp"] = None
    st = tile_ref.forward(**kw)
    out = _run_hip_forward(d, mode)
    _check_forward(d, st, out)


@pytest.mark.parametrize("deg,K", [(0, 1), (0, 4), (2, 9), (3, 16)])
def test_forward_parity_sh_degrees(deg, K):
    from oracle import tile_ref
    d, _ = small_scene(P=800, W=96, H=64, seed=20 + deg, K=K, sh_degree=deg)
    st = tile_ref.forward(**oracle_kwargs(d))
    out = _run_hip_forward(d)
    _check_forward(d, st, out)


def _backward_both(d, mode="sh_sr", seed=0, use_depth=True, use_alpha=True):
    from oracle import tile_ref
    from binocular3dgs_amd import _C
    kw = oracle_kwargs(d)
    if "col" in mode:
        kw["shs"] = None
    else:
        kw.pop("colors_precomp", None)
    if "cov" in mode:
        kw["scales"] = kw["rotations"] = None
    else:
        kw.pop("cov3D_precomp", None)
    st = tile_ref.forward(**kw)
    out = _run_hip_forward(d, mode)
    H, W = d["H"], d["W"]
    g = torch.Generator().manual_seed(100 + seed)
    gc = torch.randn(3, H, W, generator=g)
    gd = torch.randn(1, H, W, generator=g) if use_depth else None
    ga = torch.randn(1, H, W, generator=g) if use_alpha else None
    ref = tile_ref.backward(st, gc.numpy(), None if gd is None else gd.numpy(), None if ga is None else ga.numpy())
    G, sh, colors, scales, rots, cov = out["inputs"]
    e = torch.empty(0, device="cuda")
    res = _C.rasterize_gaussians_backward(
        G["bg"], G["means3D"], out["radii"], colors, scales, rots, d.get("scale_modifier", 1.0), cov, G["viewmatrix"],
        G["projmatrix"], d["tanfovx"], d["tanfovy"], gc.cuda(), e if gd is None else gd.cuda(),
        e if ga is None else ga.cuda(), sh, d["sh_degree"], G["campos"], out["geom"], out["n"], out["binning"],
        out["img"], out["alpha"], True)
    names = ("dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales",
             "dL_drotations")
    return st, ref, dict(zip(names, res))


@pytest.mark.parametrize("seed,P,W,H", [(0, 500, 64, 48), (1, 4000, 200, 120)])
def test_backward_parity(seed, P, W, H):
    d, _ = small_scene(P=P, W=W, H=H, seed=seed, near_frac=0.05)
    st, ref, got = _backward_both(d, seed=seed)
    for k in ("dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales",
              "dL_drotations"):
        e = rel_l2(got[k].cpu().numpy(), ref[k])
        assert e <= 2e-4, f"{k}: rel L2 {e:.3e}"
    # culled Gaussians receive exact zeros
    culled = torch.from_numpy(st.radii <= 0)
    for k in got:
        assert float(got[k].cpu()[culled].abs().sum()) == 0.0, k


@pytest.mark.parametrize("mode", ["col_sr", "sh_cov", "col_cov"])
def test_backward_parity_precomputed_inputs(mode):
    from binocular3dgs_amd.gaussian_model import covariance_from_scaling_rotation
    d, _ = small_scene(P=1500, W=160, H=96, seed=31)
    g = torch.Generator().manual_seed(3)
    d["colors_precomp"] = torch.rand(1500, 3, generator=g)
    d["cov3D_precomp"] = covariance_from_scaling_rotation(d["scales"], 1.0, d["rotations"])
    st, ref, got = _backward_both(d, mode=mode, seed=4)
    keys = ["dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D"]
    if "col" not in mode:
        keys.append("dL_dsh")
    if "cov" not in mode:
        keys += ["dL_dscales", "dL_drotations"]
    for k in keys:
        e = rel_l2(got[k].cpu().numpy(), ref[k])
        assert e <= 2e-4, f"{k}: rel L2 {e:.3e}"


def test_backward_color_only_and_determinism_band():
    d, _ = small_scene(P=2000, W=128, H=96, seed=41)
    st, ref, a = _backward_both(d, seed=1, use_depth=False, use_alpha=False)
    _, _, b = _backward_both(d, seed=1, use_depth=False, use_alpha=False)
    for k in ("dL_dmeans3D", "dL_dscales", "dL_dsh"):
        assert rel_l2(a[k].cpu().numpy(), ref[k]) <= 2e-4
        # fp32 atomics: run-to-run differences stay at rounding level
        assert rel_l2(a[k].cpu().numpy(), b[k].cpu().numpy()) <= 1e-5


def test_autograd_function_and_render_surface():
    """render() end to end: dict keys / shapes / dtypes of gaussian_renderer/__init__.py:97-103 and
    means2D.grad populated (train.py:179 reads viewspace_points.grad[:, :2])."""
    from oracle import tile_ref
    from binocular3dgs_amd import synth
    from binocular3dgs_amd.render import PipelineParams, render
    W, H = 160, 120
    model = synth.synth_model(3000, seed=5, device="cuda", width=W, height=H)
    cam = synth.synth_cameras(W, H, yaws=(4.0,), device="cuda")[0]
    bg = torch.tensor([0.0, 0.0, 0.0], device="cuda")
    grads = {}
    for flags in [(False, False), (True, True)]:
        pipe = PipelineParams(convert_SHs_python=flags[0], compute_cov3D_python=flags[1])
        for p in model.parameters():
            p.grad = None
        pkg = render(cam, model, pipe, bg)
        assert set(pkg) == {"render", "viewspace_points", "visibility_filter", "radii", "rendered_depth", "rendered_alpha"}
        assert pkg["render"].shape == (3, H, W) and pkg["rendered_depth"].shape == (1, H, W)
        assert pkg["rendered_alpha"].shape == (1, H, W) and pkg["radii"].dtype == torch.int32
        assert pkg["visibility_filter"].dtype == torch.bool
        loss = pkg["render"].mean() + 0.1 * pkg["rendered_depth"].mean() + 0.2 * pkg["rendered_alpha"].mean()
        loss.backward()
        vg = pkg["viewspace_points"].grad
        assert vg is not None and vg.shape == (3000, 3) and float(vg[:, 2].abs().sum()) == 0.0
        assert float(vg[~pkg["visibility_filter"]].abs().sum()) == 0.0
        grads[flags] = [p.grad.clone() for p in model.parameters()]
    # the in-rasterizer SH / covariance paths and the PyTorch-side ones give the same parameter grads
    for ga, gb in zip(grads[(False, False)], grads[(True, True)]):
        assert rel_l2(ga.cpu().numpy(), gb.cpu().numpy()) <= 5e-4
    # forward equals the oracle for the same activated inputs
    st = tile_ref.forward(means3D=model.get_xyz.detach().cpu().numpy(), opacities=model.get_opacity.detach().cpu().numpy(),
                          scales=model.get_scaling.detach().cpu().numpy(), rotations=model.get_rotation.detach().cpu().numpy(),
                          shs=model.get_features.detach().cpu().numpy(), viewmatrix=cam.world_view_transform.cpu().numpy(),
                          projmatrix=cam.full_proj_transform.cpu().numpy(), campos=cam.camera_center.cpu().numpy(),
                          bg=bg.cpu().numpy(), W=W, H=H, tanfovx=math.tan(cam.FoVx / 2), tanfovy=math.tan(cam.FoVy / 2),
                          sh_degree=1)
    pkg = render(cam, model, PipelineParams(), bg)
    np.testing.assert_array_equal(pkg["radii"].cpu().numpy(), st.radii)
    assert np.abs(pkg["render"].detach().cpu().numpy() - st.color).max() <= 2e-5


def test_cpu_tensors_are_rejected():
    from binocular3dgs_amd import GaussianRasterizationSettings, GaussianRasterizer, _lib
    d, _ = small_scene(P=10, W=16, H=16)
    rs = GaussianRasterizationSettings(16, 16, d["tanfovx"], d["tanfovy"], d["bg"], 1.0, d["viewmatrix"],
                                       d["projmatrix"], 1, d["campos"], False, False)
    with pytest.raises(_lib.B3gsError):
        GaussianRasterizer(rs)(means3D=d["means3D"], means2D=torch.zeros(10, 3), opacities=d["opacities"],
                               shs=d["shs"], scales=d["scales"], rotations=d["rotations"])


def test_empty_and_all_culled():
    from binocular3dgs_amd import _C
    d, _ = small_scene(P=64, W=48, H=32, seed=9)
    d["means3D"] = d["means3D"].clone()
    d["means3D"][:, 2] = -5.0          # everything behind the camera
    out = _run_hip_forward(d)
    assert out["n"] == 0 and int(out["radii"].abs().sum()) == 0
    bg = d["bg"].reshape(3, 1, 1).expand(3, 32, 48)
    np.testing.assert_allclose(out["color"].cpu().numpy(), bg.numpy(), rtol=0, atol=0)
    assert float(out["alpha"].abs().sum()) == 0.0
    vis = _C.mark_visible(d["means3D"].cuda(), d["viewmatrix"].cuda(), d["projmatrix"].cuda())
    assert not bool(vis.any())


def test_huge_gaussians_scale_modifier_and_single_point():
    """Edge cases: splats covering the whole image (rect = every tile, long lists in every tile),
    scale_modifier != 1, and P = 1."""
    from oracle import tile_ref
    d, _ = small_scene(P=300, W=100, H=70, seed=51, scale_mu=1.5)
    d["scale_modifier"] = 0.8
    st = tile_ref.forward(**oracle_kwargs(d))
    assert int(st.tiles_touched.max()) == 7 * 5          # some splats cover all 35 tiles
    out = _run_hip_forward(d)
    _check_forward(d, st, out)
    st2, ref, got = _backward_both(d, seed=3)
    for k in ("dL_dmeans3D", "dL_dscales", "dL_drotations", "dL_dsh", "dL_dopacity"):
        assert rel_l2(got[k].cpu().numpy(), ref[k]) <= 2e-4, k
    one, _ = small_scene(P=1, W=40, H=40, seed=52, scale_mu=0.3)
    one["means3D"] = torch.tensor([[0.0, 0.0, 5.0]])
    st1 = tile_ref.forward(**oracle_kwargs(one))
    _check_forward(one, st1, _run_hip_forward(one))


def test_tile_bits_beyond_one_byte_and_two_bytes():
    """More than 256 tiles needs 2 byte-passes of the tile sort (both test sizes do); a tall, narrow
    image exercises grid_x = 1."""
    from oracle import tile_ref
    for (W, H) in ((16, 400), (720, 16)):
        d, _ = small_scene(P=2000, W=W, H=H, seed=60 + W, scale_mu=0.05)
        st = tile_ref.forward(**oracle_kwargs(d))
        _check_forward(d, st, _run_hip_forward(d))


def test_two_word_instances_when_tile_and_index_bits_exceed_32():
    """bits(P) + bits(tiles) > 32 switches the binning from one packed (tile | index) word per instance to a
    (tile, index) pair (b3gs_packed_idx_bits): 140k Gaussians need 18 bits, 128 x 129 tiles 15.  Lists, ranges,
    images and gradients must be the same algorithm."""
    from oracle import tile_ref
    from binocular3dgs_amd import _lib
    import ctypes as C
    P, W, H = 140000, 2048, 2064
    d, _ = small_scene(P=P, W=W, H=H, seed=9, scale_mu=0.012)
    st, ref, got = _backward_both(d, seed=9)
    out = _run_hip_forward(d)
    v = _lib.B3gsDebugViews()
    _lib.check(_lib.lib().b3gs_debug_views(P, W, H, out["n"], out["geom"].data_ptr(), out["binning"].data_ptr(),
                                           out["img"].data_ptr(), C.byref(v)), "b3gs_debug_views")
    assert v.packed_idx_bits == -1 and v.tile_ids
    _check_forward(d, st, out, flip_frac=1e-5)
    for k in ("dL_dmeans2D", "dL_dopacity", "dL_dmeans3D", "dL_dsh", "dL_dscales", "dL_drotations"):
        assert rel_l2(got[k].cpu().numpy(), ref[k]) <= 2e-4, k
    # and the packed flavour just below the limit (17 + 15 bits)
    d2, _ = small_scene(P=120000, W=W, H=H, seed=10, scale_mu=0.012)
    st2 = tile_ref.forward(**oracle_kwargs(d2))
    out2 = _run_hip_forward(d2)
    _lib.check(_lib.lib().b3gs_debug_views(120000, W, H, out2["n"], out2["geom"].data_ptr(), out2["binning"].data_ptr(),
                                           out2["img"].data_ptr(), C.byref(v)), "b3gs_debug_views")
    assert v.packed_idx_bits == 17
    _check_forward(d2, st2, out2, flip_frac=1e-5)


def test_forward_capacity_entry_point_and_overflow():
    """b3gs_forward_capacity (reference-shaped scene, caller-presized buffers, N stays on the device) reproduces
    b3gs_forward; capacity overflow is visible in the device-side N."""
    import ctypes as C
    from binocular3dgs_amd import _lib
    d, _ = small_scene(P=3000, W=200, H=120, seed=21)
    out = _run_hip_forward(d)
    g, sh, colors, scales, rots, cov = out["inputs"]
    L = _lib.lib()
    P, W, H = 3000, d["W"], d["H"]
    ptr = lambda t: t.data_ptr() if t.numel() else None  # noqa: E731
    sc = _lib.B3gsScene(P, d["sh_degree"], sh.shape[1], W, H, d["tanfovx"], d["tanfovy"], 1.0, 0, 0, ptr(g["bg"]), ptr(g["means3D"]),
                        ptr(sh), None, ptr(g["opacities"]), ptr(scales), ptr(rots), None, ptr(g["viewmatrix"]),
                        ptr(g["projmatrix"]), ptr(g["campos"]))
    u8 = dict(dtype=torch.uint8, device="cuda")
    for cap in (out["n"] + 1000, max(out["n"] // 3, 1)):
        geom = torch.empty(L.b3gs_geometry_bytes(P), **u8)
        binning = torch.empty(L.b3gs_binning_bytes(P, cap), **u8)
        img = torch.empty(L.b3gs_image_bytes(W, H), **u8)
        color, depth, alpha = torch.empty(3, H, W, device="cuda"), torch.empty(1, H, W, device="cuda"), torch.empty(1, H, W, device="cuda")
        radii = torch.zeros(P, dtype=torch.int32, device="cuda")
        n = torch.zeros(1, dtype=torch.int32, device="cuda")
        rc = L.b3gs_forward_capacity(C.byref(sc), geom.data_ptr(), binning.data_ptr(), cap, img.data_ptr(), color.data_ptr(),
                                     depth.data_ptr(), alpha.data_ptr(), radii.data_ptr(), n.data_ptr(),
                                     torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "b3gs_forward_capacity")
        torch.cuda.synchronize()
        assert int(n.item()) == out["n"]                       # the true N even when it does not fit
        assert torch.equal(radii, out["radii"])
        if cap >= out["n"]:
            assert torch.equal(color, out["color"]) and torch.equal(depth, out["depth"]) and torch.equal(alpha, out["alpha"])
        else:
            assert int(n.item()) > cap and torch.isfinite(color).all()   # truncated lists: caller must grow and repeat


def test_autograd_surface_runs_sync_free_after_the_first_render_of_a_shape(monkeypatch):
    """GaussianRasterizer (the surface render() uses) pays the blocking read-back of num_rendered only for the first
    render of a (P, W, H) shape; afterwards a DIFFERENTIATED render runs b3gs_forward_capacity with twice the largest N
    seen and its N is checked at the entry of its backward (rasterizer._LazyN).  Same images and gradients bit for bit / to
    atomics order; an N above the capacity raises in backward() -- before any gradient exists, so optimizer.step() cannot
    consume a truncated render -- and the next render is complete again.  Renders nobody differentiates stay exact."""
    from binocular3dgs_amd import _lib, rasterizer, synth
    from binocular3dgs_amd.render import PipelineParams, render
    W, H, P = 160, 112, 4000
    model = synth.synth_model(P, seed=5, device="cuda", width=W, height=H)
    cam = synth.synth_view_set(W, H, device="cuda")[0][0]
    bg = torch.tensor([0.1, 0.2, 0.3], device="cuda")
    gc, gd, ga = synth.synth_pixel_grads(W, H, seed=2, device="cuda")
    lz = rasterizer._lazy
    assert lz.enabled
    key = (torch.cuda.current_device(), P, W, H)
    lz.capacity.pop(key, None)
    lz.pending.clear()
    import binocular3dgs_amd.render as R
    monkeypatch.setattr(R, "_FUSED_NODE", False)     # this test is about the reference-shaped GaussianRasterizer surface

    def run(backward=True):
        for p in model.parameters():
            p.grad = None
        pkg = render(cam, model, PipelineParams(), bg)
        if backward:
            torch.autograd.backward([pkg["render"], pkg["rendered_depth"], pkg["rendered_alpha"]], [gc, gd, ga])
        torch.cuda.synchronize()
        return ([pkg[k].detach().clone() for k in ("render", "rendered_depth", "rendered_alpha", "radii")],
                [p.grad.clone() for p in model.parameters()] + [pkg["viewspace_points"].grad.clone()] if backward else None)

    first = run()                                   # synchronous: learns N
    cap = lz.capacity[key]
    assert lz.pending == []
    pkg = render(cam, model, PipelineParams(), bg)  # sync-free: capacity path
    assert [t[0] for t in lz.pending] == [key]      # its N is on the way to the host, nobody waited for it
    torch.autograd.backward([pkg["render"], pkg["rendered_depth"], pkg["rendered_alpha"]], [gc, gd, ga])
    assert lz.pending == []                         # ... until its backward looked at it
    second = run()
    for a, b in zip(first[0], second[0]):
        assert torch.equal(a, b)
    for a, b in zip(first[1], second[1]):
        assert rel_l2(b.cpu().numpy(), a.cpu().numpy()) < 1e-5
    assert lz.capacity[key] == cap                  # N well inside the capacity: nothing grew
    # a scene that outgrew the buffer: the backward of THAT render refuses, nothing reaches the parameters' .grad
    lz.capacity[key] = 256
    with pytest.raises(_lib.B3gsError, match="B3GS_ERR_CAPACITY"):
        run()
    assert all(p.grad is None for p in model.parameters())
    assert lz.capacity[key] >= cap // 2 and lz.capacity[key] > 256 and lz.pending == []
    third = run()
    for a, b in zip(first[0], third[0]):
        assert torch.equal(a, b)
    # a truncated render whose backward never runs is caught at the next render instead
    lz.capacity[key] = 256
    run(backward=False)
    with pytest.raises(_lib.B3gsError, match="B3GS_ERR_CAPACITY"):
        run(backward=False)
    lz.pending.clear()
    # renders without gradients (evaluation loops) take the exact forward: never truncated, nothing pending
    lz.capacity[key] = 256
    with torch.no_grad():
        ev = render(cam, model, PipelineParams(), bg)
    assert lz.pending == [] and torch.equal(ev["render"], first[0][0]) and lz.capacity[key] > 256


def test_lazy_num_rendered_survives_more_renders_than_pinned_slots(monkeypatch):
    """More sync-free renders in a row than the ring of pinned read-back slots (64), none of them differentiated in the
    end: the oldest are drained before a slot is reused, every N is still checked."""
    from binocular3dgs_amd import rasterizer, synth
    from binocular3dgs_amd.render import PipelineParams, render
    W, H, P = 64, 48, 500
    model = synth.synth_model(P, seed=9, device="cuda", width=W, height=H)
    cam = synth.synth_view_set(W, H, device="cuda")[0][0]
    bg = torch.zeros(3, device="cuda")
    lz = rasterizer._lazy
    lz.pending.clear()
    import binocular3dgs_amd.render as R
    monkeypatch.setattr(R, "_FUSED_NODE", False)
    with torch.no_grad():
        ref = render(cam, model, PipelineParams(), bg)["render"].clone()
    for _ in range(lz.RING + 10):
        out = render(cam, model, PipelineParams(), bg)["render"].detach()
    assert 0 < len(lz.pending) < lz.RING
    assert torch.equal(out, ref)
    lz.poll(force=True)
    assert lz.pending == []


@pytest.mark.parametrize("K", [1, 4])
def test_render_of_a_raw_parameter_model_is_one_fused_node(K):
    """render() handed a model with the reference's raw parameters and activations (scene/gaussian_model.py:33-43) goes
    through ONE autograd node on the raw tensors (rasterizer.rasterize_raw: activations and their backward in-kernel,
    three-pass depth sort, tight binning) instead of the accessor kernels + GaussianRasterizer.  Same dict, images to the
    stated band (activation rounding may flip a 1/255 decision on a handful of pixels), gradients of the six parameters
    and of `viewspace_points` to 2e-4, accumulation over two renders through autograd, exact renders under no_grad."""
    import binocular3dgs_amd.render as R
    from binocular3dgs_amd import rasterizer, synth
    from binocular3dgs_amd.render import PipelineParams, render
    W, H, P = 200, 144, 6000
    model = synth.synth_model(P, seed=7, device="cuda", width=W, height=H, K=K)
    pairs = synth.synth_view_set(W, H, device="cuda")
    cam, scam = pairs[0][0], pairs[0][1]
    bg = torch.tensor([0.1, 0.0, 0.2], device="cuda")
    gc, gd, ga = synth.synth_pixel_grads(W, H, seed=2, device="cuda")
    assert rasterizer.raw_model_ok(model) and R._FUSED_NODE

    def run(fused):
        R._FUSED_NODE = fused
        try:
            for p in model.parameters():
                p.grad = None
            a = render(cam, model, PipelineParams(), bg)
            b = render(scam, model, PipelineParams(), bg)
            torch.autograd.backward([a["render"], a["rendered_depth"], a["rendered_alpha"], b["render"]], [gc, gd, ga, gc])
            torch.cuda.synchronize()
            return a, b, [p.grad.clone() for p in model.parameters()]
        finally:
            R._FUSED_NODE = True

    ra, rb, rg = run(False)
    fa, fb, fg = run(True)
    assert set(fa.keys()) == set(ra.keys())
    # parameters that already hold a dense .grad (a second backward into them, a gradient slab): the node adds into it
    # in place instead of handing autograd a tensor to add -- same sums
    for p in model.parameters():
        p.grad = torch.zeros_like(p)
    a = render(cam, model, PipelineParams(), bg)
    keep = [p.grad for p in model.parameters()]
    torch.autograd.backward([a["render"], a["rendered_depth"], a["rendered_alpha"]], [gc, gd, ga])
    b = render(scam, model, PipelineParams(), bg)
    torch.autograd.backward([b["render"]], [gc])
    torch.cuda.synchronize()
    for n, p, k, r in zip(["xyz", "f_dc", "f_rest", "scaling", "rotation", "opacity"], model.parameters(), keep, rg):
        assert p.grad is k                                # still the caller's tensor
        if r.numel():
            assert rel_l2(p.grad.cpu().numpy(), r.cpu().numpy()) < 2e-4, n
    for got, ref in ((fa, ra), (fb, rb)):
        assert float((got["radii"] != ref["radii"]).float().mean()) < 1e-3
        assert got["visibility_filter"].dtype == torch.bool and got["radii"].dtype == torch.int32
        for (k, scale) in (("render", 1.0), ("rendered_depth", 10.0), ("rendered_alpha", 1.0)):
            err = ((got[k] - ref[k]).abs() / (1 + ref[k].abs())).detach()
            assert got[k].shape == ref[k].shape
            assert int((err > 2e-5).sum()) <= 3 and float(err.max()) <= scale / 255, (k, float(err.max()))
    for n, g, r in zip(["xyz", "f_dc", "f_rest", "scaling", "rotation", "opacity"], fg, rg):
        if r.numel():
            assert g.shape == r.shape and rel_l2(g.cpu().numpy(), r.cpu().numpy()) < 2e-4, n
    assert rel_l2(fa["viewspace_points"].grad.cpu().numpy(), ra["viewspace_points"].grad.cpu().numpy()) < 2e-4
    assert fb["viewspace_points"].grad is not None
    # evaluation renders: exact (synchronous) forward, nothing pending, same image as the differentiated render
    lz = rasterizer._lazy
    lz.pending.clear()
    with torch.no_grad():
        ev = render(cam, model, PipelineParams(), bg)
    assert lz.pending == [] and torch.equal(ev["render"], fa["render"])
    # a pipeline flag the node does not cover falls back to the statement-by-statement path
    with torch.no_grad():
        py = render(cam, model, PipelineParams(compute_cov3D_python=True), bg)
    assert float(((py["render"] - ev["render"]).abs() / (1 + ev["render"].abs())).max()) <= 1.0 / 255


def test_fused_render_node_refuses_truncated_lists_in_backward():
    """The node's sync-free forward follows the protocol of the reference-shaped surface: N is checked inside backward()
    -- as its last act, behind the launches -- and a buffer that was too small raises out of it, before optimizer.step()
    can run; the capacity has grown and the repeated step is exact."""
    from binocular3dgs_amd import _lib, rasterizer, synth
    from binocular3dgs_amd.render import PipelineParams, render
    W, H, P = 160, 112, 4000
    model = synth.synth_model(P, seed=5, device="cuda", width=W, height=H)
    cam = synth.synth_view_set(W, H, device="cuda")[0][0]
    bg = torch.zeros(3, device="cuda")
    gc, gd, ga = synth.synth_pixel_grads(W, H, seed=2, device="cuda")
    lz = rasterizer._lazy
    lz.pending.clear()
    key = ("raw", torch.cuda.current_device(), P, W, H)
    lz.capacity.pop(key, None)

    def run():
        for p in model.parameters():
            p.grad = None
        pkg = render(cam, model, PipelineParams(), bg)
        torch.autograd.backward([pkg["render"], pkg["rendered_depth"], pkg["rendered_alpha"]], [gc, gd, ga])
        torch.cuda.synchronize()
        return pkg["render"].detach().clone()

    first = run()                                   # exact render: learns N
    assert key in lz.capacity
    second = run()                                  # sync-free
    assert torch.equal(first, second) and lz.pending == []
    lz.capacity[key] = 256
    with pytest.raises(_lib.B3gsError, match="B3GS_ERR_CAPACITY"):
        run()
    assert lz.capacity[key] > 256 and lz.pending == []
    assert torch.equal(run(), first)
