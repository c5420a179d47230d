"""CPU: host-side pieces of round 4 -- the static camera-pair block of the graph-replayed iteration, the host's knowledge of a
camera's z row (what decides whether a depth sort is launched at all), bench.py's N > 1 defaults."""
import argparse
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def test_camera_pair_slots_equal_shifted_cameras():
    from binocular3dgs_amd.camera import Camera, CameraPairSlots, look_at_orbit
    cams = [Camera(*look_at_orbit(y), 1.0, 0.8, 64, 48) for y in (5.0, -8.0, 0.0)]
    slots = CameraPairSlots(cams[0], 0.1)
    for cam, t in ((cams[1], -0.3), (cams[2], 0.25), (cams[0], 0.0)):
        slots.set(cam, t)
        s = cam.shifted(t)
        assert torch.equal(slots.cam.world_view_transform, cam.world_view_transform)
        assert torch.equal(slots.cam.full_proj_transform, cam.full_proj_transform)
        assert torch.equal(slots.cam.camera_center, cam.camera_center)
        assert torch.equal(slots.shifted.world_view_transform, s.world_view_transform)
        assert torch.equal(slots.shifted.full_proj_transform, s.full_proj_transform)
        assert torch.equal(slots.shifted.camera_center, s.camera_center)
        assert float(slots.trans_dist_dev) == float(np.float32(t))
        assert slots.shifted.same_depth_as is slots.cam and slots.cam.get_focal() == cam.get_focal()
        # the z row (every depth key is a function of it) is the input view's, bit for bit
        assert torch.equal(slots.shifted.world_view_transform[:, 2], slots.cam.world_view_transform[:, 2])


def test_shifted_camera_closed_form_against_the_reference_construction():
    """Camera.shifted() (closed form on the host, one upload) against G4 = the reference's getShiftedCamera."""
    from binocular3dgs_amd.camera import Camera
    g = np.load(os.path.join(GOLDEN, "cameras.npz"))
    for i in range(int(g["n"])):
        fx, fy, w, h = g[f"fov{i}"]
        cam = Camera(g[f"R{i}"], g[f"T{i}"], fx, fy, int(w), int(h))
        for j in range(4):
            sc = cam.shifted(float(g[f"shift{i}_{j}_t"]))
            np.testing.assert_allclose(sc.world_view_transform.numpy(), g[f"shift{i}_{j}_wvt"], rtol=0, atol=3e-6)
            np.testing.assert_allclose(sc.full_proj_transform.numpy(), g[f"shift{i}_{j}_full"], rtol=0, atol=1e-5)
            np.testing.assert_allclose(sc.camera_center.numpy(), g[f"shift{i}_{j}_center"], rtol=0, atol=1e-5)
            # only row 3 of the row-vector matrices moves
            assert torch.equal(sc.world_view_transform[:3], cam.world_view_transform[:3])
            assert torch.equal(sc.full_proj_transform[:3], cam.full_proj_transform[:3])


def test_camera_depth_key_is_the_z_row_the_constructor_produced():
    """rasterizer.camera_depth_key(): re-derived from R, T, trans, scale the way scene/cameras.py:55 derives the matrix --
    equal to the bytes of the stored z row for every golden camera (G3); Camera.shifted() inherits its parent's; a camera
    without those attributes is unknown (False)."""
    from binocular3dgs_amd.camera import Camera
    from binocular3dgs_amd.rasterizer import camera_depth_key
    g = np.load(os.path.join(GOLDEN, "cameras.npz"))
    keys = []
    for i in range(int(g["n"])):
        fx, fy, w, h = g[f"fov{i}"]
        cam = Camera(g[f"R{i}"], g[f"T{i}"], fx, fy, int(w), int(h))
        k = camera_depth_key(cam)
        assert k == np.ascontiguousarray(g[f"wvt{i}"][:, 2]).tobytes()
        assert camera_depth_key(cam.shifted(0.3)) == k and camera_depth_key(cam.shifted(-0.1).shifted(0.2)) == k
        keys.append(k)
    assert len(set(keys)) == len(keys)

    class Mini:        # scene/cameras.py:72-83 MiniCam: matrices only
        world_view_transform = torch.eye(4)
    assert camera_depth_key(Mini()) is False


def test_bench_defaults_for_one_and_for_n_ranks():
    import bench
    ns = lambda **kw: argparse.Namespace(**{**dict(optimizer=None, pipeline_ranges=-1, dp_path=False), **kw})   # noqa: E731
    a = bench.resolve_defaults(ns(), 1)
    assert (a.optimizer, a.pipeline_ranges) == ("sharded", 0)
    a = bench.resolve_defaults(ns(), 8)
    assert (a.optimizer, a.pipeline_ranges) == ("b3gs", -1)
    a = bench.resolve_defaults(ns(optimizer="sharded"), 8)
    assert (a.optimizer, a.pipeline_ranges) == ("sharded", 0)
    a = bench.resolve_defaults(ns(optimizer="b3gs", dp_path=True), 1)
    assert (a.optimizer, a.pipeline_ranges) == ("b3gs", -1)
    a = bench.resolve_defaults(ns(optimizer="b3gs", pipeline_ranges=2), 4)
    assert a.pipeline_ranges == 2
    # the rule behind -1: one all-reduce per ~48 MB of gradients (92 B per Gaussian), at most four
    from binocular3dgs_amd.step import auto_pipeline_ranges
    assert [auto_pipeline_ranges(P) for P in (30_000, 500_000, 1_000_000, 2_000_000, 8_000_000)] == [1, 1, 2, 4, 4]


def test_densification_statistics_without_boolean_indexing_keep_the_reference_bits():
    """gaussian_model.add_densification_stats / update_max_radii form the reference's sums (scene/gaussian_model.py:409-411,
    train.py:178) without boolean-mask indexing (a host sync per call on a GPU): same bits as the indexed statements."""
    import torch
    from binocular3dgs_amd.gaussian_model import GaussianModel
    torch.manual_seed(3)
    P = 777
    a, b = GaussianModel.__new__(GaussianModel), GaussianModel.__new__(GaussianModel)
    for m in (a, b):
        m._xyz = torch.zeros(P, 3)
        m.init_densification_stats()
    for _ in range(6):
        g = torch.randn(P, 3)
        vis = torch.rand(P) > 0.4
        radii = (torch.rand(P) * 60).int() * vis
        leaf = torch.zeros(P, 3, requires_grad=True)       # render()'s `viewspace_points`: the gradient is read from .grad
        leaf.grad = g
        a.add_densification_stats(leaf, vis)
        a.update_max_radii(radii, vis)
        b.xyz_gradient_accum[vis] += torch.norm(g[vis, :2], dim=-1, keepdim=True)
        b.denom[vis] += 1
        b.max_radii2D[vis] = torch.max(b.max_radii2D[vis], radii[vis].float())
        idx = torch.nonzero(vis).reshape(-1)          # an index tensor takes the indexed statements
        a.add_densification_stats(leaf, idx)
        b.xyz_gradient_accum[idx] += torch.norm(g[idx, :2], dim=-1, keepdim=True)
        b.denom[idx] += 1
    assert torch.equal(a.xyz_gradient_accum, b.xyz_gradient_accum) and torch.equal(a.denom, b.denom)
    assert torch.equal(a.max_radii2D, b.max_radii2D)


def test_lazy_output_runs_what_is_pending_at_the_first_use_and_not_for_metadata(monkeypatch):
    """rasterizer._LazyOut (what render() hands out while a differentiated render's forward waits for its partner): any torch
    function, method or operator on it launches what is pending first; shape / dtype / device-style metadata does not; the
    results of operations are plain tensors; the autograd node stays attached."""
    import torch
    import binocular3dgs_amd.rasterizer as R
    launched = []
    monkeypatch.setattr(R, "_launch_forward", lambda lst: launched.append(list(lst)))
    base = torch.ones(3, 4, 5, requires_grad=True) * 2.0
    x = base.detach().requires_grad_(True).as_subclass(R._LazyOut)

    def arm():
        R._pending_fwd.clear()
        R._pending_fwd[0] = ["view"]
        launched.clear()

    try:
        arm()
        assert x.shape == (3, 4, 5) and x.dtype == torch.float32 and x.device.type == "cpu" and x.requires_grad
        assert x.size(1) == 4 and x.dim() == 3 and len(x) == 3 and x.numel() == 60 and not x.is_cuda and x.ndim == 3
        assert launched == [] and R._pending_fwd
        for use in (lambda: x * 2.0, lambda: x[0], lambda: x.sum(), lambda: torch.stack([x]), lambda: x.detach(),
                    lambda: repr(x), lambda: x.clone(), lambda: x.cpu(), lambda: torch.nn.functional.relu(x),
                    lambda: x.unsqueeze(0), lambda: x.data_ptr(), lambda: x + x, lambda: x.tolist(),
                    lambda: x.sum().backward()):
            arm()
            r = use()
            assert launched == [["view"]] and not R._pending_fwd, use
            if torch.is_tensor(r):
                assert r is x or type(r) is torch.Tensor      # (x.cpu() of a CPU tensor is x itself)
        arm()
        R._flush_pending(1)                      # another device's list: nothing happens
        assert launched == [] and R._pending_fwd
        R._flush_pending()
        assert launched == [["view"]]
    finally:
        R._pending_fwd.clear()
