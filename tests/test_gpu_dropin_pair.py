"""GPU: what an UNCHANGED train.py:100,128 gets from render() -- the input view and its shifted partner as two render()
calls of one iteration: the second render adopts the first one's depth order (checked on the device, ABI 7), the
backward of both runs as one batch (one blend backward, one chain-rule pass) inside a plain loss.backward().

Everything is compared with the same renders done one by one (B3GS_DROPIN_ORDER_HINT / _INPLACE_GRADS switched off):
tile lists bit for bit, gradients up to fp32 summation order (per-view results added by autograd vs both views summed in
the chain-rule kernel; fp32 atomics): 1e-4 relative L2, half the suite's 2e-4 bound against the oracle."""
import os

import numpy as np
import pytest
import torch

from helpers import rel_l2

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _model(P=30000, W=208, H=144, seed=7):
    from binocular3dgs_amd import synth
    return synth.synth_model(P, seed=seed, device="cuda", width=W, height=H)


def _state(pkg, W, H):
    """Tile lists of a raw-node render, from what its autograd node saved."""
    from binocular3dgs_amd import debug
    node = pkg["render"].grad_fn
    saved = node.saved_tensors
    radii, geom, binning, img = saved[6:10]
    P = radii.shape[0]
    n = int(debug.state_views(P, W, H, 0, geom, None, img)["counts"][0].item())
    v = debug.state_views(P, W, H, n, geom, binning, img)
    return n, v


class _Cam:
    """A camera as the REFERENCE builds it: the matrices of tests/golden/cameras.npz (G3 / G4: scene/cameras.py and
    Scene.getShiftedCamera run in the build container by tests/golden/make_golden.py)."""

    def __init__(self, g, wvt, full, center, fov, uid=0):
        fx, fy, w, h = fov
        self.FoVx, self.FoVy, self.image_width, self.image_height = float(fx), float(fy), int(w), int(h)
        self.world_view_transform = torch.from_numpy(np.ascontiguousarray(wvt)).cuda()
        self.full_proj_transform = torch.from_numpy(np.ascontiguousarray(full)).cuda()
        self.camera_center = torch.from_numpy(np.ascontiguousarray(center)).cuda()
        self.uid = uid


class _Anon:
    """A camera the host knows nothing about (no R / T / same_depth_as: camera_depth_key() says False): only the tensors
    render() reads -- MiniCam-like (scene/cameras.py:72-83).  Hints for such cameras run the device-gated sort."""

    def __init__(self, cam):
        for k in ("FoVx", "FoVy", "image_width", "image_height", "world_view_transform", "full_proj_transform",
                  "camera_center"):
            setattr(self, k, getattr(cam, k))


def _golden_pairs():
    g = np.load(os.path.join(GOLDEN, "cameras.npz"))
    out = []
    for i in range(int(g["n"])):
        cam = _Cam(g, g[f"wvt{i}"], g[f"full{i}"], g[f"center{i}"], g[f"fov{i}"], uid=i)
        for j in range(4):
            out.append((cam, _Cam(g, g[f"shift{i}_{j}_wvt"], g[f"shift{i}_{j}_full"], g[f"shift{i}_{j}_center"], g[f"fov{i}"], uid=i)))
    return out


@pytest.fixture(autouse=True)
def _forward_mode(request):
    """Tests named test_lazy_* run with the default (a differentiated render's forward waits for its partner:
    rasterizer._LazyOut); all others with B3GS_DROPIN_LAZY=0 semantics -- every render() launches its own forward before it
    returns, which is what the depth-order hint / adoption mechanism they pin is about."""
    import binocular3dgs_amd.rasterizer as R
    R._flush_pending()
    R._S.lazy_adapt.clear()       # (the adaptive batch size of pending forwards starts from its initial state in every test)
    old = R._LAZY_FWD, R._LAZY_WHEN_IDLE
    R._LAZY_FWD = request.node.name.startswith("test_lazy")
    R._LAZY_WHEN_IDLE = True      # (these tests pin the pending-forward machinery itself: it must engage on an idle device too)
    yield
    R._flush_pending()
    R._LAZY_FWD, R._LAZY_WHEN_IDLE = old


def _render_pair(model, cam, scam, bg, hint, inplace):
    import binocular3dgs_amd.rasterizer as R
    from binocular3dgs_amd.render import PipelineParams, render
    R._ORDER_HINT, R._INPLACE_GRADS = hint, inplace
    R._flush_pending()
    R._order_hint.clear()
    R._last_raw_ctx.clear()
    try:
        a = render(cam, model, PipelineParams(), bg)
        b = render(scam, model, PipelineParams(), bg)
    finally:
        R._ORDER_HINT, R._INPLACE_GRADS = True, True
    return a, b


@pytest.mark.parametrize("known", [True, False])
def test_shifted_render_adopts_the_depth_order_lists_bit_identical(known):
    """The second render of a pair takes the first one's depth order -- its key-mismatch word stays zero -- and its tile
    lists, ranges and images equal a render that sorted its own keys, bit for bit.  known: the host knows the z rows are
    equal (Camera.shifted(): the sort is not launched); otherwise the sort launches are gated on the device's comparison."""
    import binocular3dgs_amd.rasterizer as R
    from binocular3dgs_amd import synth
    W, H = 208, 144
    model = _model(W=W, H=H)
    bg = torch.tensor([0.1, 0.0, 0.2], device="cuda")
    pairs = [(c, s) if known else (_Anon(c), _Anon(s)) for c, s, _ in synth.synth_view_set(W, H, device="cuda")]
    R._lazy.trust_hints = True
    for cam, scam in pairs:
        t0 = R._stats["trusted"]
        before = R._stats["hinted"]
        a1, b1 = _render_pair(model, cam, scam, bg, hint=True, inplace=True)
        assert R._stats["hinted"] == before + 1 and R._stats["trusted"] == t0 + int(known)
        words = R._order_hint[0]["words"]
        assert int(words[2].item()) == 0, "the shifted camera must have the z row of its input view"
        a0, b0 = _render_pair(model, cam, scam, bg, hint=False, inplace=True)
        for x, y in ((a1, a0), (b1, b0)):
            nx, vx = _state(x, W, H)
            ny, vy = _state(y, W, H)
            assert nx == ny and nx > 0
            for k in ("point_list", "tile_ids", "ranges", "n_contrib", "tiles_touched", "depth_bits"):
                assert torch.equal(vx[k], vy[k]), k
            for k in ("render", "rendered_depth", "rendered_alpha", "radii"):
                assert torch.equal(x[k], y[k]), k


def test_reference_built_shifted_cameras_share_the_order_when_the_z_row_survives():
    """G4: the pairs the reference's own Scene.getShiftedCamera produced (two host-side float64 matrix inversions, then a
    cast to fp32).  In 16 of the 20 golden pairs the z row of the shifted view matrix equals the input view's bit for bit
    and the order is adopted; in the other 4 the translation entry comes out one ulp off, every depth key moves, the
    mismatch word goes up and the render sorts its own keys -- what the reference would do.  Lists equal the per-view sort
    in both cases."""
    import binocular3dgs_amd.rasterizer as R
    from binocular3dgs_amd import synth
    pairs = _golden_pairs()
    assert len(pairs) == 20
    adopted = same_row = 0
    for cam, scam in pairs:
        W, H = cam.image_width, cam.image_height
        # Gaussians in front of THIS camera: the synthetic cloud lives in the view frustum of the identity camera
        model = synth.synth_model(6000, seed=3, device="cuda", width=W, height=H)
        with torch.no_grad():
            c2w = torch.linalg.inv(cam.world_view_transform)            # row-vector convention
            xyz1 = torch.cat([model._xyz, torch.ones_like(model._xyz[:, :1])], 1) @ c2w
            model._xyz.copy_(xyz1[:, :3])
        bg = torch.zeros(3, device="cuda")
        a1, b1 = _render_pair(model, cam, scam, bg, hint=True, inplace=True)
        row_equal = bool(torch.equal(cam.world_view_transform[:, 2], scam.world_view_transform[:, 2]))
        took = int(R._order_hint[0]["words"][2].item()) == 0
        assert took == row_equal, (took, row_equal)
        adopted += took
        same_row += row_equal
        a0, b0 = _render_pair(model, cam, scam, bg, hint=False, inplace=True)
        n1, v1 = _state(b1, W, H)
        n0, v0 = _state(b0, W, H)
        assert n1 == n0 and int((a1["radii"] > 0).sum()) > 100
        if n1:
            assert torch.equal(v1["point_list"], v0["point_list"]) and torch.equal(v1["ranges"], v0["ranges"])
        assert torch.equal(b1["render"], b0["render"])
    assert adopted == same_row == 16, (adopted, same_row)


def test_a_different_camera_or_moved_gaussians_fall_back_to_their_own_sort():
    """The hint is only a guess: another camera (different z row) raises the mismatch word and the render sorts itself;
    Gaussians moved behind autograd's back (same storage, same version counter) are caught by the same key comparison."""
    import binocular3dgs_amd.rasterizer as R
    from binocular3dgs_amd import synth
    from binocular3dgs_amd.render import PipelineParams, render
    W, H = 208, 144
    model = _model(W=W, H=H)
    bg = torch.zeros(3, device="cuda")
    (c0, _, _), (c1, _, _) = synth.synth_view_set(W, H, device="cuda")[:2]
    c0, c1 = _Anon(c0), _Anon(c1)                                       # (the host knows nothing about their z rows)
    R._order_hint.clear()
    render(c0, model, PipelineParams(), bg)
    other = render(c1, model, PipelineParams(), bg)                     # different yaw: every key differs
    assert int(R._order_hint[0]["words"][2].item()) == 1
    R._ORDER_HINT = False
    try:
        want = render(c1, model, PipelineParams(), bg)
    finally:
        R._ORDER_HINT = True
    n1, v1 = _state(other, W, H)
    n0, v0 = _state(want, W, H)
    assert n1 == n0 and torch.equal(v1["point_list"], v0["point_list"]) and torch.equal(other["render"], want["render"])
    # same camera, positions changed through .data (no version bump)
    R._order_hint.clear()
    render(c0, model, PipelineParams(), bg)
    model._xyz.data[:, 2] += 0.37 * torch.rand(model._xyz.shape[0], device="cuda")
    moved = render(c0, model, PipelineParams(), bg)
    assert int(R._order_hint[0]["words"][2].item()) == 1
    R._ORDER_HINT = False
    try:
        want = render(c0, model, PipelineParams(), bg)
    finally:
        R._ORDER_HINT = True
    n1, v1 = _state(moved, W, H)
    n0, v0 = _state(want, W, H)
    assert n1 == n0 and torch.equal(v1["point_list"], v0["point_list"]) and torch.equal(moved["render"], want["render"])


def _loss(a, b, gc, gd, ga):
    return (a["render"] * gc).sum() + (a["rendered_depth"] * gd).sum() + (a["rendered_alpha"] * ga).sum() + \
        (b["render"] * gc.flip(-1)).sum()


def test_pair_backward_runs_as_one_batch_and_equals_per_node_gradients():
    """loss.backward() over an (input, shifted) pair: the later render's node defers to the earlier one, which launches ONE
    blend backward + ONE chain-rule pass for both and adds into .grad; parameter gradients and both `viewspace_points`
    gradients equal the per-node path (gradients returned to autograd) -- with .grad unset, preset (accumulation) and
    under retain_graph (second backward of the same graph)."""
    import binocular3dgs_amd.rasterizer as R
    from binocular3dgs_amd import synth
    W, H = 208, 144
    model = _model(W=W, H=H)
    bg = torch.zeros(3, device="cuda")
    cam, scam, _ = synth.synth_view_set(W, H, device="cuda")[1]
    gc, gd, ga = synth.synth_pixel_grads(W, H, seed=5, device="cuda")
    # reference: per node, gradients returned to autograd
    for p in model.parameters():
        p.grad = None
    a, b = _render_pair(model, cam, scam, bg, hint=False, inplace=False)
    R._INPLACE_GRADS = False
    try:
        _loss(a, b, gc, gd, ga).backward()
    finally:
        R._INPLACE_GRADS = True
    ref = [p.grad.clone() for p in model.parameters()]
    ref_m2d = (a["viewspace_points"].grad.clone(), b["viewspace_points"].grad.clone())
    assert float(ref_m2d[1].abs().max()) > 0
    for preset in (False, True):
        for p in model.parameters():
            p.grad = torch.ones_like(p) if preset else None
        a, b = _render_pair(model, cam, scam, bg, hint=True, inplace=True)
        s0 = dict(R._stats)
        _loss(a, b, gc, gd, ga).backward(retain_graph=True)
        assert R._stats["deferred"] == s0["deferred"] + 1 and R._stats["launches"] == s0["launches"] + 1
        assert R._stats["batched_views"] == s0["batched_views"] + 2
        for n, p, r in zip("xyz f_dc f_rest scaling rotation opacity".split(), model.parameters(), ref):
            want = r + 1.0 if preset else r
            assert rel_l2(p.grad.cpu().numpy(), want.cpu().numpy()) < 1e-4, (n, preset)
        for got, want in zip((a["viewspace_points"].grad, b["viewspace_points"].grad), ref_m2d):
            assert rel_l2(got.cpu().numpy(), want.cpu().numpy()) < 1e-4
        # the same graph again: accumulates a second time
        _loss(a, b, gc, gd, ga).backward()
        for p, r in zip(model.parameters(), ref):
            want = 2.0 * r + (1.0 if preset else 0.0)
            assert rel_l2(p.grad.cpu().numpy(), want.cpu().numpy()) < 1e-4
    for pool in R._raw_scratch.values():
        for s in pool:
            assert float(s.abs().max()) == 0.0, "scratch rows must be left clean"


@pytest.mark.parametrize("K", [4, 1, 16])
def test_six_renders_of_an_iteration_run_their_backward_as_one_launch(K):
    """K = SH coefficients per channel (degree 1 / 0 / 3: features_rest empty, resp. beyond the chain-rule kernel's register
    window)."""
    from binocular3dgs_amd import synth
    import binocular3dgs_amd.rasterizer as R
    from binocular3dgs_amd.render import PipelineParams, render
    W, H = 160, 120
    model = synth.synth_model(12000, seed=7, device="cuda", width=W, height=H, K=K)
    model.active_sh_degree = {1: 0, 4: 1, 16: 3}[K]
    bg = torch.zeros(3, device="cuda")
    gc, gd, ga = synth.synth_pixel_grads(W, H, seed=2, device="cuda")
    cams = [c for cam, scam, _ in synth.synth_view_set(W, H, device="cuda") for c in (cam, scam)]

    def run(inplace):
        R._INPLACE_GRADS = inplace
        try:
            for p in model.parameters():
                p.grad = None
            pk = [render(c, model, PipelineParams(), bg) for c in cams]
            outs, grads = [], []
            for k, o in enumerate(pk):
                outs += [o["render"]] + ([o["rendered_depth"], o["rendered_alpha"]] if k % 2 == 0 else [])
                grads += [gc] + ([gd, ga] if k % 2 == 0 else [])
            torch.autograd.backward(outs, grads)
        finally:
            R._INPLACE_GRADS = True
        return [p.grad.clone() for p in model.parameters()], [o["viewspace_points"].grad.clone() for o in pk]

    ref, ref_m = run(False)
    s0 = dict(R._stats)
    got, got_m = run(True)
    assert R._stats["launches"] == s0["launches"] + 1 and R._stats["batched_views"] == s0["batched_views"] + 6
    assert R._stats["hinted"] >= s0["hinted"] + 3
    for g, r in zip(got + got_m, ref + ref_m):
        if r.numel():
            assert rel_l2(g.cpu().numpy(), r.cpu().numpy()) < 1e-4


def test_autograd_grad_and_partial_backward_leave_dot_grad_alone():
    """torch.autograd.grad(loss, params) must RETURN the gradients and must not touch .grad; backward(inputs=[one
    parameter]) must fill only that one; a hooked parameter sees its hook."""
    from binocular3dgs_amd import synth
    import binocular3dgs_amd.rasterizer as R
    W, H = 160, 120
    model = _model(P=8000, W=W, H=H)
    bg = torch.zeros(3, device="cuda")
    cam, scam, _ = synth.synth_view_set(W, H, device="cuda")[0]
    gc, gd, ga = synth.synth_pixel_grads(W, H, seed=4, device="cuda")
    params = model.parameters()
    for p in params:
        p.grad = None
    a, b = _render_pair(model, cam, scam, bg, hint=True, inplace=True)
    _loss(a, b, gc, gd, ga).backward()
    ref = [p.grad.clone() for p in params]
    # autograd.grad: returned, .grad untouched (preset to a sentinel)
    for p in params:
        p.grad = torch.full_like(p, 3.0)
    a, b = _render_pair(model, cam, scam, bg, hint=True, inplace=True)
    s0 = dict(R._stats)
    got = torch.autograd.grad(_loss(a, b, gc, gd, ga), params)
    assert R._stats["deferred"] == s0["deferred"]
    for g, r, p in zip(got, ref, params):
        assert g is not None and rel_l2(g.cpu().numpy(), r.cpu().numpy()) < 1e-4
        assert torch.equal(p.grad, torch.full_like(p, 3.0))
    # backward(inputs=[xyz]): only xyz receives
    for p in params:
        p.grad = None
    a, b = _render_pair(model, cam, scam, bg, hint=True, inplace=True)
    _loss(a, b, gc, gd, ga).backward(inputs=[model._xyz])
    assert rel_l2(model._xyz.grad.cpu().numpy(), ref[0].cpu().numpy()) < 1e-4
    assert all(p.grad is None for p in params[1:])
    # a tensor hook on one parameter is honoured (gradients go through autograd)
    seen = []
    h = model._opacity.register_hook(lambda g: seen.append(float(g.abs().sum())))
    try:
        for p in params:
            p.grad = None
        a, b = _render_pair(model, cam, scam, bg, hint=True, inplace=True)
        _loss(a, b, gc, gd, ga).backward()
    finally:
        h.remove()
    assert len(seen) >= 1 and sum(seen) > 0
    for p, r in zip(params, ref):
        assert rel_l2(p.grad.cpu().numpy(), r.cpu().numpy()) < 1e-4


def test_only_the_shifted_loss_backpropagated_still_delivers():
    """A backward that reaches only the LATER render (its partner is not part of the graph task): nothing is deferred."""
    from binocular3dgs_amd import synth
    import binocular3dgs_amd.rasterizer as R
    W, H = 160, 120
    model = _model(P=8000, W=W, H=H)
    bg = torch.zeros(3, device="cuda")
    cam, scam, _ = synth.synth_view_set(W, H, device="cuda")[0]
    gc, _, _ = synth.synth_pixel_grads(W, H, seed=4, device="cuda")
    for p in model.parameters():
        p.grad = None
    a, b = _render_pair(model, cam, scam, bg, hint=True, inplace=True)
    s0 = dict(R._stats)
    (b["render"] * gc).sum().backward()
    assert R._stats["deferred"] == s0["deferred"] and R._stats["launches"] == s0["launches"] + 1
    got = model._xyz.grad.clone()
    R._INPLACE_GRADS = False
    try:
        for p in model.parameters():
            p.grad = None
        a, b = _render_pair(model, cam, scam, bg, hint=False, inplace=False)
        R._INPLACE_GRADS = False
        (b["render"] * gc).sum().backward()
    finally:
        R._INPLACE_GRADS = True
    assert rel_l2(got.cpu().numpy(), model._xyz.grad.cpu().numpy()) < 1e-4


def test_host_knowledge_of_the_z_row_skips_the_sort_launches_and_is_still_checked():
    """camera_depth_key(): Camera.shifted() keeps its parent's z row, so the second render of a pair does not even launch
    its depth sort (`hint_trusted`); the device still compares every key.  A camera that LIES about its row (matrices that
    are not what its R / T say) is caught: the step is refused out of backward() and the knowledge is not trusted again."""
    import binocular3dgs_amd.rasterizer as R
    from binocular3dgs_amd import _lib, synth
    from binocular3dgs_amd.render import PipelineParams, render
    W, H = 208, 144
    model = _model(W=W, H=H)
    bg = torch.zeros(3, device="cuda")
    gc, _, _ = synth.synth_pixel_grads(W, H, seed=5, device="cuda")
    (c0, s0, _), (c1, _, _) = synth.synth_view_set(W, H, device="cuda")[:2]
    assert R.camera_depth_key(c0) == R.camera_depth_key(s0) != R.camera_depth_key(c1)
    assert R.camera_depth_key(c0) == c0.world_view_transform[:, 2].cpu().numpy().tobytes()
    R._lazy.trust_hints = True
    t0 = R._stats["trusted"]
    _render_pair(model, c0, s0, bg, hint=True, inplace=True)
    assert R._stats["trusted"] == t0 + 1 and int(R._order_hint[0]["words"][2].item()) == 0
    # known to differ: no hint at all
    h0 = R._stats["hinted"]
    _render_pair(model, c0, c1, bg, hint=True, inplace=True)
    assert R._stats["hinted"] == h0
    # a liar
    import copy
    liar = copy.copy(c1)
    liar._b3gs_zkey = R.camera_depth_key(c0)
    for p in model.parameters():
        p.grad = None
    R._order_hint.clear()
    a = render(c0, model, PipelineParams(), bg)
    b = render(liar, model, PipelineParams(), bg)
    with pytest.raises(_lib.B3gsError, match="depth order of another view"):
        ((a["render"] + b["render"]) * gc).sum().backward()
    assert R._lazy.trust_hints is False
    try:
        R._order_hint.clear()
        render(c0, model, PipelineParams(), bg)
        again = render(liar, model, PipelineParams(), bg)          # verified (gated) now: sorts itself
        assert int(R._order_hint[0]["words"][2].item()) == 1
        R._ORDER_HINT = False
        want = render(c1, model, PipelineParams(), bg)
        R._ORDER_HINT = True
        assert torch.equal(again["render"], want["render"])
    finally:
        R._ORDER_HINT = True
        R._lazy.trust_hints = True


def test_pair_members_of_different_image_size_and_an_empty_model():
    """The batched backward takes views of different W x H (same Gaussians); zero Gaussians render the background."""
    from binocular3dgs_amd import synth
    from binocular3dgs_amd.camera import Camera, look_at_orbit
    from binocular3dgs_amd.gaussian_model import GaussianModel
    import binocular3dgs_amd.rasterizer as R
    from binocular3dgs_amd.render import PipelineParams, render
    model = _model(P=9000, W=160, H=120)
    bg = torch.tensor([0.2, 0.1, 0.0], device="cuda")
    Rm, T = look_at_orbit(4.0)
    big = Camera(Rm, T, 1.0, 0.8, 160, 120, device="cuda")
    small = Camera(Rm, T, 1.0, 0.8, 96, 72, device="cuda")

    def run(inplace):
        R._INPLACE_GRADS = inplace
        R._order_hint.clear()
        R._last_raw_ctx.clear()
        try:
            for p in model.parameters():
                p.grad = None
            a = render(big, model, PipelineParams(), bg)
            b = render(small, model, PipelineParams(), bg)
            (a["render"].sum() * 0.5 + (b["render"] ** 2).sum() + b["rendered_alpha"].sum()).backward()
        finally:
            R._INPLACE_GRADS = True
        return [p.grad.clone() for p in model.parameters()]
    s0 = dict(R._stats)
    got = run(True)
    assert R._stats["launches"] == s0["launches"] + 1 and R._stats["batched_views"] == s0["batched_views"] + 2
    ref = run(False)
    for g, r in zip(got, ref):
        assert rel_l2(g.cpu().numpy(), r.cpu().numpy()) < 1e-4
    # P = 0
    e = lambda *sh: torch.zeros(sh, device="cuda")                                   # noqa: E731
    empty = GaussianModel.from_tensors(e(0, 3), e(0, 1, 3), e(0, 3, 3), e(0, 3), e(0, 4), e(0, 1), sh_degree=1, device="cuda")
    pkg = render(big, empty, PipelineParams(), bg)
    # (the upstream binding returns its zero-initialised image when there is nothing to rasterize, not the background)
    assert pkg["radii"].numel() == 0 and float(pkg["render"].abs().max()) == 0.0 and pkg["render"].shape == (3, 120, 160)
    pkg["render"].sum().backward()
    assert empty._xyz.grad is None or empty._xyz.grad.numel() == 0


def test_forward_on_a_side_stream_backward_called_from_the_default_stream():
    """The engine runs each node on its forward's stream; the gradients this module writes into `.grad` itself must be visible
    to the stream `backward()` was called from when it returns (the end-of-backward callback makes it wait)."""
    from binocular3dgs_amd import synth
    W, H = 160, 120
    model = _model(P=9000, W=W, H=H)
    bg = torch.zeros(3, device="cuda")
    cam, scam, _ = synth.synth_view_set(W, H, device="cuda")[0]
    gc, gd, ga = synth.synth_pixel_grads(W, H, seed=6, device="cuda")
    for p in model.parameters():
        p.grad = None
    a, b = _render_pair(model, cam, scam, bg, hint=True, inplace=True)
    _loss(a, b, gc, gd, ga).backward()
    torch.cuda.synchronize()
    ref = [p.grad.clone() for p in model.parameters()]
    side = torch.cuda.Stream()
    for _ in range(3):
        for p in model.parameters():
            p.grad = None
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            a, b = _render_pair(model, cam, scam, bg, hint=True, inplace=True)
            loss = _loss(a, b, gc, gd, ga)
        torch.cuda.current_stream().wait_stream(side)
        loss.backward()                                   # called from the default stream
        got = [p.grad.clone() for p in model.parameters()]   # read on the default stream, no device-wide sync in between
        torch.cuda.synchronize()
        for g, r in zip(got, ref):
            assert rel_l2(g.cpu().numpy(), r.cpu().numpy()) < 1e-4


def test_a_backward_that_raised_midway_leaves_nothing_behind_for_the_next_attempt():
    """retain_graph=True, a hook between the input view's image and the loss raises: the shifted render's node has already
    parked its job at the input view's node, which never runs.  The second attempt on the same graph must deliver each
    render's gradient ONCE (the parked job belongs to the failed graph task and is dropped)."""
    import binocular3dgs_amd.rasterizer as R
    from binocular3dgs_amd import synth
    W, H = 160, 120
    model = _model(P=9000, W=W, H=H)
    bg = torch.zeros(3, device="cuda")
    cam, scam, _ = synth.synth_view_set(W, H, device="cuda")[0]
    gc, gd, ga = synth.synth_pixel_grads(W, H, seed=6, device="cuda")
    for p in model.parameters():
        p.grad = None
    a, b = _render_pair(model, cam, scam, bg, hint=True, inplace=True)
    _loss(a, b, gc, gd, ga).backward()
    ref = [p.grad.clone() for p in model.parameters()]

    from binocular3dgs_amd.render import PipelineParams, render
    boom = {"on": True}

    def hook(g):
        if boom["on"]:
            raise RuntimeError("interrupted")
        return g

    R._order_hint.clear()
    R._last_raw_ctx.clear()
    a = render(cam, model, PipelineParams(), bg)
    img = a["render"] * 1.0                # (created BEFORE the shifted render: the engine runs that render's node first)
    img.register_hook(hook)
    b = render(scam, model, PipelineParams(), bg)
    loss = (img * gc).sum() + (a["rendered_depth"] * gd).sum() + (a["rendered_alpha"] * ga).sum() + \
        (b["render"] * gc.flip(-1)).sum()
    for p in model.parameters():
        p.grad = None
    d0 = R._stats["deferred"]
    with pytest.raises(RuntimeError, match="interrupted"):
        loss.backward(retain_graph=True)
    assert R._stats["deferred"] == d0 + 1, "the shifted render's job was parked before the hook fired"
    for p in model.parameters():
        p.grad = None                      # (whatever an interrupted backward left is discarded, as always)
    boom["on"] = False
    loss.backward()
    for n, p, r in zip("xyz f_dc f_rest scaling rotation opacity".split(), model.parameters(), ref):
        assert rel_l2(p.grad.cpu().numpy(), r.cpu().numpy()) < 1e-4, n
    for pool in R._raw_scratch.values():
        for s in pool:
            assert float(s.abs().max()) == 0.0, "scratch rows must be left clean"


# ---------------------------------------------------------------------------------------------------------------------
# the default: the first render's forward waits for its partner (rasterizer._LazyOut)
# ---------------------------------------------------------------------------------------------------------------------
def _plain(t):
    """the tensor behind a _LazyOut without going through torch's function dispatch (= without making it run)"""
    with torch._C.DisableTorchFunctionSubclass():
        return t.as_subclass(torch.Tensor)


def test_lazy_pair_is_one_two_view_forward_with_the_results_of_two_single_ones():
    """render(input view); render(shifted view) with nothing in between (train.py:100-128): ONE b3gs_forward_raw_batch of
    two views when the second call arrives.  Until then the first render's images hold NaN; afterwards tile lists, images,
    radii, visibility and all gradients equal the renders launched one by one."""
    import binocular3dgs_amd.rasterizer as R
    from binocular3dgs_amd import synth
    from binocular3dgs_amd.render import PipelineParams, render
    W, H = 208, 144
    model = _model(W=W, H=H)
    bg = torch.tensor([0.1, 0.0, 0.2], device="cuda")
    cam, scam, _ = synth.synth_view_set(W, H, device="cuda")[1]
    gc, gd, ga = synth.synth_pixel_grads(W, H, seed=5, device="cuda")
    R._LAZY_FWD = False
    for p in model.parameters():
        p.grad = None
    a0, b0 = _render_pair(model, cam, scam, bg, hint=False, inplace=True)
    want_state = [_state(y, W, H) for y in (a0, b0)]
    _loss(a0, b0, gc, gd, ga).backward()
    ref = [p.grad.clone() for p in model.parameters()]
    ref_m2d = (a0["viewspace_points"].grad.clone(), b0["viewspace_points"].grad.clone())
    R._LAZY_FWD = True
    for p in model.parameters():
        p.grad = None
    R._order_hint.clear()
    R._last_raw_ctx.clear()
    s0 = dict(R._stats)
    a = render(cam, model, PipelineParams(), bg)
    assert type(a["render"]) is R._LazyOut and len(R._pending_fwd[0]) == 1
    assert a["render"].shape == (3, H, W) and a["render"].requires_grad and a["radii"].dtype == torch.int32   # metadata: no launch
    assert len(R._pending_fwd[0]) == 1 and R._stats["lazy_batches"] == s0["lazy_batches"]
    assert bool(torch.isnan(_plain(a["render"])).all()) and bool(torch.isnan(_plain(a["rendered_depth"])).all())
    b = render(scam, model, PipelineParams(), bg)
    assert not R._pending_fwd and R._stats["lazy_batches"] == s0["lazy_batches"] + 1
    assert R._stats["lazy_views"] == s0["lazy_views"] + 2
    assert R._stats["shared"] == s0["shared"] + 1          # Camera.shifted(): one depth sort for the pair, keys compared
    assert type(b["render"]) is torch.Tensor                              # (the call that completed the batch)
    for x, y, (ny, vy) in ((a, a0, want_state[0]), (b, b0, want_state[1])):
        nx, vx = _state(x, W, H)
        assert nx == ny and nx > 0
        for k in ("point_list", "tile_ids", "ranges", "n_contrib", "tiles_touched", "depth_bits"):
            assert torch.equal(vx[k], vy[k]), k
        for k in ("render", "rendered_depth", "rendered_alpha", "radii", "visibility_filter"):
            assert torch.equal(x[k], y[k]), k
        # (ABI 8: the projection writes the visibility bytes itself)
        assert x["visibility_filter"].dtype == torch.bool and torch.equal(x["visibility_filter"], x["radii"] > 0)
        assert torch.equal(y["visibility_filter"], y["radii"] > 0) and 0 < int(y["visibility_filter"].sum()) < y["radii"].numel()
    _loss(a, b, gc, gd, ga).backward()
    for n, p, r in zip("xyz f_dc f_rest scaling rotation opacity".split(), model.parameters(), ref):
        assert rel_l2(p.grad.cpu().numpy(), r.cpu().numpy()) < 1e-4, n
    for got, want in zip((a["viewspace_points"].grad, b["viewspace_points"].grad), ref_m2d):
        assert rel_l2(got.cpu().numpy(), want.cpu().numpy()) < 1e-4


def _as4(t, R):
    """[C,H,W] -> [1,C,H,W] WITHOUT a torch function on the pending tensor (a view made under the dispatch guard, so that the
    compiled entry point is really the first consumer) -- still a _LazyOut."""
    with torch._C.DisableTorchFunctionSubclass():
        v = t.unsqueeze(0)
    return v.as_subclass(R._LazyOut) if type(v) is not R._LazyOut else v


@pytest.mark.parametrize("how", ["operator", "function", "method", "index", "repr", "numpy", "detach", "backward", "other_model",
                                 "no_grad_render", "other_size", "other_stream", "custom_function", "deepcopy", "conv",
                                 "compiled_l1", "compiled_ssim", "compiled_warp", "compiled_smooth"])
def test_lazy_forward_runs_at_the_first_use_of_an_output(how):
    """Whatever touches a pending output first makes the forward run before it: the values are those of an eager render."""
    import binocular3dgs_amd.rasterizer as R
    from binocular3dgs_amd import synth
    from binocular3dgs_amd.render import PipelineParams, render
    W, H = 160, 120
    model = _model(P=9000, W=W, H=H)
    bg = torch.zeros(3, device="cuda")
    cam, scam, _ = synth.synth_view_set(W, H, device="cuda")[0]
    R._LAZY_FWD = False
    want = render(cam, model, PipelineParams(), bg)
    R._LAZY_FWD = True
    R._order_hint.clear()
    pkg = render(cam, model, PipelineParams(), bg)
    img = pkg["render"]
    assert R._pending_fwd and type(img) is R._LazyOut
    s0 = R._stats["lazy_batches"]
    if how == "operator":
        got = (img * 1.0)
    elif how == "function":
        got = torch.stack([img])[0]
    elif how == "method":
        got = img.clone()
    elif how == "index":
        got = torch.cat([img[c:c + 1] for c in range(3)])
    elif how == "repr":
        repr(pkg["rendered_alpha"])
        got = img
    elif how == "numpy":
        got = torch.from_numpy(img.detach().cpu().numpy()).cuda()
    elif how == "detach":
        got = img.detach()
    elif how == "custom_function":
        class Twice(torch.autograd.Function):       # (apply() itself does not dispatch: the first torch call inside does)
            @staticmethod
            def forward(ctx, x):
                return x * 2.0

            @staticmethod
            def backward(ctx, g):
                return g * 2.0
        got = Twice.apply(img) * 0.5
    elif how == "deepcopy":
        import copy
        got = copy.deepcopy(pkg["rendered_depth"].detach())
        assert not bool(torch.isnan(got).any())
        got = img
    elif how.startswith("compiled_"):
        # the loss functions are entry points of the compiled `_C` module: they read raw pointers without going through torch's
        # function dispatch -- their python fronts launch what is pending first (rasterizer.touch_pending)
        from binocular3dgs_amd.graphics_utils import inverse_warp_images
        from binocular3dgs_amd.loss_utils import SmoothLoss, l1_loss, ssim
        other = torch.rand(3, H, W, device="cuda")
        if how == "compiled_l1":
            v, ref = l1_loss(img, other), (want["render"] - other).abs().mean()
        elif how == "compiled_ssim":
            v, ref = ssim(img, other), ssim(want["render"].detach(), other)
        elif how == "compiled_smooth":
            d = pkg["rendered_depth"]
            assert type(d) is R._LazyOut
            v = SmoothLoss().forward(disparity=_as4(d, R), image=other[None])
            ref = SmoothLoss().forward(want["rendered_depth"].detach()[None], other[None])
        else:
            v = inverse_warp_images(_as4(img, R), torch.zeros(1, 1, H, W, device="cuda")).mean()
            ref = want["render"][..., :-1].sum() / want["render"].numel()    # (a zero shift keeps every column but the last)
        assert not bool(torch.isnan(v)) and abs(float(v) - float(ref)) <= 1e-5 * max(1.0, abs(float(ref))), (float(v), float(ref))
        got = img
    elif how == "conv":
        k = torch.zeros(3, 1, 1, 1, device="cuda") + 1.0
        got = torch.nn.functional.conv2d(img.unsqueeze(0), k, groups=3)[0]
    elif how == "backward":
        for p in model.parameters():
            p.grad = None
        torch.autograd.backward([img], [torch.ones(3, H, W, device="cuda")])
        assert model._xyz.grad is not None and float(model._xyz.grad.abs().max()) > 0
        got = img
    elif how == "other_model":
        render(cam, _model(P=5000, W=W, H=H, seed=3), PipelineParams(), bg)
        got = img
    elif how == "no_grad_render":
        with torch.no_grad():
            render(scam, model, PipelineParams(), bg)
        got = img
    elif how == "other_size":
        small = synth.synth_view_set(96, 64, device="cuda")[0][0]
        render(small, model, PipelineParams(), bg)
        assert not R._pending_fwd            # (the first render of a shape is exact and synchronous: it never waits)
        got = img
    else:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            render(scam, model, PipelineParams(), bg)
        torch.cuda.current_stream().wait_stream(side)
        got = img
    assert R._stats["lazy_batches"] >= s0 + 1
    got = _plain(got) if type(got) is R._LazyOut else got
    assert not bool(torch.isnan(got).any())
    if how == "conv":            # (a library convolution need not be bit-exact on x * 1)
        assert float((got - want["render"]).abs().max()) < 1e-5
    else:
        assert torch.equal(got, want["render"])
    assert torch.equal(pkg["visibility_filter"], want["visibility_filter"]) and torch.equal(pkg["radii"], want["radii"])
    R._flush_pending()


def test_lazy_switch_off_and_direct_callers_are_eager():
    """B3GS_DROPIN_LAZY=0 (module switch), rasterize_raw() called without lazy_outputs, and renders that will not be
    differentiated never wait."""
    import binocular3dgs_amd.rasterizer as R
    from binocular3dgs_amd import synth
    from binocular3dgs_amd.render import PipelineParams, render
    W, H = 160, 120
    model = _model(P=9000, W=W, H=H)
    bg = torch.zeros(3, device="cuda")
    cam, _, _ = synth.synth_view_set(W, H, device="cuda")[0]
    render(cam, model, PipelineParams(), bg)["render"].sum().item()      # (the shape's capacity is known from here on)
    with torch.no_grad():
        pkg = render(cam, model, PipelineParams(), bg)
    assert type(pkg["render"]) is torch.Tensor and not R._pending_fwd
    R._LAZY_FWD = False
    pkg = render(cam, model, PipelineParams(), bg)
    assert type(pkg["render"]) is torch.Tensor and not R._pending_fwd
    R._LAZY_FWD = True
    pkg = render(cam, model, PipelineParams(), bg)
    assert type(pkg["render"]) is R._LazyOut and R._pending_fwd
    R._flush_pending()
    assert not bool(torch.isnan(pkg["render"]).any())


def test_lazy_pair_with_a_camera_that_lies_about_its_z_row_is_refused():
    """Two pending renders whose cameras claim the same z row share one depth sort (depth_order_from + hint_trusted): the
    projection compares the keys, a difference raises bit 3 -- backward() refuses the step, the claim is not believed again
    and the next pair sorts per view (images of a per-view render)."""
    import copy
    import binocular3dgs_amd.rasterizer as R
    from binocular3dgs_amd import _lib, synth
    from binocular3dgs_amd.render import PipelineParams, render
    W, H = 208, 144
    model = _model(W=W, H=H)
    bg = torch.zeros(3, device="cuda")
    gc, _, _ = synth.synth_pixel_grads(W, H, seed=5, device="cuda")
    (c0, _, _), (c1, _, _) = synth.synth_view_set(W, H, device="cuda")[:2]
    R._LAZY_FWD = False
    want = render(c1, model, PipelineParams(), bg)["render"].detach().clone()
    R._LAZY_FWD = True
    liar = copy.copy(c1)
    liar._b3gs_zkey = R.camera_depth_key(c0)
    R._lazy.trust_hints = True
    try:
        for p in model.parameters():
            p.grad = None
        sh = R._stats["shared"]
        a = render(c0, model, PipelineParams(), bg)
        b = render(liar, model, PipelineParams(), bg)
        assert R._stats["shared"] == sh + 1
        with pytest.raises(_lib.B3gsError, match="depth order of another view"):
            ((a["render"] + b["render"]) * gc).sum().backward()
        assert R._lazy.trust_hints is False
        a = render(c0, model, PipelineParams(), bg)
        b = render(liar, model, PipelineParams(), bg)
        assert R._stats["shared"] == sh + 1 and torch.equal(b["render"], want)
    finally:
        R._lazy.trust_hints = True


def test_lazy_pair_then_a_single_render_adopts_the_order_of_the_view_that_sorted():
    """After a two-view forward whose second view borrowed the first one's depth order, the order offered to the NEXT render
    is the first view's (the borrower has no sorted arrays of its own): a third render of the same camera row, launched
    alone, must produce the lists of a render that sorted for itself."""
    import binocular3dgs_amd.rasterizer as R
    from binocular3dgs_amd import synth
    from binocular3dgs_amd.render import PipelineParams, render
    W, H = 208, 144
    model = _model(W=W, H=H)
    bg = torch.zeros(3, device="cuda")
    cam, scam, _ = synth.synth_view_set(W, H, device="cuda")[1]
    R._LAZY_FWD, R._ORDER_HINT = False, False
    want = render(cam, model, PipelineParams(), bg)
    R._LAZY_FWD, R._ORDER_HINT = True, True
    R._lazy.trust_hints = True
    sh = R._stats["shared"]
    a = render(cam, model, PipelineParams(), bg)
    b = render(scam, model, PipelineParams(), bg)
    assert R._stats["shared"] == sh + 1 and not R._pending_fwd
    t0 = R._stats["trusted"]
    c = render(cam, model, PipelineParams(), bg)          # waits alone ...
    c["render"].sum()                                     # ... launched alone, with the pair's order as a trusted hint
    assert R._stats["trusted"] == t0 + 1
    nc, vc = _state(c, W, H)
    nw, vw = _state(want, W, H)
    assert nc == nw
    for k in ("point_list", "tile_ids", "ranges", "n_contrib"):
        assert torch.equal(vc[k], vw[k]), k
    assert torch.equal(c["render"], want["render"]) and b["render"].shape == a["render"].shape


def test_lazy_max_above_two_batches_every_pending_render_of_the_iteration():
    """B3GS_DROPIN_LAZY_MAX = 6: the six renders of an iteration wait for ONE six-view forward (three pairs, three depth
    sorts), launched by the backward; gradients equal renders launched one by one."""
    import binocular3dgs_amd.rasterizer as R
    from binocular3dgs_amd import synth
    from binocular3dgs_amd.render import PipelineParams, render
    W, H = 160, 120
    model = _model(P=12000, W=W, H=H)
    bg = torch.zeros(3, device="cuda")
    cams = [c for cam, scam, _ in synth.synth_view_set(W, H, device="cuda")[:3] for c in (cam, scam)]
    gc, gd, ga = synth.synth_pixel_grads(W, H, seed=4, device="cuda")

    def run():
        for p in model.parameters():
            p.grad = None
        R._order_hint.clear()
        R._last_raw_ctx.clear()
        pk = [render(c, model, PipelineParams(), bg) for c in cams]
        outs = [x for i, k in enumerate(pk) for x in ((k["render"], k["rendered_depth"], k["rendered_alpha"]) if i % 2 == 0
                                                      else (k["render"],))]
        grads = [g for i in range(6) for g in ((gc, gd, ga) if i % 2 == 0 else (gc,))]
        torch.autograd.backward(outs, grads)
        return pk, [p.grad.clone() for p in model.parameters()]

    R._LAZY_FWD = False
    pk0, ref = run()
    R._LAZY_FWD, old = True, R._LAZY_MAX
    R._LAZY_MAX = 6
    try:
        s0 = dict(R._stats)
        pk, got = run()
        assert R._stats["lazy_batches"] == s0["lazy_batches"] + 1 and R._stats["lazy_views"] == s0["lazy_views"] + 6
        assert R._stats["shared"] == s0["shared"] + 3
    finally:
        R._LAZY_MAX = old
    for a, b in zip(pk, pk0):
        assert torch.equal(a["render"], b["render"]) and torch.equal(a["radii"], b["radii"])
        assert torch.equal(a["visibility_filter"], b["visibility_filter"])
    for n, g, r in zip("xyz f_dc f_rest scaling rotation opacity".split(), got, ref):
        assert rel_l2(g.cpu().numpy(), r.cpu().numpy()) < 1e-4, n


# ---------------------------------------------------------------------------------------------------------------------
# round 5: hardening of the zero-change surface
# ---------------------------------------------------------------------------------------------------------------------
def test_lazy_batch_size_follows_the_loop():
    """Round 6: the number of renders a pending forward waits for adapts to the loop (rasterizer._LAZY_ADAPT).
    (i) a loop that renders its six views before it consumes any: two renders per forward in the first iteration, ONE six-view
        forward from the second on, the same gradients;
    (ii) the same six renders consumed pair by pair: the first early touch shrinks the batch back to a pair, growth is blocked;
    (iii) train.py's shape (a pair, consumed at once) never leaves two."""
    import binocular3dgs_amd.rasterizer as R
    from binocular3dgs_amd import synth
    from binocular3dgs_amd.render import PipelineParams, render
    assert R._LAZY_ADAPT and R._LAZY_MAX == 2
    W, H = 160, 120
    model = _model(P=8000, W=W, H=H)
    bg = torch.zeros(3, device="cuda")
    pairs = synth.synth_view_set(W, H, device="cuda")
    gc, gd, ga = synth.synth_pixel_grads(W, H, seed=1, device="cuda")
    pipe = PipelineParams()

    def iteration(consume_per_pair):
        for p in model.parameters():
            p.grad = None
        outs, total = [], 0.0
        for cam, scam, _ in pairs:
            a, b = render(cam, model, pipe, bg), render(scam, model, pipe, bg)
            if consume_per_pair:
                total = total + _loss(a, b, gc, gd, ga)
            else:
                outs.append((a, b))
        for a, b in outs:
            total = total + _loss(a, b, gc, gd, ga)
        total.backward()
        return [p.grad.clone() for p in model.parameters()]

    s0 = R._stats["lazy_batches"]
    g1 = iteration(False)
    assert R._stats["lazy_batches"] == s0 + 3 and R._lazy_max(0) == 6          # three pairs; the rule saw six renders, all full
    g2 = iteration(False)
    assert R._stats["lazy_batches"] == s0 + 4                                    # ONE six-view forward
    g3 = iteration(False)
    assert R._stats["lazy_batches"] == s0 + 5 and R._lazy_max(0) == 6
    for a, b, c in zip(g1, g2, g3):
        assert rel_l2(b.cpu().numpy(), a.cpu().numpy()) < 1e-5 and rel_l2(c.cpu().numpy(), a.cpu().numpy()) < 1e-5
    # (ii) the consumer now touches every pair as soon as it exists
    s1 = R._stats["lazy_batches"]
    g4 = iteration(True)
    assert R._stats["lazy_batches"] == s1 + 3 and R._lazy_max(0) == 2 and R._S.lazy_adapt[0]["hold"] >= 4
    g5 = iteration(True)
    assert R._stats["lazy_batches"] == s1 + 6 and R._lazy_max(0) == 2
    for a, b in zip(g1, g5):
        assert rel_l2(b.cpu().numpy(), a.cpu().numpy()) < 1e-5
    iteration(False)                         # growth stays blocked for a while after an early touch
    assert R._lazy_max(0) == 2
    # (iii) one pair per iteration, consumed at once
    R._S.lazy_adapt.clear()
    cam, scam, _ = pairs[0]
    for _ in range(4):
        for p in model.parameters():
            p.grad = None
        a, b = render(cam, model, pipe, bg), render(scam, model, pipe, bg)
        assert not R._pending_fwd               # the second render launched the pair when it returned
        _loss(a, b, gc, gd, ga).backward()
        assert R._lazy_max(0) == 2


def test_lazy_outputs_become_plain_tensors_once_launched_and_survive_save_dlpack_numpy_deepcopy(tmp_path):
    """A pending output handed out by render() is a private subclass only WHILE it is pending: whatever launches the
    forward turns every handed-out object back into torch.Tensor.  torch.save / load, dlpack, __cuda_array_interface__,
    .numpy() and copy.deepcopy of a pending output all make it run first and see finite pixels."""
    import copy
    import binocular3dgs_amd.rasterizer as R
    from binocular3dgs_amd import synth
    from binocular3dgs_amd.render import PipelineParams, render
    W, H = 160, 120
    model = _model(P=9000, W=W, H=H)
    bg = torch.zeros(3, device="cuda")
    cam, scam, _ = synth.synth_view_set(W, H, device="cuda")[0]
    R._LAZY_FWD = False
    want = render(cam, model, PipelineParams(), bg)["render"].detach().clone()
    R._LAZY_FWD = True

    def pending():
        R._order_hint.clear()
        pkg = render(cam, model, PipelineParams(), bg)
        assert R._pending_fwd and type(pkg["render"]) is R._LazyOut
        return pkg

    pkg = pending()
    f = tmp_path / "img.pt"
    torch.save(pkg["render"].detach(), f)                       # detach() is the first use: launches
    assert all(type(pkg[k]) is torch.Tensor for k in ("render", "radii", "rendered_depth", "rendered_alpha", "visibility_filter"))
    back = torch.load(f)
    assert type(back) is torch.Tensor and torch.equal(back.cuda(), want)
    pkg = pending()
    torch.save(pkg, f)                                          # the whole dict, pending outputs included
    back = torch.load(f, weights_only=False)
    assert type(back["render"]) is torch.Tensor and torch.equal(back["render"].cuda(), want) and not R._pending_fwd
    pkg = pending()
    cap = torch.utils.dlpack.to_dlpack(pkg["render"].detach())
    assert torch.equal(torch.utils.dlpack.from_dlpack(cap), want)
    pkg = pending()
    assert torch.equal(torch.from_dlpack(pkg["rendered_alpha"].detach()), pkg["rendered_alpha"])
    assert not bool(torch.isnan(pkg["rendered_alpha"]).any())
    pkg = pending()
    iface = pkg["radii"].__cuda_array_interface__               # (an int tensor that requires no grad: the protocol allows it)
    assert iface["shape"] == (9000,) and not R._pending_fwd
    pkg = pending()
    arr = pkg["render"].detach().cpu().numpy()
    assert np.array_equal(arr, want.cpu().numpy())
    pkg = pending()
    dc = copy.deepcopy({k: v.detach() for k, v in pkg.items() if k != "viewspace_points"})
    assert type(dc["render"]) is torch.Tensor and torch.equal(dc["render"], want)
    pkg = pending()
    n = pkg["render"].data_ptr()                                # a raw pointer is a use: what it points to is rendered
    assert n != 0 and not R._pending_fwd and not bool(torch.isnan(pkg["render"]).any())


def test_lazy_forward_launched_from_another_stream_orders_the_consumer_behind_it():
    """render() on stream A (forward pending), first use of the output on stream B.  The launch goes to A (where
    the parameters are valid); B must wait for it -- B.wait_stream(A) issued BEFORE the first use saw an empty queue."""
    import binocular3dgs_amd.rasterizer as R
    from binocular3dgs_amd import synth
    from binocular3dgs_amd.render import PipelineParams, render
    W, H = 320, 240
    model = _model(P=60000, W=W, H=H)
    bg = torch.zeros(3, device="cuda")
    cam, _, _ = synth.synth_view_set(W, H, device="cuda")[0]
    R._LAZY_FWD = False
    want = render(cam, model, PipelineParams(), bg)["render"].detach().clone()
    R._LAZY_FWD = True
    torch.cuda.synchronize()
    for _ in range(5):
        R._order_hint.clear()
        pkg = render(cam, model, PipelineParams(), bg)
        assert R._pending_fwd
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())          # correct by the usual rules -- and not enough by itself
        with torch.cuda.stream(side):
            got = pkg["render"].detach().clone()               # first use, on the side stream
        side.synchronize()
        assert not bool(torch.isnan(got).any()) and torch.equal(got, want)


def test_missing_private_engine_hooks_fall_back_to_returned_gradients(monkeypatch):
    """The batched backward leans on torch._C._will_engine_execute_node / _current_graph_task_id / queue_callback, the
    pending forward on torch._C.DisableTorchFunctionSubclass.  A torch without them must give the same gradients through the
    plain path (every node launches for itself and returns its gradients), not an AttributeError inside backward()."""
    import binocular3dgs_amd.rasterizer as R
    from binocular3dgs_amd import synth
    from binocular3dgs_amd.render import PipelineParams, render
    W, H = 208, 144
    model = _model(W=W, H=H)
    bg = torch.zeros(3, device="cuda")
    cam, scam, _ = synth.synth_view_set(W, H, device="cuda")[1]
    gc, gd, ga = synth.synth_pixel_grads(W, H, seed=5, device="cuda")
    R._LAZY_FWD = True

    def run():
        for p in model.parameters():
            p.grad = None
        R._order_hint.clear()
        R._last_raw_ctx.clear()
        a = render(cam, model, PipelineParams(), bg)
        b = render(scam, model, PipelineParams(), bg)
        _loss(a, b, gc, gd, ga).backward()
        return [p.grad.clone() for p in model.parameters()], a["viewspace_points"].grad.clone(), type(a["render"])

    s0 = dict(R._stats)
    ref, ref_m2d, _ = run()
    assert R._stats["launches"] == s0["launches"] + 1           # (hooks present: one batched backward for the pair)
    for name in ("_will_engine_execute_node", "_current_graph_task_id", "DisableTorchFunctionSubclass"):
        monkeypatch.delattr(torch._C, name)
        assert R._refresh_probes() != (True, True)
        try:
            s0 = dict(R._stats)
            got, got_m2d, ty = run()
            if name == "DisableTorchFunctionSubclass":
                assert ty is torch.Tensor and R._stats["lazy_batches"] == s0["lazy_batches"]      # no pending forwards
            else:
                assert R._stats["launches"] == s0["launches"] + 2 and R._stats["deferred"] == s0["deferred"]
            for n, g, r in zip("xyz f_dc f_rest scaling rotation opacity".split(), got, ref):
                assert rel_l2(g.cpu().numpy(), r.cpu().numpy()) < 1e-4, (name, n)
            assert rel_l2(got_m2d.cpu().numpy(), ref_m2d.cpu().numpy()) < 1e-4
        finally:
            monkeypatch.undo()
            assert R._refresh_probes() == (True, True)


def test_two_python_threads_render_two_models_on_two_streams():
    """The surface's shared state sits behind one lock (rasterizer._DropinState): two threads, each with its own model,
    stream and training loop, get the gradients a single-threaded run gives."""
    import threading
    import binocular3dgs_amd.rasterizer as R
    from binocular3dgs_amd import synth
    from binocular3dgs_amd.render import PipelineParams, render
    W, H = 160, 120
    bg = torch.zeros(3, device="cuda")
    pairs = synth.synth_view_set(W, H, device="cuda")
    gc, gd, ga = synth.synth_pixel_grads(W, H, seed=2, device="cuda")
    models = [_model(P=7000, W=W, H=H, seed=11), _model(P=9000, W=W, H=H, seed=12)]
    R._LAZY_FWD = True

    def loop(m, k, out, stream=None, iters=6):
        try:
            with torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream()):
                acc = None
                for it in range(iters):
                    for p in m.parameters():
                        p.grad = None
                    cam, scam, _ = pairs[(it + k) % len(pairs)]
                    a = render(cam, m, PipelineParams(), bg)
                    b = render(scam, m, PipelineParams(), bg)
                    _loss(a, b, gc, gd, ga).backward()
                    g = torch.cat([p.grad.reshape(-1) for p in m.parameters()])
                    acc = g.clone() if acc is None else acc + g
                torch.cuda.current_stream().synchronize()
                out[k] = acc
        except BaseException as exc:      # noqa: BLE001
            out[k] = exc

    ref = {}
    for k, m in enumerate(models):
        loop(m, k, ref)
    tasks_before = set(R._S.tasks)        # (backward() calls of earlier tests that raised inside a node leave their entry)
    got = {}
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for s in streams:
        s.wait_stream(torch.cuda.current_stream())
    th = [threading.Thread(target=loop, args=(m, k, got, streams[k])) for k, m in enumerate(models)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    R._flush_pending()
    for k in range(2):
        assert not isinstance(got[k], BaseException), got[k]
        assert rel_l2(got[k].cpu().numpy(), ref[k].cpu().numpy()) < 1e-4, k
    assert set(R._S.tasks) <= tasks_before            # every backward() cleaned its entry up


def test_a_refused_step_leaves_no_created_gradients_behind():
    """the capacity / key-span verdict of the raw node is raised at the END of backward(), when its launches have
    written gradients from truncated lists: every .grad that backward() created is set back to None and the message says
    what to do about the ones it added into."""
    import binocular3dgs_amd.rasterizer as R
    from binocular3dgs_amd import _lib, synth
    from binocular3dgs_amd.render import PipelineParams, render
    W, H = 208, 144
    model = _model(W=W, H=H)
    bg = torch.zeros(3, device="cuda")
    cam, _, _ = synth.synth_view_set(W, H, device="cuda")[0]
    gc, _, _ = synth.synth_pixel_grads(W, H, seed=5, device="cuda")
    R._LAZY_FWD = False
    render(cam, model, PipelineParams(), bg)["render"].sum().item()
    key = ("raw", 0, model.get_xyz.shape[0], W, H)
    assert key in R._lazy.capacity
    old = R._lazy.capacity[key]
    R._lazy.capacity[key] = 4096                     # far too small: the sync-free forward truncates its lists
    try:
        for p in model.parameters():
            p.grad = None
        pkg = render(cam, model, PipelineParams(), bg)
        with pytest.raises(_lib.B3gsError, match="zero_grad"):
            (pkg["render"] * gc).sum().backward()
        assert all(p.grad is None for p in model.parameters()) and pkg["viewspace_points"].grad is None
        assert R._lazy.capacity[key] > 4096          # grown from the N that came back
    finally:
        R._lazy.capacity[key] = max(old, R._lazy.capacity[key])
