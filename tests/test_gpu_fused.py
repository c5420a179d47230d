"""GPU: the fused-activation / persistent-scratch path (b3gs_forward_raw, b3gs_backward_raw) against
the drop-in path on the same parameters: same images, same parameter gradients."""
import numpy as np
import pytest
import torch

from helpers import rel_l2

pytestmark = pytest.mark.gpu


def _setup(P=6000, W=200, H=144, K=4):
    from binocular3dgs_amd import synth
    model = synth.synth_model(P, seed=7, device="cuda", width=W, height=H, K=K)
    pairs = synth.synth_view_set(W, H, device="cuda")
    bg = torch.tensor([0.1, 0.0, 0.2], device="cuda")
    return model, pairs, bg


@pytest.mark.parametrize("K", [1, 4, 16])
def test_fused_equals_dropin(K):
    from binocular3dgs_amd import synth
    from binocular3dgs_amd.fused import FusedRasterizer
    from binocular3dgs_amd.render import PipelineParams, render
    W, H = 200, 144
    model, pairs, bg = _setup(W=W, H=H, K=K)
    model.active_sh_degree = {1: 0, 4: 1, 16: 3}[K]
    gc, gd, ga = synth.synth_pixel_grads(W, H, seed=3, device="cuda")
    cam = pairs[1][0]
    # drop-in
    for p in model.parameters():
        p.grad = None
    pkg = render(cam, model, PipelineParams(), bg)
    torch.autograd.backward([pkg["render"], pkg["rendered_depth"], pkg["rendered_alpha"]], [gc, gd, ga])
    ref = [p.grad.clone() for p in model.parameters()]
    ref_img = [pkg[k].detach().clone() for k in ("render", "rendered_depth", "rendered_alpha")]
    ref_m2d = pkg["viewspace_points"].grad.clone()
    ref_radii = pkg["radii"].clone()
    # fused, twice into the same grads: accumulation (+=) and the self-cleaning scratch
    fr = FusedRasterizer(model, W, H, num_slots=2)
    for p in model.parameters():
        p.grad = torch.zeros_like(p)
    for rep in range(2):
        out = fr.render(cam, bg, slot=rep)
        torch.autograd.backward([out["render"], out["rendered_depth"], out["rendered_alpha"]], [gc, gd, ga])
    torch.cuda.synchronize()
    assert not fr.overflowed()
    # activations are evaluated by different code (torch vs in-kernel expf): allow borderline radius flips
    assert float((out["radii"] != ref_radii).float().mean()) < 1e-3
    # the stated contract (tests/test_gpu_parity.py): images within 2e-5 (1 + |x|), apart from pixels where the 1-ulp
    # activation difference flips a 1/255-rule decision (at most one minimal contribution, a handful of pixels)
    # (same bound as tests/test_gpu_fullsize_oracle.py: one minimal contribution is 1/255 of colour / alpha and z/255 of
    # the un-normalised depth, z <= 10 in these scenes)
    for (a, b), scale in zip(zip((out["render"], out["rendered_depth"], out["rendered_alpha"]), ref_img), (1.0, 10.0, 1.0)):
        err = ((a - b).abs() / (1 + b.abs())).detach()
        assert int((err > 2e-5).sum()) <= 3 and float(err.max()) <= scale / 255, (float(err.max()), int((err > 2e-5).sum()))
    names = ["xyz", "f_dc", "f_rest", "scaling", "rotation", "opacity"]
    for n, p, r in zip(names, model.parameters(), ref):
        if r.numel() == 0:
            continue
        assert rel_l2(p.grad.cpu().numpy(), 2.0 * r.cpu().numpy()) < 2e-4, n
    assert rel_l2(out["viewspace_points_grad"].cpu().numpy(), ref_m2d.cpu().numpy()) < 2e-4
    for sl in fr.slots:
        assert float(sl.scratch.abs().max()) == 0.0, "scratch must be left clean"


def test_step_with_fused_path_matches_dropin_step():
    from binocular3dgs_amd import synth
    from binocular3dgs_amd.fused import FusedRasterizer
    from binocular3dgs_amd.step import ViewShardedStep
    W, H = 160, 120
    gc, gd, ga = synth.synth_pixel_grads(W, H, seed=1, device="cuda")

    def grad_fn(i, pkg, spkg):
        return [(pkg["render"], gc), (pkg["rendered_depth"], gd), (pkg["rendered_alpha"], ga), (spkg["render"], gc)]

    flats = []
    for use_fused in (False, True):
        model, pairs, bg = _setup(P=4000, W=W, H=H)
        fr = FusedRasterizer(model, W, H, num_slots=2 * len(pairs)) if use_fused else None
        st = ViewShardedStep(model, pairs, bg, fused=fr)
        st.step(pair_grad_fn=grad_fn)
        st.step(pair_grad_fn=grad_fn)      # slab zeroed between steps
        torch.cuda.synchronize()
        flats.append(st.slab.flat.clone())
    assert rel_l2(flats[1].cpu().numpy(), flats[0].cpu().numpy()) < 2e-4


def test_schedules_give_the_same_gradients():
    """The three ways of issuing the 6 views of an iteration -- one launch per stage for all views with the
    binocular pairs sharing a depth sort ("batched"), one stream per view ("streams"), one after the other
    ("serial") -- give the same tile lists, hence the same slab."""
    from binocular3dgs_amd import synth
    from binocular3dgs_amd.fused import FusedRasterizer
    from binocular3dgs_amd.step import ViewShardedStep
    W, H = 160, 120
    gc, gd, ga = synth.synth_pixel_grads(W, H, seed=1, device="cuda")

    def grad_fn(i, pkg, spkg):
        return [(pkg["render"], gc), (pkg["rendered_depth"], gd), (pkg["rendered_alpha"], ga), (spkg["render"], gc)]

    flats = []
    for schedule in ("serial", "streams", "batched"):
        model, pairs, bg = _setup(P=20000, W=W, H=H)
        fr = FusedRasterizer(model, W, H, num_slots=2 * len(pairs), schedule=schedule)
        st = ViewShardedStep(model, pairs, bg, fused=fr)
        for _ in range(3):
            st.step(pair_grad_fn=grad_fn)
        torch.cuda.synchronize()
        flats.append(st.slab.flat.clone())
    assert float(flats[0].abs().max()) > 0
    assert rel_l2(flats[1].cpu().numpy(), flats[0].cpu().numpy()) < 1e-4
    assert rel_l2(flats[2].cpu().numpy(), flats[0].cpu().numpy()) < 1e-4


def test_batched_forward_is_bit_identical_to_per_view_forward():
    """b3gs_forward_raw_batch (multi-view projection, batched radix passes, shared depth order for the
    shifted camera) must produce exactly the per-view results: same N, radii, point lists (hence images)."""
    from binocular3dgs_amd.fused import FusedRasterizer
    W, H = 208, 144
    model, pairs, bg = _setup(P=30000, W=W, H=H)
    views = []
    for i, (cam, scam, _t) in enumerate(pairs):
        views += [(cam, 2 * i), (scam, 2 * i + 1)]
    res = {}
    for schedule in ("serial", "batched"):
        fr = FusedRasterizer(model, W, H, num_slots=len(views), schedule=schedule, seg1_fraction=0.0)   # one binning round
        with torch.no_grad():
            outs = fr.render_batch(views, bg)
        torch.cuda.synchronize()
        n = fr.num_rendered()
        res[schedule] = [(n[k], o["radii"].clone(), o["render"].clone(), o["rendered_depth"].clone(),
                          fr.slots[k].binning[:4 * n[k]].clone()) for k, o in enumerate(outs)]
    for a, b in zip(res["serial"], res["batched"]):
        assert a[0] == b[0] and a[0] > 0
        assert torch.equal(a[1], b[1])
        assert torch.equal(a[4], b[4])          # point_list (val[0] sits at offset 0 of the binning buffer)
        assert torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])


def test_capacity_overflow_is_detected_and_recovered():
    from binocular3dgs_amd.fused import FusedRasterizer
    model, pairs, bg = _setup(P=5000, W=160, H=120)
    fr = FusedRasterizer(model, 160, 120, num_slots=1, binning_capacity=1000)
    with torch.no_grad():
        fr.render(pairs[0][0], bg, slot=0)
    assert fr.overflowed()
    fr.grow()
    with torch.no_grad():
        out = fr.render(pairs[0][0], bg, slot=0)
    assert not fr.overflowed() and float(out["rendered_alpha"].mean()) > 0.1


def test_densification_statistics_match_reference_formula():
    """a11: max_radii2D / xyz_gradient_accum / denom updated inside the fused multi-view accumulate equal
    the reference's PyTorch update (train.py:178-179, scene/gaussian_model.py:409-411) applied to the
    drop-in path's outputs, primary views only."""
    from binocular3dgs_amd import synth
    from binocular3dgs_amd.fused import FusedRasterizer
    from binocular3dgs_amd.render import PipelineParams, render
    from binocular3dgs_amd.step import ViewShardedStep
    W, H = 160, 120
    gc, gd, ga = synth.synth_pixel_grads(W, H, seed=1, device="cuda")

    def grad_fn(i, pkg, spkg):
        return [(pkg["render"], gc), (pkg["rendered_depth"], gd), (pkg["rendered_alpha"], ga), (spkg["render"], gc)]

    model, pairs, bg = _setup(P=8000, W=W, H=H)
    model.init_densification_stats()
    fr = FusedRasterizer(model, W, H, num_slots=2 * len(pairs))
    st = ViewShardedStep(model, pairs, bg, fused=fr)
    for _ in range(2):
        st.step(pair_grad_fn=grad_fn)
    torch.cuda.synchronize()
    got = (model.max_radii2D.clone(), model.xyz_gradient_accum.clone(), model.denom.clone())

    ref, _, _ = _setup(P=8000, W=W, H=H)
    ref.init_densification_stats()
    for _ in range(2):
        for cam, scam, t in pairs:
            for p in ref.parameters():
                p.grad = None
            pkg = render(cam, ref, PipelineParams(), bg)
            spkg = render(scam, ref, PipelineParams(), bg)
            torch.autograd.backward([pkg["render"], pkg["rendered_depth"], pkg["rendered_alpha"], spkg["render"]],
                                    [gc, gd, ga, gc])
            vis = pkg["visibility_filter"]
            ref.update_max_radii(pkg["radii"], vis)
            ref.add_densification_stats(pkg["viewspace_points"], vis)
    # a handful of Gaussians may flip visibility (in-kernel vs torch activations): compare where both agree
    same = got[2].reshape(-1) == ref.denom.reshape(-1)
    assert float(same.float().mean()) > 0.999
    assert float(got[2].max()) == 2.0 * len(pairs)
    assert rel_l2(got[1].reshape(-1)[same].cpu().numpy(), ref.xyz_gradient_accum.reshape(-1)[same].cpu().numpy()) < 2e-4
    assert float((got[0][same] != ref.max_radii2D[same]).float().mean()) < 1e-3


def test_fused_adam_matches_torch_adam_and_opacity_decay():
    from binocular3dgs_amd.step import FusedAdam
    torch.manual_seed(0)
    shapes = [(5000, 3), (5000, 1, 3), (5000, 3, 3), (5000, 3), (5000, 4), (5000, 1)]
    lrs = [1.6e-4, 2.5e-3, 1.25e-4, 5e-3, 1e-3, 0.05]
    a = [torch.nn.Parameter(torch.randn(s, device="cuda")) for s in shapes]
    b = [torch.nn.Parameter(p.detach().clone()) for p in a]
    ref = torch.optim.Adam([{"params": [p], "lr": lr} for p, lr in zip(a, lrs)], eps=1e-15)
    mine = FusedAdam(b, lrs, eps=1e-15)
    for it in range(5):
        for p, q in zip(a, b):
            g = torch.randn_like(p) * (10.0 ** (it - 3))
            p.grad = g.clone()
            q.grad = g.clone()
        ref.step()
        mine.step()
    torch.cuda.synchronize()
    assert int(mine.step_count.item()) == 5
    for p, q in zip(a, b):
        assert float((p - q).abs().max()) < 2e-6 * max(1.0, float(p.abs().max()))
    # opacity decay on the last tensor: o <- logit(sigmoid(o_updated) * 0.995)
    dec = FusedAdam(b, lrs, eps=1e-15, opacity_decay=0.995, opacity_index=5)
    before = b[5].detach().clone()
    for q in b:
        q.grad = torch.zeros_like(q)
    dec.step()
    torch.cuda.synchronize()
    s = torch.sigmoid(before) * 0.995
    assert float((b[5] - torch.log(s / (1 - s))).abs().max()) < 1e-5


def test_pipelined_data_parallel_tail_equals_the_plain_tail():
    """Chain rule / all-reduce / Adam walked over Gaussian ranges (range-major gradient slab, the 8-GPU tail) must
    give the parameters and gradients of the plain accumulate -> all-reduce -> Adam sequence."""
    from binocular3dgs_amd import synth
    from binocular3dgs_amd.fused import FusedRasterizer
    from binocular3dgs_amd.step import FusedAdam, ViewShardedStep
    W, H = 160, 120
    gc, gd, ga = synth.synth_pixel_grads(W, H, seed=1, device="cuda")
    fn = lambda i, pkg, spkg: [(pkg["render"], gc), (pkg["rendered_depth"], gd), (pkg["rendered_alpha"], ga), (spkg["render"], gc)]  # noqa: E731
    res = []
    for K in (0, 3, 7):
        model, pairs, bg = _setup(P=10007, W=W, H=H)
        model.init_densification_stats()
        opt = FusedAdam(model.parameters(), [1.6e-4, 2.5e-3, 1.25e-4, 5e-3, 1e-3, 0.05], eps=1e-15)
        fr = FusedRasterizer(model, W, H, num_slots=2 * len(pairs), want_means2D=False)
        # (the gradients are compared below: the plain tail must leave DENSE rows in the slab)
        st = ViewShardedStep(model, pairs, bg, optimizer=opt, fused=fr, pipeline_ranges=K, sparse_grad_rows=False)
        assert (st.range_slab is not None) == (K > 1)
        for _ in range(3):
            st.step(pair_grad_fn=fn)
        torch.cuda.synchronize()
        grads = st.range_slab.gather() if K > 1 else [p.grad.clone() for p in model.parameters()]
        res.append(([p.detach().clone() for p in model.parameters()], grads, int(opt.step_count.item()), model.denom.clone()))
    for params, grads, steps, denom in res[1:]:
        assert steps == res[0][2] == 3
        assert torch.equal(denom, res[0][3])
        for a, b in zip(grads, res[0][1]):
            assert rel_l2(a.cpu().numpy(), b.cpu().numpy()) < 1e-5
        for a, b in zip(params, res[0][0]):
            assert rel_l2(a.cpu().numpy(), b.cpu().numpy()) < 1e-6


@pytest.mark.parametrize("P,W,H", [(1, 16, 16), (63, 16, 16), (65, 40, 24), (257, 200, 16), (1000, 16, 200), (3000, 33, 17)])
def test_fused_batch_edge_sizes_match_dropin(P, W, H):
    """Tiny Gaussian counts, a single tile (no tile-sort pass: tile_ranges kernel), one-tile-wide and one-tile-high
    images, ragged borders: the batched fused path against the drop-in path, images and gradients."""
    from binocular3dgs_amd import synth
    from binocular3dgs_amd.fused import FusedRasterizer
    from binocular3dgs_amd.render import PipelineParams, render
    model = synth.synth_model(P, seed=P, device="cuda", width=W, height=H)
    with torch.no_grad():
        model._scaling += 1.0
    pairs = synth.synth_view_set(W, H, device="cuda")
    bg = torch.tensor([0.2, 0.1, 0.0], device="cuda")
    gc, gd, ga = synth.synth_pixel_grads(W, H, seed=2, device="cuda")
    views = [pairs[0][0], pairs[0][1], pairs[1][0]]
    ref_imgs, ref_grads = [], None
    for p in model.parameters():
        p.grad = None
    for cam in views:
        pkg = render(cam, model, PipelineParams(), bg)
        torch.autograd.backward([pkg["render"], pkg["rendered_depth"], pkg["rendered_alpha"]], [gc, gd, ga])
        ref_imgs.append([pkg[k].detach().clone() for k in ("render", "rendered_depth", "rendered_alpha")])
    ref_grads = [p.grad.clone() for p in model.parameters()]
    for p in model.parameters():
        p.grad = None
    fr = FusedRasterizer(model, W, H, num_slots=3)
    outs = fr.render_batch([(c, k) for k, c in enumerate(views)], bg)
    flat_o, flat_g = [], []
    for o in outs:
        flat_o += [o["render"], o["rendered_depth"], o["rendered_alpha"]]
        flat_g += [gc, gd, ga]
    torch.autograd.backward(flat_o, flat_g)
    torch.cuda.synchronize()
    assert not fr.overflowed()
    for o, r in zip(outs, ref_imgs):
        for k, t in zip(("render", "rendered_depth", "rendered_alpha"), r):
            assert float((o[k] - t).abs().max()) <= 2e-5 * (1 + float(t.abs().max())), k
    for p, r in zip(model.parameters(), ref_grads):
        if float(r.abs().max()) == 0:
            assert float(p.grad.abs().max()) == 0
        else:
            assert rel_l2(p.grad.cpu().numpy(), r.cpu().numpy()) < 2e-4


def test_views_of_different_tile_sort_depth_in_one_forward_batch():
    """b3gs_forward_raw_batch with W x H that need a different number of tile-sort passes (the launcher then bins the
    views one by one): same results as two single-view forwards."""
    import ctypes as C
    import math
    from binocular3dgs_amd import _lib, synth
    from binocular3dgs_amd.camera import Camera, look_at_orbit
    P = 4000
    model = synth.synth_model(P, seed=4, device="cuda", width=64, height=48)
    L = _lib.lib()
    rp = _lib.B3gsRawParams()
    rp.xyz, rp.features_dc, rp.features_rest = model._xyz.data_ptr(), model._features_dc.data_ptr(), model._features_rest.data_ptr()
    rp.scaling, rp.rotation, rp.opacity = model._scaling.data_ptr(), model._rotation.data_ptr(), model._opacity.data_ptr()
    bg = torch.zeros(3, device="cuda")
    sizes = [(16, 16), (320, 240)]            # 1 tile (0 passes) and 300 tiles (2 passes)
    K = model._features_dc.shape[1] + model._features_rest.shape[1]

    def make(W, H):
        R, T = look_at_orbit(2.0)
        fovx = math.radians(60.0)
        cam = Camera(R, T, fovx, synth.fovy_from(fovx, W, H), W, H, device="cuda")
        sc = _lib.B3gsScene(P, 1, K, W, H, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), 1.0, 0, 0, bg.data_ptr(), None, None,
                            None, None, None, None, None, cam.world_view_transform.data_ptr(),
                            cam.full_proj_transform.data_ptr(), cam.camera_center.data_ptr())
        u8 = dict(dtype=torch.uint8, device="cuda")
        bufs = dict(geom=torch.empty(L.b3gs_geometry_bytes(P), **u8), binning=torch.empty(L.b3gs_binning_bytes(P, 200000), **u8),
                    img=torch.empty(L.b3gs_image_bytes(W, H), **u8), color=torch.empty(3, H, W, device="cuda"),
                    depth=torch.empty(1, H, W, device="cuda"), alpha=torch.empty(1, H, W, device="cuda"),
                    radii=torch.zeros(P, dtype=torch.int32, device="cuda"), n=torch.zeros(1, dtype=torch.int32, device="cuda"))
        return cam, sc, bufs

    res = {}
    for mode in ("single", "batch"):
        items = [make(W, H) for W, H in sizes]
        arr = (_lib.B3gsForwardView * len(items))()
        for k, (_cam, sc, b) in enumerate(items):
            arr[k].view = C.pointer(sc)
            arr[k].geometry, arr[k].binning, arr[k].image = b["geom"].data_ptr(), b["binning"].data_ptr(), b["img"].data_ptr()
            arr[k].binning_capacity = 200000
            arr[k].out_color, arr[k].out_depth, arr[k].out_alpha = b["color"].data_ptr(), b["depth"].data_ptr(), b["alpha"].data_ptr()
            arr[k].radii, arr[k].device_num_rendered, arr[k].depth_order_from = b["radii"].data_ptr(), b["n"].data_ptr(), -1
        stream = torch.cuda.current_stream().cuda_stream
        if mode == "batch":
            _lib.check(L.b3gs_forward_raw_batch(len(items), arr, C.byref(rp), 3, stream), "batch")
        else:
            for k in range(len(items)):
                one = (_lib.B3gsForwardView * 1)(arr[k])
                _lib.check(L.b3gs_forward_raw_batch(1, one, C.byref(rp), 3, stream), "single")
        torch.cuda.synchronize()
        res[mode] = [(int(b["n"].item()), b["color"].clone(), b["radii"].clone()) for _c, _s, b in items]
    for a, b in zip(res["single"], res["batch"]):
        assert a[0] == b[0] and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    assert res["batch"][1][0] > 0


def test_single_view_raw_entry_points_match_the_batched_path():
    """b3gs_forward_raw + b3gs_backward_raw (one view per call, both backward phases in one call, gradients
    ACCUMULATED into the given buffers) against the batched entry points the FusedRasterizer uses."""
    import ctypes as C
    from binocular3dgs_amd import _lib, synth
    from binocular3dgs_amd.fused import FusedRasterizer
    W, H = 176, 128
    model, pairs, bg = _setup(P=7000, W=W, H=H)
    gc, gd, ga = synth.synth_pixel_grads(W, H, seed=4, device="cuda")
    cam = pairs[2][0]
    fr = FusedRasterizer(model, W, H, num_slots=1)
    out = fr.render(cam, bg, slot=0)
    torch.autograd.backward([out["render"], out["rendered_depth"], out["rendered_alpha"]], [gc, gd, ga])
    torch.cuda.synchronize()
    ref = [p.grad.clone() for p in model.parameters()]
    ref_img = out["render"].detach().clone()
    ref_m2d = out["viewspace_points_grad"].clone()
    # the same through the single-view C entry points, into fresh buffers
    L = _lib.lib()
    P = model.get_xyz.shape[0]
    sc = fr._scene(dict(cam=cam, bg=bg, scaling_modifier=1.0, debug=False))
    rp = fr._bind_params()
    u8 = dict(dtype=torch.uint8, device="cuda")
    geom = torch.empty(L.b3gs_geometry_bytes(P), **u8)
    binning = torch.empty(L.b3gs_binning_bytes(P, 400000), **u8)
    img = torch.empty(L.b3gs_image_bytes(W, H), **u8)
    color, depth, alpha = torch.empty(3, H, W, device="cuda"), torch.empty(1, H, W, device="cuda"), torch.empty(1, H, W, device="cuda")
    radii = torch.zeros(P, dtype=torch.int32, device="cuda")
    n = torch.zeros(1, dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    _lib.check(L.b3gs_forward_raw(C.byref(sc), C.byref(rp), geom.data_ptr(), binning.data_ptr(), 400000, img.data_ptr(),
                                  color.data_ptr(), depth.data_ptr(), alpha.data_ptr(), radii.data_ptr(), n.data_ptr(), 3,
                                  stream), "b3gs_forward_raw")
    assert torch.equal(color, ref_img) and 0 < int(n.item()) <= 400000
    grads = [torch.zeros_like(p) for p in model.parameters()]              # += semantics: two calls give twice the gradient
    gr = _lib.B3gsRawGrads()
    for name, g in zip(("xyz", "features_dc", "features_rest", "scaling", "rotation", "opacity"), grads):
        setattr(gr, name, g.data_ptr())
    scratch = torch.zeros(L.b3gs_backward_scratch_floats(P), device="cuda")
    m2d = torch.empty(P, 3, device="cuda")
    for _ in range(2):                                                       # twice: the scratch must come back clean
        _lib.check(L.b3gs_backward_raw(C.byref(sc), C.byref(rp), radii.data_ptr(), geom.data_ptr(), binning.data_ptr(),
                                       img.data_ptr(), gc.data_ptr(), gd.data_ptr(), ga.data_ptr(), scratch.data_ptr(),
                                       C.byref(gr), m2d.data_ptr(), 3, stream), "b3gs_backward_raw")
    torch.cuda.synchronize()
    assert float(scratch.abs().max()) == 0.0
    assert rel_l2(m2d.cpu().numpy(), ref_m2d.cpu().numpy()) < 1e-5
    for g, r in zip(grads, ref):
        assert rel_l2((g / 2).cpu().numpy(), r.cpu().numpy()) < 2e-5


def test_adam_reference_decay_order_empty_segment_and_error_text():
    """(i) decay_first: the reference decays the opacity logits BEFORE optimizer.step() (train.py:171-173 vs :196-198):
    p <- logit(sigmoid(p) f) - delta(g); (ii) a zero-width tensor (features_rest at SH degree 0: [P,0,3], data_ptr 0)
    is an empty segment, not an error; (iii) a failing call leaves a message in b3gs_last_error."""
    import ctypes as C
    from binocular3dgs_amd import _lib
    from binocular3dgs_amd.step import FusedAdam
    torch.manual_seed(1)
    shapes = [(4000, 3), (4000, 1, 3), (4000, 0, 3), (4000, 3), (4000, 4), (4000, 1)]
    lrs = [1.6e-4, 2.5e-3, 1.25e-4, 5e-3, 1e-3, 0.05]
    a = [torch.nn.Parameter(torch.randn(s, device="cuda")) for s in shapes]
    b = [torch.nn.Parameter(p.detach().clone()) for p in a]
    ref = torch.optim.Adam([{"params": [p], "lr": lr} for p, lr in zip(a, lrs) if p.numel()], eps=1e-15)
    mine = FusedAdam(b, lrs, eps=1e-15, opacity_decay=0.995, opacity_index=5, decay_first=True)
    for it in range(3):
        for p, q in zip(a, b):
            g = torch.randn_like(p) * 0.01
            p.grad, q.grad = g.clone(), g.clone()
        with torch.no_grad():      # train.py:171-173: opacity_decay() on .data, then optimizer.step()
            o = torch.sigmoid(a[5]) * 0.995
            a[5].copy_(torch.log(o / (1 - o)))
        ref.step()
        mine.step()
    torch.cuda.synchronize()
    for p, q in zip(a, b):
        if p.numel():
            assert float((p - q).abs().max()) < 5e-6 * max(1.0, float(p.abs().max()))
    seg = (_lib.B3gsAdamSegment * 1)()
    seg[0].count, seg[0].lr = 5, 1e-3          # non-empty segment with NULL pointers
    rc = _lib.lib().b3gs_adam_step(1, seg, mine.step_count.data_ptr(), 0.9, 0.999, 1e-15, 0.0, -1, 0, 1, None, None, None)
    assert rc == -1 and b"b3gs_adam_step" in _lib.lib().b3gs_last_error()
    io = _lib.B3gsDensifyIO()
    io.P, io.M = 4, 0
    assert _lib.lib().b3gs_densify_classify(C.byref(io), None, None) == -1
    assert b"densify" in _lib.lib().b3gs_last_error()


def test_sharded_adam_one_rank_equals_fused_adam_and_survives_densification():
    """ShardedAdam with one rank = the one-launch Adam on flat buffers: same parameters as FusedAdam step for step;
    the parameters live in one flat buffer (views), and densify_and_prune re-flattens them with the moments carried."""
    from binocular3dgs_amd import synth
    from binocular3dgs_amd.fused import FusedRasterizer
    from binocular3dgs_amd.step import FusedAdam, ShardedAdam, ViewShardedStep
    W, H = 160, 120
    gc, gd, ga = synth.synth_pixel_grads(W, H, seed=1, device="cuda")
    fn = lambda i, pkg, spkg: [(pkg["render"], gc), (pkg["rendered_depth"], gd), (pkg["rendered_alpha"], ga), (spkg["render"], gc)]  # noqa: E731
    lrs = [1.6e-4, 2.5e-3, 1.25e-4, 5e-3, 1e-3, 0.05]
    res = []
    for cls in (FusedAdam, ShardedAdam):
        model, pairs, bg = _setup(P=9001, W=W, H=H)
        model.init_densification_stats()
        opt = cls(model.parameters(), lrs, eps=1e-15, opacity_decay=0.995, opacity_index=5)
        fr = FusedRasterizer(model, W, H, num_slots=2 * len(pairs), want_means2D=False)
        st = ViewShardedStep(model, pairs, bg, optimizer=opt, fused=fr)
        for _ in range(3):
            st.step(pair_grad_fn=fn)
        noise = torch.randn(2, 9001, 3, generator=torch.Generator().manual_seed(3)).cuda()
        thr = float((model.xyz_gradient_accum / model.denom.clamp(min=1)).quantile(0.7))
        newP = st.densify_and_prune(thr, 0.005, 5.0, noise=noise)
        assert newP == model.get_xyz.shape[0] and newP != 9001
        for _ in range(2):
            st.step(pair_grad_fn=fn)
        torch.cuda.synchronize()
        if cls is ShardedAdam:
            lo = opt.pflat.data_ptr()
            assert all(lo <= p.data_ptr() < lo + 4 * opt.padded_numel for p in model.parameters())
            assert st.slab.flat.numel() == opt.padded_numel
        res.append([p.detach().clone() for p in model.parameters()] + [model.denom.clone()])
    for x, y in zip(*res):
        assert x.shape == y.shape and rel_l2(x.cpu().numpy(), y.cpu().numpy()) < 1e-6


def test_step_detects_binning_overflow_and_grows():
    """The step object never trains on truncated tile lists unknowingly: the largest device-side N since the last
    check is compared with the capacity at check_capacity() (every overflow_check_every steps / at densification)."""
    from binocular3dgs_amd import _lib, synth
    from binocular3dgs_amd.fused import FusedRasterizer
    from binocular3dgs_amd.step import ViewShardedStep
    W, H = 160, 120
    model, pairs, bg = _setup(P=6000, W=W, H=H)
    gc, gd, ga = synth.synth_pixel_grads(W, H, seed=1, device="cuda")
    fn = lambda i, pkg, spkg: [(pkg["render"], gc), (pkg["rendered_depth"], gd), (pkg["rendered_alpha"], ga), (spkg["render"], gc)]  # noqa: E731
    fr = FusedRasterizer(model, W, H, num_slots=2 * len(pairs), binning_capacity=1000)
    st = ViewShardedStep(model, pairs, bg, fused=fr, overflow_check_every=2)
    need = max(fr.num_rendered())
    assert fr.capacity >= need > 1000            # fit_capacity() at construction sized the buffers from the actual N
    st.step(pair_grad_fn=fn)
    st.step(pair_grad_fn=fn)                     # second step runs the periodic check: nothing to report
    fr.capacity = 1000                           # simulate a scene that outgrew its buffers
    for i in range(len(fr.slots)):
        fr.slots[i] = fr._new_slot(fr.slots[i].stream, i)
    st.step(pair_grad_fn=fn)
    with pytest.raises(_lib.B3gsError, match="B3GS_ERR_CAPACITY"):
        st.step(pair_grad_fn=fn)
    assert fr.capacity >= need
    st.step(pair_grad_fn=fn)
    st.check_capacity()


@pytest.mark.parametrize("opt_kind", ["fused", "sharded"])
def test_overflowing_steps_are_dropped_on_the_device(opt_kind):
    """A step rendered from truncated tile lists cannot corrupt the model: the binning
    kernels raise a sticky device flag when N exceeds the capacity; from that step on the Adam launch (parameters,
    moments, step counter) and the densification statistics skip their update on the device, with no host round trip.
    When check_capacity() finally looks, the model is bit for bit in the state of the last complete step; after the
    buffers have grown, repeating the steps gives what a run that never overflowed gives."""
    from binocular3dgs_amd import _lib, synth
    from binocular3dgs_amd.fused import FusedRasterizer
    from binocular3dgs_amd.step import FusedAdam, ShardedAdam, ViewShardedStep
    W, H = 160, 120
    gc, gd, ga = synth.synth_pixel_grads(W, H, seed=1, device="cuda")
    fn = lambda i, pkg, spkg: [(pkg["render"], gc), (pkg["rendered_depth"], gd), (pkg["rendered_alpha"], ga), (spkg["render"], gc)]  # noqa: E731
    lrs = [1.6e-4, 2.5e-3, 1.25e-4, 5e-3, 1e-3, 0.05]

    def state(model, opt):
        return [p.detach().clone() for p in model.parameters()] + [opt.exp_avg.clone(), opt.exp_avg_sq.clone(),
                opt.step_count.clone(), model.denom.clone(), model.xyz_gradient_accum.clone(), model.max_radii2D.clone()]

    def build():
        model, pairs, bg = _setup(P=6000, W=W, H=H)
        model.init_densification_stats()
        cls = FusedAdam if opt_kind == "fused" else ShardedAdam
        opt = cls(model.parameters(), lrs, eps=1e-15, opacity_decay=0.995, opacity_index=5, decay_first=True)
        fr = FusedRasterizer(model, W, H, num_slots=2 * len(pairs), want_means2D=False)
        st = ViewShardedStep(model, pairs, bg, optimizer=opt, fused=fr, overflow_check_every=0)
        return model, opt, fr, st

    # reference: four complete steps
    model, opt, fr, st = build()
    for _ in range(4):
        st.step(pair_grad_fn=fn)
    want = state(model, opt)

    model, opt, fr, st = build()
    assert opt.skip_flag is fr.overflow_flag
    for _ in range(2):
        st.step(pair_grad_fn=fn)
    good = state(model, opt)
    fr.capacity = 1000                           # the scene outgrows its buffers
    for i in range(len(fr.slots)):
        fr.slots[i] = fr._new_slot(fr.slots[i].stream, i)
    for _ in range(3):                           # nobody has looked yet: three steps on truncated lists
        st.step(pair_grad_fn=fn)
    torch.cuda.synchronize()
    assert int(fr.overflow_flag.item()) == 1
    for a, b in zip(state(model, opt), good):    # ... and none of them touched the model
        assert torch.equal(a, b)
    with pytest.raises(_lib.B3gsError, match="B3GS_ERR_CAPACITY"):
        st.check_capacity()
    assert int(fr.overflow_flag.item()) == 0 and fr.capacity > 1000
    for _ in range(2):                           # repeat from the last complete step
        st.step(pair_grad_fn=fn)
    st.check_capacity()
    for a, b in zip(state(model, opt), want):
        if a.dtype == torch.int32:
            assert torch.equal(a, b)
        else:
            assert rel_l2(a.cpu().numpy(), b.cpu().numpy()) < 1e-5   # fp32 atomics in the gradients


@pytest.mark.parametrize("scene", ["dense", "thin"])
def test_two_round_binning_equals_one_round(scene):
    """Termination-aware binning (B3gsForwardView.seg1_fraction): bin the nearest fraction of the depth order (plus, into the
    tiles predicted open -- the ones the previous forward of the slot left unterminated -- everything behind it), blend, bin
    the rest only into the tiles that are not finished, blend those again.  Every pixel walks the same list prefix in the
    same order as with one-round binning, so images, final_T and n_contrib are BIT-identical and the gradients equal up
    to the order of the fp32 atomics -- for any fraction, including ones so small that almost every tile needs the second
    round, and for a scene whose tiles never saturate ("thin": low opacities)."""
    from binocular3dgs_amd import synth
    from binocular3dgs_amd.debug import state_views
    from binocular3dgs_amd.fused import FusedRasterizer
    W, H, P = 208, 144, 30000
    model, pairs, bg = _setup(P=P, W=W, H=H)
    with torch.no_grad():
        model._scaling += 0.6
        if scene == "thin":
            model._opacity -= 4.0
    gc, gd, ga = synth.synth_pixel_grads(W, H, seed=4, device="cuda")
    views = []
    for i, (cam, scam, _t) in enumerate(pairs):
        views += [(cam, 2 * i, True), (scam, 2 * i + 1, False)]

    # the same cameras on other slots: the open-tile prediction a slot carries over then comes from a DIFFERENT view
    rot = [(views[(k + 2) % len(views)][0], views[k][1], views[(k + 2) % len(views)][2]) for k in range(len(views))]

    def one(fr, vs):
        model.init_densification_stats()
        for p in model.parameters():
            p.grad = torch.zeros_like(p)
        outs = fr.render_batch(vs, bg)
        o, g = [], []
        for x, v in zip(outs, vs):
            o.append(x["render"]); g.append(gc)
            if v[2]:
                o += [x["rendered_depth"], x["rendered_alpha"]]; g += [gd, ga]
        torch.autograd.backward(o, g)
        torch.cuda.synchronize()
        assert not fr.overflowed()
        # the chain rule's scan reads -- and resets -- only the scratch rows of the Gaussians the blend forward marked
        # (GeomView::staged: list positions below a tile's deepest used one, both binning rounds): a flush into an unmarked
        # row would be left behind here (and its gradient missing below)
        for v in vs:
            assert float(fr.slots[v[1]].scratch.abs().max()) == 0.0, "a scratch row outside the marked set received a gradient"
        imgs = [[x[k].detach().clone() for k in ("render", "rendered_depth", "rendered_alpha")] for x in outs]
        st = [state_views(P, W, H, fr.capacity, fr.slots[v[1]].geom, fr.slots[v[1]].binning, fr.slots[v[1]].img) for v in vs]
        aux = [(v["final_T"].clone(), v["n_contrib"].clone(), int(v["counts"][0]), int(v["counts"][2])) for v in st]
        n = fr.num_rendered()
        return imgs, aux, [p.grad.clone() for p in model.parameters()], model.denom.clone(), [n[v[1]] for v in vs]

    def run(frac):
        fr = FusedRasterizer(model, W, H, num_slots=len(views), seg1_fraction=frac)
        fr.fit_capacity(views, bg)
        for sl in fr.slots:       # (fit_capacity settles the prediction with a forward of its own: start from none)
            sl.img.zero_()
        # first forward: no prediction yet, every unterminated tile is repaired by the second round; second forward:
        # those tiles are predicted open and get their complete list in round 1; third: the prediction of another view
        return one(fr, views), one(fr, views), one(fr, rot)

    ref = run(0.0)
    assert all(a[3] == 0 for r in ref for a in r[1])
    for frac in (0.5, 0.125, 0.01):
        got = run(frac)
        second = [sum(a[3] for a in r[1]) for r in got]
        if frac == 0.01 or scene == "thin":
            assert second[0] > 0, "the second round must have had work"
            assert second[1] == 0, "same view again: every open tile was predicted, nothing left to repair"
        for (imgs, aux, grads, denom, n), (ref_imgs, ref_aux, ref_grads, ref_denom, ref_n) in zip(got, ref):
            for (a, b) in zip(imgs, ref_imgs):
                for x, y in zip(a, b):
                    assert torch.equal(x, y)
            for a, b, nn, rn in zip(aux, ref_aux, n, ref_n):
                assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
                assert nn == a[2] + a[3] <= rn          # never more instances than one-round binning
            assert torch.equal(denom, ref_denom)
            for x, y in zip(grads, ref_grads):
                if y.numel():
                    assert rel_l2(x.cpu().numpy(), y.cpu().numpy()) < 5e-5   # fp32 atomics: the chunking of the lists differs


def test_auto_rule_goes_back_to_one_round_when_the_prediction_keeps_missing():
    """The binning kernels count two-round forwards and the ones whose open-tile prediction missed (image-header words 11 /
    10).  When more than `two_round_max_miss_rate` of the forwards since the last check needed the repair round -- here the
    prediction is wiped before every forward, as a slot whose camera changes completely would see it -- check_overflow()
    sends "auto" back to one round; images never depended on any of it."""
    from binocular3dgs_amd.fused import FusedRasterizer
    W, H, P = 208, 144, 30000
    model, pairs, bg = _setup(P=P, W=W, H=H)
    with torch.no_grad():
        model._scaling += 0.6
    views = [(pairs[0][0], 0, True), (pairs[0][1], 1, False)]
    fr = FusedRasterizer(model, W, H, num_slots=2)
    fr.two_round_min_instances = 0                  # make the rule pick two rounds on this small scene
    fr.fit_capacity(views, bg)
    assert 0.0 < fr.seg1_fraction <= 0.125
    with torch.no_grad():
        want = [o["render"].clone() for o in fr.render_batch(views, bg)]
    fr.repair_rate(reset=True)
    for _ in range(10):
        for sl in fr.slots:                         # forget the prediction: this forward's second round has work
            sl.img[64 * 4:].zero_()
        with torch.no_grad():
            outs = fr.render_batch(views, bg)
        for a, b in zip(outs, want):
            assert torch.equal(a["render"], b)
    missed, total = fr.repair_rate()
    assert total == 10 and missed == 10
    assert fr.check_overflow() == 0 and fr.seg1_fraction == 0.0 and fr.two_round_disabled == (10, 10)
    with torch.no_grad():
        outs = fr.render_batch(views, bg)           # one round from now on
    for a, b in zip(outs, want):
        assert torch.equal(a["render"], b)
    # a settled prediction keeps two rounds
    fr2 = FusedRasterizer(model, W, H, num_slots=2)
    fr2.two_round_min_instances = 0
    fr2.fit_capacity(views, bg)
    fr2.repair_rate(reset=True)
    with torch.no_grad():
        for _ in range(10):
            fr2.render_batch(views, bg)
    assert fr2.check_overflow() == 0 and fr2.seg1_fraction > 0.0 and fr2.two_round_disabled is None


def _lists(fr, P, W, H, slots):
    from binocular3dgs_amd.debug import state_views
    out = []
    for s in slots:
        sl = fr.slots[s]
        v = state_views(P, W, H, fr.capacity, sl.geom, sl.binning, sl.img)
        n = int(v["counts"][0])
        out.append((n, v["point_list"][:n].clone(), v["ranges"].clone()))
    return out


@pytest.mark.parametrize("P,W,H", [(30000, 208, 144), (3, 16, 16), (4097, 64, 48)])
def test_three_pass_depth_sort_equals_four_pass(P, W, H):
    """B3gsForwardView::depth_key_bits = 27: three 9-bit radix passes over (key - bits(0.2f)) give the permutation of the
    four 8-bit passes over all 32 key bits -- tile lists, ranges and images bit for bit, culled Gaussians (behind the
    near plane) included, ties (equal depths) in index order."""
    from binocular3dgs_amd.fused import FusedRasterizer
    model, pairs, bg = _setup(P=P, W=W, H=H)
    with torch.no_grad():
        model._xyz[: P // 7, 2] -= 12.0                    # a good part behind the camera: culled keys sink to the end
        model._xyz[P // 2: P // 2 + P // 9] = model._xyz[P // 2: P // 2 + 1]   # identical positions: equal keys
    views = [(pairs[0][0], 0, True), (pairs[0][1], 1, False), (pairs[1][0], 2, True)]
    res = []
    for bits in (27, 0):
        fr = FusedRasterizer(model, W, H, num_slots=3, seg1_fraction=0.0)
        fr.depth_key_bits = bits
        with torch.no_grad():
            outs = fr.render_batch(views, bg, _span_checked=True)    # (the span verdict is read right below)
        torch.cuda.synchronize()
        assert int(fr.overflow_flag.item()) == 0
        res.append(([o["render"].clone() for o in outs], _lists(fr, P, W, H, (0, 1, 2))))
    for a, b in zip(res[0][0], res[1][0]):
        assert torch.equal(a, b)
    for (na, la, ra), (nb, lb, rb) in zip(res[0][1], res[1][1]):
        assert na == nb and torch.equal(la, lb) and torch.equal(ra, rb)


def test_depth_key_outside_the_27_bit_span_is_detected_and_falls_back():
    """A visible Gaussian farther than z ~ 13107 does not fit the 27-bit key span: the projection raises bit 1 of the
    overflow word (the step is dropped on the device like a capacity overflow), check_overflow() switches the rasterizer
    to the full 32-bit sort, and the repeated render equals a rasterizer that sorted 32 bits from the start."""
    from binocular3dgs_amd.fused import FusedRasterizer
    W, H, P = 160, 120, 5000
    model, pairs, bg = _setup(P=P, W=W, H=H)
    cam = pairs[0][0]
    with torch.no_grad():
        # push a few Gaussians 20000 units out along the viewing direction (camera space +z), scaled up to stay visible
        R = cam.world_view_transform[:3, :3]               # row-vector convention: x_cam = x_world @ R + t
        fwd = R[:, 2]
        model._xyz[:8] = cam.camera_center + 20000.0 * fwd + 50.0 * torch.randn(8, 3, device="cuda")
        model._scaling[:8] += 9.0
    views = [(cam, 0, True)]
    ref = FusedRasterizer(model, W, H, num_slots=1, seg1_fraction=0.0)
    ref.depth_key_bits = 0
    with torch.no_grad():
        want = ref.render_batch(views, bg)[0]["render"].clone()
    fr = FusedRasterizer(model, W, H, num_slots=1, seg1_fraction=0.0)
    assert fr.depth_key_bits == 27
    with torch.no_grad():
        # a render nobody will differentiate (evaluation) has no reader of the span verdict: it sorts all 32 bits and is
        # right without any check
        got_eval = fr.render_batch(views, bg)[0]["render"].clone()
    assert torch.equal(got_eval, want) and int(fr.overflow_flag.item()) == 0
    fr.render_batch(views, bg)            # a training render: three passes on the checked span
    torch.cuda.synchronize()
    assert int(fr.overflow_flag.item()) & 2
    assert fr.check_overflow() and fr.depth_key_bits == 0 and int(fr.overflow_flag.item()) == 0
    with torch.no_grad():
        got = fr.render_batch(views, bg)[0]["render"]
    assert torch.equal(got, want) and not fr.check_overflow()


@pytest.mark.parametrize("P,W,H", [(2, 16, 16), (65, 16, 16), (257, 40, 24), (3000, 33, 17), (5000, 272, 16), (4097, 40, 24),
                                   (8192, 33, 17), (12289, 272, 16), (20000, 16, 16),
                                   pytest.param(8_500_000, 96, 64, id="8.5M_two_scan_tiles_per_chunk")])
def test_two_round_binning_edge_sizes(P, W, H):
    """Two-round binning at the edges.  Segment 1 is a whole number of 4096-Gaussian tiles of the depth order, so fewer than
    4097 Gaussians are always binned in one round (the first four sizes: the request must simply be honoured as one round);
    above: ONE Gaussian behind segment 1, a last tile that is exactly full / holds one Gaussian, a single image tile (no
    tile-sort pass: the ranges of segment 2 come from tile_ranges with the device-side offset), ragged borders.
    Bit-identical images to one round."""
    from binocular3dgs_amd import synth
    from binocular3dgs_amd.fused import FusedRasterizer
    model = synth.synth_model(P, seed=P, device="cuda", width=W, height=H, requires_grad=False)
    with torch.no_grad():
        model._scaling += 1.0
        model._opacity -= 2.0          # tiles stay open: the second round has work
    pairs = synth.synth_view_set(W, H, device="cuda")
    bg = torch.tensor([0.2, 0.1, 0.0], device="cuda")
    views = [(pairs[0][0], 0), (pairs[0][1], 1), (pairs[1][0], 2)]
    res = []
    for frac in (0.0, 0.3, 0.01):
        fr = FusedRasterizer(model, W, H, num_slots=3, seg1_fraction=frac)
        with torch.no_grad():
            outs = fr.render_batch(views, bg)
        torch.cuda.synchronize()
        assert not fr.overflowed()
        res.append([[o[k].clone() for k in ("render", "rendered_depth", "rendered_alpha", "radii")] for o in outs])
    for other in res[1:]:
        for a, b in zip(other, res[0]):
            for x, y in zip(a, b):
                assert torch.equal(x, y)


def test_two_round_binning_overflow_is_detected():
    """Segment 2 sits behind segment 1 in the same arrays: N1 + N2 above the capacity truncates the lists (no write past
    the buffers) and shows up in the device-side N / high-water mark like a one-round overflow."""
    from binocular3dgs_amd.fused import FusedRasterizer
    model, pairs, bg = _setup(P=5000, W=160, H=120)
    with torch.no_grad():
        model._opacity -= 4.0
    full = FusedRasterizer(model, 160, 120, num_slots=1, seg1_fraction=0.0)
    with torch.no_grad():
        full.render(pairs[0][0], bg, slot=0)
    n_full = full.num_rendered()[0]
    fr = FusedRasterizer(model, 160, 120, num_slots=1, binning_capacity=max(n_full // 2, 64), seg1_fraction=0.2)
    with torch.no_grad():
        out = fr.render(pairs[0][0], bg, slot=0)
    torch.cuda.synchronize()
    assert torch.isfinite(out["render"]).all()
    assert fr.num_rendered()[0] > fr.capacity and fr.overflowed() and fr.check_overflow() > 0
    with torch.no_grad():
        out2 = fr.render(pairs[0][0], bg, slot=0)
        ref = full.render(pairs[0][0], bg, slot=0)
    torch.cuda.synchronize()
    assert not fr.overflowed() and torch.equal(out2["render"], ref["render"])


@pytest.mark.parametrize("P", [4096, 5003])
def test_adam_row_mask_skips_untouched_rows_bit_exact(P):
    """b3gs_adam_step(row_mask): the gradient of a Gaussian whose bit is clear is 0 WITHOUT being read -- the masked-out
    rows hold NaN here -- and the result is bit-identical to the dense step on gradients zeroed by the same mask.
    P = 4096: every tensor 16-byte aligned (float4 kernel); P = 5003: scalar kernel, ragged last bitmap word."""
    from binocular3dgs_amd.step import FusedAdam
    torch.manual_seed(P)
    shapes = [(P, 3), (P, 1, 3), (P, 15, 3), (P, 3), (P, 4), (P, 1)]
    lrs = [1.6e-4, 2.5e-3, 1.25e-4, 5e-3, 1e-3, 0.05]
    a = [torch.nn.Parameter(torch.randn(s, device="cuda")) for s in shapes]
    b = [torch.nn.Parameter(p.detach().clone()) for p in a]
    dense = FusedAdam(a, lrs, eps=1e-15, opacity_decay=0.995, opacity_index=5, decay_first=True)
    sparse = FusedAdam(b, lrs, eps=1e-15, opacity_decay=0.995, opacity_index=5, decay_first=True)
    for it in range(4):
        live = torch.rand(P, device="cuda") < (0.2 if it else 0.0)           # first step: nothing touched at all
        bits = torch.zeros(((P + 63) // 64) * 64, dtype=torch.int64, device="cuda")
        bits[:P] = live.to(torch.int64)
        words = (bits.view(-1, 64) << torch.arange(64, device="cuda")).sum(1)   # wraps into the sign bit: fine
        for p, q in zip(a, b):
            g = torch.randn_like(p) * 0.01
            sel = live.view(-1, *([1] * (p.dim() - 1)))
            p.grad = torch.where(sel, g, torch.zeros_like(g))
            q.grad = torch.where(sel, g, torch.full_like(g, float("nan")))
        dense.step()
        sparse.step(row_mask=words)
    torch.cuda.synchronize()
    for p, q in zip(a, b):
        assert torch.equal(p, q)
    assert torch.equal(dense.exp_avg, sparse.exp_avg) and torch.equal(dense.exp_avg_sq, sparse.exp_avg_sq)
    assert not any(bool(torch.isnan(q).any()) for q in b)


@pytest.mark.parametrize("P", [9001, 9024])
def test_sparse_gradient_rows_equal_dense_rows(P):
    """Sparse-row gradient slab (B3gsRawGrads.touched_rows, single rank): the chain-rule pass stores only the rows of
    Gaussians that received a gradient and a bitmap of them; Adam reads the bitmap.  (i) the stored rows equal the dense
    pass, every skipped row is a zero row of the dense pass and keeps its sentinel; (ii) training with it gives the dense
    run's parameters (same atomics-order tolerance as two dense runs)."""
    from binocular3dgs_amd import synth
    from binocular3dgs_amd.fused import FusedRasterizer
    from binocular3dgs_amd.step import FusedAdam, ShardedAdam, ViewShardedStep
    W, H = 160, 120
    gc, gd, ga = synth.synth_pixel_grads(W, H, seed=1, device="cuda")
    fn = lambda i, pkg, spkg: [(pkg["render"], gc), (pkg["rendered_depth"], gd), (pkg["rendered_alpha"], ga), (spkg["render"], gc)]  # noqa: E731
    lrs = [1.6e-4, 2.5e-3, 1.25e-4, 5e-3, 1e-3, 0.05]
    # (i) one backward, no optimiser: dense vs sparse rows
    rows = {}
    for sparse in (False, True):
        model, pairs, bg = _setup(P=P, W=W, H=H)
        fr = FusedRasterizer(model, W, H, num_slots=2 * len(pairs), want_means2D=False)
        st = ViewShardedStep(model, pairs, bg, fused=fr, sparse_grad_rows=False)
        st.slab.flat.fill_(-7.0)                                     # sentinel: a skipped row keeps it
        st.slab.rebind()
        words = torch.zeros((P + 63) // 64, dtype=torch.int64, device="cuda")
        fr.begin_deferred()
        pkgs = st._render_views()
        outs, grads = [], []
        for k in range(0, len(pkgs), 2):
            for o, g in fn(k // 2, pkgs[k], pkgs[k + 1]):
                outs.append(o)
                grads.append(g)
        torch.autograd.backward(outs, grads)
        fr.finish_deferred(overwrite=True, touched_rows=words if sparse else None)
        torch.cuda.synchronize()
        rows[sparse] = ([p.grad.detach().clone().reshape(P, -1) for p in model.parameters()], words)
    bit = ((rows[True][1][torch.arange(P, device="cuda") >> 6] >> (torch.arange(P, device="cuda") & 63)) & 1).bool()
    assert 0 < int(bit.sum()) < P
    for d, s in zip(*[r[0] for r in rows.values()]):
        assert rel_l2(s[bit].cpu().numpy(), d[bit].cpu().numpy()) < 1e-5
        assert float(d[~bit].abs().max()) == 0.0 and bool((s[~bit] == -7.0).all())
    # (ii) the training step
    res = []
    for cls, sparse in ((FusedAdam, False), (FusedAdam, True), (ShardedAdam, True)):
        model, pairs, bg = _setup(P=P, W=W, H=H)
        model.init_densification_stats()
        opt = cls(model.parameters(), lrs, eps=1e-15, opacity_decay=0.995, opacity_index=5)
        fr = FusedRasterizer(model, W, H, num_slots=2 * len(pairs), want_means2D=False)
        st = ViewShardedStep(model, pairs, bg, optimizer=opt, fused=fr, sparse_grad_rows=sparse)
        assert (st._sparse_rows() is not None) == sparse
        for _ in range(4):
            st.step(pair_grad_fn=fn)
        torch.cuda.synchronize()
        res.append([p.detach().clone() for p in model.parameters()] + [model.denom.clone(), model.xyz_gradient_accum.clone()])
    for other in res[1:]:
        for x, y in zip(res[0], other):
            assert rel_l2(x.cpu().numpy(), y.cpu().numpy()) < 1e-6


@pytest.mark.parametrize("K", [4, 16])
def test_overwrite_chain_rule_of_a_batch_equals_accumulation_per_view(K):
    """`accumulate_chain_kernel` deals the (Gaussian, view) pairs of a batch to its waves as dense chunks and sums them in
    an LDS row per Gaussian; SH coefficients beyond degree 1 (K = 16) go to memory with atomic adds after the row's own
    thread zeroed them (overwrite mode).  One overwrite pass over six views must equal six single-view passes ADDED into
    zeroed gradients, on top of sentinel-filled gradients, for every parameter tensor."""
    from binocular3dgs_amd import synth
    from binocular3dgs_amd.fused import FusedRasterizer
    W, H, P = 160, 120, 20000
    model, pairs, bg = _setup(P=P, W=W, H=H, K=K)
    model.active_sh_degree = {4: 1, 16: 3}[K]
    gc, gd, ga = synth.synth_pixel_grads(W, H, seed=1, device="cuda")
    views = []
    for i, (cam, scam, _t) in enumerate(pairs):
        views += [(cam, 2 * i, True), (scam, 2 * i + 1, False)]

    def backward(outs):
        o, g = [], []
        for x, v in zip(outs, views):
            o.append(x["render"]); g.append(gc)
            if v[2]:
                o += [x["rendered_depth"], x["rendered_alpha"]]; g += [gd, ga]
        torch.autograd.backward(o, g)

    fr = FusedRasterizer(model, W, H, num_slots=len(views))
    for p in model.parameters():
        p.grad = torch.full_like(p, -7.0)                      # overwritten, not added to
    fr.begin_deferred()
    backward(fr.render_batch(views, bg))
    fr.finish_deferred(overwrite=True)
    torch.cuda.synchronize()
    got = [p.grad.clone() for p in model.parameters()]
    for p in model.parameters():
        p.grad = torch.zeros_like(p)
    for k, v in enumerate(views):                              # one view at a time, added
        out = fr.render(v[0], bg, slot=v[1])
        torch.autograd.backward([out["render"]] + ([out["rendered_depth"], out["rendered_alpha"]] if v[2] else []),
                                [gc] + ([gd, ga] if v[2] else []))
    torch.cuda.synchronize()
    for n, a, p in zip(["xyz", "f_dc", "f_rest", "scaling", "rotation", "opacity"], got, model.parameters()):
        if p.numel():
            assert float(p.grad.abs().max()) > 0
            assert rel_l2(a.cpu().numpy(), p.grad.cpu().numpy()) < 1e-5, n


def test_sparse_row_and_row_mask_argument_errors():
    """(i) sparse-row accumulate over a Gaussian range must start at a multiple of 64 (the bitmap is written word-wise);
    (ii) a masked Adam step rejects negative row geometry; both leave a message in b3gs_last_error."""
    import ctypes as C
    from binocular3dgs_amd import _lib
    from binocular3dgs_amd.fused import FusedRasterizer
    from binocular3dgs_amd.step import FusedAdam
    W, H, P = 96, 64, 1000
    model, pairs, bg = _setup(P=P, W=W, H=H)
    fr = FusedRasterizer(model, W, H, num_slots=2)
    for p in model.parameters():
        p.grad = torch.zeros_like(p)
    fr.begin_deferred()
    outs = fr.render_batch([(pairs[0][0], 0, True), (pairs[0][1], 1, False)], bg)
    torch.autograd.backward([o["render"] for o in outs], [torch.ones_like(o["render"]) for o in outs])
    pend = fr.take_deferred()
    words = torch.zeros((P + 63) // 64, dtype=torch.int64, device="cuda")
    gr = fr._bind_grads()
    with pytest.raises(_lib.B3gsError, match="multiple of 64"):
        fr._accumulate(pend, True, gr, first=32, count=P - 32, touched_rows=words)
    fr._accumulate(pend, True, gr, first=64, count=P - 64, touched_rows=words)     # aligned start: fine
    fr._accumulate(pend, True, gr, first=0, count=64, touched_rows=words)
    torch.cuda.synchronize()
    assert int(words[1:].abs().sum()) != 0
    opt = FusedAdam(list(model.parameters()), [1e-3] * 6)
    seg = (_lib.B3gsAdamSegment * 1)()
    p0 = opt.params[0]
    seg[0].param, seg[0].grad = p0.data_ptr(), p0.grad.data_ptr()
    seg[0].exp_avg, seg[0].exp_avg_sq = opt.exp_avg.data_ptr(), opt.exp_avg_sq.data_ptr()
    seg[0].count, seg[0].lr, seg[0].row_len, seg[0].first_row = p0.numel(), 1e-3, -3, 0
    rc = _lib.lib().b3gs_adam_step(1, seg, opt.step_count.data_ptr(), 0.9, 0.999, 1e-15, 0.0, -1, 0, 1, words.data_ptr(), None, None)
    assert rc == -1 and b"row_len" in _lib.lib().b3gs_last_error()


def test_forward_reuses_the_backward_tile_order_without_changing_the_images():
    """The batched forward follows the longest-tile-first order the previous backward of the same batch shape left in
    view 0's image buffer (signature-checked; placement only).  Same images bit for bit before and after an order exists,
    and after the batch shape changes (the old order must be ignored, not misread)."""
    from binocular3dgs_amd import synth
    from binocular3dgs_amd.fused import FusedRasterizer
    W, H, P = 208, 144, 20000
    model, pairs, bg = _setup(P=P, W=W, H=H)
    gc, _, _ = synth.synth_pixel_grads(W, H, seed=4, device="cuda")
    views = []
    for i, (cam, scam, _t) in enumerate(pairs):
        views += [(cam, 2 * i, True), (scam, 2 * i + 1, False)]
    fr = FusedRasterizer(model, W, H, num_slots=len(views))

    def fwd(vs, backward):
        for p in model.parameters():
            p.grad = torch.zeros_like(p)
        outs = fr.render_batch(vs, bg)
        imgs = [o["render"].detach().clone() for o in outs]
        if backward:
            torch.autograd.backward([o["render"] for o in outs], [gc] * len(outs))
        torch.cuda.synchronize()
        return imgs

    first = fwd(views, True)            # no order yet -> default placement; the backward leaves one for 6 views
    second = fwd(views, True)           # follows it
    for a, b in zip(first, second):
        assert torch.equal(a, b)
    sub = fwd(views[:4], True)          # 4 views: another shape, the 6-view order must be ignored
    for a, b in zip(first[:4], sub):
        assert torch.equal(a, b)
    again = fwd(views[:4], False)       # follows the 4-view order
    for a, b in zip(sub, again):
        assert torch.equal(a, b)
    third = fwd(views, False)           # back to 6 views with a 4-view order in the buffer
    for a, b in zip(first, third):
        assert torch.equal(a, b)


def test_a_pair_that_claims_a_shared_depth_order_it_does_not_have_is_reported():
    """render_batch() sorts ONE depth order for a camera and the camera that names it as `same_depth_as` (Camera.shifted()).
    For an adjacent pair the projection compares the two depth keys of every Gaussian (ABI 8): a camera that claims a parent
    whose z row it does not have raises bit 3 of the overflow word -- check_overflow() says so instead of letting the
    device drop every following step in silence; honest pairs leave the word zero."""
    import copy
    from binocular3dgs_amd import _lib
    from binocular3dgs_amd.fused import FusedRasterizer
    W, H = 200, 144
    model, pairs, bg = _setup(W=W, H=H)
    (c0, s0, _), (c1, _, _) = pairs[:2]
    fr = FusedRasterizer(model, W, H, num_slots=2)
    with torch.no_grad():
        fr.render_batch([(c0, 0), (s0, 1)], bg, _span_checked=True)
    assert fr.check_overflow() == 0
    liar = copy.copy(c1)
    liar.same_depth_as = c0
    with torch.no_grad():
        fr.render_batch([(c0, 0), (liar, 1)], bg, _span_checked=True)
    with pytest.raises(_lib.B3gsError, match="same_depth_as"):
        fr.check_overflow()
    assert int(fr.overflow_flag.item()) == 0          # cleared: the next honest pair is fine
    with torch.no_grad():
        fr.render_batch([(c0, 0), (s0, 1)], bg, _span_checked=True)
    assert fr.check_overflow() == 0
