"""GPU: the reference's loss functions and per-iteration model methods ONE BY ONE behind their own signatures (ABI 9,
binocular3dgs_amd/loss_utils.py, graphics_utils.py, optim.py, GaussianModel methods) against
  (i)  golden G10 = the reference's own functions called one by one (value + EVERY input gradient),
  (ii) golden G6 = the composed block of train.py:130-148 written with the reference's statements,
  (iii) the PyTorch statements of binocular3dgs_amd/loss.py on ragged and full-size images,
  (iv) golden G8 / G10 for opacity_decay, add_densification_stats, three optimizer.step() calls."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
NAMES = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")


@pytest.fixture(scope="module")
def g():
    return np.load(os.path.join(GOLD, "functions.npz"))


def _t(a, grad=True):
    return torch.from_numpy(np.array(a)).cuda().requires_grad_(grad)


def _close(got, ref, rtol=2e-5, what="", flips=0.0):
    got = got.detach().cpu().numpy() if torch.is_tensor(got) else np.asarray(got)
    ref = np.asarray(ref)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    scale = max(float(np.abs(ref).max()), 1e-30)
    d = np.abs(got - ref)
    if flips:       # |.| terms: a residual one ulp from zero may take the other sign in a handful of isolated pixels
        assert float((d > rtol * scale).mean()) <= flips, (what, float((d > rtol * scale).mean()))
        assert np.linalg.norm(d) <= 2e-3 * np.linalg.norm(ref), what
    else:
        assert float(d.max()) <= rtol * scale, (what, float(d.max()), scale)


def test_l1_loss_golden(g):
    from binocular3dgs_amd.loss_utils import l1_loss
    x, y = _t(g["l1_x"]), _t(g["l1_y"])
    v = l1_loss(x, y)
    (1.7 * v).backward()
    _close(v, g["l1_val"]), _close(x.grad, g["l1_gx"]), _close(y.grad, g["l1_gy"])
    x, y, m = _t(g["l1m_x"]), _t(g["l1m_y"]), _t(g["l1m_m"])
    v = l1_loss(x, y, mask=m)
    (0.6 * v).backward()
    _close(v, g["l1m_val"]), _close(x.grad, g["l1m_gx"]), _close(y.grad, g["l1m_gy"]), _close(m.grad, g["l1m_gm"])
    # a mask of the images' own shape, and shapes that only broadcast
    x, y = _t(g["l1m_x"]), _t(g["l1m_y"])
    mf = torch.from_numpy(g["l1m_m"]).cuda().expand(-1, 3, -1, -1).contiguous().requires_grad_(True)
    v = l1_loss(x, y, mask=mf)
    (0.6 * v).backward()
    _close(v, g["l1m_val"]), _close(x.grad, g["l1m_gx"]), _close(mf.grad.sum(1, keepdim=True), g["l1m_gm"], 1e-4)
    x = _t(g["l1m_x"])
    y1 = torch.from_numpy(g["l1m_y"][:1]).cuda().requires_grad_(True)
    ref_x = torch.from_numpy(g["l1m_x"]).requires_grad_(True)
    ref_y = torch.from_numpy(g["l1m_y"][:1]).requires_grad_(True)
    torch.abs(ref_x - ref_y).mean().backward()
    l1_loss(x, y1).backward()
    _close(x.grad, ref_x.grad.numpy()), _close(y1.grad, ref_y.grad.numpy())


def test_ssim_golden(g):
    from binocular3dgs_amd.loss_utils import ssim
    a, b = _t(g["ss_a"]), _t(g["ss_b"])
    v = ssim(a, b)
    (1.3 * v).backward()
    _close(v, g["ss_val"]), _close(a.grad, g["ss_ga"], 1e-4), _close(b.grad, g["ss_gb"], 1e-4)
    a, b = _t(g["ssb_a"]), _t(g["ssb_b"])
    v = ssim(a, b, size_average=False)
    assert v.shape == (2,)
    (v * torch.from_numpy(g["ssb_w"]).cuda()).sum().backward()
    _close(v, g["ssb_val"]), _close(a.grad, g["ssb_ga"], 1e-4), _close(b.grad, g["ssb_gb"], 1e-4)
    # value only (metrics.py / training_report call it under no_grad), and only one input differentiated
    with torch.no_grad():
        _close(ssim(a, b), np.array(g["ssb_val"].mean(), dtype=np.float32))
    a2 = _t(g["ss_a"])
    (1.3 * ssim(a2, torch.from_numpy(g["ss_b"]).cuda())).backward()
    _close(a2.grad, g["ss_ga"], 1e-4)
    with pytest.raises(Exception):
        ssim(a2, a2, window_size=7)


def test_smooth_loss_golden(g):
    from binocular3dgs_amd.loss_utils import SmoothLoss
    sm = SmoothLoss()
    assert list(sm.parameters()) == []
    d, im = _t(g["sm_d"]), _t(g["sm_im"])
    v = sm.forward(disparity=d, image=im)
    (2.2 * v).backward()
    _close(v, g["sm_val"]), _close(d.grad, g["sm_gd"], 1e-4, flips=1e-3), _close(im.grad, g["sm_gim"], 1e-4, flips=1e-3)


def test_inverse_warp_golden(g):
    from binocular3dgs_amd.graphics_utils import inverse_warp_images
    H, W = g["iw_d"].shape[-2:]
    rows, cols = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    im, d = _t(g["iw_im"]), _t(g["iw_d"])
    o = inverse_warp_images(im, d, rows.cuda(), cols.cuda())
    (o * torch.from_numpy(g["iw_up"]).cuda()).sum().backward()
    _close(o, g["iw_out"]), _close(im.grad, g["iw_gim"]), _close(d.grad, g["iw_gd"])
    d2 = _t(g["iw_d"])
    o = inverse_warp_images(torch.ones(2, 1, H, W).cuda(), d2, rows.cuda(), cols.cuda())
    (o * torch.from_numpy(g["iw_up"][:, :1]).cuda()).sum().backward()
    _close(o, g["iwm_out"])
    assert float(d2.grad.abs().max()) <= 1e-7
    # a second backward through a retained graph (the scatter buffer of the first one is gone)
    im, d = _t(g["iw_im"]), _t(g["iw_d"])
    o = inverse_warp_images(im, d)
    s = (o * torch.from_numpy(g["iw_up"]).cuda()).sum()
    s.backward(retain_graph=True)
    s.backward()
    _close(im.grad, 2 * g["iw_gim"]), _close(d.grad, 2 * g["iw_gd"])


def test_the_loss_block_through_the_schedule_driver_golden_g6():
    """The loss block of one binocular iteration as binocular3dgs_amd/schedule.py drives it (golden G11 pins that sequence
    against the reference's loop), every function its HIP drop-in, on the tensors of golden G6: the values and the pixel
    gradients the reference's own Python produced for train.py:130-148."""
    import types
    from binocular3dgs_amd.schedule import IterationSchedule, default_ops
    g = np.load(os.path.join(GOLD, "loss_block.npz"))
    first = {"render": _t(g["image"]), "rendered_depth": _t(g["depth"]), "rendered_alpha": _t(g["alpha"]),
             "radii": torch.zeros(4, dtype=torch.int32).cuda(), "visibility_filter": torch.zeros(4, dtype=torch.bool).cuda(),
             "viewspace_points": torch.zeros(4, 3).cuda()}
    second = {"render": _t(g["shifted"])}
    fx, shift, lam = [float(x) for x in g["scalars"]]
    hip, seen = default_ops(), {"warp": [], "l1": []}
    renders = iter([first, second])

    def warp(*a):
        seen["warp"].append(hip.inverse_warp_images(*a))
        return seen["warp"][-1]

    def l1(*a, **k):
        seen["l1"].append(hip.l1_loss(*a, **k))
        return seen["l1"][-1]

    ops = types.SimpleNamespace(render=lambda *a: next(renders), l1_loss=l1, ssim=hip.ssim, SmoothLoss=hip.SmoothLoss,
                                inverse_warp_images=warp)
    cam = types.SimpleNamespace(image_height=first["render"].shape[-2], image_width=first["render"].shape[-1],
                                original_image=torch.from_numpy(g["gt"]).cuda(),
                                gt_alpha_mask=torch.from_numpy(g["gt_alpha_mask"]).cuda(), get_focal=lambda: (fx, fx))
    quiet = lambda *a, **k: None    # noqa: E731
    model = types.SimpleNamespace(update_learning_rate=quiet, oneupSHdegree=quiet, opacity_decay=quiet,
                                  add_densification_stats=quiet, max_radii2D=torch.zeros(4).cuda(),
                                  optimizer=types.SimpleNamespace(step=quiet, zero_grad=quiet))
    scene = types.SimpleNamespace(getTrainCameras=lambda: [cam], getShiftedCamera=lambda c, t: c, cameras_extent=1.0)
    sched = IterationSchedule(model, scene, None, torch.zeros(3).cuda(), ops=ops, iterations=10, shift_cam_start=0,
                              lambda_dssim=lam, opacity_decay_factor=None)
    total = sched.run_iteration(1, 0, shift)
    _close(seen["warp"][0], g["warped"]), _close(seen["warp"][1], g["shift_mask"])
    np.testing.assert_allclose(float(total), float(g["total"]), rtol=3e-5)
    np.testing.assert_allclose(float(seen["l1"][1]), float(g["Ll1"]), rtol=3e-5)
    for ten, key in ((first["render"], "g_image"), (first["rendered_depth"], "g_depth"), (first["rendered_alpha"], "g_alpha"),
                     (second["render"], "g_shifted")):
        ref = g[key]
        assert np.abs(ten.grad.cpu().numpy() - ref).max() <= 2e-4 * np.abs(ref).max() + 2e-9, key


@pytest.mark.parametrize("W,H", [(203, 157), (800, 600), (33, 17), (64, 64)])
def test_functions_match_the_pytorch_statements_on_other_sizes(W, H):
    from binocular3dgs_amd import loss as ref
    from binocular3dgs_amd.graphics_utils import inverse_warp_images
    from binocular3dgs_amd.loss_utils import SmoothLoss, l1_loss, ssim
    gen = torch.Generator().manual_seed(W * 7 + H)
    r = lambda *s: torch.rand(*s, generator=gen).cuda()  # noqa: E731
    gt = r(1, 3, H, W)
    base = dict(img=(gt + 0.2 * (r(1, 3, H, W) - 0.5)).clamp(0, 1), disp=20 * r(1, 1, H, W) - 10, mask=r(1, 1, H, W))
    res = []
    for fns in ((ref.l1_loss, ref.ssim, ref.smooth_loss, ref.inverse_warp_images),
                (l1_loss, ssim, SmoothLoss().forward, inverse_warp_images)):
        l1, ss, sm, iw = fns
        t = {k: v.clone().requires_grad_(True) for k, v in base.items()}
        w = iw(t["img"], t["disp"])
        parts = [l1(w, gt, mask=t["mask"]), ss(t["img"], gt), sm(t["disp"] * t["mask"], gt), l1(t["img"], gt)]
        total = parts[0] + 0.3 * parts[1] + 0.05 * parts[2] + 0.8 * parts[3]
        total.backward()
        res.append(([float(p) for p in parts], {k: v.grad.clone() for k, v in t.items()}))
    for a, b in zip(res[0][0], res[1][0]):
        assert b == pytest.approx(a, rel=3e-5)
    for k in base:
        _close(res[1][1][k], res[0][1][k].cpu().numpy(), 2e-4, what=k, flips=3e-5)
    # deterministic scalars: the same bits on a second call (per-workgroup partial sums folded in index order)
    v1 = ssim(base["img"], gt), l1_loss(base["img"], gt), SmoothLoss()(base["disp"], gt)
    v2 = ssim(base["img"], gt), l1_loss(base["img"], gt), SmoothLoss()(base["disp"], gt)
    assert all(torch.equal(a, b) for a, b in zip(v1, v2))


def test_host_tensors_are_refused():
    from binocular3dgs_amd._lib import B3gsError
    from binocular3dgs_amd.graphics_utils import inverse_warp_images
    from binocular3dgs_amd.loss_utils import SmoothLoss, l1_loss, ssim
    a = torch.rand(1, 3, 8, 8)
    for fn in (lambda: l1_loss(a, a), lambda: ssim(a, a), lambda: SmoothLoss()(a[:, :1], a),
               lambda: inverse_warp_images(a, a[:, :1])):
        with pytest.raises(B3gsError):
            fn()


def test_model_methods_golden(g):
    from test_golden_functions import model_from_golden
    m = model_from_golden(g, device="cuda")
    # three optimizer.step() calls with the reference's gradients and schedule (train.py:83,196-198)
    groups = {gr["name"]: gr for gr in m.optimizer.param_groups}
    for k in range(3):
        m.update_learning_rate(100 * (k + 1))
        for n in NAMES:
            groups[n]["params"][0].grad = torch.from_numpy(g[f"opt_g{k}_{n}"]).cuda()
        m.optimizer.step()
        m.optimizer.zero_grad(set_to_none=True)
        assert all(groups[n]["params"][0].grad is None for n in NAMES)
    for n in NAMES:
        p = groups[n]["params"][0]
        st = m.optimizer.state[p]
        _close(p, g[f"opt_p3_{n}"], 2e-6, what=n), _close(st["exp_avg"], g[f"opt_m3_{n}"], 2e-6), _close(st["exp_avg_sq"], g[f"opt_v3_{n}"], 3e-5)
        assert float(st["step"]) == float(g["opt_step3"])
    # a parameter without a gradient is left alone (torch's rule), the others step
    before = m._rotation.detach().clone()
    for n in NAMES:
        groups[n]["params"][0].grad = None if n == "rotation" else torch.zeros_like(groups[n]["params"][0])
    m.optimizer.step()
    assert torch.equal(m._rotation, before) and float(m.optimizer.state[m._rotation]["step"]) == 3.0
    assert float(m.optimizer.state[m._xyz]["step"]) == 4.0
    # state_dict round trip through torch's own optimiser: the layout is torch.optim.Adam's
    sd = m.optimizer.state_dict()
    ref = torch.optim.Adam([{"params": [p], "lr": 0.0, "name": n} for n, p in ((n, groups[n]["params"][0]) for n in NAMES)],
                           lr=0.0, eps=1e-15)
    ref.load_state_dict(sd)
    assert torch.equal(ref.state[m._xyz]["exp_avg"], m.optimizer.state[m._xyz]["exp_avg"])
    # add_densification_stats, the reference's signature
    leaf = torch.zeros(203, 3, device="cuda", requires_grad=True)
    for k in range(2):
        leaf.grad = torch.from_numpy(g[f"ads_grad{k}"]).cuda()
        m.add_densification_stats(leaf, torch.from_numpy(g[f"ads_filter{k}"]).cuda())
    _close(m.xyz_gradient_accum, g["ads_accum"], 1e-6), _close(m.denom, g["ads_denom"], 0.0)
    # opacity_decay (G8)
    d = np.load(os.path.join(GOLD, "densify.npz"))
    m._opacity = torch.nn.Parameter(torch.from_numpy(d["decay_in"]).cuda())
    v0 = m._opacity._version
    m.opacity_decay(factor=0.995)
    _close(m._opacity, d["decay_out"], 2e-6)
    assert m._opacity._version > v0


def test_densify_and_prune_method_keeps_the_optimizer_in_step(g):
    """GaussianModel.densify_and_prune(max_grad, min_opacity, extent, max_screen_size) -- the reference's signature -- on
    the torch-compatible optimiser: golden G8 case a (rows, moments), and the optimiser keeps stepping the NEW tensors."""
    from binocular3dgs_amd.gaussian_model import GaussianModel
    d = np.load(os.path.join(GOLD, "densify.npz"))
    t = lambda k: torch.from_numpy(d[k]).cuda()  # noqa: E731
    m = GaussianModel.from_tensors(t("a_in_xyz"), t("a_in_f_dc"), t("a_in_f_rest"), t("a_in_scaling"), t("a_in_rotation"),
                                   t("a_in_opacity"), sh_degree=1, device="cuda")
    import types
    m.spatial_lr_scale = 1.0
    m.training_setup(types.SimpleNamespace(percent_dense=0.01, position_lr_init=1.6e-4, position_lr_final=1.6e-6,
                                           position_lr_delay_mult=0.01, position_lr_max_steps=30000, feature_lr=2.5e-3,
                                           opacity_lr=0.05, scaling_lr=5e-3, rotation_lr=1e-3))
    groups = {gr["name"]: gr for gr in m.optimizer.param_groups}
    for n in NAMES:
        p = groups[n]["params"][0]
        m.optimizer.state[p] = {"step": torch.tensor(2.0), "exp_avg": t(f"a_in_{n}_exp_avg").clone(),
                                "exp_avg_sq": t(f"a_in_{n}_exp_avg_sq").clone()}
    m.xyz_gradient_accum, m.denom, m.max_radii2D = t("a_accum"), t("a_denom"), t("a_max_radii2D")
    thr, min_op, extent, pdense, size_thr = [float(x) for x in d["a_scalars"]]
    # the split noise of the fixture is indexed (child k, j-th selected Gaussian); densify.densify_and_prune wants [2, P, 3]
    # addressed by the ORIGINAL index: rebuild it from the selection the classification makes
    P = 400
    grads = (m.xyz_gradient_accum / m.denom).nan_to_num(0.0).squeeze()
    sel = (grads >= thr) & (m.get_scaling.max(dim=1).values > pdense * extent)
    idx = torch.nonzero(sel).squeeze(1)
    n_sel = idx.numel()
    noise = torch.zeros(2, P, 3, device="cuda")
    nz = t("a_noise")
    noise[0, idx], noise[1, idx] = nz[:n_sel], nz[n_sel:2 * n_sel]
    m.split_noise = noise                   # (the attribute replicas / lock-step runs use to share the split offsets: used once)
    m.densify_and_prune(thr, min_op, extent, None)
    assert m.split_noise is None
    newP = d["a_out_xyz"].shape[0]
    assert m.get_xyz.shape[0] == newP
    groups = {gr["name"]: gr for gr in m.optimizer.param_groups}
    attr = dict(xyz="_xyz", f_dc="_features_dc", f_rest="_features_rest", opacity="_opacity", scaling="_scaling", rotation="_rotation")
    for n in NAMES:
        p = groups[n]["params"][0]
        assert p is getattr(m, attr[n])
        _close(p, d[f"a_out_{n}"], 1e-6, what=n)
        _close(m.optimizer.state[p]["exp_avg"], d[f"a_out_{n}_exp_avg"], 1e-6), _close(m.optimizer.state[p]["exp_avg_sq"], d[f"a_out_{n}_exp_avg_sq"], 1e-6)
    for p in m.parameters():
        p.grad = torch.full_like(p, 1e-3)
    before = m._xyz.detach().clone()
    m.optimizer.step()
    assert float((m._xyz - before).abs().max()) > 0 and float(m.optimizer.state[m._xyz]["step"]) == 3.0
