"""GPU: edge cases of the reference-signature functions of round 5 (loss_utils / graphics_utils / optim / GaussianModel
methods) against the PyTorch statements of binocular3dgs_amd/loss.py (which golden G10 pins to the reference's own functions):
images smaller than the SSIM window, one-pixel and one-row images, batches, views that are not contiguous, only some inputs
differentiated, empty models, zero-size inputs, gradients of other dtypes / layouts handed to Adam."""
import copy
import types

import pytest
import torch

pytestmark = pytest.mark.gpu
_OPT = types.SimpleNamespace(percent_dense=0.01, position_lr_init=1.6e-4, position_lr_final=1.6e-6, position_lr_delay_mult=0.01,
                             position_lr_max_steps=30000, feature_lr=2.5e-3, opacity_lr=0.05, scaling_lr=5e-3, rotation_lr=1e-3)


def _leaf(t, dev):
    return t.detach().clone().to(dev).requires_grad_(True)


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-30))


@pytest.mark.parametrize("B,C,H,W", [(1, 3, 1, 1), (1, 3, 5, 7), (1, 1, 10, 10), (2, 3, 11, 3), (3, 2, 1, 40), (1, 3, 12, 13)])
def test_ssim_and_l1_on_images_smaller_than_the_window(B, C, H, W):
    """The 11x11 window with zero padding reaches outside a small image on every side (utils/loss_utils.py:47-52)."""
    from binocular3dgs_amd import loss as ref
    from binocular3dgs_amd.loss_utils import l1_loss, ssim
    gen = torch.Generator().manual_seed(B * 1000 + C * 100 + H * 10 + W)
    a0, b0 = torch.rand(B, C, H, W, generator=gen), torch.rand(B, C, H, W, generator=gen)
    for size_average in (True, False):
        out = []
        for fn, dev in ((ref.ssim, "cpu"), (ssim, "cuda")):
            a, b = _leaf(a0, dev), _leaf(b0, dev)
            v = fn(a, b, size_average=size_average)
            (v * torch.arange(1, v.numel() + 1, device=dev).reshape(v.shape).float()).sum().backward()
            out.append((v, a.grad, b.grad))
        assert out[0][0].shape == out[1][0].shape
        for x, y in zip(out[1], out[0]):
            assert _rel(x, y) <= 2e-4
    out = []
    for fn, dev in ((ref.l1_loss, "cpu"), (l1_loss, "cuda")):
        a, b = _leaf(a0, dev), _leaf(b0, dev)
        v = fn(a, b)
        v.backward()
        out.append((v, a.grad, b.grad))
    for x, y in zip(out[1], out[0]):
        assert _rel(x, y) <= 2e-5
    # the 3-D form train.py uses ([C,H,W]) gives the mean over the same numbers as a batch of one
    if B == 1:
        assert _rel(ssim(a0[0].cuda(), b0[0].cuda()), ref.ssim(a0[0], b0[0])) <= 3e-5


@pytest.mark.parametrize("B,C,H,W", [(1, 3, 3, 3), (2, 1, 3, 9), (3, 3, 17, 4)])
def test_smooth_loss_and_warp_on_small_batches(B, C, H, W):
    from binocular3dgs_amd import loss as ref
    from binocular3dgs_amd.graphics_utils import inverse_warp_images
    from binocular3dgs_amd.loss_utils import SmoothLoss
    gen = torch.Generator().manual_seed(B * 1000 + C * 100 + H * 10 + W)
    im0 = torch.rand(B, C, H, W, generator=gen)
    d0 = (torch.rand(B, 1, H, W, generator=gen) - 0.5) * 2.5 * W      # most columns leave the image on one side or the other
    up = torch.rand(B, C, H, W, generator=gen)
    out = []
    for sm, iw, dev in ((ref.smooth_loss, ref.inverse_warp_images, "cpu"), (SmoothLoss(), inverse_warp_images, "cuda")):
        im, d = _leaf(im0, dev), _leaf(d0, dev)
        w = iw(im, d)
        s = sm(d * 0.1, im)
        ((w * up.to(dev)).sum() + 3.0 * s).backward()
        out.append((w, s, im.grad, d.grad))
    for k, (x, y) in enumerate(zip(out[1], out[0])):
        assert _rel(x, y) <= 2e-4, k
    # an integer disparity: x1 - d = 1, d - x0 = 0 -- the neighbour's weight is an exact zero, its column may be outside
    d_int = torch.full((B, 1, H, W), -1.0)
    assert torch.equal(inverse_warp_images(im0.cuda(), d_int.cuda()).cpu(), ref.inverse_warp_images(im0, d_int))
    with pytest.raises((RuntimeError, ValueError)):
        SmoothLoss()(d0[..., :2, :].cuda(), im0[..., :2, :].cuda())     # the reference's 3x3 convolution raises too


def test_views_broadcasts_and_partial_gradients():
    """Inputs that are views (a channel slice, a transposed image, an expanded mask) are taken as their values; inputs that do
    not ask for a gradient get none; a scalar upstream gradient that is not contiguous float32 is accepted."""
    from binocular3dgs_amd import loss as ref
    from binocular3dgs_amd.graphics_utils import inverse_warp_images
    from binocular3dgs_amd.loss_utils import SmoothLoss, l1_loss, ssim
    gen = torch.Generator().manual_seed(5)
    big = torch.rand(1, 5, 40, 48, generator=gen)
    gt = torch.rand(1, 3, 48, 40, generator=gen)
    res = []
    for l1, ss, sm, iw, dev in ((ref.l1_loss, ref.ssim, ref.smooth_loss, ref.inverse_warp_images, "cpu"),
                                (l1_loss, ssim, SmoothLoss(), inverse_warp_images, "cuda")):
        src = _leaf(big, dev)
        img = src[:, 1:4].transpose(2, 3)                       # [1,3,48,40], neither contiguous nor at offset 0
        disp = (src[:, :1].transpose(2, 3) * 6 - 3)
        m = src[:, 4:5].transpose(2, 3).expand(1, 3, 48, 40)    # a mask of the images' own shape through expand()
        g = gt.to(dev)                                          # no gradient asked for
        total = l1(img, g, mask=m) + 0.4 * ss(img, g) + 0.2 * sm(disp, g) + (iw(img, disp) * g).mean()
        total = total.double() * 1.5                            # the upstream gradient arrives as float64
        total.backward()
        res.append((total, src.grad))
        assert g.grad is None
    assert _rel(res[1][0], res[0][0]) <= 3e-5 and _rel(res[1][1], res[0][1]) <= 2e-4
    # value under no_grad, and the value-only form with inputs that do require a gradient
    a = torch.rand(1, 3, 20, 20, generator=gen).cuda().requires_grad_(True)
    with torch.no_grad():
        v = ssim(a, gt[..., :20, :20].cuda())
    assert not v.requires_grad and _rel(v, ref.ssim(a.detach().cpu(), gt[..., :20, :20])) <= 3e-5
    # half-precision inputs are computed in float32 (the reference would run its convolutions in half)
    h = l1_loss(a.detach().half(), gt[..., :20, :20].cuda().half())
    assert h.dtype == torch.float32 and _rel(h, ref.l1_loss(a.detach().half().float().cpu(), gt[..., :20, :20].half().float())) <= 1e-5


def test_zero_size_inputs_and_empty_models():
    from binocular3dgs_amd.gaussian_model import GaussianModel
    from binocular3dgs_amd.loss_utils import l1_loss
    z = torch.zeros(1, 3, 0, 8, device="cuda")
    assert torch.isnan(l1_loss(z, z))                           # torch.abs(x - y).mean() of nothing
    m = GaussianModel.from_tensors(torch.zeros(0, 3), torch.zeros(0, 1, 3), torch.zeros(0, 3, 3), torch.zeros(0, 3),
                                   torch.zeros(0, 4), torch.zeros(0, 1), sh_degree=1, device="cuda")
    m.spatial_lr_scale = 1.0
    m.training_setup(_OPT)
    m.opacity_decay(factor=0.995)
    leaf = torch.zeros(0, 3, device="cuda", requires_grad=True)
    leaf.grad = torch.zeros(0, 3, device="cuda")
    m.add_densification_stats(leaf, torch.zeros(0, dtype=torch.bool, device="cuda"))
    for p in m.parameters():
        p.grad = torch.zeros_like(p)
    m.optimizer.step()                                          # nothing to update, nothing raised
    m.optimizer.zero_grad(set_to_none=True)
    assert m.denom.shape == (0, 1) and float(m.optimizer.state[m._xyz]["step"]) == 1.0


def test_adam_takes_gradients_torch_adam_would_take():
    """optim.Adam next to torch.optim.Adam on the same values: a gradient that is a non-contiguous view, a float64
    gradient (torch refuses a dtype mismatch at assignment, so only layouts vary), groups added later with their own betas /
    eps, a learning rate held in a tensor; the state dict of one continues in the other."""
    from binocular3dgs_amd.optim import Adam
    gen = torch.Generator().manual_seed(11)
    shapes = [(257, 3), (64, 1, 3), (1000,), (33, 4)]
    vals = [torch.randn(*s, generator=gen) for s in shapes]
    mine = [torch.nn.Parameter(v.clone().cuda()) for v in vals]
    ref = [torch.nn.Parameter(v.clone().cuda()) for v in vals]
    oa = Adam([{"params": [mine[0]], "lr": 1e-2, "name": "a"}, {"params": [mine[1]], "lr": torch.tensor(3e-3), "name": "b"}],
              lr=0.0, eps=1e-15)
    ob = torch.optim.Adam([{"params": [ref[0]], "lr": 1e-2, "name": "a"}, {"params": [ref[1]], "lr": torch.tensor(3e-3), "name": "b"}],
                          lr=0.0, eps=1e-15)
    oa.add_param_group({"params": [mine[2], mine[3]], "lr": 5e-3, "betas": (0.8, 0.99), "eps": 1e-8, "name": "late"})
    ob.add_param_group({"params": [ref[2], ref[3]], "lr": 5e-3, "betas": (0.8, 0.99), "eps": 1e-8, "name": "late"})
    for k in range(4):
        for p, q in zip(mine, ref):
            gr = torch.randn(*p.shape, generator=gen).cuda()
            if p.dim() == 2 and k % 2:                          # a transposed buffer viewed back: same values, other strides
                gr = gr.t().contiguous().t()
                assert not gr.is_contiguous() or gr.shape[1] == 1
            p.grad, q.grad = gr, gr.clone()
        if k == 2:                                              # the later group skips a step for one tensor (torch's rule)
            mine[3].grad = ref[3].grad = None
        oa.step(), ob.step()
    for p, q in zip(mine, ref):
        assert _rel(p, q) <= 2e-6
        assert float(oa.state[p]["step"]) == float(ob.state[q]["step"])
        assert _rel(oa.state[p]["exp_avg_sq"], ob.state[q]["exp_avg_sq"]) <= 3e-5
    # hand the state to torch's optimiser and back: both continue on the same numbers
    ob2 = torch.optim.Adam([{"params": [ref[0]], "lr": 0.0, "name": "a"}, {"params": [ref[1]], "lr": 0.0, "name": "b"},
                            {"params": [ref[2], ref[3]], "lr": 0.0, "name": "late"}], lr=0.0, eps=1e-15)
    for p, q in zip(mine, ref):
        q.data.copy_(p.data)
    ob2.load_state_dict(copy.deepcopy(oa.state_dict()))   # (torch keeps the host `step` tensors of the dict it is handed: two live
    #                                                       optimisers must not share them)
    assert [g_["betas"] for g_ in ob2.param_groups][2] == (0.8, 0.99) and ob2.param_groups[2]["eps"] == 1e-8
    for p, q in zip(mine, ref):
        gr = torch.randn(*p.shape, generator=gen).cuda()
        p.grad, q.grad = gr, gr.clone()
    oa.step(), ob2.step()
    for p, q in zip(mine, ref):
        assert _rel(p, q) <= 2e-6
    # step hooks and the profiler range are torch's (this class skips torch's wrapper while neither is in use)
    calls = []
    small = torch.nn.Parameter(torch.ones(5, device="cuda"))
    os_ = Adam([small], lr=0.1)
    h1 = os_.register_step_pre_hook(lambda o, a, k: calls.append("pre"))
    h2 = os_.register_step_post_hook(lambda o, a, k: calls.append("post"))
    small.grad = torch.ones(5, device="cuda")
    os_.step()
    assert calls == ["pre", "post"] and float(small[0]) == pytest.approx(0.9, rel=1e-5)
    h1.remove(), h2.remove()
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU]) as prof:
        os_.step()
        os_.zero_grad()
    assert small.grad is None and any("Optimizer.step#Adam.step" in e.key for e in prof.key_averages())
    assert calls == ["pre", "post"] and float(os_.state[small]["step"]) == 2.0
    # a parameter replaced behind the optimiser's back (what a densification does) leaves the plan of checked tensors
    new = torch.nn.Parameter(torch.ones(9, device="cuda"))
    st = os_.state.pop(small)
    os_.param_groups[0]["params"][0] = new
    os_.state[new] = {"step": st["step"], "exp_avg": torch.zeros(9, device="cuda"), "exp_avg_sq": torch.zeros(9, device="cuda")}
    new.grad = torch.ones(9, device="cuda")
    os_.step()
    assert [id(k) for k in os_._b3gs_plan] == [id(new)]
    # host parameters: no CPU path
    from binocular3dgs_amd._lib import B3gsError
    hp = torch.nn.Parameter(torch.zeros(3))
    hp.grad = torch.ones(3)
    with pytest.raises(B3gsError):
        Adam([hp], lr=1e-3).step()
    with pytest.raises(NotImplementedError):
        Adam([mine[0]], lr=1e-3, weight_decay=0.1)


def test_add_densification_stats_row_layouts():
    """train.py:179 hands `viewspace_point_tensor` (a [P,3] leaf whose .grad the rasterizer wrote) and a bool filter; an index
    tensor as the filter (torch's own indexing semantics) or a gradient with another row stride give the same statistics."""
    from binocular3dgs_amd.gaussian_model import GaussianModel
    P = 1000
    gen = torch.Generator().manual_seed(3)
    m = GaussianModel.from_tensors(torch.zeros(P, 3), torch.zeros(P, 1, 3), torch.zeros(P, 3, 3), torch.zeros(P, 3),
                                   torch.zeros(P, 4), torch.zeros(P, 1), sh_degree=1, device="cuda")
    g4 = torch.randn(P, 4, generator=gen).cuda()
    filt = (torch.rand(P, generator=gen) < 0.4).cuda()
    with pytest.raises(RuntimeError, match="training_setup"):   # (the reference's statistics are empty until then, too)
        m._accumulate_stats(g4[:, :3], filt)
    m.spatial_lr_scale = 1.0
    m.training_setup(_OPT)
    leaf = torch.zeros(P, 3, device="cuda", requires_grad=True)
    want_acc, want_den = torch.zeros(P, 1, device="cuda"), torch.zeros(P, 1, device="cuda")
    for f, gr in ((filt, g4[:, :3]), (filt, g4[:, :3].contiguous()), (torch.nonzero(filt).squeeze(1), g4[:, :3]),
                  (filt.clone(), g4[:, :3].contiguous())):
        leaf.grad = gr                                          # (a [P,3] view of a [P,4] buffer: row stride 4)
        m.add_densification_stats(leaf, f)
        want_acc[filt] += torch.norm(g4[filt, :2], dim=-1, keepdim=True)
        want_den[filt] += 1
    assert _rel(m.xyz_gradient_accum, want_acc) <= 1e-6 and torch.equal(m.denom, want_den)
