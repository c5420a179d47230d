"""GPU: the fused loss block (b3gs_binocular_loss) against (i) the golden fixture generated from
the reference's own Python (tests/golden/loss_block.npz: train.py:123-148 values and pixel gradients) and
(ii) the PyTorch statement of the same block (binocular3dgs_amd/loss.py) with autograd, on ragged sizes."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_golden_loss_block_values_and_pixel_gradients():
    from binocular3dgs_amd.fused_loss import binocular_loss_fused
    g = np.load(os.path.join(GOLD, "loss_block.npz"))
    t = lambda k: torch.from_numpy(g[k]).cuda().requires_grad_(True)  # noqa: E731
    image, depth, alpha, shifted = t("image"), t("depth"), t("alpha"), t("shifted")
    focal_x, trans, lam = [float(x) for x in g["scalars"]]
    total, parts = binocular_loss_fused(image, depth, alpha, torch.from_numpy(g["gt"]).cuda(), lambda_dssim=lam,
                                        shifted_image=shifted, focal_x=focal_x, trans_dist=trans,
                                        gt_alpha_mask=torch.from_numpy(g["gt_alpha_mask"]).cuda(), return_parts=True)
    total.backward()
    parts = parts.cpu().numpy()
    for i, k in ((0, "total"), (1, "Ll1"), (2, "ssim"), (3, "l1_masked"), (4, "smooth"), (5, "alpha_loss")):
        np.testing.assert_allclose(parts[i], g[k], rtol=3e-5, atol=1e-7, err_msg=k)
    for ten, key in ((image, "g_image"), (depth, "g_depth"), (alpha, "g_alpha"), (shifted, "g_shifted")):
        got, ref = ten.grad.cpu().numpy(), g[key]
        assert np.abs(ref).max() > 0
        assert np.abs(got - ref).max() <= 2e-4 * np.abs(ref).max() + 2e-9, key


@pytest.mark.parametrize("W,H,shift,aw", [(203, 157, True, "bg"), (64, 48, False, None), (800, 600, True, "alpha"),
                                           (33, 17, True, None)])
def test_matches_the_pytorch_statement(W, H, shift, aw):
    from binocular3dgs_amd.fused_loss import binocular_loss_fused
    from binocular3dgs_amd.loss import binocular_loss
    gen = torch.Generator().manual_seed(W * 1000 + H)
    r = lambda *s: torch.rand(*s, generator=gen)  # noqa: E731
    gt = r(3, H, W).cuda()
    base = dict(image=(gt.cpu() + 0.2 * (r(3, H, W) - 0.5)).clamp(0, 1), depth=2.0 + 6.0 * r(1, H, W), alpha=r(1, H, W),
                shifted=r(3, H, W))
    mask = (r(1, H, W) > 0.5).float().cuda()
    kw = dict(lambda_dssim=0.2)
    if shift:
        kw.update(focal_x=0.8 * W, trans_dist=-0.23)
    if aw == "bg":
        kw["bg_mask"] = mask
    elif aw == "alpha":
        kw["gt_alpha_mask"] = mask
    res = []
    for fn in (lambda *a, **k: binocular_loss(*a, **k)[0], binocular_loss_fused):
        t = {k: v.clone().cuda().requires_grad_(True) for k, v in base.items()}
        total = fn(t["image"], t["depth"], t["alpha"], gt, shifted_image=t["shifted"] if shift else None, **kw)
        (2.5 * total).backward()           # a non-unit upstream gradient
        res.append((float(total), {k: (None if v.grad is None else v.grad.cpu().numpy()) for k, v in t.items()}))
    (ref_total, ref_g), (got_total, got_g) = res
    assert got_total == pytest.approx(ref_total, rel=2e-5)
    for k in ("image", "depth", "alpha", "shifted"):
        if ref_g[k] is None or np.abs(ref_g[k]).max() == 0:
            assert got_g[k] is None or np.abs(got_g[k]).max() == 0, k
            continue
        # |.| terms: a pixel whose residual is ~1 ulp from zero may take the other sign in one of the two
        # implementations (different rounding of the warp / convolution): allow a few isolated pixels, bound the rest
        d = np.abs(got_g[k] - ref_g[k])
        scale = np.abs(ref_g[k]).max()
        assert float((d > 2e-4 * scale).mean()) <= 2e-5, (k, float((d > 2e-4 * scale).mean()))
        assert np.linalg.norm(d) <= 2e-3 * np.linalg.norm(ref_g[k]), k


def test_unit_grad_shortcut_and_slots_are_independent():
    from binocular3dgs_amd.fused_loss import binocular_loss_fused
    gen = torch.Generator().manual_seed(3)
    H, W = 40, 56
    gts = [torch.rand(3, H, W, generator=gen).cuda() for _ in range(2)]
    ims = [torch.rand(3, H, W, generator=gen).cuda().requires_grad_(True) for _ in range(2)]
    dp = [(2 + torch.rand(1, H, W, generator=gen)).cuda().requires_grad_(True) for _ in range(2)]
    al = [torch.rand(1, H, W, generator=gen).cuda().requires_grad_(True) for _ in range(2)]
    sh = [torch.rand(3, H, W, generator=gen).cuda().requires_grad_(True) for _ in range(2)]
    tot = sum(binocular_loss_fused(ims[i], dp[i], al[i], gts[i], shifted_image=sh[i], focal_x=40.0, trans_dist=0.3,
                                   slot=i, unit_grad=True) for i in range(2))
    tot.backward()
    assert ims[0].grad is not None and ims[1].grad is not None
    assert float((ims[0].grad - ims[1].grad).abs().max()) > 0     # two slots, two different pairs
    x = ims[1].detach().clone().requires_grad_(True)
    binocular_loss_fused(x, dp[1].detach(), al[1].detach(), gts[1], shifted_image=sh[1].detach(), focal_x=40.0,
                         trans_dist=0.3, slot=7).backward()
    assert torch.allclose(x.grad, ims[1].grad, rtol=1e-5, atol=1e-10)


def test_batched_pairs_equal_single_pair_calls():
    from binocular3dgs_amd.fused_loss import binocular_loss_fused, binocular_loss_fused_batch
    gen = torch.Generator().manual_seed(11)
    H, W = 72, 100
    mk = lambda *s: torch.rand(*s, generator=gen).cuda()  # noqa: E731
    pairs = []
    for k in range(3):
        pairs.append(dict(image=mk(3, H, W), depth=2 + 5 * mk(1, H, W), alpha=mk(1, H, W), gt_image=mk(3, H, W),
                          shifted_image=None if k == 1 else mk(3, H, W), focal_x=70.0, trans_dist=0.3 - 0.25 * k,
                          bg_mask=(mk(1, H, W) > 0.5).float() if k != 2 else None))
    names = ("image", "depth", "alpha", "shifted_image")
    res = []
    for batched in (False, True):
        ps = [{k: (v.clone().requires_grad_(True) if k in names and v is not None else v) for k, v in p.items()} for p in pairs]
        if batched:
            total, parts = binocular_loss_fused_batch(ps, return_parts=True)
        else:
            total = sum(binocular_loss_fused(p["image"], p["depth"], p["alpha"], p["gt_image"], shifted_image=p["shifted_image"],
                                             focal_x=p["focal_x"], trans_dist=p["trans_dist"], bg_mask=p["bg_mask"], slot=k)
                        for k, p in enumerate(ps))
        (1.5 * total).backward()
        res.append((float(total), [[None if p[n] is None or p[n].grad is None else p[n].grad.clone() for n in names] for p in ps]))
    assert res[0][0] == pytest.approx(res[1][0], rel=1e-6)
    for ga, gb in zip(res[0][1], res[1][1]):
        for a, b in zip(ga, gb):
            assert (a is None) == (b is None)
            if a is not None:
                assert torch.allclose(a, b, rtol=1e-5, atol=1e-10)
    assert parts.shape == (3, 8) and float(parts[1, 3]) == 0.0      # pair 1 has no binocular term
