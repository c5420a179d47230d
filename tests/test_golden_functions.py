"""CPU: golden G10 (tests/golden/functions.npz, the reference's loss functions and per-iteration model methods called one
by one under the import shim) against the PyTorch statements of this build (binocular3dgs_amd/loss.py) and the host logic
of GaussianModel.training_setup / update_learning_rate / add_densification_stats.  The HIP versions of the same functions
face the same fixture in tests/test_gpu_lossfn.py."""
import os
import types

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(__file__), "golden", "functions.npz")
NAMES = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")


@pytest.fixture(scope="module")
def g():
    return np.load(GOLD)


def _t(a, grad=True):
    return torch.from_numpy(np.array(a)).requires_grad_(grad)


def _close(got, ref, rtol=2e-5, what=""):
    got, ref = np.asarray(got), np.asarray(ref)
    scale = max(float(np.abs(ref).max()), 1e-30)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    assert float(np.abs(got - ref).max()) <= rtol * scale, (what, float(np.abs(got - ref).max()), scale)


def test_l1_loss_statement(g):
    from binocular3dgs_amd.loss import l1_loss
    x, y = _t(g["l1_x"]), _t(g["l1_y"])
    v = l1_loss(x, y)
    (1.7 * v).backward()
    _close(v.detach(), g["l1_val"], what="value")
    _close(x.grad, g["l1_gx"]), _close(y.grad, g["l1_gy"])
    x, y, m = _t(g["l1m_x"]), _t(g["l1m_y"]), _t(g["l1m_m"])
    v = l1_loss(x, y, mask=m)
    (0.6 * v).backward()
    _close(v.detach(), g["l1m_val"])
    _close(x.grad, g["l1m_gx"]), _close(y.grad, g["l1m_gy"]), _close(m.grad, g["l1m_gm"])


def test_ssim_statement(g):
    from binocular3dgs_amd.loss import ssim
    a, b = _t(g["ss_a"]), _t(g["ss_b"])
    v = ssim(a, b)
    (1.3 * v).backward()
    _close(v.detach(), g["ss_val"])
    _close(a.grad, g["ss_ga"], 1e-4), _close(b.grad, g["ss_gb"], 1e-4)
    a, b = _t(g["ssb_a"]), _t(g["ssb_b"])
    v = ssim(a, b, size_average=False)
    (v * torch.from_numpy(g["ssb_w"])).sum().backward()
    _close(v.detach(), g["ssb_val"])
    _close(a.grad, g["ssb_ga"], 1e-4), _close(b.grad, g["ssb_gb"], 1e-4)


def test_smooth_loss_statement(g):
    from binocular3dgs_amd.loss import smooth_loss
    d, im = _t(g["sm_d"]), _t(g["sm_im"])
    v = smooth_loss(d, im)
    (2.2 * v).backward()
    _close(v.detach(), g["sm_val"])
    _close(d.grad, g["sm_gd"], 1e-4), _close(im.grad, g["sm_gim"], 1e-4)


def test_inverse_warp_statement(g):
    from binocular3dgs_amd.loss import inverse_warp_images
    im, d = _t(g["iw_im"]), _t(g["iw_d"])
    o = inverse_warp_images(im, d)
    (o * torch.from_numpy(g["iw_up"])).sum().backward()
    _close(o.detach(), g["iw_out"])
    _close(im.grad, g["iw_gim"]), _close(d.grad, g["iw_gd"])
    d2 = _t(g["iw_d"])
    o = inverse_warp_images(torch.ones(2, 1, *g["iw_d"].shape[-2:]), d2)
    (o * torch.from_numpy(g["iw_up"][:, :1])).sum().backward()
    _close(o.detach(), g["iwm_out"])
    assert float(np.abs(g["iwm_gd"]).max()) == 0.0 and float(d2.grad.abs().max()) <= 1e-7     # the shift mask has no gradient


def _args(g):
    a = g["ts_args"]
    return types.SimpleNamespace(percent_dense=float(a[0]), position_lr_init=float(a[1]), position_lr_final=float(a[2]),
                                 position_lr_delay_mult=float(a[3]), position_lr_max_steps=int(a[4]), feature_lr=float(a[5]),
                                 opacity_lr=float(a[6]), scaling_lr=float(a[7]), rotation_lr=float(a[8])), float(a[9])


def model_from_golden(g, device="cpu"):
    from binocular3dgs_amd.gaussian_model import GaussianModel
    t = lambda n: torch.from_numpy(g[f"opt_p0_{n}"])  # noqa: E731
    m = GaussianModel.from_tensors(t("xyz"), t("f_dc"), t("f_rest"), t("scaling"), t("rotation"), t("opacity"), sh_degree=1,
                                   device=device)
    args, scale = _args(g)
    m.spatial_lr_scale = scale
    m.training_setup(args)
    return m


def test_training_setup_groups_and_learning_rate_schedule(g):
    m = model_from_golden(g)
    gs = m.optimizer.param_groups
    assert [x["name"] for x in gs] == list(g["ts_names"]) == list(NAMES)
    np.testing.assert_allclose([x["lr"] for x in gs], g["ts_lrs"], rtol=1e-12)
    assert gs[0]["eps"] == g["ts_eps_betas"][0] and tuple(gs[0]["betas"]) == tuple(g["ts_eps_betas"][1:])
    assert all(len(x["params"]) == 1 for x in gs)
    assert isinstance(m.optimizer, torch.optim.Adam) and m.percent_dense == 0.01
    assert m.xyz_gradient_accum.shape == (203, 1) and m.denom.shape == (203, 1)
    for it, ref in zip(g["ulr_its"], g["ulr_vals"]):
        lr = m.update_learning_rate(int(it))
        assert lr == pytest.approx(float(ref), rel=1e-12) and m.optimizer.param_groups[0]["lr"] == lr
    # the state_dict layout is torch.optim.Adam's own (what the reference's capture() stores)
    sd = m.optimizer.state_dict()
    ref = torch.optim.Adam([{"params": [torch.nn.Parameter(torch.zeros(1))], "lr": 0.1, "name": "xyz"}], lr=0.0, eps=1e-15).state_dict()
    assert set(sd["param_groups"][0]) == set(ref["param_groups"][0])


def test_add_densification_stats_reference_signature(g):
    m = model_from_golden(g)
    leaf = torch.zeros(203, 3, requires_grad=True)
    for k in range(2):
        leaf.grad = torch.from_numpy(g[f"ads_grad{k}"])
        m.add_densification_stats(leaf, torch.from_numpy(g[f"ads_filter{k}"]))
    assert np.array_equal(m.xyz_gradient_accum.numpy(), g["ads_accum"]) and np.array_equal(m.denom.numpy(), g["ads_denom"])
    with pytest.raises(AttributeError):
        m.add_densification_stats(torch.zeros(203, 3), torch.from_numpy(g["ads_filter0"]))      # a tensor without .grad


def test_optimizer_has_no_cpu_path(g):
    from binocular3dgs_amd._lib import B3gsError
    m = model_from_golden(g)
    for p in m.parameters():
        p.grad = torch.zeros_like(p)
    with pytest.raises(B3gsError):
        m.optimizer.step()
