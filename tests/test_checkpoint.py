"""Checkpoint tuple (scene/gaussian_model.py:61-93, train.py:41-43,200-202).

CPU: the tuple this build captures has the STRUCTURE the reference's `capture()` produced (G9: element kinds and shapes,
state_dict keys in order, group names / learning rates / key order), carries the same numbers when fed the reference's
moments, and survives torch.save / torch.load; when /root/reference exists (build container only) the reference's own
`restore()` consumes it.  GPU: capture -> restore into a fresh model + optimiser -> continue equals the uninterrupted
run bit for bit, for FusedAdam and ShardedAdam, and one resumed step reproduces the reference's resumed step (G9)."""
import json
import os

import numpy as np
import pytest
import torch

from binocular3dgs_amd import checkpoint
from binocular3dgs_amd.gaussian_model import GaussianModel
from binocular3dgs_amd.step import FusedAdam, ShardedAdam

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "checkpoint.npz")
REF = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")
MODEL_ORDER = ("xyz", "f_dc", "f_rest", "scaling", "rotation", "opacity")


def gold():
    z = np.load(GOLD)
    return z, json.loads(str(z["structure_json"]))


def model_from_gold(z, dev="cpu"):
    t = lambda n: torch.from_numpy(z[f"p_{n}"])  # noqa: E731
    m = GaussianModel.from_tensors(t("xyz"), t("f_dc"), t("f_rest"), t("scaling"), t("rotation"), t("opacity"),
                                   sh_degree=1, device=dev)
    m.max_radii2D = torch.from_numpy(z["max_radii2D"]).to(dev)
    m.xyz_gradient_accum = torch.from_numpy(z["xyz_gradient_accum"]).to(dev)
    m.denom = torch.from_numpy(z["denom"]).to(dev)
    m.spatial_lr_scale = float(z["spatial_lr_scale"])
    return m


def lrs_model_order(st):
    by = dict(zip(st["group_names"], st["group_lrs"]))
    return [by[n] for n in MODEL_ORDER]


def load_moments(opt, z, dev="cpu"):
    """put the reference's moments (tensor-major, model order) into a FusedAdam"""
    opt.exp_avg.copy_(torch.cat([torch.from_numpy(z[f"m_{n}"]).reshape(-1) for n in MODEL_ORDER]).to(dev))
    opt.exp_avg_sq.copy_(torch.cat([torch.from_numpy(z[f"v_{n}"]).reshape(-1) for n in MODEL_ORDER]).to(dev))
    opt.step_count.fill_(2)


def kind(x):
    if torch.is_tensor(x):
        return ["Parameter" if isinstance(x, torch.nn.Parameter) else "Tensor", list(x.shape), str(x.dtype)]
    return [type(x).__name__, x if isinstance(x, (int, float)) else None]


def test_capture_has_the_reference_structure_and_numbers():
    z, st = gold()
    m = model_from_gold(z)
    opt = FusedAdam(m.parameters(), lrs_model_order(st), eps=st["group_eps"])
    load_moments(opt, z)
    tup = m.capture(opt)
    assert len(tup) == st["len"] == 12
    for i, (x, want) in enumerate(zip(tup, st["elements"])):
        if i == 10:
            assert list(x.keys()) == want[1]
        else:
            assert kind(x) == want, (i, kind(x), want)
    sd = tup[10]
    assert [int(k) for k in sd["state"].keys()] == st["state_keys"]
    assert list(sd["state"][0].keys()) == st["state_entry_keys"]
    assert list(sd["param_groups"][0].keys()) == st["group_keys"]
    assert [g["name"] for g in sd["param_groups"]] == st["group_names"] == list(REF)
    assert [g["params"] for g in sd["param_groups"]] == st["group_params"]
    np.testing.assert_allclose([g["lr"] for g in sd["param_groups"]], st["group_lrs"], rtol=0, atol=0)
    assert sd["param_groups"][0]["eps"] == st["group_eps"] and list(sd["param_groups"][0]["betas"]) == st["group_betas"]
    for i, n in enumerate(REF):
        e = sd["state"][i]
        assert kind(e["step"]) + [float(e["step"])] == st["state_step"][i]
        assert np.array_equal(e["exp_avg"].numpy(), z[f"m_{n}"]) and np.array_equal(e["exp_avg_sq"].numpy(), z[f"v_{n}"])
        assert e["exp_avg"].shape == getattr(m, checkpoint._ATTR[n]).shape
    # a fresh optimiser has no state entries, like torch.optim.Adam before its first step
    sd0 = checkpoint.optimizer_state_dict(m, FusedAdam(m.parameters(), lrs_model_order(st)))
    assert sd0["state"] == {} and len(sd0["param_groups"]) == 6


def test_torch_adam_accepts_the_captured_state(tmp_path):
    """What the reference's restore() does with the tuple: training_setup -> torch.optim.Adam -> load_state_dict."""
    z, st = gold()
    m = model_from_gold(z)
    opt = FusedAdam(m.parameters(), lrs_model_order(st), eps=st["group_eps"])
    load_moments(opt, z)
    path = str(tmp_path / "chkpnt7.pth")
    checkpoint.save(path, m, opt, 7)
    tup, it = checkpoint.load(path)
    assert it == 7
    params = {n: torch.nn.Parameter(tup[1 + ["xyz", "f_dc", "f_rest", "scaling", "rotation", "opacity"].index(n)].detach().clone())
              for n in REF}
    ref_opt = torch.optim.Adam([{"params": [params[n]], "lr": 0.123, "name": n} for n in REF], lr=0.0, eps=1e-15)
    ref_opt.load_state_dict(tup[10])
    for n in REF:
        params[n].grad = torch.from_numpy(z[f"g_{n}"])
    ref_opt.step()
    for n in REF:   # one resumed step = what the reference's own resumed optimiser produced (G9)
        np.testing.assert_allclose(params[n].detach().numpy(), z[f"after_{n}"], rtol=1e-6, atol=1e-9)


def test_restore_roundtrip_cpu_structures():
    z, st = gold()
    m = model_from_gold(z)
    opt = FusedAdam(m.parameters(), lrs_model_order(st), eps=st["group_eps"])
    load_moments(opt, z)
    tup = m.capture(opt)
    m2 = GaussianModel.from_tensors(*[torch.zeros_like(p) for p in m.parameters()], sh_degree=1, active_sh_degree=0)
    opt2 = FusedAdam(m2.parameters(), [1.0] * 6)
    m2.restore(tup, None, optimizer=opt2)
    assert m2.active_sh_degree == 1 and m2.spatial_lr_scale == 1.0 and m2.optimizer is opt2
    for a, b in zip(m.parameters(), m2.parameters()):
        assert torch.equal(a, b)
    assert torch.equal(opt.exp_avg, opt2.exp_avg) and torch.equal(opt.exp_avg_sq, opt2.exp_avg_sq)
    assert int(opt2.step_count) == 2 and opt2.lrs == opt.lrs
    assert torch.equal(m2.denom, m.denom) and torch.equal(m2.max_radii2D, m.max_radii2D)
    # a different Gaussian count (checkpoint taken after a densification): parameters and moments are re-created
    m3 = GaussianModel.from_tensors(*[torch.zeros((5,) + tuple(p.shape[1:])) for p in m.parameters()], sh_degree=1)
    opt3 = FusedAdam(m3.parameters(), [1.0] * 6)
    m3.restore(tup, None, optimizer=opt3)
    assert m3.get_xyz.shape[0] == m.get_xyz.shape[0] and torch.equal(opt3.exp_avg, opt.exp_avg)
    # mismatching moments are refused
    bad = list(tup)
    bad[10] = {"state": {k: dict(v, exp_avg=v["exp_avg"][:3]) for k, v in tup[10]["state"].items()},
               "param_groups": tup[10]["param_groups"]}
    with pytest.raises(ValueError):
        m2.restore(tuple(bad), None, optimizer=opt2)


def test_restore_into_a_torch_optimizer_in_model_order_pairs_groups_by_name():
    """load_state_dict() pairs groups by POSITION; a torch.optim.Adam built over model.parameters() (xyz, f_dc, f_rest,
    scaling, rotation, opacity) must still receive each tensor's own moments from a checkpoint in the reference's order."""
    z, st = gold()
    m = model_from_gold(z)
    opt = FusedAdam(m.parameters(), lrs_model_order(st), eps=st["group_eps"])
    load_moments(opt, z)
    tup = m.capture(opt)
    m2 = GaussianModel.from_tensors(*[torch.zeros_like(p) for p in m.parameters()], sh_degree=1)
    topt = torch.optim.Adam([{"params": [p], "lr": 0.5, "name": n} for p, n in zip(m2.parameters(), MODEL_ORDER)], lr=0.0, eps=1e-15)
    m2.restore(tup, None, optimizer=topt)
    for p, n in zip(m2.parameters(), MODEL_ORDER):
        assert np.array_equal(topt.state[p]["exp_avg"].numpy(), z[f"m_{n}"]), n
        assert np.array_equal(topt.state[p]["exp_avg_sq"].numpy(), z[f"v_{n}"]), n
        p.grad = torch.from_numpy(z[f"g_{n}"])
    by = dict(zip(st["group_names"], st["group_lrs"]))
    assert [g["lr"] for g in topt.param_groups] == [by[n] for n in MODEL_ORDER]
    # ... and capture() of that torch optimiser emits the reference's order again
    sd = m2.capture(topt)[10]
    assert [g["name"] for g in sd["param_groups"]] == list(REF) and [g["params"] for g in sd["param_groups"]] == [[i] for i in range(6)]
    for i, n in enumerate(REF):
        assert np.array_equal(sd["state"][i]["exp_avg"].numpy(), z[f"m_{n}"]), n
    topt.step()
    for p, n in zip(m2.parameters(), MODEL_ORDER):
        np.testing.assert_allclose(p.detach().numpy(), z[f"after_{n}"], rtol=1e-6, atol=1e-9)


def test_restore_refuses_a_torch_optimizer_built_for_another_gaussian_count():
    """a checkpoint with other shapes installs fresh Parameters; a torch optimiser built before that owns the old
    tensors and load_state_dict() would attach the moments to them silently.  restore() raises; optimizer_factory works."""
    z, st = gold()
    m = model_from_gold(z)
    opt = FusedAdam(m.parameters(), lrs_model_order(st), eps=st["group_eps"])
    load_moments(opt, z)
    tup = m.capture(opt)
    m3 = GaussianModel.from_tensors(*[torch.zeros((5,) + tuple(p.shape[1:])) for p in m.parameters()], sh_degree=1)
    stale = torch.optim.Adam([{"params": [p], "lr": 0.5, "name": n} for p, n in zip(m3.parameters(), MODEL_ORDER)], lr=0.0, eps=1e-15)
    with pytest.raises(ValueError, match="optimizer_factory"):
        m3.restore(tup, None, optimizer=stale)

    def factory(model):
        return torch.optim.Adam([{"params": [p], "lr": 0.5, "name": n} for p, n in zip(model.parameters(), MODEL_ORDER)],
                                lr=0.0, eps=1e-15)
    m4 = GaussianModel.from_tensors(*[torch.zeros((5,) + tuple(p.shape[1:])) for p in m.parameters()], sh_degree=1)
    topt = m4.restore(tup, None, optimizer_factory=factory)
    for p, n in zip(m4.parameters(), MODEL_ORDER):
        assert p.shape[0] == m.get_xyz.shape[0] and np.array_equal(topt.state[p]["exp_avg"].numpy(), z[f"m_{n}"]), n


def test_restore_with_training_args_is_the_reference_flow():
    """train.py:41-43 / scene/gaussian_model.py:77-93: `restore(model_params, opt)` rebuilds the optimiser through
    training_setup(opt) -- which zeroes the statistics -- and THEN installs the captured statistics and optimiser state."""
    from types import SimpleNamespace
    z, st = gold()
    m = model_from_gold(z)
    opt = FusedAdam(m.parameters(), lrs_model_order(st), eps=st["group_eps"])
    load_moments(opt, z)
    tup = m.capture(opt)
    ta = SimpleNamespace(percent_dense=0.01, position_lr_init=0.00016, position_lr_final=0.0000016, position_lr_delay_mult=0.01,
                         position_lr_max_steps=30000, feature_lr=0.0025, opacity_lr=0.05, scaling_lr=0.005, rotation_lr=0.001)
    m2 = GaussianModel(sh_degree=1)
    m2.restore(tup, ta)
    assert [g["name"] for g in m2.optimizer.param_groups] == list(REF)
    assert isinstance(m2.optimizer, torch.optim.Adam)
    assert torch.equal(m2.xyz_gradient_accum, m.xyz_gradient_accum) and m2.xyz_gradient_accum.abs().sum() > 0
    assert torch.equal(m2.denom, m.denom) and torch.equal(m2.max_radii2D, m.max_radii2D)
    by = dict(zip(st["group_names"], st["group_lrs"]))
    for g in m2.optimizer.param_groups:
        n = g["name"]
        assert g["lr"] == by[n] and g["eps"] == st["group_eps"], n
        e = m2.optimizer.state[g["params"][0]]
        assert np.array_equal(e["exp_avg"].numpy(), z[f"m_{n}"]) and np.array_equal(e["exp_avg_sq"].numpy(), z[f"v_{n}"]), n
        assert float(e["step"]) == 2.0
    # the state_dict of the rebuilt optimiser is, key for key, what the reference's torch.optim.Adam(l, lr=0.0, eps=1e-15) holds
    sd = m2.optimizer.state_dict()
    assert list(sd["param_groups"][0].keys()) == st["group_keys"]
    ref_like = torch.optim.Adam([{"params": [torch.nn.Parameter(torch.zeros(1))], "lr": 0.1, "name": "xyz"}], lr=0.0, eps=1e-15)
    want = ref_like.state_dict()["param_groups"][0]
    got = sd["param_groups"][0]
    for k in want:
        if k not in ("lr", "params"):
            assert got[k] == want[k], (k, got[k], want[k])


@pytest.mark.skipif(not os.path.isdir("/root/reference/scene"), reason="the reference tree exists in the build container only")
def test_reference_restore_consumes_our_tuple():
    """Live check (build container): the reference's own GaussianModel.restore() takes the tuple this build captured and
    its next optimiser step lands where the reference's own resumed run landed (G9)."""
    import subprocess
    import sys
    code = r'''
import sys, types, numpy as np, torch
sys.path.insert(0, "tests/golden"); sys.path.insert(0, ".")
from make_golden import CudaToCpu, install_shim
from make_golden_checkpoint import ARGS
from make_golden_densify import NAMES
import importlib
tc = importlib.import_module("tests.test_checkpoint")
z, st = tc.gold()
m = tc.model_from_gold(z)
from binocular3dgs_amd.step import FusedAdam
opt = FusedAdam(m.parameters(), tc.lrs_model_order(st), eps=st["group_eps"]); tc.load_moments(opt, z)
tup = m.capture(opt)
install_shim({}); torch.nn.Module.cuda = lambda self, *a, **k: self
with CudaToCpu():
    from scene.gaussian_model import GaussianModel as RefModel
    gm = RefModel(1)
    gm.restore(tuple(torch.nn.Parameter(x.detach().clone()) if 1 <= i <= 6 else x for i, x in enumerate(tup)),
               types.SimpleNamespace(**ARGS))
    groups = {g["name"]: g for g in gm.optimizer.param_groups}
    for n in NAMES:
        groups[n]["params"][0].grad = torch.from_numpy(z["g_" + n])
    gm.optimizer.step()
    for n in NAMES:
        np.testing.assert_allclose(groups[n]["params"][0].detach().numpy(), z["after_" + n], rtol=1e-6, atol=1e-9)
print("REF_RESTORE_OK")
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=300)
    assert "REF_RESTORE_OK" in res.stdout, res.stdout[-2000:] + res.stderr[-2000:]


# ---------------------------------------------------------------------------------------------------------
def _train_steps(model, opt, n, seed0):
    """n optimiser steps with seeded dense gradients (the Adam kernels; the rasterizer is not needed here)"""
    from binocular3dgs_amd.step import FlatGradSlab
    slab = FlatGradSlab(model.parameters(), getattr(opt, "padded_numel", 0))   # p.grad = views into one flat buffer
    for k in range(n):
        g = torch.Generator().manual_seed(seed0 + k)
        for p in model.parameters():
            p.grad.copy_((1e-3 * torch.randn(p.shape, generator=g)).to(p.device))
        if isinstance(opt, ShardedAdam):
            opt.step(slab)
        else:
            opt.step()


@pytest.mark.gpu
@pytest.mark.parametrize("kind_", ["fused", "sharded"])
def test_resume_equals_uninterrupted_gpu(kind_, tmp_path):
    z, st = gold()
    dev = torch.device("cuda", 0)
    lrs = lrs_model_order(st)

    def make(m):
        cls = FusedAdam if kind_ == "fused" else ShardedAdam
        return cls(m.parameters(), lrs, eps=1e-15, opacity_decay=0.995, opacity_index=5, decay_first=True)

    a = model_from_gold(z, dev)
    oa = make(a)
    _train_steps(a, oa, 3, 100)
    path = str(tmp_path / "chk.pth")
    checkpoint.save(path, a, oa, 3)
    _train_steps(a, oa, 2, 103)                      # uninterrupted
    tup, it = checkpoint.load(path, map_location="cpu")
    assert it == 3
    b = GaussianModel.from_tensors(*[torch.zeros_like(p) for p in a.parameters()], sh_degree=1, device=dev)
    ob = make(b)
    b.restore(tup, None, optimizer=ob)
    _train_steps(b, ob, 2, 103)                      # resumed
    for pa, pb in zip(a.parameters(), b.parameters()):
        assert torch.equal(pa, pb)
    ma, va = checkpoint._full_moments(oa)[:2]
    mb, vb = checkpoint._full_moments(ob)[:2]
    assert torch.equal(ma, mb) and torch.equal(va, vb) and int(oa.step_count) == int(ob.step_count) == 5


@pytest.mark.gpu
def test_resumed_step_matches_the_reference_gpu():
    """restore the reference-made state (G9) into FusedAdam, take the step the reference took after ITS restore()"""
    z, st = gold()
    dev = torch.device("cuda", 0)
    m = model_from_gold(z, dev)
    opt = FusedAdam(m.parameters(), lrs_model_order(st), eps=st["group_eps"])
    load_moments(opt, z, dev)
    tup = m.capture(opt)
    m2 = GaussianModel.from_tensors(*[torch.zeros_like(p) for p in m.parameters()], sh_degree=1, device=dev)
    opt2 = FusedAdam(m2.parameters(), [0.5] * 6, eps=st["group_eps"])
    m2.restore(tup, None, optimizer=opt2)
    for n, p in zip(MODEL_ORDER, m2.parameters()):
        p.grad = torch.from_numpy(z[f"g_{n}"]).to(dev)
    opt2.step()
    for n, p in zip(MODEL_ORDER, m2.parameters()):
        np.testing.assert_allclose(p.detach().cpu().numpy(), z[f"after_{n}"], rtol=1e-5, atol=1e-7)


# ---------------------------------------------------------------------------------------------------------
# sharded optimiser state over two ranks (gloo, CPU): the moments live 1/N per rank; capture() gathers them into the
# reference's state_dict, restore() hands every rank its shard back
def _ckpt_rank(rank, world, port, out):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tests"))
    import torch.distributed as dist
    from test_dp_views_gloo import torch_adam_impl
    from binocular3dgs_amd.step import FlatGradSlab
    if world > 1:
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    z, st = gold()
    lrs = lrs_model_order(st)

    def steps(model, opt, n, seed0):
        slab = FlatGradSlab(model.parameters(), opt.padded_numel)
        for k in range(n):
            g = torch.Generator().manual_seed(seed0 + k)
            for p in model.parameters():
                p.grad.copy_(1e-3 * torch.randn(p.shape, generator=g))     # the same (already summed) gradient on every rank
            if world > 1:
                slab.flat.div_(world)                                       # ... so that the reduce-scatter SUM gives it back
            opt.step(slab)

    a = model_from_gold(z)
    oa = ShardedAdam(a.parameters(), lrs, eps=1e-15, opacity_decay=0.995, opacity_index=5, decay_first=True,
                     adam_impl=torch_adam_impl)
    steps(a, oa, 3, 100)
    tup = a.capture(oa)                       # collective: all-gathers the moments
    import io                                 # (capture() returns the LIVE tensors, like the reference's: train.py saves at once)
    buf = io.BytesIO()
    torch.save(tup, buf)
    buf.seek(0)
    tup = torch.load(buf, weights_only=False)
    steps(a, oa, 2, 103)                      # uninterrupted
    b = GaussianModel.from_tensors(*[torch.zeros_like(p) for p in a.parameters()], sh_degree=1)
    ob = ShardedAdam(b.parameters(), lrs, eps=1e-15, opacity_decay=0.995, opacity_index=5, decay_first=True,
                     adam_impl=torch_adam_impl)
    b.restore(tup, None, optimizer=ob)
    steps(b, ob, 2, 103)                      # resumed
    same = all(torch.equal(x, y) for x, y in zip(a.parameters(), b.parameters())) and \
        torch.equal(oa.exp_avg, ob.exp_avg) and torch.equal(oa.exp_avg_sq, ob.exp_avg_sq) and int(ob.step_count) == 5
    if rank == 0:
        sd = tup[10]
        torch.save(dict(same=same, state={k: {kk: vv.clone() for kk, vv in v.items()} for k, v in sd["state"].items()},
                        names=[g["name"] for g in sd["param_groups"]], params=[p.detach().clone() for p in a.parameters()]), out)
    else:
        assert same
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_sharded_moments_are_captured_and_restored_across_ranks(tmp_path):
    """ShardedAdam keeps 1/N of exp_avg / exp_avg_sq per rank: capture() on 2 ranks (gloo) yields the SAME reference-format
    state_dict as one process, and restore() + two more steps equals the uninterrupted run on every rank."""
    import socket
    import torch.multiprocessing as mp
    outs = []
    for world in (1, 2):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        out = str(tmp_path / f"w{world}.pt")
        if world == 1:
            _ckpt_rank(0, 1, port, out)
        else:
            mp.spawn(_ckpt_rank, args=(world, port, out), nprocs=world, join=True)
        outs.append(torch.load(out, weights_only=False))
    one, two = outs
    assert one["same"] and two["same"] and one["names"] == two["names"] == list(REF)
    for k in one["state"]:
        for kk in ("exp_avg", "exp_avg_sq", "step"):
            assert torch.allclose(one["state"][k][kk], two["state"][k][kk], rtol=1e-6, atol=1e-12), (k, kk)
    for x, y in zip(one["params"], two["params"]):
        assert torch.allclose(x, y, rtol=1e-6, atol=1e-9)
