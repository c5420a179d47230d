"""Densify / clone / split / prune of the Gaussian set with the optimiser state carried along
(scene/gaussian_model.py:258-407, called from train.py:180-185 every densification_interval).

One classification kernel, three prefix sums (torch.cumsum: plumbing), ONE host read-back of the three totals
(the new P sizes every buffer, as in the reference) and one scatter kernel that writes all surviving / new rows
of the six parameter tensors and their Adam moments -- instead of ~100 masked gathers / torch.cat calls.

Data parallel: every replica must take the same decisions -- call ViewShardedStep.sync_densify_stats() first and
pass the same `noise` (or equally seeded generators) everywhere.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch
from torch import nn

from . import _lib

_ATTRS = ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity")


def _adam_state(optimizer, params):
    """-> (list of exp_avg, list of exp_avg_sq) per parameter, or (None, None)."""
    if optimizer is None:
        return None, None
    if hasattr(optimizer, "gather_full_state"):   # step.ShardedAdam: this rank holds 1/N of the moments
        fm, fv = optimizer.gather_full_state()
        return [fm[a:b] for a, b in optimizer.bounds], [fv[a:b] for a, b in optimizer.bounds]
    if hasattr(optimizer, "exp_avg") and torch.is_tensor(optimizer.exp_avg):   # step.FusedAdam: flat buffers
        m, v, off = [], [], 0
        for p in optimizer.params:
            m.append(optimizer.exp_avg[off:off + p.numel()])
            v.append(optimizer.exp_avg_sq[off:off + p.numel()])
            off += p.numel()
        return m, v
    m, v = [], []
    for p in params:
        st = optimizer.state.get(p, None)
        if not st:
            return None, None      # no step taken yet: nothing to carry
        m.append(st["exp_avg"])
        v.append(st["exp_avg_sq"])
    return m, v


def densify_and_prune(model, optimizer, max_grad: float, min_opacity: float, extent: float,
                      max_screen_size: Optional[float] = None, percent_dense: float = 0.01,
                      noise: Optional[torch.Tensor] = None, generator: Optional[torch.Generator] = None) -> int:
    """In-place equivalent of GaussianModel.densify_and_prune(max_grad, min_opacity, extent, max_screen_size):
    replaces the model's six parameters (new nn.Parameter objects), re-creates the densification statistics as
    zeros and moves the optimiser state (step.FusedAdam or torch.optim.Adam with one parameter per group).
    noise: [2, P, 3] standard-normal samples for the split offsets (default: drawn here).  Returns the new P."""
    L = _lib.lib()
    params = [getattr(model, a) for a in _ATTRS]
    xyz = params[0]
    if not xyz.is_cuda:
        raise _lib.B3gsError("densify_and_prune needs the model on the HIP device (no CPU fallback)")
    dev, P = xyz.device, xyz.shape[0]
    M = model._features_dc.shape[1] + model._features_rest.shape[1]
    widths = [3, 3, 3 * (M - 1), 3, 4, 1]
    m_in, v_in = _adam_state(optimizer, params)
    if noise is None:
        noise = torch.randn((2, P, 3), device=dev, generator=generator)
    noise = noise.to(device=dev, dtype=torch.float32).contiguous()
    assert noise.shape == (2, P, 3)
    io = _lib.B3gsDensifyIO()
    io.P, io.M = P, M
    keep_alive = []
    for t, p in enumerate(params):
        d = p.detach().contiguous()
        keep_alive.append(d)
        io.param[t] = d.data_ptr() if d.numel() else None
        if m_in is not None:
            mm, vv = m_in[t].contiguous(), v_in[t].contiguous()
            keep_alive += [mm, vv]
            io.exp_avg[t] = mm.data_ptr() if mm.numel() else None
            io.exp_avg_sq[t] = vv.data_ptr() if vv.numel() else None
    acc = model.xyz_gradient_accum.reshape(-1).contiguous()
    den = model.denom.reshape(-1).contiguous()
    io.xyz_gradient_accum, io.denom = acc.data_ptr(), den.data_ptr()
    io.grad_threshold, io.min_opacity, io.extent = float(max_grad), float(min_opacity), float(extent)
    io.percent_dense = float(percent_dense)
    io.max_screen_size = float(max_screen_size) if max_screen_size else -1.0
    stream = torch.cuda.current_stream(dev).cuda_stream
    flags = torch.empty(P, dtype=torch.int32, device=dev)
    _lib.check(L.b3gs_densify_classify(C.byref(io), flags.data_ptr(), stream), "b3gs_densify_classify")
    bits = torch.stack([flags & 1, (flags >> 1) & 1, (flags >> 2) & 1])           # [3,P]
    incl = torch.cumsum(bits, dim=1, dtype=torch.int32)
    offs = (incl - bits).contiguous()                                               # exclusive
    n_keep, n_clone, n_split = (int(x) for x in incl[:, -1].tolist()) if P else (0, 0, 0)   # the one read-back
    newP = n_keep + n_clone + 2 * n_split
    f = dict(dtype=torch.float32, device=dev)
    out_p = [torch.empty((newP, w), **f) for w in widths]
    carry = m_in is not None
    tot = sum(newP * w for w in widths)
    flat_m = torch.empty(tot, **f) if carry else None
    flat_v = torch.empty(tot, **f) if carry else None
    PtrArr = C.c_void_p * 6
    a_p, a_m, a_v = PtrArr(), PtrArr(), PtrArr()
    off = 0
    for t, w in enumerate(widths):
        a_p[t] = out_p[t].data_ptr() if out_p[t].numel() else None
        if carry:
            a_m[t] = flat_m.data_ptr() + 4 * off if newP * w else None
            a_v[t] = flat_v.data_ptr() + 4 * off if newP * w else None
        off += newP * w
    rc = L.b3gs_densify_scatter(C.byref(io), flags.data_ptr(), offs[0].data_ptr(), offs[1].data_ptr(), offs[2].data_ptr(),
                                n_keep, n_clone, n_split, noise.data_ptr(), a_p, a_m if carry else None,
                                a_v if carry else None, stream)
    _lib.check(rc, "b3gs_densify_scatter")
    shapes = [(newP, 3), (newP, 1, 3), (newP, M - 1, 3), (newP, 3), (newP, 4), (newP, 1)]
    new_params = [nn.Parameter(t.view(s), requires_grad=True) for t, s in zip(out_p, shapes)]
    old_params = params
    for a, p in zip(_ATTRS, new_params):
        setattr(model, a, p)
    model.xyz_gradient_accum = torch.zeros((newP, 1), device=dev)
    model.denom = torch.zeros((newP, 1), device=dev)
    model.max_radii2D = torch.zeros((newP,), device=dev)
    if optimizer is not None:
        _rebind_optimizer(optimizer, old_params, new_params, flat_m, flat_v, widths, newP)
    del keep_alive
    return newP


def _rebind_optimizer(optimizer, old_params, new_params, flat_m, flat_v, widths, newP):
    if hasattr(optimizer, "rebuild"):             # step.ShardedAdam: re-flatten the parameters, keep this rank's share
        optimizer.rebuild(new_params, flat_m, flat_v)
        return
    if hasattr(optimizer, "exp_avg") and torch.is_tensor(optimizer.exp_avg):   # step.FusedAdam
        order = {id(p): k for k, p in enumerate(old_params)}
        assert [order[id(p)] for p in optimizer.params] == list(range(6)), "FusedAdam must own the six tensors in model order"
        optimizer.params = list(new_params)
        optimizer.exp_avg, optimizer.exp_avg_sq = flat_m, flat_v
        return
    off = 0
    views = []
    for w in widths:
        views.append((None, None) if flat_m is None else (flat_m[off:off + newP * w], flat_v[off:off + newP * w]))
        off += newP * w
    for group in optimizer.param_groups:
        assert len(group["params"]) == 1, "one tensor per parameter group (scene/gaussian_model.py:154-161)"
        old = group["params"][0]
        k = [i for i, p in enumerate(old_params) if p is old]
        if not k:
            continue
        k = k[0]
        st = optimizer.state.pop(old, None)
        group["params"][0] = new_params[k]
        if st and views[k][0] is not None:
            st["exp_avg"] = views[k][0].view_as(new_params[k]).clone()
            st["exp_avg_sq"] = views[k][1].view_as(new_params[k]).clone()
            optimizer.state[new_params[k]] = st
