"""Checkpoint capture / restore in the reference's format (SURVEY.md section 8 f4 "PLY / checkpoint formats").

The reference saves `torch.save((gaussians.capture(), iteration), "chkpnt<iter>.pth")` (train.py:200-202) and resumes
with `gaussians.restore(model_params, opt)` (train.py:41-43).  `capture()` is the 12-tuple of
scene/gaussian_model.py:61-75:

    (active_sh_degree, _xyz, _features_dc, _features_rest, _scaling, _rotation, _opacity, max_radii2D,
     xyz_gradient_accum, denom, optimizer.state_dict(), spatial_lr_scale)

and the optimiser is `torch.optim.Adam` over six single-tensor groups in the order xyz, f_dc, f_rest, opacity, scaling,
rotation (scene/gaussian_model.py:154-163).  This module maps the build's optimisers -- `step.FusedAdam` (flat moments,
device-side step counter), `step.ShardedAdam` (moments sharded over the ranks) and plain `torch.optim.Adam` -- to that
`state_dict()` layout and back, so a run of either code base can be resumed by the other:

    capture(model, optimizer, ...)             -> the 12-tuple (tensors detached; ShardedAdam moments are gathered)
    restore(model, model_args, optimizer, ...) -> parameters, statistics and Adam state put back in place
    save(path, model, optimizer, iteration) / load(path) -> the reference's (tuple, iteration) file

The moments of the build's optimisers are tensor-major in `model.parameters()` order (xyz, f_dc, f_rest, scaling,
rotation, opacity); the reference's state indices follow ITS group order, hence the permutation below.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch

# the reference's param-group order (scene/gaussian_model.py:154-161) and where each group sits in model.parameters()
REF_GROUPS = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")
MODEL_ORDER = ("xyz", "f_dc", "f_rest", "scaling", "rotation", "opacity")
_ATTR = {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity",
         "scaling": "_scaling", "rotation": "_rotation"}


def _group_template(betas, eps) -> dict:
    """The key set (and order) `torch.optim.Adam(..., lr=0.0, eps=...)`.state_dict() gives a param group under the
    installed torch -- built from a real optimiser so that the reference's load_state_dict accepts the result."""
    p = torch.nn.Parameter(torch.zeros(1))
    g = torch.optim.Adam([{"params": [p], "lr": 0.0, "name": "x"}], lr=0.0, betas=tuple(betas), eps=float(eps)).state_dict()
    return dict(g["param_groups"][0])


def _full_moments(optimizer) -> Tuple[torch.Tensor, torch.Tensor, int, Sequence[float], tuple, float]:
    """(exp_avg, exp_avg_sq) flat and tensor-major in model.parameters() order, step, lrs, betas, eps."""
    from .step import FusedAdam, ShardedAdam
    if isinstance(optimizer, ShardedAdam):
        m, v = optimizer.gather_full_state()          # all-gather when the state is sharded over ranks
        return m[:optimizer.numel], v[:optimizer.numel], int(optimizer.step_count.item()), optimizer.lrs, optimizer.betas, optimizer.eps
    if isinstance(optimizer, FusedAdam):
        return optimizer.exp_avg, optimizer.exp_avg_sq, int(optimizer.step_count.item()), optimizer.lrs, optimizer.betas, optimizer.eps
    raise TypeError(f"unsupported optimizer {type(optimizer).__name__}")


def optimizer_state_dict(model, optimizer, lrs_by_name: Optional[Dict[str, float]] = None) -> dict:
    """`optimizer.state_dict()` as the reference's torch.optim.Adam would return it.  A torch optimiser whose groups
    carry the reference's names is passed through unchanged."""
    if isinstance(optimizer, torch.optim.Optimizer):
        sd = optimizer.state_dict()
        gs = sd["param_groups"]
        names = [g.get("name") for g in gs]
        if sorted(n for n in names if n) == sorted(REF_GROUPS) and all(len(g["params"]) == 1 for g in gs) and \
                names != list(REF_GROUPS):
            # named groups in another order (e.g. model.parameters()): emit them in the reference's order, so that the
            # reference's positional load_state_dict() pairs every tensor with its own moments
            by = dict(zip(names, gs))
            state, groups = {}, []
            for idx, n in enumerate(REF_GROUPS):
                g = dict(by[n])
                st = sd["state"].get(g["params"][0])
                if st is not None:
                    state[idx] = st
                g["params"] = [idx]
                groups.append(g)
            return {"state": state, "param_groups": groups}
        return sd
    m, v, step, lrs, betas, eps = _full_moments(optimizer)
    params = {n: getattr(model, _ATTR[n]) for n in REF_GROUPS}
    offs, off = {}, 0
    for n in MODEL_ORDER:
        offs[n] = off
        off += params[n].numel()
    lr_of = dict(zip(MODEL_ORDER, lrs))
    if lrs_by_name:
        lr_of.update(lrs_by_name)
    tmpl = _group_template(betas, eps)
    state, groups = {}, []
    for idx, n in enumerate(REF_GROUPS):
        p = params[n]
        if step > 0:          # torch creates a parameter's state at its first step
            a, b = offs[n], offs[n] + p.numel()
            state[idx] = {"step": torch.tensor(float(step)),
                          "exp_avg": m[a:b].detach().clone().view(p.shape),
                          "exp_avg_sq": v[a:b].detach().clone().view(p.shape)}
        g = dict(tmpl)
        g["lr"], g["name"], g["params"] = float(lr_of[n]), n, [idx]
        groups.append(g)
    return {"state": state, "param_groups": groups}


def capture(model, optimizer=None, spatial_lr_scale: Optional[float] = None,
            lrs_by_name: Optional[Dict[str, float]] = None) -> tuple:
    """scene/gaussian_model.py:61-75.  Tensors are the live parameters / statistics (as in the reference); the
    optimiser entry is a fresh state_dict."""
    optimizer = optimizer if optimizer is not None else getattr(model, "optimizer", None)
    if optimizer is None:
        raise ValueError("capture() needs the optimizer (the reference stores optimizer.state_dict())")
    if getattr(model, "denom", None) is None:
        model.init_densification_stats()
    scale = getattr(model, "spatial_lr_scale", 0) if spatial_lr_scale is None else spatial_lr_scale
    return (model.active_sh_degree, model._xyz, model._features_dc, model._features_rest, model._scaling,
            model._rotation, model._opacity, model.max_radii2D, model.xyz_gradient_accum, model.denom,
            optimizer_state_dict(model, optimizer, lrs_by_name), scale)


def _moments_from_state_dict(model, opt_dict) -> Tuple[torch.Tensor, torch.Tensor, int, Dict[str, float]]:
    """Flat tensor-major (model.parameters() order) moments, the step and the learning rates by group name."""
    groups = opt_dict["param_groups"]
    by_name = {}
    for gi, g in enumerate(groups):
        name = g.get("name", REF_GROUPS[gi] if gi < len(REF_GROUPS) else None)
        by_name[name] = g
    missing = [n for n in REF_GROUPS if n not in by_name]
    if missing:
        raise ValueError(f"checkpoint optimizer state lacks the groups {missing}")
    dev = model._xyz.device
    ms, vs, steps = [], [], []
    for n in MODEL_ORDER:
        p = getattr(model, _ATTR[n])
        st = opt_dict["state"].get(by_name[n]["params"][0])
        if st is None:
            ms.append(torch.zeros(p.numel(), device=dev))
            vs.append(torch.zeros(p.numel(), device=dev))
            continue
        if tuple(st["exp_avg"].shape) != tuple(p.shape):
            raise ValueError(f"checkpoint moments of '{n}' have shape {tuple(st['exp_avg'].shape)}, parameter {tuple(p.shape)}")
        ms.append(st["exp_avg"].to(dev, torch.float32).reshape(-1))
        vs.append(st["exp_avg_sq"].to(dev, torch.float32).reshape(-1))
        steps.append(int(float(st["step"])))
    if steps and min(steps) != max(steps):
        raise ValueError("the groups of the checkpoint disagree on the step count; the one-launch Adam keeps one counter")
    return torch.cat(ms), torch.cat(vs), (steps[0] if steps else 0), {n: float(by_name[n]["lr"]) for n in REF_GROUPS}


def _by_group_name(optimizer, opt_dict) -> dict:
    """torch's load_state_dict() pairs parameter groups BY POSITION.  The checkpoint's groups are in the reference's
    order (xyz, f_dc, f_rest, opacity, scaling, rotation); a torch optimiser built over `model.parameters()` has scaling /
    rotation / opacity elsewhere.  When both sides name their single-tensor groups, the checkpoint is re-ordered to the
    target's order first (each group keeps the TARGET's hyper-parameter keys it lacks, e.g. after a torch upgrade)."""
    tgt = optimizer.state_dict()["param_groups"]
    src = opt_dict["param_groups"]
    if not (all("name" in g for g in tgt) and all("name" in g for g in src) and
            all(len(g["params"]) == 1 for g in tgt) and all(len(g["params"]) == 1 for g in src)):
        return opt_dict
    by = {g["name"]: g for g in src}
    if sorted(by) != sorted(g["name"] for g in tgt):
        raise ValueError(f"checkpoint groups {sorted(by)} do not match the optimizer's {sorted(g['name'] for g in tgt)}")
    groups, state = [], {}
    for g in tgt:
        sg = dict(g)
        sg.update({k: v for k, v in by[g["name"]].items() if k != "params"})
        groups.append(sg)
        st = opt_dict["state"].get(by[g["name"]]["params"][0])
        if st is not None:
            state[g["params"][0]] = st
    return {"state": state, "param_groups": groups}


def restore(model, model_args: tuple, optimizer=None, optimizer_factory=None):
    """scene/gaussian_model.py:77-93: put a captured tuple back.  `optimizer` (FusedAdam / ShardedAdam / torch Adam built
    over model.parameters()) receives the Adam state; alternatively `optimizer_factory(model) -> optimizer` is called
    after the parameters are in place (the reference calls training_setup there).  Returns the optimizer."""
    (model.active_sh_degree, xyz, f_dc, f_rest, scaling, rotation, opacity, max_radii2D, xyz_gradient_accum, denom,
     opt_dict, model.spatial_lr_scale) = model_args
    dev = model._xyz.device if model._xyz.numel() else xyz.device
    same_shape = all(getattr(model, a).shape == t.shape for a, t in
                     (("_xyz", xyz), ("_features_dc", f_dc), ("_features_rest", f_rest), ("_scaling", scaling),
                      ("_rotation", rotation), ("_opacity", opacity)))
    with torch.no_grad():
        for attr, t in (("_xyz", xyz), ("_features_dc", f_dc), ("_features_rest", f_rest), ("_scaling", scaling),
                        ("_rotation", rotation), ("_opacity", opacity)):
            if same_shape:      # keep the storage (flat buffers of ShardedAdam, gradient slab views)
                getattr(model, attr).copy_(t.detach().to(dev))
            else:
                setattr(model, attr, torch.nn.Parameter(t.detach().to(dev, torch.float32).clone().contiguous(),
                                                        requires_grad=True))
    model.max_radii2D = max_radii2D.detach().to(dev).clone()
    model.xyz_gradient_accum = xyz_gradient_accum.detach().to(dev).clone()
    model.denom = denom.detach().to(dev).clone()
    if optimizer is None and optimizer_factory is not None:
        optimizer = optimizer_factory(model)
        # (the reference's order, scene/gaussian_model.py:90-92: training_setup() -- which zeroes the statistics -- first, the
        # captured statistics after it)
        model.max_radii2D = max_radii2D.detach().to(dev).clone()
        model.xyz_gradient_accum = xyz_gradient_accum.detach().to(dev).clone()
        model.denom = denom.detach().to(dev).clone()
    if optimizer is None:
        return None
    if isinstance(optimizer, torch.optim.Optimizer):
        # the optimiser must own THESE parameter tensors: when the checkpoint's shapes differed from the model's, fresh
        # Parameters were just installed and an optimiser built before that still holds the old ones -- load_state_dict()
        # validates no shapes, so the moments would be attached to stale tensors and the restored parameters never stepped
        # (the reference rebuilds its optimiser in training_setup() before load_state_dict, scene/gaussian_model.py:90-93)
        owned = [p for g in optimizer.param_groups for p in g["params"]]
        mine = list(model.parameters())
        if not (len(owned) == len(mine) and {p.data_ptr() for p in owned if p.numel()} == {p.data_ptr() for p in mine if p.numel()}
                and sorted(tuple(p.shape) for p in owned) == sorted(tuple(p.shape) for p in mine)):
            raise ValueError("restore(): the torch optimizer does not own the model's (restored) parameter tensors -- the "
                             "checkpoint changed their shapes; pass optimizer_factory=lambda model: <build the optimizer> "
                             "instead of an optimizer built for the old Gaussian count")
        optimizer.load_state_dict(_by_group_name(optimizer, opt_dict))
        model.optimizer = optimizer
        return optimizer
    from .step import FusedAdam, ShardedAdam
    m, v, step, lrs = _moments_from_state_dict(model, opt_dict)
    if isinstance(optimizer, ShardedAdam):
        if not same_shape or any(p.data_ptr() != q.data_ptr() for p, q in zip(optimizer.params, model.parameters())):
            optimizer.rebuild(model.parameters(), None, None)
        pad = optimizer.padded_numel - m.numel()
        fm = torch.cat([m, torch.zeros(pad, device=m.device)]) if pad else m
        fv = torch.cat([v, torch.zeros(pad, device=v.device)]) if pad else v
        lo = optimizer.rank * optimizer.chunk
        optimizer.exp_avg.copy_(fm[lo:lo + optimizer.chunk])
        optimizer.exp_avg_sq.copy_(fv[lo:lo + optimizer.chunk])
    elif isinstance(optimizer, FusedAdam):
        if optimizer.exp_avg.numel() != m.numel():
            optimizer.params = list(model.parameters())
            optimizer.exp_avg = torch.zeros_like(m)
            optimizer.exp_avg_sq = torch.zeros_like(v)
        optimizer.exp_avg.copy_(m)
        optimizer.exp_avg_sq.copy_(v)
    else:
        raise TypeError(f"unsupported optimizer {type(optimizer).__name__}")
    optimizer.step_count.fill_(step)
    optimizer.lrs = [lrs[n] for n in MODEL_ORDER]
    model.optimizer = optimizer
    return optimizer


def save(path: str, model, optimizer, iteration: int, **kw):
    """train.py:200-202: torch.save((gaussians.capture(), iteration), path)."""
    torch.save((capture(model, optimizer, **kw), int(iteration)), path)


def load(path: str, map_location=None):
    """train.py:42: (model_params, first_iter) = torch.load(path)."""
    model_params, first_iter = torch.load(path, map_location=map_location, weights_only=False)
    return model_params, int(first_iter)
