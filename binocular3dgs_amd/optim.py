"""`torch.optim.Adam`-compatible optimiser whose `step()` is ONE launch of libb3gs_raster.so for all parameter groups
(`b3gs_adam_step_at`, csrc/optim.hip, reached through the compiled `_C` module: csrc/host/optim.cpp) -- what sits behind `gaussians.optimizer.step()` / `.zero_grad(set_to_none=True)` of
an unchanged train.py:196-198 and behind the optimiser-state surgery of scene/gaussian_model.py:258-340.

It IS a torch.optim.Adam (subclass): `param_groups` (the reference's six named groups, scene/gaussian_model.py:154-161),
`state[param] = {"step", "exp_avg", "exp_avg_sq"}`, `state_dict()` / `load_state_dict()` (the checkpoint tuple of
scene/gaussian_model.py:61-93, golden G9), `add_param_group`, `zero_grad` are torch's own; only the arithmetic of `step()`
moves: torch issues two foreach kernels per group plus scalar bookkeeping (12+ launches, ~0.35 ms per iteration at 1M
Gaussians), here every group's tensors are segments of one launch (28 bytes per parameter float at 5-6.5 TB/s).
No CPU path: host parameters raise.  amsgrad / weight_decay / maximize are not what the reference uses and raise.
"""
from __future__ import annotations

import torch

from torch.optim.optimizer import _global_optimizer_post_hooks as _global_post_hooks
from torch.optim.optimizer import _global_optimizer_pre_hooks as _global_pre_hooks

from . import _C, _lib


def _hooked_step(self, closure):
    """step() as torch.optim.Optimizer.profile_hook_step would have wrapped it (hooks registered, or a profiler running)"""
    args, kwargs = (self,) if closure is None else (self, closure), {}
    with torch.autograd.profiler.record_function(f"Optimizer.step#{self.__class__.__name__}.step"):
        for hook in (*_global_pre_hooks.values(), *self._optimizer_step_pre_hooks.values()):
            result = hook(self, args, kwargs)
            if result is not None:
                if isinstance(result, tuple) and len(result) == 2:
                    args, kwargs = result
                else:
                    raise RuntimeError(f"{hook} must return None or a tuple of (new_args, new_kwargs), but got {result}.")
        out = Adam._step(*args, **kwargs)
        self._optimizer_step_code()
        for hook in (*self._optimizer_step_post_hooks.values(), *_global_post_hooks.values()):
            hook(self, args, kwargs)
        return out


class Adam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, **kw):
        kw.pop("foreach", None), kw.pop("fused", None), kw.pop("capturable", None)
        if kw.pop("maximize", False) or kw.pop("differentiable", False):
            raise NotImplementedError("binocular3dgs_amd.optim.Adam: maximize / differentiable are not supported")
        if kw:
            raise TypeError(f"unexpected arguments {sorted(kw)}")
        if weight_decay or amsgrad:
            raise NotImplementedError("binocular3dgs_amd.optim.Adam: weight_decay / amsgrad are not supported "
                                      "(scene/gaussian_model.py:163 uses neither)")
        # (foreach / fused stay None, torch's defaults: state_dict() then carries exactly the hyper-parameters the reference's
        # own torch.optim.Adam(l, lr=0.0, eps=1e-15) carries -- this class's step() consults neither)
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False)

    def _init_state(self, p):
        st = self.state[p]
        if len(st) == 0:
            # torch.optim.Adam's own layout (non-capturable: the step counter is a host tensor)
            st["step"] = torch.tensor(0.0, dtype=torch.float32)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return st

    def step(self, closure=None):
        """torch.optim.Optimizer wraps every subclass's step() in a profiler range + hook dispatch (~40 us of host time per call,
        a third of this step's own); this method is marked `hooked` so the wrapper is not installed, and does what it did:
        pre / post hooks (global and per optimiser) when any are registered, the profiler range while a profiler runs."""
        if _global_pre_hooks or _global_post_hooks or self._optimizer_step_pre_hooks or self._optimizer_step_post_hooks \
                or torch.autograd._profiler_enabled():
            return _hooked_step(self, closure)
        out = self._step(closure)
        self._optimizer_step_code()
        return out

    step.hooked = True

    def zero_grad(self, set_to_none: bool = True) -> None:
        if not set_to_none or torch.autograd._profiler_enabled():
            return super().zero_grad(set_to_none)
        for group in self.param_groups:       # (torch's loop without its profiler range and foreach bookkeeping)
            for p in group["params"]:
                if p.grad is not None:
                    p.grad = None

    @torch.no_grad()
    def _step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        # segments grouped by (device, betas, eps, step number): ONE launch per group of up to 8 tensors -- the reference's
        # six single-tensor groups share everything but the learning rate, which travels per segment
        buckets = {}
        plan = self.__dict__.setdefault("_b3gs_plan", {})     # parameter -> (data_ptr, state dict, exp_avg, exp_avg_sq): checked once
        state = self.state
        purged = False
        for group in self.param_groups:
            if group.get("weight_decay", 0) or group.get("amsgrad", False) or group.get("maximize", False):
                raise NotImplementedError("binocular3dgs_amd.optim.Adam: weight_decay / amsgrad / maximize are not supported")
            b1, b2 = group["betas"]
            lr = group["lr"]
            lr = float(lr.item()) if torch.is_tensor(lr) else float(lr)
            eps = float(group["eps"])
            for p in group["params"]:
                g = p.grad
                if g is None:
                    continue
                ent = plan.get(p)
                st = state.get(p)
                if ent is None or st is None or ent[1] is not st or ent[0] != p.data_ptr() or st.get("exp_avg") is not ent[2] \
                        or st.get("exp_avg_sq") is not ent[3]:
                    if not p.is_cuda:
                        raise _lib.B3gsError("binocular3dgs_amd.optim.Adam: parameters must live on the HIP device (no CPU path)")
                    if not purged:      # parameters that left the optimiser (a densification replaced them) leave the plan too
                        purged = True
                        live = {id(q) for gr in self.param_groups for q in gr["params"]}
                        for q in [q for q in plan if id(q) not in live]:
                            del plan[q]
                    if p.dtype != torch.float32 or not p.is_contiguous():
                        raise _lib.B3gsError("binocular3dgs_amd.optim.Adam: parameters must be contiguous float32 tensors")
                    st = self._init_state(p)
                    m, v = st["exp_avg"], st["exp_avg_sq"]
                    if not (m.is_contiguous() and v.is_contiguous() and m.dtype == torch.float32 and v.dtype == torch.float32
                            and m.device == p.device and v.device == p.device and m.numel() == p.numel() == v.numel()):
                        raise _lib.B3gsError("binocular3dgs_amd.optim.Adam: exp_avg / exp_avg_sq must be contiguous float32 "
                                             "tensors of the parameter's size on its device")
                    ent = plan[p] = (p.data_ptr(), st, m, v)
                if g.is_sparse:
                    raise RuntimeError("Adam does not support sparse gradients")
                step_t = st["step"]
                step_t += 1
                key = (p.device.index, b1, b2, eps, int(step_t))
                b = buckets.get(key)
                if b is None:
                    b = buckets[key] = []
                b.append((p, g, ent[2], ent[3], lr))
        # one call into the compiled module per bucket (csrc/host/optim.cpp: adam_step_at fills the B3gsAdamSegment array, launches
        # up to 8 tensors at a time and bumps the parameters' version counters -- the kernel wrote through raw pointers, and
        # saved-tensor checks / the depth-order hint of rasterizer._RasterizeRaw key on the position tensor's version)
        for (_di, b1, b2, eps, step), segs_py in buckets.items():
            _C.adam_step_at([e[0] for e in segs_py], [e[1] for e in segs_py], [e[2] for e in segs_py], [e[3] for e in segs_py],
                            [e[4] for e in segs_py], step, b1, b2, eps)
        return loss
